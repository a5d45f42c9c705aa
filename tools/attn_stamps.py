"""Diagnostic: phase stamps of the fused decode-attention kernel (workgroup 0) at Llama-2-7B shapes.
Needs the diagnostic build of the library (`python lit-parrot_amd/_build.py --diag`): the stamp / tuning hooks
(`parrot_tune_attn_stamps` ...) are not compiled into the shipped one."""
import ctypes as C
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from lit_parrot_amd import _hip, ops  # noqa: E402

DEV = torch.device("cuda", 0)


def main():
    lib = _hip.load()
    lib.parrot_tune_attn_stamps.argtypes = [C.c_void_p]
    BF = torch.bfloat16
    for n_groups, q_per_kv, hs, S, ctx in ((32, 1, 128, 273, 200), (32, 1, 128, 1024, 900), (8, 16, 64, 273, 200)):
        n_head = n_groups * q_per_kv
        qkv = torch.randn(1, n_groups * (q_per_kv + 2) * hs, device=DEV).to(BF)
        cos = torch.randn(2048, hs, device=DEV).to(torch.float16)
        sin = torch.randn(2048, hs, device=DEV).to(torch.float16)
        kc = torch.randn(n_groups, S, hs, device=DEV).to(BF)
        vc = torch.randn(n_groups, S, hs, device=DEV).to(BF)
        y = torch.empty(1, n_head * hs, dtype=BF, device=DEV)
        pos = torch.tensor([ctx], dtype=torch.int32, device=DEV)
        nsplit = ops.attn_nsplit(n_groups, S, q_per_kv)
        ws = ops.attn_workspace(1, n_head, hs, nsplit, DEV)
        tickets = torch.zeros(n_head, dtype=torch.int32, device=DEV)
        dbg = torch.zeros(8, dtype=torch.int64, device=DEV)
        for _ in range(5):
            ops.attn_fused_decode(qkv, cos, sin, hs, pos, kc, vc, n_groups, q_per_kv, hs, S, nsplit, ws, tickets, y)
        lib.parrot_tune_attn_stamps(dbg.data_ptr())
        ops.attn_fused_decode(qkv, cos, sin, hs, pos, kc, vc, n_groups, q_per_kv, hs, S, nsplit, ws, tickets, y)
        torch.cuda.synchronize()
        lib.parrot_tune_attn_stamps(None)
        d = dbg.cpu().double() * 0.01
        print(f"groups {n_groups} q/kv {q_per_kv} hs {hs} S {S} ctx {ctx} nsplit {nsplit}: rope+sync {float(d[1] - d[0]):5.2f} | key loop {float(d[2] - d[1]):5.2f} | "
              f"wave merge+sync {float(d[3] - d[2]):5.2f} | final merge+store {float(d[4] - d[3]):5.2f} | total {float(d[4] - d[0]):5.2f} us")


if __name__ == "__main__":
    main()
