#!/usr/bin/env python3
"""Diagnostic: the prompt (M-row) Linears of a config one shape at a time - per-kernel time from the library's profiling sink,
weights rotated through more than the 256 MiB Infinity Cache so every call streams them from HBM as the prefill does.

    python tools/gemm_probe.py Llama-2-7b-hf w4 128 [iters]
    python tools/gemm_probe.py stablelm-base-alpha-3b bf16 512
"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from lit_parrot_amd import _hip, ops  # noqa: E402
from lit_parrot_amd.config import Config  # noqa: E402

DEV = torch.device("cuda", 0)
name, mode, M = sys.argv[1], sys.argv[2], int(sys.argv[3])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 20
cfg = Config.from_name(name)
G = 128
g = torch.Generator(device="cpu").manual_seed(5)


def w4_image(N, K):
    q = torch.randint(0, 256, (K // 2, N), dtype=torch.uint8, generator=g).to(DEV).t()  # (N, K/2) with strides (1, N)
    s = (torch.rand((N, K // G), generator=g) * 0.02 + 0.005).to(torch.bfloat16).to(DEV)
    z = torch.randint(0, 16, (N, K // G), generator=g).to(torch.bfloat16).to(DEV)
    p = torch.empty((ops.w4_packed_bytes(N, K, G),), dtype=torch.uint8, device=DEV)
    ops.w4_repack(q, s, z, N, K, G, p, 0)
    return p


shapes = dict(cfg.linear_shapes())
shapes.pop("lm_head")
pair = "mlp.fc_1" in shapes
if pair:
    shapes.pop("mlp.fc_2")
for lname, (N, K) in shapes.items():
    swi = pair and lname == "mlp.fc_1"
    nbytes = N * K // 2 if mode == "w4" else N * K * 2
    copies = max(2, int(400e6 // (nbytes * (2 if swi else 1))) + 1)
    if mode == "w4":
        Ws = [(w4_image(N, K), w4_image(N, K) if swi else None) for _ in range(copies)]
    else:
        Ws = [((torch.randn((N, K), generator=g) * 0.02).to(torch.bfloat16).to(DEV), None) for _ in range(copies)]
    x = torch.randn((M, K), generator=g).to(torch.bfloat16).to(DEV)
    out = torch.empty((M, N), dtype=torch.bfloat16, device=DEV)

    def call(i):
        w, w2 = Ws[i % copies]
        if mode == "w4":
            ops.w4_linear(w, N, K, G, x, out, epilogue=ops.EPI_SWIGLU if swi else ops.EPI_NONE, packed2=w2)
        else:
            ops.bf16_linear(w, x, out)

    for i in range(3):
        call(i)
    torch.cuda.synchronize()
    _hip.prof_begin()
    for i in range(iters):
        call(i)
    stats = _hip.prof_end()
    tot = sum(v[0] for v in stats.values()) / iters * 1e3
    flops = 2.0 * M * N * K * (2 if swi else 1)
    parts = ", ".join(f"{k} {v[0] / v[1] * 1e3:.1f} us x{v[1] // iters}" for k, v in sorted(stats.items(), key=lambda kv: -kv[1][0]))
    print(f"{lname:10s} M={M} N={N}{' x2 (SwiGLU)' if swi else ''} K={K}: {tot:7.1f} us = {flops / tot / 1e6:6.0f} TFLOP/s, "
          f"{nbytes * (2 if swi else 1) / tot / 1e6:5.2f} TB/s of weights | {parts}")
    del Ws
    torch.cuda.empty_cache()
