"""Diagnostic: device-clock timeline of the int4 GEMV launches inside one hipGraph replay of the decode step.
Every GEMV launch gets its own stamp buffer (3 workgroups x 8 stamps of the 100 MHz clock: entry, norm ready, dots done,
barrier, exit); prints, per launch of the first blocks, the span and the gap to the next GEMV's entry.
Needs the diagnostic build of the library (`python lit-parrot_amd/_build.py --diag`): the stamp / tuning hooks
(`parrot_tune_w4_stamps` ...) are not compiled into the shipped one."""
import argparse
import ctypes as C
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import lit_parrot_amd as L  # noqa: E402
from lit_parrot_amd import _hip, ops  # noqa: E402
from lit_parrot_amd.config import Config  # noqa: E402
from lit_parrot_amd.generate import base as gb  # noqa: E402
from lit_parrot_amd.synth import build_synthetic_model, synthetic_prompt  # noqa: E402


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="Llama-2-7b-hf")
    ap.add_argument("--mode", default="gptq.int4-g128")
    ap.add_argument("--blocks", type=int, default=3)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    cfg = Config.from_name(args.config)
    lib = _hip.load()
    lib.parrot_tune_w4_stamps.argtypes = [C.c_void_p]
    model = build_synthetic_model(cfg, args.mode, seed=1, device=dev)
    prompt = synthetic_prompt(cfg, 128, seed=1, device="cpu")
    bufs = []
    real = ops.w4_linear

    def stamped(*a, **k):
        b = torch.zeros(24, dtype=torch.int64, device=dev)
        bufs.append(b)
        lib.parrot_tune_w4_stamps(b.data_ptr())
        try:
            return real(*a, **k)
        finally:
            lib.parrot_tune_w4_stamps(None)

    with torch.no_grad():
        sess = gb.DecodeSession(model, 512, 512, True)
        logits = sess.prefill(prompt.to(dev))
        L.ops.argmax_advance(logits, sess.tokens, sess.pos)
        sess._step()  # warm-up (lazy repacks) without stamps
        torch.cuda.synchronize()
        ops.w4_linear = stamped
        import lit_parrot_amd.quantize.gptq as gq

        gq.ops.w4_linear = stamped
        sess.capture()
        n_launch = len(bufs) // 2  # capture() runs the step twice (warm-up + capture); the second half is in the graph
        graph_bufs = bufs[n_launch:]
        for _ in range(10):
            sess.step()
        torch.cuda.synchronize()
    names = ["qkv", "proj", "fc", "down"]
    rows = [b.cpu().view(3, 8).double() * 0.01 for b in graph_bufs]
    t0 = float(rows[0][:, 0].min())
    print(f"{'launch':>10} {'entry':>8} {'exit':>8} {'span':>6} {'gap to next gemv entry':>24}")
    for i in range(min(4 * args.blocks, len(rows) - 1)):
        a, b = rows[i], rows[i + 1]
        ent, ext = float(a[:, 0].min()), float(a[:, 4].max())
        print(f"{i:3d} {names[i % 4]:>6} {ent - t0:8.2f} {ext - t0:8.2f} {ext - ent:6.2f} {float(b[:, 0].min()) - ext:10.2f}")
    per_block = (float(rows[4 * (cfg.n_layer - 1)][:, 0].min()) - t0) / (cfg.n_layer - 1)
    print(f"mean per block: {per_block:.2f} us; step (first entry .. lm_head exit): {float(rows[-1][:, 4].max()) - t0:.1f} us")


if __name__ == "__main__":
    main()
