#!/usr/bin/env python3
"""Diagnostic: per-kernel time of the prefill pass (profiling sink of the library)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from lit_parrot_amd import _hip  # noqa: E402
from lit_parrot_amd.config import Config  # noqa: E402
from lit_parrot_amd.generate import base as gb  # noqa: E402
from lit_parrot_amd.synth import build_synthetic_model, synthetic_prompt  # noqa: E402

name, mode, T = sys.argv[1], (None if sys.argv[2] == "bf16" else sys.argv[2]), int(sys.argv[3])
cfg = Config.from_name(name)
dev = torch.device("cuda", 0)
model = build_synthetic_model(cfg, mode, device=dev)
with torch.no_grad():
    sess = gb.DecodeSession(model, T + 8, T + 8, True, use_graph=False)
    prompt = synthetic_prompt(cfg, T).to(dev)
    sess.prefill(prompt)
    torch.cuda.synchronize()
    _hip.prof_begin()
    sess.prefill(prompt)
    stats = _hip.prof_end()
tot = sum(v[0] for v in stats.values())
for k, v in sorted(stats.items(), key=lambda kv: -kv[1][0]):
    print(f"{k:20s} {v[0]:9.3f} ms  {v[1]:5d} launches  {v[0] / v[1] * 1e3:9.1f} us each")
flops = 2.0 * cfg.n_linear_params() * T - 2.0 * cfg.padded_vocab_size * cfg.n_embd * (T - 1)
print(f"total {tot:.2f} ms; linear FLOPs {flops / 1e12:.2f} T -> {flops / tot / 1e9:.0f} TFLOP/s over the whole prefill")
