#!/usr/bin/env python3
"""Diagnostic: per-step logit difference between the stream engine and the multi-launch step (same forced tokens)."""
import sys
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "tests"))
import test_engine_gpu as T  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "tiny-llama"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
cfg, qsd, model = T.int4_model(name)
prompt = T.synthetic_prompt(cfg, 9, 3)
tok_a, log_a = T.run_session(model, prompt, n, engine=False)
tok_b, log_b = T.run_session(model, prompt, n, engine=True, follow=tok_a.to(T.DEV))
for i in range(log_a.shape[0]):
    d = (log_a[i] - log_b[i]).abs()
    print(f"step {i}: max |diff| {float(d.max()):.4f} mean {float(d.mean()):.5f}  argmax a {int(log_a[i].argmax())} b {int(log_b[i].argmax())}")
