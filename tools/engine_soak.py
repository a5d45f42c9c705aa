#!/usr/bin/env python3
"""Full-model soak of the stream engine: N greedy steps twice from the same prompt - identical tokens, error word clear.
python tools/engine_soak.py <workload> [steps]"""
import sys
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
from bench import WORKLOADS  # noqa: E402
import lit_parrot_amd as L  # noqa: E402
from lit_parrot_amd.config import Config  # noqa: E402
from lit_parrot_amd.generate import base as gb  # noqa: E402
from lit_parrot_amd.synth import build_synthetic_model, synthetic_prompt  # noqa: E402

name = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
cfg_name, mode, T, _ = WORKLOADS[name]
cfg = Config.from_name(cfg_name)
dev = torch.device("cuda", 0)
model = build_synthetic_model(cfg, mode, seed=1234, device=dev)
prompt = synthetic_prompt(cfg, T, seed=1234)
total = min(cfg.block_size, T + steps + 2)
runs = []
with torch.no_grad():
    for rep in range(2):
        model.reset_cache()
        model.__dict__.pop("_decode_sessions", None)
        sess = gb.DecodeSession(model, total, total, True, engine=True)
        assert sess.eng is not None, "model not supported by the engine"
        logits = sess.prefill(prompt.to(dev))
        L.ops.argmax_advance(logits, sess.tokens, sess.pos)
        sess.capture()
        for _ in range(total - T - 2):
            sess.step()
        torch.cuda.synchronize()
        sess.eng.check_error()
        runs.append(sess.tokens[: total - 1].cpu().clone())
same = torch.equal(runs[0], runs[1])
print(f"{name}: {total - T - 2} steps x 2, identical tokens: {same}, distinct tokens in the run: {int(runs[0].unique().numel())}")
sys.exit(0 if same else 1)
