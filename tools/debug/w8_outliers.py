"""Diagnostic: how many outlier columns (|x| >= 6) do the rows of a prompt have in each LLM.int8 Linear of the synthetic model?"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from lit_parrot_amd.config import Config  # noqa: E402
from lit_parrot_amd.generate import base as gb  # noqa: E402
from lit_parrot_amd.quantize.bnb import InferenceLinear8bitLt  # noqa: E402
from lit_parrot_amd.synth import build_synthetic_model, synthetic_prompt  # noqa: E402

cfg = Config.from_name("Llama-2-7b-hf")
dev = torch.device("cuda", 0)
model = build_synthetic_model(cfg, "bnb.int8", device=dev)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 128
with torch.no_grad():
    sess = gb.DecodeSession(model, T + 8, T + 8, True, use_graph=False)
    sess.prefill(synthetic_prompt(cfg, T).to(dev))
    torch.cuda.synchronize()
for name, mod in list(model.named_modules()):
    if isinstance(mod, InferenceLinear8bitLt) and mod._act is not None and ("h.0." in name or "h.31." in name or name == "lm_head"):
        n = mod._act.nout.float()
        print(f"{name:32s} rows {mod._act.M:4d} K {mod._act.K:6d}: outlier columns per row mean {float(n.mean()):8.1f} max {int(n.max()):6d}")
