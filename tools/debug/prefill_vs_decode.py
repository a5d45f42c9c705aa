"""Diagnostic: prompt path vs token-by-token decode path on full-width Llama-2-7B int4 models of growing depth."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from lit_parrot_amd.config import Config, name_to_config  # noqa: E402
from lit_parrot_amd.generate import base as gb  # noqa: E402
from lit_parrot_amd.synth import build_synthetic_model, synthetic_prompt  # noqa: E402

dev = torch.device("cuda", 0)
mode = sys.argv[1] if len(sys.argv) > 1 else "gptq.int4-g128"
mode = None if mode == "bf16" else mode
for L in (1, 2, 8, 32):
    cfg = Config(**{**name_to_config["Llama-2-7b-hf"], "n_layer": L})
    model = build_synthetic_model(cfg, mode, seed=1234, device=dev)
    T, S = 48, 96
    prompt = synthetic_prompt(cfg, T, seed=99).to(dev)
    with torch.no_grad():
        sess = gb.DecodeSession(model, S, S, greedy=False, use_graph=False)
        lp = sess.prefill(prompt).float().view(-1).clone()
        model.reset_cache()
        sess = gb.DecodeSession(model, S, S, greedy=False, use_graph=False)
        sess.tokens[:T].copy_(prompt)
        for t in range(T):
            sess.pos.fill_(t)
            ld = sess.step().clone()
        ld = ld.float().view(-1)
    d = (lp - ld).abs()
    print(f"L={L:2d} scale {float(ld.abs().max()):.3f} rms {float(ld.pow(2).mean().sqrt()):.3f} | max diff {float(d.max()):.4f} mean {float(d.mean()):.5f} "
          f"rel rms {float(d.pow(2).mean().sqrt() / ld.pow(2).mean().sqrt()):.4f} | argmax same {int(lp.argmax()) == int(ld.argmax())} "
          f"gap {float(ld.max() - ld[lp.argmax()]):.4f}", flush=True)
    del model, sess
    torch.cuda.empty_cache()
