#!/usr/bin/env python3
"""Diagnostic: where workgroup 0 of the persistent step spends its time (100 MHz stamps, separate diagnostic run)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import lit_parrot_amd as L  # noqa: E402
from lit_parrot_amd.config import Config  # noqa: E402
from lit_parrot_amd.generate import base as gb  # noqa: E402
from lit_parrot_amd.synth import build_synthetic_model, synthetic_prompt  # noqa: E402

cfg = Config.from_name(sys.argv[1] if len(sys.argv) > 1 else "Llama-2-7b-hf")
dev = torch.device("cuda", 0)
model = build_synthetic_model(cfg, "gptq.int4-g128", device=dev)
T, total = 128, 300
with torch.no_grad():
    sess = gb.DecodeSession(model, total, total, True, use_graph=False, persistent=True)
    logits = sess.prefill(synthetic_prompt(cfg, T).to(dev))
    L.ops.argmax_advance(logits, sess.tokens, sess.pos)
    for _ in range(5):
        sess.step()
    dbg = sess.pk.enable_stamps()
    for _ in range(3):
        sess.step()
    torch.cuda.synchronize()
    sess.pk.check_error()
d = dbg.cpu().view(-1, 8).double() * 10e-3  # us
names = ["wait_done", "x_in_lds", "x_regs+norm", "dots+reduce", "epilogue_done", "stores_drained", "arrived", "loop_top"]
print("op  type   " + "  ".join(f"{n:>14s}" for n in ["wait", "x->lds", "norm+regs", "dots", "epilogue", "drain", "prefetch+arrive", "total"]))
nops = d.shape[0]
tot = 0
for k in range(min(nops - 1, 12)):
    t = d[k]
    nxt_top = d[k + 1][7]
    row = [t[0] - t[7], t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], t[5] - t[4], t[6] - t[5], nxt_top - t[7]]
    print(f"{k:3d} " + "  ".join(f"{float(v):14.2f}" for v in row))
print("token span us:", float(d[nops - 2][6] - d[0][7]))
