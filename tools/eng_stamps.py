#!/usr/bin/env python3
"""Diagnostic: where workgroup 0 of the stream engine spends a token (100 MHz stamps of its first consumer wave:
op enter, input vector ready, last own unit done).  python tools/eng_stamps.py [workload] [steps]"""
import sys
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
from bench import WORKLOADS  # noqa: E402
from lit_parrot_amd.config import Config  # noqa: E402
from lit_parrot_amd.generate import base as gb  # noqa: E402
from lit_parrot_amd.synth import build_synthetic_model, synthetic_prompt  # noqa: E402
import lit_parrot_amd as L  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "llama2-7b-int4"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
cfg_name, mode, T, _ = WORKLOADS[name]
cfg = Config.from_name(cfg_name)
dev = torch.device("cuda", 0)
model = build_synthetic_model(cfg, mode, seed=1234, device=dev)
prompt = synthetic_prompt(cfg, T, seed=1234)
total = T + steps + 8
with torch.no_grad():
    sess = gb.DecodeSession(model, total, total, True, use_graph=False, engine=True)
    assert sess.eng is not None, "model not supported by the engine"
    logits = sess.prefill(prompt.to(dev))
    L.ops.argmax_advance(logits, sess.tokens, sess.pos)
    for _ in range(3):
        sess.step()
    dbg = sess.eng.enable_stamps()
    acc = None
    for _ in range(steps):
        sess.step()
        torch.cuda.synchronize()
        d = dbg.cpu().view(-1, 16).double()
        raw = dbg.cpu().view(-1, 16)
        acc = d if acc is None else acc + d
    sess.eng.check_error()
    allc = sess.eng.dbg_all.cpu().view(-1, 256, 2).double() / 100.0  # last step: [op][cu][ready, done] in us
d = acc / steps
t0 = d[0, 0]
names = []
for i in range(cfg.n_layer):
    from lit_parrot_amd.engine import StreamEngine
    nch = len(StreamEngine._down_chunks(cfg))
    downs = ["down"] if nch == 1 else [f"down{j}" for j in range(nch)]
    names += (["qkv", "fc0", "attn1", "fc1", "attn2", "fc2", "proj"] if cfg.parallel_residual else ["qkv", "attn", "proj", "fc"]) + downs
names.append("lm_head")
agg = {}
for k, n in enumerate(names):
    enter, ready, done = (d[k, 0] - t0) / 100.0, (d[k, 1] - t0) / 100.0, (d[k, 2] - t0) / 100.0
    nxt = (d[k + 1, 0] - t0) / 100.0 if k + 1 < len(names) else done
    a = agg.setdefault(n, [0.0, 0.0, 0.0, 0, 0.0, 0.0, 0.0])
    a[6] += d[k, 12] / 100.0
    a[0] += ready - enter
    a[1] += done - ready
    a[2] += nxt - done
    a[3] += 1
    gate_spins = int(raw[k, 3]) >> 40
    d[k, 3] = float(int(raw[k, 3]) & ((1 << 40) - 1))
    a[4] += d[k, 3] / 100.0
    lead = (d[k, 0] - d[k, 4]) / 100.0  # how long before the consumers the loader reached this op
    a[5] += lead
    if k < 11 or k >= len(names) - 6:
        extra = ""
        if not n.startswith("attn") and k > 0:
            extra = f" | gate +{(d[k, 6] - d[k, 0]) / 100.0:.2f} sweep +{(d[k, 7] - d[k, 6]) / 100.0:.2f} norm+bars +{(d[k, 1] - d[k, 7]) / 100.0:.2f} gate polls {gate_spins}"
            if d[k, 8] > 0:
                t9 = d[k, 9] if d[k, 9] > 0 else d[k, 8]
                extra += f" [stat1 +{(d[k, 8] - d[k, 7]) / 100.0:.2f} stat2 +{(t9 - d[k, 8]) / 100.0:.2f} norm slot +{(d[k, 10] - t9) / 100.0:.2f} apply +{(d[k, 11] - d[k, 10]) / 100.0:.2f} bar +{(d[k, 1] - d[k, 11]) / 100.0:.2f}]"
            else:
                extra += f" [apply +{(d[k, 11] - d[k, 7]) / 100.0:.2f} bar +{(d[k, 1] - d[k, 11]) / 100.0:.2f}]"
        if n in ("attn", "attn1"):
            extra = f" | units +{(d[k, 6] - d[k, 1]) / 100.0:.2f} merge+bar +{(d[k, 7] - d[k, 6]) / 100.0:.2f} tail +{(d[k, 2] - d[k, 7]) / 100.0:.2f}"
        print(f"op {k:3d} {n:8s} enter {enter:9.2f} us  input ready +{ready - enter:6.2f}  units done +{done - ready:6.2f} (waiting for slots {d[k, 3] / 100.0:5.2f})"
              f"  loader lead {lead:6.2f} us, seq/pub at reach {int(raw[k, 5]) & 0xffffffff}/{int(raw[k, 5]) >> 32}{extra}")
print("mean per op type (us): wait+gather | own units (of which waiting for slots) | to next op | loader lead | loader 0 stalled on a full ring while issuing the op")
for n, a in agg.items():
    print(f"  {n:8s} {a[0] / a[3]:7.2f} {a[1] / a[3]:7.2f} ({a[4] / a[3]:5.2f}) {a[2] / a[3]:7.2f} {a[5] / a[3]:7.2f} {a[6] / a[3]:7.2f}   x{a[3]}")
print(f"token (workgroup 0, wave 1): {(d[len(names) - 1, 2] - t0) / 100.0:.1f} us")

# skew across the 256 workgroups (last step): spread of the "units done" and "input ready" times per op type
import collections
sk = collections.defaultdict(lambda: [0.0, 0.0, 0, collections.Counter(), collections.Counter()])
for k, n in enumerate(names):
    done, ready = allc[k, :, 1], allc[k, :, 0]
    a = sk[n]
    a[0] += float(done.max() - done.min())
    a[1] += float(ready.max() - ready.min())
    a[2] += 1
    a[3][int(done.argmax())] += 1
    a[4][int(done.argmin())] += 1
print("skew over workgroups (us): units-done spread | input-ready spread | most often last / first to finish")
for n, a in sk.items():
    print(f"  {n:8s} {a[0] / a[2]:6.2f} {a[1] / a[2]:6.2f}   last {a[3].most_common(3)}  first {a[4].most_common(3)}")
k = 3  # one fc op in detail: units-done time by workgroup, relative to the earliest
dd = allc[k, :, 1] - allc[k, :, 1].min()
print("fc op 3, units-done offset by workgroup (us), 16 per row:")
for r0 in range(0, 256, 16):
    print("  " + " ".join(f"{float(x):5.2f}" for x in dd[r0:r0 + 16]))

# is a CU's pace systematic?  Per op type: units time (done - ready) of every workgroup relative to the op's mean, averaged over
# the layers, then summarised per XCD (workgroup id % 8) and as the correlation between the first and second half of the layers
import numpy as np
dur = (allc[:, :, 1] - allc[:, :, 0]).numpy()  # [op][cu]
for n in sorted(set(names)):
    idx = [k for k, m in enumerate(names) if m == n]
    if len(idx) < 4 or not n.startswith(("qkv", "fc", "down", "lm")):
        continue
    rel = np.stack([dur[k] / dur[k].mean() for k in idx])  # [layer][cu]
    a, b = rel[: len(idx) // 2].mean(0), rel[len(idx) // 2:].mean(0)
    corr = float(np.corrcoef(a, b)[0, 1])
    m = rel.mean(0)
    per_xcd = [float(m[x::8].mean()) for x in range(8)]
    print(f"  {n:6s} relative units time per XCD: " + " ".join(f"{v:5.3f}" for v in per_xcd) + f" | per-CU spread {m.min():.2f} .. {m.max():.2f}, half-vs-half correlation {corr:+.2f}")
