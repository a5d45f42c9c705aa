#!/usr/bin/env python3
"""Diagnostic: where two calls of the int4 prompt GEMM on the same operands differ (tile-relative rows / columns, size)."""
import sys
from pathlib import Path
from collections import Counter

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from lit_parrot_amd import ops  # noqa: E402

DEV = torch.device("cuda", 0)
G = 128
g = torch.Generator(device="cpu").manual_seed(5)


def w4_image(N, K):
    q = torch.randint(0, 256, (K // 2, N), dtype=torch.uint8, generator=g).to(DEV).t()
    s = (torch.rand((N, K // G), generator=g) * 0.02 + 0.005).to(torch.bfloat16).to(DEV)
    z = torch.randint(0, 16, (N, K // G), generator=g).to(torch.bfloat16).to(DEV)
    p = torch.empty((ops.w4_packed_bytes(N, K, G),), dtype=torch.uint8, device=DEV)
    ops.w4_repack(q, s, z, N, K, G, p, 0)
    return p


M, N, K, swi = 512, 11008, 4096, True
w, w2 = w4_image(N, K), w4_image(N, K)
x = torch.randn((M, K), generator=g).to(torch.bfloat16).to(DEV)
outs = []
for rep in range(4):
    out = torch.empty((M, N), dtype=torch.bfloat16, device=DEV)
    ops.w4_linear(w, N, K, G, x, out, epilogue=ops.EPI_SWIGLU, packed2=w2)
    outs.append(out.float().cpu())
a, b = outs[0], outs[1]
d = (a - b).abs()
idx = (d > 0).nonzero()
print("differing", len(idx), "max |d|", float(d.max()), "mean |a|", float(a.abs().mean()))
print("rows%128 -> 32-blocks", sorted(Counter(((idx[:, 0] % 128) // 32).tolist()).items()))
print("cols%128 -> 32-blocks", sorted(Counter(((idx[:, 1] % 128) // 32).tolist()).items()))
print("m tiles", sorted(Counter((idx[:, 0] // 128).tolist()).items()))
nt = Counter((idx[:, 1] // 128).tolist())
print("n tiles with differences", len(nt), "of", (N + 127) // 128, "most", nt.most_common(5))
rel = d[d > 0] / a.abs().mean()
print("relative size quantiles", [float(rel.quantile(q)) for q in (0.5, 0.9, 0.99, 1.0)])
