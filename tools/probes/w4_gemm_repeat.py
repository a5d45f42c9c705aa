#!/usr/bin/env python3
"""Diagnostic: the int4 prompt GEMM called repeatedly on the same operands - every call must return the same bits."""
import sys
from pathlib import Path

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


for M in [int(a) for a in sys.argv[1:]] or [1100, 512, 128, 40]:
    for (N, K, swi) in ((12288, 4096, False), (4096, 4096, False), (11008, 4096, True), (4096, 11008, False)):
        w, w2 = w4_image(N, K), (w4_image(N, K) if swi else None)
        x = torch.randn((M, K), generator=g).to(torch.bfloat16).to(DEV)
        outs = []
        for rep in range(6):
            out = torch.empty((M, N), dtype=torch.bfloat16, device=DEV)
            ops.w4_linear(w, N, K, G, x, out, epilogue=ops.EPI_SWIGLU if swi else ops.EPI_NONE, packed2=w2)
            outs.append(out)
        torch.cuda.synchronize()
        bad = [int((outs[0] != o).sum()) for o in outs[1:]]
        rows = sorted(set((outs[0] != outs[-1]).nonzero()[:, 0].tolist()))[:8] if bad[-1] else []
        print(f"M={M} N={N}{'x2' if swi else ''} K={K}: differing elements per repeat {bad} rows {rows} nan {int(torch.isnan(outs[0].float()).sum())}")
