"""Probe of the codebook GEMV (diagnostic): constant-nibble matrices make every stage separately visible."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from lit_parrot_amd.quantize import bnb as P  # noqa: E402

BF = torch.bfloat16
torch.manual_seed(0)


def run(N, K, M, wfn, label, qt="nf4"):
    w = wfn(N, K).to(BF)
    lin = P.Linear4bit(K, N, False, quant_type=qt, compress_statistics=False)
    lin.load_state_dict({"weight": w})
    lin = lin.to("cuda").to(BF)
    x = torch.randn(M, K).to(BF)
    out = lin(x.cuda()).float().cpu()
    wd = lin.dequantized_weight().float().cpu()
    want = x.float() @ wd.t()
    print(f"{label:28s} N={N} K={K} M={M}: got[0,:4]={out[0,:4].tolist()} want[0,:4]={want[0,:4].tolist()} maxerr={float((out-want).abs().max()):.4f}", flush=True)


for M in (1, 2):
    run(64, 256, M, lambda n, k: torch.ones(n, k), "all +absmax (code 15)")
    run(64, 256, M, lambda n, k: -torch.ones(n, k), "all -absmax (code 0)")
    run(64, 256, M, lambda n, k: torch.cat([torch.ones(n, 1), torch.zeros(n, k - 1)], 1), "first weight only")
    run(64, 256, M, lambda n, k: torch.randn(n, k) * 0.02, "random nf4")
    run(64, 256, M, lambda n, k: torch.randn(n, k) * 0.02, "random fp4", "fp4")
    run(520, 1024, M, lambda n, k: torch.randn(n, k) * 0.02, "random nf4")
    run(256, 4096, M, lambda n, k: torch.randn(n, k) * 0.02, "random nf4")
