#!/usr/bin/env python3
"""Measured bf16 parity against the reference-generated golden logits (tests/golden/model_*.npz): per family and case,
max / mean |hip - golden_bf16| in bf16 ulps of the golden value, the share of bit-identical logits and the largest absolute
distance, for both RMSNorm rsqrt roundings.  Prints a markdown table and writes gpurun_out/parity_table.json."""
import json
import sys
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(REPO))
sys.path.insert(0, str(REPO / "tests"))
import lit_parrot_amd as L  # noqa: E402
from helpers import bf16_ulp_distance  # noqa: E402
from lit_parrot_amd import ops  # noqa: E402
from lit_parrot_amd.config import Config  # noqa: E402
from lit_parrot_amd.synth import synthetic_state_dict  # noqa: E402

DEV, BF = torch.device("cuda", 0), torch.bfloat16
TINY = ["tiny-neox", "tiny-llama", "tiny-llama-gqa", "tiny-llama-hs128", "tiny-falcon-gqa", "tiny-falcon-mqa"]
T_PROMPT, MAX_SEQ, WINDOW = 7, 16, 10


def cases(model, g):
    tokens = torch.from_numpy(g["tokens"])
    prompt, forced = tokens[:T_PROMPT].to(DEV), tokens[T_PROMPT:].to(DEV)
    out = {}
    with torch.no_grad():
        out["nocache"] = model(prompt.view(1, -1))[0]
        pos = torch.arange(T_PROMPT, device=DEV)
        out["prefill"] = model(prompt.view(1, -1), MAX_SEQ, pos)[0]
        rows = []
        for i in range(4):
            pos = pos[-1:] + 1
            rows.append(model(forced[i].view(1, 1), MAX_SEQ, pos)[0])
        out["decode"] = torch.cat(rows)
        model.reset_cache()
        pos = torch.arange(T_PROMPT, device=DEV)
        model(prompt.view(1, -1), WINDOW, pos)
        rows = []
        for i in range(8):
            pos = pos[-1:] + 1
            rows.append(model(forced[i].view(1, 1), WINDOW, pos)[0])
        out["window"] = torch.cat(rows)
        model.reset_cache()
    return {k: v.float().cpu() for k, v in out.items()}


def main():
    table = {}
    for name in TINY:
        g = np.load(REPO / "tests" / "golden" / f"model_{name}.npz")
        cfg = Config.from_name(name)
        sd = {k: v.to(BF) for k, v in synthetic_state_dict(cfg, 4321, perturb=True).items()}
        model = L.GPT(cfg)
        model.load_state_dict(sd)
        model = model.to(BF).to(DEV).eval()
        # mode = (RMSNorm rsqrt rounding, softmax probabilities): "0|0" the defaults, "1|1" both as the CPU-run reference computes
        modes = ((0, 0), (1, 0), (0, 1), (1, 1)) if cfg._norm_class == "RMSNorm" else ((0, 0), (0, 1))
        for rs_mode, sm_mode in modes:
            mode = f"{rs_mode}/{sm_mode}"
            ops.RMSNORM_RSQRT_MODE, ops.ATTN_SOFTMAX_MODE = rs_mode, sm_mode
            got = cases(model, g)
            for case, hip in got.items():
                ref = torch.from_numpy(g[case + "_bf16"])
                f32 = torch.from_numpy(g[case + "_f32"])
                a = (hip - ref).abs()
                ulp = lambda x: 2.0 ** (torch.floor(torch.log2(x.abs().clamp_min(2.0 ** -100))) - 7)  # noqa: E731
                ulp_ref = ulp(ref)                                   # a bf16 ulp at the golden value itself
                ulp_row = ulp(ref.abs().amax(dim=-1, keepdim=True))  # ... at the row's largest logit (the scale the sums round at)
                big = ref.abs() >= 0.125
                table[f"{name}|{mode}|{case}"] = dict(
                    identical=float((a == 0).float().mean()), max_abs=float(a.max()), mean_abs=float(a.mean()),
                    max_in_row_ulp=float((a / ulp_row).max()), mean_in_row_ulp=float((a / ulp_row).mean()),
                    k_row=float(((a - 1e-3).clamp_min(0) / ulp_row).max()),
                    k_ref_big=float(((a - 1e-3).clamp_min(0) / ulp_ref)[big].max()) if bool(big.any()) else 0.0,
                    k_ref_all=float(((a - 1e-3).clamp_min(0) / ulp_ref).max()),
                    ref_err_max=float((ref - f32).abs().max()), hip_err_max=float((hip - f32).abs().max()), max_logit=float(ref.abs().max()))
        ops.RMSNORM_RSQRT_MODE, ops.ATTN_SOFTMAX_MODE = 0, 0
    out = REPO / "gpurun_out"
    out.mkdir(exist_ok=True)
    (out / "parity_table.json").write_text(json.dumps(table, indent=1))
    print("| family | rsqrt / softmax mode | case | bit-identical | max abs | max (row ulp) | mean (row ulp) | k_row | k_ref (|ref| >= 1/8) | k_ref (all) | ref bf16-vs-fp32 max | hip-vs-fp32 max | max |logit| |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|---|")
    for k, v in table.items():
        n, m, c = k.split("|")
        print(f"| {n} | {m} | {c} | {v['identical']:.3f} | {v['max_abs']:.4f} | {v['max_in_row_ulp']:.2f} | {v['mean_in_row_ulp']:.3f} | {v['k_row']:.2f} | {v['k_ref_big']:.2f} | {v['k_ref_all']:.1f} | "
              f"{v['ref_err_max']:.4f} | {v['hip_err_max']:.4f} | {v['max_logit']:.3f} |")
    for key in ("k_row", "k_ref_big", "k_ref_all"):
        print(f"largest {key}:", max(v[key] for v in table.values()))
    # per family and mode: the worst case of the four
    print("| family | rsqrt / softmax mode | worst k_row | least bit-identical share | largest max abs |")
    print("|---|---|---|---|---|")
    fam = {}
    for k, v in table.items():
        n, m, c = k.split("|")
        a = fam.setdefault((n, m), [0.0, 1.0, 0.0])
        a[0], a[1], a[2] = max(a[0], v["k_row"]), min(a[1], v["identical"]), max(a[2], v["max_abs"])
    for (n, m), a in fam.items():
        print(f"| {n} | {m} | {a[0]:.2f} | {a[1]:.3f} | {a[2]:.4f} |")


if __name__ == "__main__":
    main()
