"""Diagnostic: timeline of the chained single-token step (workgroup 0 of every launch stamps the 100 MHz device clock at
start / wait done / compute done / signalled).  Prints per launch, relative to the first launch of the token, in us."""
import argparse
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import lit_parrot_amd as L  # noqa: E402
from lit_parrot_amd.config import Config  # noqa: E402
from lit_parrot_amd.generate import base as gb  # noqa: E402
from lit_parrot_amd.synth import build_synthetic_model, synthetic_prompt  # noqa: E402


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="Llama-2-7b-hf")
    ap.add_argument("--mode", default="gptq.int4-g128")
    ap.add_argument("--layers", type=int, default=3, help="how many blocks to print")
    ap.add_argument("--eager", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    cfg = Config.from_name(args.config)
    model = build_synthetic_model(cfg, args.mode, seed=1, device=dev)
    prompt = synthetic_prompt(cfg, 128, seed=1, device="cpu")
    with torch.no_grad():
        sess = gb.DecodeSession(model, 512, 512, True, use_graph=not args.eager, chained=True)
        assert sess.chain is not None, gb.chain_supported(model)
        stamps = sess.chain.enable_stamps()
        logits = sess.prefill(prompt.to(dev))
        L.ops.argmax_advance(logits, sess.tokens, sess.pos)
        sess.capture()
        for _ in range(20):
            sess.step()
        torch.cuda.synchronize()
        sess.chain.check()
    s = stamps.cpu().double() / 100.0  # us
    t0 = float(s[0, 0])
    names = ["qkv", "attn", "proj", "fc", "down"]
    n = 5 * args.layers
    print(f"{'launch':>10} {'start':>8} {'waited':>8} {'computed':>9} {'signalled':>9} | wait   work")
    for i in range(n):
        a, b, c, d = (float(v) - t0 for v in s[i])
        print(f"{i:3d} {names[i % 5]:>6} {a:8.2f} {b:8.2f} {c:9.2f} {d:9.2f} | {b - a:5.2f} {c - b:6.2f}")
    last = s.shape[0] - 1
    print(f"lm_head start {float(s[last, 0]) - t0:.1f} us, end {float(s[last, 3]) - t0:.1f} us")
    per_layer = (float(s[5 * (cfg.n_layer - 1), 0]) - t0) / (cfg.n_layer - 1)
    print(f"mean per block: {per_layer:.2f} us")


if __name__ == "__main__":
    main()
