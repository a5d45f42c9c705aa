"""Measurement for the chat row: tokens/s of the streaming chat generator (device-side stop check, one
host read per CHUNK tokens) next to the plain decode loop, Llama-2-7B int4 g128, greedy, 128-token prompt."""
import json
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import lit_parrot_amd as L  # noqa: E402
from lit_parrot_amd.chat import base as chat  # noqa: E402
from lit_parrot_amd.config import Config  # noqa: E402
from lit_parrot_amd.synth import build_synthetic_model, synthetic_prompt  # noqa: E402


def main() -> None:
    dev = torch.device("cuda", 0)
    cfg = Config.from_name("Llama-2-7b-hf")
    model = build_synthetic_model(cfg, "gptq.int4-g128", seed=1234, device=dev)
    prompt = synthetic_prompt(cfg, 128, seed=1234, device="cpu").to(dev)
    n_new, total = 256, 128 + 256
    out = {}
    with torch.no_grad():
        for name, fn in (
            ("generate", lambda: L.generate(model, prompt, total, total, top_k=1)),
            ("chat.generate, 2 stop sequences (never hit)", lambda: list(chat.generate(model, prompt, total, total, top_k=1, stop_tokens=([31999, 31998], [31997, 31996, 31995])))),
            ("chat.generate, per-token host check (CHUNK=1)", None),
        ):
            if fn is None:
                chat.CHUNK = 1
                fn = lambda: list(chat.generate(model, prompt, total, total, top_k=1, stop_tokens=([31999, 31998], [31997, 31996, 31995])))  # noqa: E731
            model.reset_cache()
            fn()  # warm-up: capture
            model.reset_cache()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            out[name] = {"seconds": dt, "new_tokens_per_s_incl_prefill": n_new / dt}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
