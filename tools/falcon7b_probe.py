import sys, torch, time
sys.path.insert(0, "/root/repo")
import lit_parrot_amd as L
from lit_parrot_amd.config import Config
from lit_parrot_amd.synth import build_synthetic_model, synthetic_prompt
cfg = Config.from_name("falcon-7b")
dev = torch.device("cuda", 0)
for mode in (None, "bnb.int8", "gptq.int4", "gptq.int4-g128"):
    try:
        model = build_synthetic_model(cfg, mode, seed=1234, device=dev)
        prompt = synthetic_prompt(cfg, 64, 1).to(dev)
        with torch.no_grad():
            y = L.generate(model, prompt, 64 + 33, 64 + 33, top_k=1)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            model.reset_cache()
            y = L.generate(model, prompt, 64 + 129, 64 + 129, top_k=1)
            torch.cuda.synchronize(); el = time.perf_counter() - t0
        print(mode, "ok", y[64:72].tolist(), f"{128 / el:.1f} tok/s incl. prefill")
        del model
        torch.cuda.empty_cache()
    except Exception as e:
        print(mode, "FAILED:", type(e).__name__, str(e)[:300])
