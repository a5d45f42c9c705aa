"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE (build container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference (/root/reference, read-only) is pure Python but imports three packages that are absent here
(lightning, lightning_utilities, nltk); they are stubbed in sys.modules before the import — none of the stubbed
names is on the computed path.  Nothing of the reference travels: the outputs are small .npz files holding inputs
and expected outputs only.  Weights are NOT stored: they are regenerated from (config, seed) by
``lit_parrot_amd.synth.synthetic_state_dict`` (torch CPU generator), which is what the tests do as well.
"""
import sys
import types
from pathlib import Path

import numpy as np
import torch

REPO = Path(__file__).resolve().parents[2]
REFERENCE = Path("/root/reference")
OUT = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def import_reference():
    class RequirementCache:
        def __init__(self, req, *a, **k):
            self.req = req

        def __bool__(self):
            return self.req.startswith(("torch", "lightning"))

        def __str__(self):
            return f"stubbed {self.req}"

    _stub("lightning_utilities")
    _stub("lightning_utilities.core")
    _stub("lightning_utilities.core.imports", RequirementCache=RequirementCache)
    _stub("lightning", Fabric=object, seed_everything=lambda s: torch.manual_seed(s))
    _stub("lightning.fabric", Fabric=object)
    _stub("lightning.fabric.loggers", CSVLogger=object)
    _stub("lightning.fabric.strategies", FSDPStrategy=object)
    _stub("nltk", sent_tokenize=lambda t: [t])
    sys.path.insert(0, str(REFERENCE))
    import lit_gpt  # noqa: F401
    import generate.base as gen_base
    import quantize.gptq as gptq
    from lit_gpt import GPT, Config
    from lit_gpt.rmsnorm import RMSNorm
    from lit_gpt.model import apply_rope, build_rope_cache

    return dict(GPT=GPT, Config=Config, generate=gen_base.generate, gptq=gptq, RMSNorm=RMSNorm, apply_rope=apply_rope,
                build_rope_cache=build_rope_cache)


def f32(t):
    return t.detach().float().numpy()


TINY = ["tiny-neox", "tiny-llama", "tiny-llama-gqa", "tiny-llama-hs128", "tiny-falcon-gqa", "tiny-falcon-mqa"]
MODEL_SEED = 4321
T_PROMPT = 7
MAX_SEQ = 16
WINDOW = 10  # max_seq_length of the sliding-window case (exercises the cache roll, model.py:238-242)


def ref_model(ref, cfg_dict, sd, dtype):
    cfg = ref["Config"](**cfg_dict)
    model = ref["GPT"](cfg)
    model.load_state_dict(sd, strict=True)
    return model.to(dtype).eval()


@torch.no_grad()
def golden_models(ref):
    from lit_parrot_amd.config import name_to_config, Config
    from lit_parrot_amd.synth import synthetic_prompt, synthetic_state_dict

    for name in TINY:
        cfg_dict = dict(name_to_config[name])
        my_cfg = Config(**cfg_dict)
        sd = synthetic_state_dict(my_cfg, MODEL_SEED, perturb=True)
        tokens = synthetic_prompt(my_cfg, T_PROMPT + 8, MODEL_SEED)
        prompt, forced = tokens[:T_PROMPT], tokens[T_PROMPT:]
        out = {"tokens": tokens.numpy()}
        for dtype, tag in ((torch.bfloat16, "bf16"), (torch.float32, "f32")):
            torch.set_default_dtype(dtype)  # Fabric's precision plugin does this for "bf16-true"
            try:
                model = ref_model(ref, cfg_dict, sd, dtype)
                out[f"nocache_{tag}"] = f32(model(prompt.view(1, -1))[0])
                # KV-cached: prefill, then teacher-forced single-token steps
                model.reset_cache()
                pos = torch.arange(0, T_PROMPT)
                steps = [f32(model(prompt.view(1, -1), MAX_SEQ, pos)[0])]
                for i in range(4):
                    pos = pos[-1:] + 1
                    steps.append(f32(model(forced[i].view(1, 1), MAX_SEQ, pos)[0]))
                out[f"prefill_{tag}"] = steps[0]
                out[f"decode_{tag}"] = np.concatenate(steps[1:], axis=0)
                # sliding window: max_seq_length = WINDOW, positions run past it
                model.reset_cache()
                model.mask_cache = None
                pos = torch.arange(0, T_PROMPT)
                model(prompt.view(1, -1), WINDOW, pos)
                win = []
                for i in range(8):
                    pos = pos[-1:] + 1
                    win.append(f32(model(forced[i].view(1, 1), WINDOW, pos)[0]))
                out[f"window_{tag}"] = np.concatenate(win, axis=0)
            finally:
                torch.set_default_dtype(torch.float32)
        np.savez_compressed(OUT / f"model_{name}.npz", **out)
        print("model", name, {k: v.shape for k, v in out.items()})


@torch.no_grad()
def golden_gptq(ref):
    gptq = ref["gptq"]
    gen = torch.Generator().manual_seed(99)
    out = {}
    N, K = 48, 256
    for tile_cols in (-1, 128, 64):
        for dtype, tag in ((torch.float32, "f32"), (torch.bfloat16, "bf16")):
            torch.set_default_dtype(dtype)
            try:
                lin = gptq.ColBlockQuantizedLinear(K, N, True, bits=4, tile_cols=tile_cols)
            finally:
                torch.set_default_dtype(torch.float32)
            ng = lin.scales.shape[1]
            # power-of-two scales: (q - z) * s / s + z is exact, so the reference's truncating pack is exact too
            scales = torch.pow(2.0, -torch.randint(3, 8, (N, ng), generator=gen).float())
            zeros = torch.randint(0, 16, (N, ng), generator=gen).float()
            q = torch.randint(0, 16, (N, K), generator=gen)
            tc = lin.tile_cols
            g = torch.arange(K) // tc
            weight = (q.float() - zeros[:, g]) * scales[:, g]
            lin.scales.copy_(scales)
            lin.zeros.copy_(zeros)
            lin.bias.copy_(torch.randn(N, generator=gen) * 0.1)
            lin.pack_weight(weight)
            x = (torch.randn(3, K, generator=gen)).to(dtype)
            key = f"g{tile_cols}_{tag}"
            out[key + "_q"] = q.numpy().astype(np.uint8)
            out[key + "_scales"] = f32(lin.scales)
            out[key + "_zeros"] = f32(lin.zeros)
            out[key + "_bias"] = f32(lin.bias)
            out[key + "_x"] = f32(x)
            # bytes in MEMORY order [K/2][N]
            out[key + "_qw_mem"] = lin.quant_weight.t().contiguous().numpy()
            assert lin.quant_weight.stride() == (1, N)
            out[key + "_get_weight"] = f32(lin.get_weight(dtype))
            out[key + "_forward"] = f32(lin(x))
    # find_params_weight (gptq.py:317-347)
    lin = torch.nn.Linear(128, 16)
    qz = gptq.GPTQQuantizer(lin, bits=4)
    blk = torch.randn(16, 128, generator=gen) * 0.02
    blk[3] = 0.0  # the all-zero row special case
    blk[5] = blk[5].abs()  # min clamps to 0
    s, z = qz.find_params_weight(blk)
    out["fp_x"], out["fp_scale"], out["fp_zero"] = f32(blk), f32(s), f32(z)
    out["fp_quantized"] = f32(gptq.GPTQQuantizer.quantize_weight(blk, s, z, 15))
    np.savez_compressed(OUT / "gptq_linear.npz", **out)
    print("gptq", len(out), "arrays")


@torch.no_grad()
def golden_pieces(ref):
    gen = torch.Generator().manual_seed(7)
    out = {}
    x = torch.randn(3, 128, generator=gen).to(torch.bfloat16)
    norm = ref["RMSNorm"](128, eps=1e-5)
    norm.weight.data = (1 + 0.1 * torch.randn(128, generator=gen))
    norm = norm.to(torch.bfloat16)
    out["rms_x"], out["rms_w"], out["rms_out"] = f32(x), f32(norm.weight), f32(norm(x))
    out["rms_out_f32"] = f32(ref["RMSNorm"](128, eps=1e-5)(x.float()))
    for n_elem in (8, 64):
        cos, sin = ref["build_rope_cache"](128, n_elem, torch.bfloat16, torch.device("cpu"))
        assert cos.dtype == torch.float16
        out[f"rope_cos_{n_elem}"], out[f"rope_sin_{n_elem}"] = f32(cos), f32(sin)
        xr = torch.randn(1, 2, 5, n_elem, generator=gen).to(torch.bfloat16)
        pos = torch.tensor([0, 1, 17, 100, 127])
        out[f"rope_x_{n_elem}"] = f32(xr)
        out[f"rope_out_{n_elem}"] = f32(ref["apply_rope"](xr, cos.index_select(0, pos), sin.index_select(0, pos)))
    out["rope_pos"] = np.array([0, 1, 17, 100, 127])
    np.savez_compressed(OUT / "pieces.npz", **out)
    print("pieces", len(out), "arrays")


@torch.no_grad()
def golden_generate(ref):
    """BASELINE.json configs[0]: Pythia-160M random-init, generate/base.py greedy 128 -> 64 tokens on the CPU (fp32)."""
    from lit_parrot_amd.config import name_to_config, Config
    from lit_parrot_amd.synth import synthetic_prompt, synthetic_state_dict

    cfg_dict = dict(name_to_config["pythia-160m"])
    my_cfg = Config(**cfg_dict)
    sd = synthetic_state_dict(my_cfg, 1234)
    model = ref_model(ref, cfg_dict, sd, torch.float32)
    prompt = synthetic_prompt(my_cfg, 128, 1234)
    torch.manual_seed(1234)
    y = ref["generate"](model, prompt, 192, max_seq_length=192, temperature=1.0, top_k=1)
    out = {"prompt": prompt.numpy(), "tokens": y.numpy()}
    # eos behaviour: stop at the 5th generated token
    model.reset_cache()
    eos = int(y[128 + 4])
    first = int((y[128:] == eos).nonzero()[0])
    torch.manual_seed(1234)
    y2 = ref["generate"](model, prompt, 192, max_seq_length=192, temperature=1.0, top_k=1, eos_id=eos)
    out["eos_id"], out["eos_tokens"], out["eos_first_index"] = np.array(eos), y2.numpy(), np.array(first)
    # sampled path (top_k=5, temperature 0.8) for a tiny model: same torch seed -> same draws
    cfg_dict = dict(name_to_config["tiny-llama"])
    tiny = Config(**cfg_dict)
    sd = synthetic_state_dict(tiny, MODEL_SEED, perturb=True)
    model = ref_model(ref, cfg_dict, sd, torch.float32)
    p = synthetic_prompt(tiny, 6, MODEL_SEED)
    torch.manual_seed(1234)
    out["sampled_tiny_llama"] = ref["generate"](model, p, 24, max_seq_length=24, temperature=0.8, top_k=5).numpy()
    np.savez_compressed(OUT / "generate.npz", **out)
    print("generate", y[128:].tolist()[:8], "... eos stops at", len(y2), "first idx", first)


@torch.no_grad()
def golden_generate_bf16(ref):
    """north_star: "token-for-token greedy match at bf16".  The reference's OWN bf16 greedy run of Pythia-160M (the model in
    bfloat16 under set_default_dtype(bfloat16), as Fabric's bf16-true runs it: fp16 RoPE tables, bf16 KV caches): the 64
    tokens, and per step the two largest logits of the row the token was drawn from (values and indices) - the margin that
    says whether another correct bf16 pipeline may legitimately pick a different token there."""
    from lit_parrot_amd.config import name_to_config, Config
    from lit_parrot_amd.synth import synthetic_prompt, synthetic_state_dict

    cfg_dict = dict(name_to_config["pythia-160m"])
    my_cfg = Config(**cfg_dict)
    sd = synthetic_state_dict(my_cfg, 1234)
    prompt = synthetic_prompt(my_cfg, 128, 1234)
    torch.set_default_dtype(torch.bfloat16)
    try:
        model = ref_model(ref, cfg_dict, sd, torch.bfloat16)
        rows = []
        inner = model.forward

        def recording_forward(*a, **k):
            out = inner(*a, **k)
            rows.append(out[0, -1].float().clone())
            return out

        model.forward = recording_forward
        torch.manual_seed(1234)
        y = ref["generate"](model, prompt, 192, max_seq_length=192, temperature=1.0, top_k=1)
    finally:
        torch.set_default_dtype(torch.float32)
    assert len(rows) == 64 and y.shape == (192,)
    top = [r.topk(2) for r in rows]
    out = {"prompt": prompt.numpy(), "tokens": y.numpy(),
           "top2_values": np.stack([t.values.numpy() for t in top]).astype(np.float32),
           "top2_indices": np.stack([t.indices.numpy() for t in top]).astype(np.int64)}
    np.savez_compressed(OUT / "generate_bf16.npz", **out)
    m = out["top2_values"][:, 0] - out["top2_values"][:, 1]
    print("generate bf16", y[128:136].tolist(), "... steps with a zero top-2 margin:", int((m == 0).sum()))


@torch.no_grad()
def golden_chat(ref):
    """chat/base.py::generate (the streaming generator with multi-token stop sequences) on a tiny fp32 model, greedy.
    Stored: for every case the stop sequences and the list of yielded items (each item flattened to a list of ints)."""
    import chat.base as chat_base
    from lit_parrot_amd.config import name_to_config, Config
    from lit_parrot_amd.synth import synthetic_prompt, synthetic_state_dict

    cfg_dict = dict(name_to_config["tiny-llama"])
    tiny = Config(**cfg_dict)
    sd = synthetic_state_dict(tiny, MODEL_SEED, perturb=True)
    model = ref_model(ref, cfg_dict, sd, torch.float32)
    p = synthetic_prompt(tiny, 6, MODEL_SEED)
    MAXR = 40

    def run(stop):
        model.reset_cache()
        torch.manual_seed(1234)
        items = []
        for y in chat_base.generate(model, p, MAXR, max_seq_length=MAXR, temperature=1.0, top_k=1, stop_tokens=stop):
            items.append([int(v) for v in y.reshape(-1).tolist()])
        return items

    free = [it[0] for it in run(())]  # no stop sequence: one token per yield
    assert len(free) == MAXR - 6
    cases = {
        "none": (),
        "single": ([free[9]],),                                        # the 10th generated token, as a 1-token stop
        "pair": ([free[12], free[13]],),                               # a 2-token stop sequence
        "pair_and_long": ([free[20], free[21]], [1, 2, 3, 4]),         # buffer of 4, the 2-token one hits: leftovers are yielded
        "early": ([free[0], free[1]], [free[5], free[6], free[7]]),    # hit while the buffer is still filling
        "never": ([499, 498, 497],),                                   # never matches: the tail of the buffer is never yielded
    }
    out = {"prompt": p.numpy(), "max_returned": np.array(MAXR), "free": np.array(free)}
    for name, stop in cases.items():
        items = run(stop)
        out[f"{name}_stop_flat"] = np.array([t for sq in stop for t in sq] or [-1])
        out[f"{name}_stop_lens"] = np.array([len(sq) for sq in stop] or [0])
        out[f"{name}_items_flat"] = np.array([t for it in items for t in it] or [-1])
        out[f"{name}_items_lens"] = np.array([len(it) for it in items] or [0])
        print("chat", name, "yields", len(items), "items", [len(it) for it in items if len(it) != 1])
    np.savez_compressed(OUT / "chat.npz", **out)


@torch.no_grad()
def golden_gptq_quantizer(ref):
    """quantize/gptq.py::GPTQQuantizer (:267-444) on a small fp32 Linear with Hessians from random calibration rows:
    per-channel (groupsize -1, the only mode of the reference that runs), with and without the activation-order trick."""
    gptq = ref["gptq"]
    gen = torch.Generator().manual_seed(321)
    N, K = 48, 256
    out = {}
    W = torch.randn(N, K, generator=gen) * 0.02
    W[:, 7] *= 6.0   # a few heavy columns so that the activation order differs from the natural one
    bias = torch.randn(N, generator=gen) * 0.1
    X = torch.randn(3, 1, 40, K, generator=gen)
    X[..., 11] *= 5.0
    X[..., 200] = 0.0  # a dead input column (H diagonal 0: gptq.py:378-380)
    out["W"], out["bias"], out["X"] = f32(W), f32(bias), f32(X)
    for actorder in (False, True):
        lin = torch.nn.Linear(K, N, bias=True)
        lin.weight.copy_(W)
        lin.bias.copy_(bias)
        qz = gptq.GPTQQuantizer(lin, bits=4, groupsize=-1, actorder=actorder)
        for b in range(X.shape[0]):
            qz.collect_input_stats(None, (X[b],), None)
        if not actorder:
            out["H"] = f32(qz.H.clone())  # (quantize() goes on to modify H in place)
        qmod, err = qz.quantize()
        tag = f"act{int(actorder)}"
        out[f"{tag}_scales"], out[f"{tag}_zeros"] = f32(qmod.scales), f32(qmod.zeros)
        out[f"{tag}_weight"] = f32(qmod.get_weight(torch.float32))
        out[f"{tag}_qw_mem"] = qmod.quant_weight.t().contiguous().numpy()
        out[f"{tag}_error"] = np.array(err)
        print("gptq quantizer", tag, "error", err)
    np.savez_compressed(OUT / "gptq_quantizer.npz", **out)


@torch.no_grad()
def golden_convert(ref):
    """scripts/convert_hf_checkpoint.py: the three copy functions on synthetic HF-named state dicts of tiny shapes (regenerated
    from a seed in the tests).  The conversion only renames tensors and permutes rows, so the expected lit-side state dict is
    stored as {family|lit name: sha256 of the fp16 bytes} plus {family|lit name|shape}: exact, and a few KB."""
    import hashlib
    import importlib.util

    spec = importlib.util.spec_from_file_location("ref_convert", str(REFERENCE / "scripts" / "convert_hf_checkpoint.py"))
    conv = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(conv)
    from lit_parrot_amd.checkpoint import synthetic_hf_state_dict
    from lit_parrot_amd.config import name_to_config

    out = {}
    for family, cfg_name in (("llama", "tiny-llama-gqa"), ("falcon-7b", "tiny-falcon-mqa"), ("falcon-40b", "tiny-falcon-gqa"), ("neox", "tiny-neox")):
        cfg = ref["Config"](**dict(name_to_config[cfg_name]))
        hf = synthetic_hf_state_dict(family, name_to_config[cfg_name], seed=77)
        sd = {}
        if family == "llama":
            # two shards, q/k/v of layer 1 split across them (convert_hf_checkpoint.py keeps the holder between files)
            names = list(hf)
            cut = next(i for i, n in enumerate(names) if n.startswith("model.layers.1.self_attn.k_proj"))
            holder = {}
            conv.copy_weights_hf_llama(cfg, holder, sd, {n: hf[n] for n in names[:cut]})
            conv.copy_weights_hf_llama(cfg, holder, sd, {n: hf[n] for n in names[cut:]})
            assert not holder
        elif family.startswith("falcon"):
            conv.copy_weights_falcon("40b" if family == "falcon-40b" else "7b", sd, hf)
        else:
            conv.copy_weights_gpt_neox(sd, hf)
        for k, v in sd.items():
            a = np.ascontiguousarray(v.to(torch.float16).numpy())
            out[f"{family}|{k}"] = np.frombuffer(hashlib.sha256(a.tobytes()).digest(), dtype=np.uint8)
            out[f"{family}|{k}|shape"] = np.asarray(a.shape, dtype=np.int64)
        print("convert", family, len(hf), "->", len(sd), "tensors")
    np.savez_compressed(OUT / "convert_hf.npz", **out)


if __name__ == "__main__":
    ref = import_reference()
    if "--convert-only" in sys.argv:
        golden_convert(ref)
        sys.exit(0)
    if "--gptq-quantizer-only" in sys.argv:
        golden_gptq_quantizer(ref)
        sys.exit(0)
    if "--chat-only" in sys.argv:
        golden_chat(ref)
        sys.exit(0)
    if "--generate-bf16-only" in sys.argv:
        golden_generate_bf16(ref)
        sys.exit(0)
    golden_pieces(ref)
    golden_gptq(ref)
    golden_models(ref)
    golden_generate(ref)
    golden_generate_bf16(ref)
    golden_chat(ref)
    golden_gptq_quantizer(ref)
    golden_convert(ref)
