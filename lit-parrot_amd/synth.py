"""Synthetic weights and prompts for tests and the benchmark (there are no checkpoints offline).

Values follow ``GPT._init_weights`` (reference lit_gpt/model.py:41-54): Linear and embedding weights ~ N(0, 0.02);
norm weights 1 and biases 0 — or, with ``perturb=True``, norm weights 1 + 0.1 N(0,1) and biases 0.02 N(0,1) so that
tests exercise the bias / affine paths.  Keys and shapes are the reference model's state-dict keys, generated in a
fixed order from one seeded generator, so the same (config, seed) gives the same tensors wherever it runs on the
same device type.
"""
from typing import Dict

import torch

from .config import Config


def synthetic_state_dict(config: Config, seed: int = 1234, *, dtype: torch.dtype = torch.float32, device="cpu",
                         perturb: bool = False) -> Dict[str, torch.Tensor]:
    gen = torch.Generator(device=device).manual_seed(seed)
    c = config

    def normal(*shape: int, std: float = 0.02) -> torch.Tensor:
        return (torch.randn(shape, generator=gen, device=device, dtype=torch.float32) * std).to(dtype)

    def norm(prefix: str, sd: Dict[str, torch.Tensor]) -> None:
        w = torch.ones(c.n_embd, device=device)
        if perturb:
            w = w + normal(c.n_embd, std=0.1).float()
        sd[prefix + ".weight"] = w.to(dtype)
        if c._norm_class == "LayerNorm":
            sd[prefix + ".bias"] = normal(c.n_embd) if perturb else torch.zeros(c.n_embd, device=device, dtype=dtype)

    def linear(prefix: str, out_f: int, in_f: int, bias: bool, sd: Dict[str, torch.Tensor]) -> None:
        sd[prefix + ".weight"] = normal(out_f, in_f)
        if bias:
            sd[prefix + ".bias"] = normal(out_f) if perturb else torch.zeros(out_f, device=device, dtype=dtype)

    sd: Dict[str, torch.Tensor] = {}
    linear("lm_head", c.padded_vocab_size, c.n_embd, False, sd)
    sd["transformer.wte.weight"] = normal(c.padded_vocab_size, c.n_embd)
    for i in range(c.n_layer):
        p = f"transformer.h.{i}"
        norm(f"{p}.norm_1", sd)
        linear(f"{p}.attn.attn", c.qkv_size, c.n_embd, c.bias, sd)
        linear(f"{p}.attn.proj", c.n_embd, c.n_embd, c.bias, sd)
        if not c.shared_attention_norm:
            norm(f"{p}.norm_2", sd)
        if c._mlp_class == "LLaMAMLP":
            linear(f"{p}.mlp.fc_1", c.intermediate_size, c.n_embd, c.bias, sd)
            linear(f"{p}.mlp.fc_2", c.intermediate_size, c.n_embd, c.bias, sd)
        else:
            linear(f"{p}.mlp.fc", c.intermediate_size, c.n_embd, c.bias, sd)
        linear(f"{p}.mlp.proj", c.n_embd, c.intermediate_size, c.bias, sd)
    norm("transformer.ln_f", sd)
    return sd


def synthetic_prompt(config: Config, T: int, seed: int = 1234, device="cpu") -> torch.Tensor:
    gen = torch.Generator(device=device).manual_seed(seed + 7919)
    return torch.randint(0, config.vocab_size, (T,), generator=gen, device=device, dtype=torch.int64)


LINEAR_SUFFIXES = ("attn.attn", "attn.proj", "mlp.fc", "mlp.fc_1", "mlp.fc_2", "mlp.proj")


def is_linear_key(key: str) -> bool:
    """True for the ``.weight`` of a module that ``quantization()`` replaces (every Linear incl. lm_head)."""
    if not key.endswith(".weight"):
        return False
    stem = key[: -len(".weight")]
    return stem == "lm_head" or stem.endswith(LINEAR_SUFFIXES)


def build_synthetic_model(config: Config, mode=None, seed: int = 1234, device="cuda", tile_cols: int = 128):
    """Random-init model of ``config`` on ``device``, built layer by layer without ever holding fp32 copies of all
    weights.  ``mode``: None (bf16), "gptq.int4[-gN]" (RTN-quantised with find_params_weight semantics and packed in
    the reference's format), or "bnb.int8".  Returns the model in eval mode, bf16, ready for ``generate``.
    """
    import re

    from .model import GPT
    from .quantize.gptq import ColBlockQuantizedLinear, pack_nibbles, rtn_quantize
    from .utils import quantization

    dev = torch.device(device)
    gen = torch.Generator(device=dev).manual_seed(seed)
    with torch.device(dev), quantization(mode):
        model = GPT(config)
    for module in model.modules():
        if isinstance(module, ColBlockQuantizedLinear):
            w = (torch.randn((module.out_features, module.in_features), generator=gen, device=dev) * 0.02).to(torch.bfloat16)
            q, s, z = rtn_quantize(w, module.tile_cols)
            module.quant_weight.copy_(pack_nibbles(q))
            module.scales = s
            module.zeros = z
            if module.bias is not None:
                module.bias.zero_()
            module._packed = None
            module.image_epoch += 1
            del w, q
        elif isinstance(module, torch.nn.Linear):
            w = torch.randn((module.out_features, module.in_features), generator=gen, device=dev) * 0.02
            if hasattr(module, "_quantize_weight"):  # LLM.int8 / NF4 / FP4: quantise on arrival, like loading a checkpoint
                module._quantize_weight(w.to(torch.bfloat16) if module.weight.dtype != torch.int8 and hasattr(module, "quant_type") else w)
            else:
                module.weight.data.copy_(w)
            if module.bias is not None:
                module.bias.data.zero_()
            del w
        elif isinstance(module, torch.nn.Embedding):
            module.weight.data.copy_(torch.randn(module.weight.shape, generator=gen, device=dev) * 0.02)
        elif hasattr(module, "weight") and module.weight is not None and module.weight.dim() == 1:  # norms
            module.weight.data.fill_(1.0)
            if getattr(module, "bias", None) is not None:
                module.bias.data.zero_()
            module.eps = config.norm_eps
    return model.to(torch.bfloat16).eval()
