"""Checkpoints in and out of the decode path.

* ``convert_hf_state_dict`` — Hugging Face parameter names and layouts -> the lit-gpt names this package (and the reference)
  uses, including the per-group interleave of the fused QKV weight (reference scripts/convert_hf_checkpoint.py:18-167:
  ``copy_weights_gpt_neox`` / ``copy_weights_falcon`` / ``copy_weights_hf_llama``).  Works shard by shard: q/k/v of a layer may
  arrive in different files.
* ``lazy_load`` — a checkpoint file opened without reading it (``torch.load(mmap=True)``: tensors are backed by the file's
  pages); the reference does the same with a custom unpickler (lit_gpt/utils.py:89-225).
* ``stream_load`` — fill a (possibly quantised) model on the HIP device tensor by tensor: a dense checkpoint going into int4
  Linears is quantised round-to-nearest on the way (or copied as is when it already holds quant_weight/scales/zeros), into
  LLM.int8 Linears row-quantised by the HIP kernel, into NF4 / FP4 Linears block-quantised; at no time is more than one dense matrix resident next to the model.

The renaming rules are written as (pattern, template) pairs; ``None`` drops a tensor (rotary tables, attention masks).
"""
import re
from pathlib import Path
from typing import Dict, Iterable, List, Optional, Tuple, Union

import torch

from .config import Config

# (regex on the HF name, lit name template with \\1 = layer index | None to drop)
_L = r"(\d+)"
HF_RULES: Dict[str, List[Tuple[str, Optional[str]]]] = {
    "neox": [
        (r"gpt_neox\.embed_in\.weight", "transformer.wte.weight"),
        (rf"gpt_neox\.layers\.{_L}\.input_layernorm\.(weight|bias)", r"transformer.h.\1.norm_1.\2"),
        (rf"gpt_neox\.layers\.{_L}\.post_attention_layernorm\.(weight|bias)", r"transformer.h.\1.norm_2.\2"),
        (rf"gpt_neox\.layers\.{_L}\.attention\.query_key_value\.(weight|bias)", r"transformer.h.\1.attn.attn.\2"),
        (rf"gpt_neox\.layers\.{_L}\.attention\.dense\.(weight|bias)", r"transformer.h.\1.attn.proj.\2"),
        (rf"gpt_neox\.layers\.{_L}\.attention\.(rotary_emb\.inv_freq|bias|masked_bias)", None),
        (rf"gpt_neox\.layers\.{_L}\.mlp\.dense_h_to_4h\.(weight|bias)", r"transformer.h.\1.mlp.fc.\2"),
        (rf"gpt_neox\.layers\.{_L}\.mlp\.dense_4h_to_h\.(weight|bias)", r"transformer.h.\1.mlp.proj.\2"),
        (r"gpt_neox\.final_layer_norm\.(weight|bias)", r"transformer.ln_f.\1"),
        (r"embed_out\.weight", "lm_head.weight"),
    ],
    # Falcon checkpoints store the fused QKV already interleaved per group (the layout model.py:208-214 expects)
    "falcon-7b": [
        (r"transformer\.word_embeddings\.weight", "transformer.wte.weight"),
        (rf"transformer\.h\.{_L}\.self_attention\.query_key_value\.weight", r"transformer.h.\1.attn.attn.weight"),
        (rf"transformer\.h\.{_L}\.self_attention\.dense\.weight", r"transformer.h.\1.attn.proj.weight"),
        (rf"transformer\.h\.{_L}\.mlp\.dense_h_to_4h\.weight", r"transformer.h.\1.mlp.fc.weight"),
        (rf"transformer\.h\.{_L}\.mlp\.dense_4h_to_h\.weight", r"transformer.h.\1.mlp.proj.weight"),
        (rf"transformer\.h\.{_L}\.input_layernorm\.(weight|bias)", r"transformer.h.\1.norm_1.\2"),
        (r"transformer\.ln_f\.(weight|bias)", r"transformer.ln_f.\1"),
        (r"lm_head\.weight", "lm_head.weight"),
    ],
    "llama": [
        (r"model\.embed_tokens\.weight", "transformer.wte.weight"),
        (rf"model\.layers\.{_L}\.input_layernorm\.weight", r"transformer.h.\1.norm_1.weight"),
        (rf"model\.layers\.{_L}\.post_attention_layernorm\.weight", r"transformer.h.\1.norm_2.weight"),
        (rf"model\.layers\.{_L}\.self_attn\.o_proj\.weight", r"transformer.h.\1.attn.proj.weight"),
        (rf"model\.layers\.{_L}\.self_attn\.rotary_emb\.inv_freq", None),
        (rf"model\.layers\.{_L}\.mlp\.gate_proj\.weight", r"transformer.h.\1.mlp.fc_1.weight"),
        (rf"model\.layers\.{_L}\.mlp\.up_proj\.weight", r"transformer.h.\1.mlp.fc_2.weight"),
        (rf"model\.layers\.{_L}\.mlp\.down_proj\.weight", r"transformer.h.\1.mlp.proj.weight"),
        (r"model\.norm\.weight", "transformer.ln_f.weight"),
        (r"lm_head\.weight", "lm_head.weight"),
    ],
}
# Falcon-40B differs from -7B only in its two per-branch layer norms
HF_RULES["falcon-40b"] = [r for r in HF_RULES["falcon-7b"] if "input_layernorm" not in r[0]] + [
    (rf"transformer\.h\.{_L}\.ln_attn\.(weight|bias)", r"transformer.h.\1.norm_1.\2"),
    (rf"transformer\.h\.{_L}\.ln_mlp\.(weight|bias)", r"transformer.h.\1.norm_2.\2"),
]
_QKV_PART = re.compile(rf"model\.layers\.{_L}\.self_attn\.([qkv])_proj\.weight")


def hf_family(config: Config) -> str:
    """Which Hugging Face layout a lit-gpt config corresponds to (convert_hf_checkpoint.py:193-200)."""
    if "falcon" in config.name.lower():
        return "falcon-40b" if config.n_embd == 8192 or not config.shared_attention_norm and config.n_query_groups > 1 else "falcon-7b"
    return "llama" if config._mlp_class == "LLaMAMLP" else "neox"


def interleave_qkv(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, config: Config) -> torch.Tensor:
    """Separate q/k/v projection weights -> the fused weight with rows [q x q_per_kv, k, v] per query group
    (convert_hf_checkpoint.py:153-166; consumed by model.py:208-214)."""
    hs, per = config.head_size, config.n_head // config.n_query_groups
    groups = config.n_query_groups
    if q.shape[0] != groups * per * hs or k.shape[0] != groups * hs or v.shape[0] != groups * hs:
        raise ValueError(f"q/k/v row counts {q.shape[0]}/{k.shape[0]}/{v.shape[0]} do not match the config")
    qg, kg, vg = q.view(groups, per * hs, -1), k.view(groups, hs, -1), v.view(groups, hs, -1)
    return torch.cat([qg, kg, vg], dim=1).reshape(groups * (per + 2) * hs, -1)


class HFConverter:
    """Feed it the tensors of one or more Hugging Face shards (any order); ``state_dict`` fills up with lit-gpt names."""

    def __init__(self, config: Config, family: Optional[str] = None) -> None:
        self.config = config
        self.family = family or hf_family(config)
        if self.family not in HF_RULES:
            raise ValueError(f"unknown checkpoint family {self.family!r}")
        self.rules = [(re.compile(p), t) for p, t in HF_RULES[self.family]]
        self.state_dict: Dict[str, torch.Tensor] = {}
        self._qkv: Dict[int, Dict[str, torch.Tensor]] = {}

    def add(self, name: str, tensor: torch.Tensor) -> None:
        if self.family == "llama":
            m = _QKV_PART.fullmatch(name)
            if m:
                layer, part = int(m.group(1)), m.group(2)
                parts = self._qkv.setdefault(layer, {})
                parts[part] = tensor
                if len(parts) == 3:
                    self.state_dict[f"transformer.h.{layer}.attn.attn.weight"] = interleave_qkv(parts["q"], parts["k"], parts["v"], self.config)
                    del self._qkv[layer]
                return
        for pat, template in self.rules:
            m = pat.fullmatch(name)
            if m:
                if template is not None:
                    self.state_dict[m.expand(template)] = tensor
                return
        raise KeyError(f"no conversion rule for {name!r} ({self.family})")

    def add_shard(self, tensors: Union[Dict[str, torch.Tensor], Iterable[Tuple[str, torch.Tensor]]]) -> None:
        for name, t in (tensors.items() if hasattr(tensors, "items") else tensors):
            self.add(name, t)

    def finish(self) -> Dict[str, torch.Tensor]:
        if self._qkv:
            raise ValueError(f"incomplete q/k/v for layers {sorted(self._qkv)}")
        return self.state_dict


def convert_hf_state_dict(shards: Iterable[Dict[str, torch.Tensor]], config: Config, family: Optional[str] = None) -> Dict[str, torch.Tensor]:
    conv = HFConverter(config, family)
    for shard in shards:
        conv.add_shard(shard)
    return conv.finish()


def lazy_load(path: Union[str, Path]) -> Dict[str, torch.Tensor]:
    """Open a torch checkpoint without reading it: tensors are views of the memory-mapped file."""
    return torch.load(str(path), map_location="cpu", mmap=True, weights_only=True)


@torch.no_grad()
def stream_load(model: torch.nn.Module, state_dict: Dict[str, torch.Tensor]) -> List[str]:
    """Fill ``model`` (already on the HIP device, possibly built under ``quantization(...)``) from ``state_dict`` one tensor at
    a time.  Dense ``<linear>.weight`` entries going into int4 / int8 Linears are quantised on the device on the way in.
    Returns the checkpoint keys that were not used."""
    from .quantize.bnb import InferenceLinear8bitLt, Linear4bit
    from .quantize.gptq import ColBlockQuantizedLinear, pack_nibbles, rtn_quantize

    modules = dict(model.named_modules())
    for m in modules.values():  # (an int4 Linear whose reference-layout buffers a decode session released: resident again)
        if isinstance(m, ColBlockQuantizedLinear):
            m.restore_reference()
    own = dict(model.state_dict(keep_vars=True))
    unused = []
    for key, value in state_dict.items():
        prefix, _, leaf = key.rpartition(".")
        mod = modules.get(prefix)
        if isinstance(mod, ColBlockQuantizedLinear) and leaf == "weight":
            dev = mod.quant_weight.device
            w = value.to(device=dev, dtype=mod.scales.dtype if mod.scales.is_floating_point() else torch.bfloat16)
            q, s, z = rtn_quantize(w, mod.tile_cols)
            mod.scales.copy_(s)
            mod.zeros.copy_(z)
            mod.quant_weight.copy_(pack_nibbles(q))
            mod._packed = None
            mod.image_epoch += 1
            del w, q, s, z
        elif isinstance(mod, (InferenceLinear8bitLt, Linear4bit)) and leaf == "weight" and value.is_floating_point():
            mod._quantize_weight(value.to(mod.weight.device))
        elif key in own:
            tgt = own[key]
            if tgt.shape != value.shape:
                raise ValueError(f"{key}: checkpoint shape {tuple(value.shape)} != model shape {tuple(tgt.shape)}")
            tgt.data.copy_(value.to(device=tgt.device, dtype=tgt.dtype))
            if isinstance(mod, ColBlockQuantizedLinear):
                mod._packed = None
        else:
            unused.append(key)
    return unused


# ------------------------------------------------------------------------------------------------ test / tooling helper
def synthetic_hf_state_dict(family: str, config_kwargs: dict, seed: int = 0) -> Dict[str, torch.Tensor]:
    """A Hugging Face-named state dict of random fp16 tensors with the shapes of ``Config(**config_kwargs)``: what the
    conversion is tested with (no real checkpoints offline)."""
    c = Config(**dict(config_kwargs))
    g = torch.Generator().manual_seed(seed)

    def t(*shape):
        return (torch.randn(*shape, generator=g) * 0.02).to(torch.float16)

    d, V, I = c.n_embd, c.padded_vocab_size, c.intermediate_size
    hs, per, groups = c.head_size, c.n_head // c.n_query_groups, c.n_query_groups
    out: Dict[str, torch.Tensor] = {}
    if family == "llama":
        out["model.embed_tokens.weight"] = t(V, d)
        for i in range(c.n_layer):
            p = f"model.layers.{i}."
            out[p + "input_layernorm.weight"] = t(d)
            out[p + "self_attn.q_proj.weight"] = t(groups * per * hs, d)
            out[p + "self_attn.k_proj.weight"] = t(groups * hs, d)
            out[p + "self_attn.v_proj.weight"] = t(groups * hs, d)
            out[p + "self_attn.o_proj.weight"] = t(d, d)
            out[p + "self_attn.rotary_emb.inv_freq"] = t(hs // 2)
            out[p + "post_attention_layernorm.weight"] = t(d)
            out[p + "mlp.gate_proj.weight"] = t(I, d)
            out[p + "mlp.up_proj.weight"] = t(I, d)
            out[p + "mlp.down_proj.weight"] = t(d, I)
        out["model.norm.weight"] = t(d)
        out["lm_head.weight"] = t(V, d)
    elif family in ("falcon-7b", "falcon-40b"):
        out["transformer.word_embeddings.weight"] = t(V, d)
        for i in range(c.n_layer):
            p = f"transformer.h.{i}."
            if family == "falcon-7b":
                out[p + "input_layernorm.weight"], out[p + "input_layernorm.bias"] = t(d), t(d)
            else:
                for n in ("ln_attn", "ln_mlp"):
                    out[p + n + ".weight"], out[p + n + ".bias"] = t(d), t(d)
            out[p + "self_attention.query_key_value.weight"] = t(groups * (per + 2) * hs, d)
            out[p + "self_attention.dense.weight"] = t(d, d)
            out[p + "mlp.dense_h_to_4h.weight"] = t(I, d)
            out[p + "mlp.dense_4h_to_h.weight"] = t(d, I)
        out["transformer.ln_f.weight"], out["transformer.ln_f.bias"] = t(d), t(d)
        out["lm_head.weight"] = t(V, d)
    elif family == "neox":
        out["gpt_neox.embed_in.weight"] = t(V, d)
        for i in range(c.n_layer):
            p = f"gpt_neox.layers.{i}."
            for n in ("input_layernorm", "post_attention_layernorm"):
                out[p + n + ".weight"], out[p + n + ".bias"] = t(d), t(d)
            out[p + "attention.query_key_value.weight"], out[p + "attention.query_key_value.bias"] = t(3 * d, d), t(3 * d)
            out[p + "attention.dense.weight"], out[p + "attention.dense.bias"] = t(d, d), t(d)
            out[p + "attention.rotary_emb.inv_freq"] = t(hs // 2)
            out[p + "attention.bias"], out[p + "attention.masked_bias"] = t(1, 1, 8, 8), t(1)
            out[p + "mlp.dense_h_to_4h.weight"], out[p + "mlp.dense_h_to_4h.bias"] = t(I, d), t(I)
            out[p + "mlp.dense_4h_to_h.weight"], out[p + "mlp.dense_4h_to_h.bias"] = t(d, I), t(d)
        out["gpt_neox.final_layer_norm.weight"], out["gpt_neox.final_layer_norm.bias"] = t(d), t(d)
        out["embed_out.weight"] = t(V, d)
    else:
        raise ValueError(family)
    return out
