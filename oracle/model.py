"""CPU restatement of the reference model + decode loop (TEST INFRASTRUCTURE ONLY, see oracle/__init__.py).

Functional style over a state dict with the reference's keys; every function cites the reference lines it follows.
Dtype choreography is the reference's: everything runs in the dtype of the weights (bf16 under Fabric's bf16-true,
fp32 otherwise) with torch's CPU kernels, the RoPE tables are fp16 for 16-bit models.
"""
import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

from oracle import int4 as o_int4
from oracle import int8 as o_int8


# ------------------------------------------------------------------------------------------------ pieces
def rope_tables(seq_len: int, n_elem: int, dtype: torch.dtype, condense_ratio: int = 1, base: int = 10000,
                math_dtype: torch.dtype = torch.float32):
    """lit_gpt/model.py:304-327 build_rope_cache.

    ``math_dtype`` is the ambient torch default dtype at the first forward: the reference builds theta, the position
    index and their outer product with int/int true divisions, which come out in the DEFAULT dtype.  Under Fabric's
    ``bf16-true`` the forward runs with default dtype bf16 (generate/base.py:196,226), so the whole table — positions
    included — is computed in bf16 before the final ``.half()``; in an fp32 session it is computed in fp32.
    """
    theta = 1.0 / (base ** (torch.arange(0, n_elem, 2).to(math_dtype) / n_elem))
    seq_idx = torch.arange(seq_len).to(math_dtype) / condense_ratio
    idx_theta = torch.outer(seq_idx, theta).repeat(1, 2)
    cos, sin = torch.cos(idx_theta), torch.sin(idx_theta)
    if dtype in (torch.float16, torch.bfloat16, torch.int8):
        return cos.half(), sin.half()
    return cos, sin


def apply_rope(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    """lit_gpt/model.py:330-336: rotate-half; bf16 x fp16 promotes to fp32, result cast back."""
    n = x.size(-1)
    rotated = torch.cat((-x[..., n // 2:], x[..., : n // 2]), dim=-1)
    return ((x * cos) + (rotated * sin)).type_as(x)


def rms_norm(x: torch.Tensor, weight: torch.Tensor, eps: float) -> torch.Tensor:
    """lit_gpt/rmsnorm.py:17-21 — in x's dtype, no upcast."""
    norm_x = torch.mean(x * x, dim=-1, keepdim=True)
    return weight * (x * torch.rsqrt(norm_x + eps))


def gelu(x: torch.Tensor) -> torch.Tensor:
    return F.gelu(x)  # exact erf (lit_gpt/model.py:286)


class OracleGPT:
    """``GPT`` of the reference restated over a plain state dict.

    linear_mode: "dense" (weights as given), "gptq" (state dict holds quant_weight/scales/zeros per Linear, forward =
    get_weight + F.linear, quantize/gptq.py:243-264) or "int8" (float weights quantised like quantize/bnb.py:52-60,
    forward = LLM.int8 restatement in oracle/int8.py).
    """

    def __init__(self, config, state_dict: Dict[str, torch.Tensor], linear_mode: str = "dense", tile_cols: int = -1,
                 threshold: float = 6.0) -> None:
        self.config, self.sd, self.mode, self.tile_cols, self.threshold = config, state_dict, linear_mode, tile_cols, threshold
        self.rope_cache = None
        self.mask_cache = None
        self.kv_caches: List[Tuple[torch.Tensor, torch.Tensor]] = []
        self.dtype = state_dict["transformer.wte.weight"].dtype
        self._int8 = {}
        if linear_mode == "int8":
            for k in list(state_dict):
                if k.endswith(".weight") and (k.startswith("lm_head") or ".attn." in k or ".mlp." in k):
                    self._int8[k[: -len(".weight")]] = o_int8.quantize_weight_rows(state_dict[k])

    def reset_cache(self) -> None:
        self.kv_caches.clear()

    # -- Linear dispatch (what `quantization()` swapped in, lit_gpt/utils.py:80-83)
    def linear(self, name: str, x: torch.Tensor) -> torch.Tensor:
        bias = self.sd.get(name + ".bias")
        if self.mode == "dense":
            return F.linear(x, self.sd[name + ".weight"], bias)
        if self.mode == "gptq":
            w = o_int4.get_weight(self.sd[name + ".quant_weight"], self.sd[name + ".scales"], self.sd[name + ".zeros"],
                                  self._tile_cols(name), x.dtype)
            return F.linear(x, w, bias)
        if self.mode == "int8":
            CB, SCB = self._int8[name]
            return o_int8.linear(x, CB, SCB, bias, self.threshold)
        raise ValueError(self.mode)

    def _tile_cols(self, name: str) -> int:
        in_features = self.sd[name + ".quant_weight"].shape[1] * 2
        return in_features if self.tile_cols == -1 else self.tile_cols

    def norm(self, name: str, x: torch.Tensor) -> torch.Tensor:
        c = self.config
        if c._norm_class == "RMSNorm":
            return rms_norm(x, self.sd[name + ".weight"], c.norm_eps)
        return F.layer_norm(x, (c.n_embd,), self.sd[name + ".weight"], self.sd.get(name + ".bias"), c.norm_eps)

    # -- CausalSelfAttention.forward, lit_gpt/model.py:194-254
    def attention(self, i: int, x, cos, sin, max_seq_length, mask, input_pos, kv_cache, heads_only: bool = False):
        c = self.config
        B, T, C = x.size()
        p = f"transformer.h.{i}.attn"
        qkv = self.linear(p + ".attn", x)
        q_per_kv = c.n_head // c.n_query_groups
        qkv = qkv.view(B, T, c.n_query_groups, q_per_kv + 2, c.head_size).permute(0, 2, 3, 1, 4)  # :208-211
        q, k, v = qkv.split((q_per_kv, 1, 1), dim=2)
        if c.n_query_groups != 1:  # :217-220 (MQA keeps one head)
            k = k.repeat_interleave(q_per_kv, dim=2)
            v = v.repeat_interleave(q_per_kv, dim=2)
        q = q.reshape(B, -1, T, c.head_size)
        k = k.reshape(B, -1, T, c.head_size)
        v = v.reshape(B, -1, T, c.head_size)
        n_elem = int(c.rotary_percentage * c.head_size)
        q = torch.cat((apply_rope(q[..., :n_elem], cos, sin), q[..., n_elem:]), dim=-1)  # :226-232
        k = torch.cat((apply_rope(k[..., :n_elem], cos, sin), k[..., n_elem:]), dim=-1)
        if kv_cache is not None:  # :234-245
            cache_k, cache_v = kv_cache
            cache_k, cache_v = cache_k.to(dtype=k.dtype), cache_v.to(dtype=v.dtype)
            if input_pos[-1] >= max_seq_length:
                input_pos = torch.tensor(max_seq_length - 1)
                cache_k = torch.roll(cache_k, -1, dims=2)
                cache_v = torch.roll(cache_v, -1, dims=2)
            k = cache_k.index_copy_(2, input_pos, k)
            v = cache_v.index_copy_(2, input_pos, v)
            kv_cache = k, v
        scale = 1.0 / math.sqrt(c.head_size)
        y = F.scaled_dot_product_attention(q, k, v, attn_mask=mask, dropout_p=0.0, scale=scale, is_causal=mask is None)
        y = y.transpose(1, 2).contiguous().view(B, T, C)
        if heads_only:  # tests: the heads before the output projection
            return y
        return self.linear(p + ".proj", y), kv_cache

    def mlp(self, i: int, x: torch.Tensor) -> torch.Tensor:
        p = f"transformer.h.{i}.mlp"
        if self.config._mlp_class == "LLaMAMLP":  # :297-301
            return self.linear(p + ".proj", F.silu(self.linear(p + ".fc_1", x)) * self.linear(p + ".fc_2", x))
        return self.linear(p + ".proj", gelu(self.linear(p + ".fc", x)))  # :284-287

    # -- Block.forward, lit_gpt/model.py:158-180
    def block(self, i: int, x, rope, max_seq_length, mask=None, input_pos=None, kv_cache=None):
        c = self.config
        p = f"transformer.h.{i}"
        n_1 = self.norm(p + ".norm_1", x)
        h, new_kv = self.attention(i, n_1, rope[0], rope[1], max_seq_length, mask, input_pos, kv_cache)
        if c.parallel_residual:
            n_2 = n_1 if c.shared_attention_norm else self.norm(p + ".norm_2", x)
            x = x + h + self.mlp(i, n_2)
        else:
            x = x + h
            x = x + self.mlp(i, self.norm(p + ".norm_2", x))
        return x, new_kv

    # -- GPT.forward, lit_gpt/model.py:63-111
    def __call__(self, idx: torch.Tensor, max_seq_length: Optional[int] = None, input_pos: Optional[torch.Tensor] = None):
        c = self.config
        B, T = idx.size()
        use_kv_cache = input_pos is not None
        if max_seq_length is None:
            max_seq_length = c.block_size
        if use_kv_cache:
            assert max_seq_length >= T
        assert max_seq_length <= c.block_size and c.block_size >= T
        if self.rope_cache is None:
            self.rope_cache = rope_tables(c.block_size, int(c.rotary_percentage * c.head_size), self.dtype, c.condense_ratio,
                                          math_dtype=self.dtype)
        if use_kv_cache and self.mask_cache is None:
            ones = torch.ones((c.block_size, c.block_size), dtype=torch.bool)
            self.mask_cache = torch.tril(ones).unsqueeze(0).unsqueeze(0)
        cos, sin = self.rope_cache
        if use_kv_cache:
            cos, sin = cos.index_select(0, input_pos), sin.index_select(0, input_pos)
            mask = self.mask_cache.index_select(2, input_pos)[:, :, :, :max_seq_length]
        else:
            cos, sin, mask = cos[:T], sin[:T], None
        x = F.embedding(idx, self.sd["transformer.wte.weight"])
        if use_kv_cache and not self.kv_caches:
            heads = 1 if c.n_query_groups == 1 else c.n_head  # :132: GQA stored expanded
            shape = (B, heads, max_seq_length, c.head_size)
            self.kv_caches = [(torch.zeros(shape, dtype=self.dtype), torch.zeros(shape, dtype=self.dtype)) for _ in range(c.n_layer)]
        for i in range(c.n_layer):
            if use_kv_cache:
                x, self.kv_caches[i] = self.block(i, x, (cos, sin), max_seq_length, mask, input_pos, self.kv_caches[i])
            else:
                x, _ = self.block(i, x, (cos, sin), max_seq_length)
        x = self.norm("transformer.ln_f", x)
        return self.linear("lm_head", x)


@torch.no_grad()
def generate(model, idx: torch.Tensor, max_returned_tokens: int, max_seq_length: int, *, temperature: float = 1.0,
             top_k: Optional[int] = None, eos_id: Optional[int] = None, greedy_ties_lowest: bool = False,
             logits_log: Optional[list] = None) -> torch.Tensor:
    """generate/base.py:92-159.  ``greedy_ties_lowest`` replaces the multinomial draw by argmax (lowest index on
    ties) — what the draw does with top_k=1 whenever the maximum is unique."""
    T = idx.size(0)
    assert max_returned_tokens > T
    buf = torch.empty(max_returned_tokens, dtype=idx.dtype)
    buf[:T] = idx
    idx = buf
    input_pos = torch.arange(0, T)
    for _ in range(max_returned_tokens - T):
        x = idx.index_select(0, input_pos).view(1, -1)
        logits = model(x, max_seq_length, input_pos)
        logits = logits[0, -1] / temperature
        if logits_log is not None:
            logits_log.append(logits.clone())
        if greedy_ties_lowest:
            idx_next = torch.argmax(logits.float(), dim=-1, keepdim=True).to(idx.dtype)
        else:
            if top_k is not None:
                v, _ = torch.topk(logits, min(top_k, logits.size(-1)))
                logits = torch.where(logits < v[[-1]], -float("Inf"), logits)
            probs = F.softmax(logits, dim=-1)
            idx_next = torch.multinomial(probs, num_samples=1).to(dtype=idx.dtype)
        input_pos = input_pos[-1:] + 1
        idx = idx.index_copy(0, input_pos, idx_next)
        if idx_next == eos_id:
            return idx[:input_pos]  # as the reference: this slice ends BEFORE the eos token (its comment says otherwise)
    return idx
