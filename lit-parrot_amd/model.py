"""GPT decoder behind the reference's model surface, executed by the HIP kernels.

Interface (reference lit_gpt/model.py): ``GPT(config)`` with ``forward(idx, max_seq_length=None, input_pos=None)``,
``reset_cache``, ``build_rope_cache``, ``build_mask_cache``, ``build_kv_caches``, ``from_name`` and the attributes
``config``, ``rope_cache``, ``mask_cache``, ``kv_caches`` (:23-144); ``Block.forward(x, rope, max_seq_length, mask,
input_pos, kv_cache)`` (:158-180); ``CausalSelfAttention`` (:183-275); ``GptNeoxMLP`` / ``LLaMAMLP`` (:278-301);
``build_rope_cache`` / ``apply_rope`` (:304-336).  Module and parameter names are the reference's, so its
checkpoints (``lit_model.pth``, ``lit_model_gptq.4bit.pth``) load by key.

Execution is different by design: a token row never goes through ~30 small torch ops per layer.  Every module
contributes its parameters to a fixed sequence of fused kernels per block (``_block_rows``):

    [norm_1 + QKV linear] -> [split + RoPE + KV append + attention + split combine] -> [proj linear + residual]
        -> [norm_2 + MLP up linear(s) + GELU | SwiGLU] -> [MLP down linear + residual]          (5 launches / layer)

All intermediates live in a per-row-count ``Workspace`` that is allocated once, the position is a device scalar and
no step reads anything back to the host, so one decode step is a static launch sequence that ``generate`` captures
in a hipGraph.  There is no CPU or torch fallback: tensors must be bf16 on a HIP device.

Deliberate differences from the reference (documented in DESIGN.md): the KV cache is GQA-native
``(B, n_query_groups, S, head_size)`` (the reference expands GQA to n_head, :132); when the position passes
``max_seq_length`` the cache is used as a ring (slot = pos % S) instead of being rolled left (:238-242) — the same
window of keys, in a different slot order; ``input_pos`` must be consecutive positions.
"""
import math
import os
from typing import Any, List, Optional, Tuple

import torch
import torch.nn as nn

from . import ops
from ._hip import EPI_GELU, EPI_NONE, EPI_RESIDUAL, EPI_SWIGLU, ParrotHipError
from .config import Config
from .rmsnorm import RMSNorm

# parallel-residual blocks: overlap the MLP up-projection with the attention branch (M = 1); PARROT_PARALLEL_BRANCHES=0 disables
PARALLEL_BRANCHES = os.environ.get("PARROT_PARALLEL_BRANCHES", "1") != "0"
ATTN_PREFILL_MFMA = os.environ.get("PARROT_ATTN_PREFILL_MFMA", "1") != "0"  # prompts from position 0: parrot_attn_prefill


RoPECache = Tuple[torch.Tensor, torch.Tensor]
KVCache = Tuple[torch.Tensor, torch.Tensor]


def norm_class(config: Config):
    return RMSNorm if config._norm_class == "RMSNorm" else nn.LayerNorm


class Workspace:
    """Every intermediate of ``M`` token rows through the network (allocated once per M)."""

    def __init__(self, config: Config, M: int, device, lm_rows: int) -> None:
        def buf(cols: int, rows: int = M) -> torch.Tensor:
            return torch.empty((rows, cols), dtype=torch.bfloat16, device=device)

        c = config
        self.M, self.lm_rows = M, lm_rows
        self.x, self.t = buf(c.n_embd), buf(c.n_embd)
        self.qkv = buf(c.qkv_size)
        self.q = buf(c.n_head * c.head_size)
        self.y = buf(c.n_embd)
        self.h = buf(c.intermediate_size)
        self.logits = buf(c.padded_vocab_size, lm_rows)
        self.zero_pos = torch.zeros((1,), dtype=torch.int32, device=device)
        self.side_stream = None  # parallel-residual blocks: the MLP's first Linear runs beside the attention branch
        self.tickets = torch.zeros((c.n_head,), dtype=torch.int32, device=device)  # fused attention arrival counters
        self._attn_ws = {}

    def attn_ws(self, config: Config, nsplit: int) -> torch.Tensor:
        if nsplit not in self._attn_ws:
            self._attn_ws[nsplit] = ops.attn_workspace(self.M, config.n_head, config.head_size, nsplit, self.x.device)
        return self._attn_ws[nsplit]


def _fused_norm(mod: Optional[nn.Module]) -> Optional[ops.Norm]:
    """Describe a norm module so that the following Linear applies it to its input rows on the fly."""
    if mod is None:
        return None
    if isinstance(mod, RMSNorm):
        return ops.Norm(1, mod.weight.data, None, mod.eps)
    if isinstance(mod, nn.LayerNorm):
        return ops.Norm(2, mod.weight.data, None if mod.bias is None else mod.bias.data, mod.eps)
    raise ParrotHipError(f"no HIP kernel for norm class {type(mod).__name__}")


# Calibration (quantize/gptq.py::blockwise_quantization): id(Linear) -> callable(rows) that receives the rows the Linear is
# applied to.  The fused pipeline never materialises a normalised activation, so when a Linear is observed its input norm is
# run as a stand-alone kernel first (the reference hooks ``module.forward`` instead, quantize/gptq.py:500).
LINEAR_OBSERVERS: dict = {}


def _linear(mod: nn.Module, x: torch.Tensor, out: torch.Tensor, *, epilogue: int = EPI_NONE, residual=None,
            partner: Optional[nn.Module] = None, norm: Optional[nn.Module] = None) -> torch.Tensor:
    """Run one Linear-shaped module on rows with a fused epilogue — and optionally the norm module in front of it
    fused as a prologue — whatever class ``quantization()`` installed."""
    if LINEAR_OBSERVERS and (id(mod) in LINEAR_OBSERVERS or (partner is not None and id(partner) in LINEAR_OBSERVERS)):
        if norm is not None:
            x = _norm(norm, x, torch.empty_like(x))
            norm = None
        for m in (mod, partner):
            if m is not None and id(m) in LINEAR_OBSERVERS:
                LINEAR_OBSERVERS[id(m)](x)
    norm = _fused_norm(norm)
    if hasattr(mod, "hip_linear"):
        return mod.hip_linear(x, out, epilogue=epilogue, residual=residual, partner=partner, norm=norm)
    if isinstance(mod, nn.Linear):
        if partner is not None and not (isinstance(partner, nn.Linear) and not hasattr(partner, "hip_linear")):
            raise ParrotHipError("SwiGLU partner must be the same Linear class")
        return ops.bf16_linear(mod.weight.data, x, out, bias=None if mod.bias is None else mod.bias.data,
                               epilogue=epilogue, residual=residual,
                               weight2=None if partner is None else partner.weight.data, norm=norm)
    raise ParrotHipError(f"no HIP kernel for Linear class {type(mod).__name__}")


def _norm(mod: nn.Module, x: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    if isinstance(mod, RMSNorm):
        return ops.rmsnorm(x, mod.weight.data, mod.eps, out)
    if isinstance(mod, nn.LayerNorm):
        return ops.layernorm(x, mod.weight.data, None if mod.bias is None else mod.bias.data, mod.eps, out)
    raise ParrotHipError(f"no HIP kernel for norm class {type(mod).__name__}")


class GPT(nn.Module):
    def __init__(self, config: Config) -> None:
        super().__init__()
        assert config.padded_vocab_size is not None
        self.config = config
        # nn.Linear is looked up at call time: inside `with quantization(mode):` it is the quantized class
        self.lm_head = nn.Linear(config.n_embd, config.padded_vocab_size, bias=False)
        self.transformer = nn.ModuleDict(
            dict(
                wte=nn.Embedding(config.padded_vocab_size, config.n_embd),
                h=nn.ModuleList(Block(config) for _ in range(config.n_layer)),
                ln_f=norm_class(config)(config.n_embd, eps=config.norm_eps),
            )
        )
        self.rope_cache: Optional[RoPECache] = None
        self.mask_cache: Optional[torch.Tensor] = None
        self.kv_caches: List[KVCache] = []
        self._workspaces = {}

    def _init_weights(self, module: nn.Module) -> None:
        """N(0, 0.02) Linears and embeddings, unit norms (reference model.py:41-54)."""
        if isinstance(module, (nn.Linear, nn.Embedding)) and module.weight.is_floating_point():
            nn.init.normal_(module.weight, mean=0.0, std=0.02)
            if getattr(module, "bias", None) is not None:
                nn.init.zeros_(module.bias)
        elif isinstance(module, (nn.LayerNorm, RMSNorm)):
            nn.init.ones_(module.weight)
            if getattr(module, "bias", None) is not None:
                nn.init.zeros_(module.bias)
            module.eps = self.config.norm_eps

    @classmethod
    def from_name(cls, name: str, **kwargs: Any) -> "GPT":
        return cls(Config.from_name(name, **kwargs))

    def reset_cache(self) -> None:
        self.kv_caches.clear()

    # ------------------------------------------------------------------------------------------ caches
    def build_rope_cache(self, idx: torch.Tensor) -> RoPECache:
        return build_rope_cache(
            seq_len=self.config.block_size,
            n_elem=self.config.rope_n_elem,
            dtype=self.transformer.wte.weight.dtype,  # bf16 model -> fp16 tables, as under Fabric's bf16-true
            device=idx.device,
            condense_ratio=self.config.condense_ratio,
        )

    def build_mask_cache(self, idx: torch.Tensor) -> torch.Tensor:
        """Kept for interface parity (model.py:126-128); the kernels derive the causal limit from the position."""
        n = self.config.block_size
        return torch.tril(torch.ones((n, n), device=idx.device, dtype=torch.bool)).unsqueeze(0).unsqueeze(0)

    def build_kv_caches(self, idx: torch.Tensor, max_seq_length: int, rope_cache_length: int) -> List[KVCache]:
        c = self.config
        shape = (idx.size(0), c.n_query_groups, max_seq_length, c.head_size)
        return [
            (torch.zeros(shape, device=idx.device, dtype=torch.bfloat16), torch.zeros(shape, device=idx.device, dtype=torch.bfloat16))
            for _ in range(c.n_layer)
        ]

    def workspace(self, M: int, device, lm_rows: Optional[int] = None) -> Workspace:
        lm_rows = M if lm_rows is None else lm_rows
        key = (M, lm_rows, str(device))
        if key not in self._workspaces:
            if len(self._workspaces) > 8:
                self._workspaces.clear()
            self._workspaces[key] = Workspace(self.config, M, device, lm_rows)
        return self._workspaces[key]

    # ------------------------------------------------------------------------------------------ execution
    def run_rows(self, ws: Workspace, tokens: torch.Tensor, tok_pos: Optional[torch.Tensor], pos: torch.Tensor,
                 S: int, caches: List[KVCache], rope: RoPECache, *, rope_local: bool = False) -> torch.Tensor:
        """Embed ``ws.M`` tokens (``tokens[tok_pos + m]``), run every block, final norm and lm_head.

        ``pos`` (device int32[1]) is the position of row 0; ``caches[i]`` are (n_groups, S, hs) views.  Returns
        ``ws.logits``: all rows, or only the last row when the workspace was built with ``lm_rows == 1``.
        """
        M = ws.M
        ops.embedding(self.transformer.wte.weight.data, tokens, tok_pos, M, ws.x)
        nsplit = ops.attn_nsplit(self.config.n_query_groups, S, self.config.q_per_kv, M)
        for block, (kc, vc) in zip(self.transformer.h, caches):
            block.run_rows(ws, pos, S, kc, vc, rope, nsplit, rope_local)
        last = ws.x if ws.lm_rows == M else ws.x[M - 1:M]
        logits = _linear(self.lm_head, last, ws.logits, norm=self.transformer.ln_f)  # ln_f fused into lm_head
        return logits

    def forward(self, idx: torch.Tensor, max_seq_length: Optional[int] = None,
                input_pos: Optional[torch.Tensor] = None) -> torch.Tensor:
        B, T = idx.size()
        use_kv_cache = input_pos is not None
        block_size = self.config.block_size
        if max_seq_length is None:
            max_seq_length = block_size
        if use_kv_cache:
            assert max_seq_length >= T, f"Cannot forward sequence of length {T}, max seq length is only {max_seq_length}"
        assert max_seq_length <= block_size, f"Cannot attend to {max_seq_length}, block size is only {block_size}"
        assert block_size >= T, f"Cannot forward sequence of length {T}, block size is only {block_size}"
        if not idx.is_cuda:
            raise ParrotHipError("GPT.forward runs on the HIP device only (no CPU fallback); move the model and idx to cuda")

        if self.rope_cache is None:
            self.rope_cache = self.build_rope_cache(idx)
        if use_kv_cache:
            self.kv_caches = self.kv_caches or self.build_kv_caches(idx, max_seq_length, self.rope_cache[0].size(-1))
            S = self.kv_caches[0][0].size(2)
            if input_pos.numel() != T:
                raise ParrotHipError("input_pos must hold one position per token")
            if not input_pos.is_cuda and int(input_pos[-1]) >= block_size:  # a host value: check it (the reference raises in index_select)
                raise IndexError(f"input_pos {int(input_pos[-1])} is past the RoPE tables (block_size {block_size})")
            pos = input_pos[:1].to(torch.int32)
        else:
            S = T
            pos = None
        ws = self.workspace(T, idx.device)
        idx = idx.to(torch.int64).contiguous()
        out = torch.empty((B, T, self.config.padded_vocab_size), dtype=torch.bfloat16, device=idx.device)
        for b in range(B):
            if use_kv_cache:
                caches = [(k[b], v[b]) for k, v in self.kv_caches]
            else:
                c = self.config
                caches = [tuple(torch.empty((c.n_query_groups, S, c.head_size), dtype=torch.bfloat16, device=idx.device)
                                for _ in range(2))] * 1
                caches = caches * c.n_layer  # the same scratch pair serves every layer in turn
            ws.pos_is_zero = not use_kv_cache  # without a cache the rows are positions 0 .. T-1; with one, input_pos lives on the device
            logits = self.run_rows(ws, idx[b], None, pos if pos is not None else ws.zero_pos, S, caches, self.rope_cache)
            ws.pos_is_zero = False
            out[b].copy_(logits)
        return out


class Block(nn.Module):
    def __init__(self, config: Config) -> None:
        super().__init__()
        self.norm_1 = norm_class(config)(config.n_embd, eps=config.norm_eps)
        self.attn = CausalSelfAttention(config)
        if not config.shared_attention_norm:
            self.norm_2 = norm_class(config)(config.n_embd, eps=config.norm_eps)
        self.mlp = LLaMAMLP(config) if config._mlp_class == "LLaMAMLP" else GptNeoxMLP(config)
        self.config = config

    def run_rows(self, ws: Workspace, pos: torch.Tensor, S: int, k_cache: torch.Tensor, v_cache: torch.Tensor,
                 rope: RoPECache, nsplit: int, rope_local: bool = False) -> None:
        """One block over ``ws.x`` in place (reference Block.forward, model.py:158-180)."""
        c = self.config
        # norm_1 is fused into the QKV linear, norm_2 into the MLP's first linear: no normalised copy is materialised
        if c.parallel_residual:
            # x + h + mlp(n_2): the first sum is rounded to bf16 before the second, as in the reference (:171).
            # The two branches only meet at that sum, so for a single token the MLP's up-projection is enqueued on a side
            # stream next to [QKV -> attention -> proj]: inside the captured graph they become parallel nodes and the
            # weight stream of one covers the dispatch / latency gaps of the other.
            mlp_norm = self.norm_1 if c.shared_attention_norm else self.norm_2
            if PARALLEL_BRANCHES and ws.M == 1:
                if ws.side_stream is None:
                    ws.side_stream = torch.cuda.Stream(device=ws.x.device)
                main = torch.cuda.current_stream(ws.x.device)
                ws.side_stream.wait_stream(main)  # x of this block is final
                with torch.cuda.stream(ws.side_stream):
                    self.mlp.run_up(ws, ws.x, norm=mlp_norm)  # -> ws.h
                self.attn.run_rows(ws, ws.x, pos, S, k_cache, v_cache, rope, nsplit, rope_local, norm=self.norm_1)
                _linear(self.attn.proj, ws.y, ws.t, epilogue=EPI_RESIDUAL, residual=ws.x)
                main.wait_stream(ws.side_stream)
                self.mlp.run_down(ws, residual=ws.t, out=ws.x)
                return
            self.attn.run_rows(ws, ws.x, pos, S, k_cache, v_cache, rope, nsplit, rope_local, norm=self.norm_1)  # -> ws.y
            _linear(self.attn.proj, ws.y, ws.t, epilogue=EPI_RESIDUAL, residual=ws.x)
            self.mlp.run_rows(ws, ws.x, residual=ws.t, out=ws.x, norm=mlp_norm)
        else:
            if c.shared_attention_norm:
                raise NotImplementedError(
                    "No checkpoint amongst the ones we support uses this configuration"
                    " (non-parallel residual and shared attention norm)."
                )
            self.attn.run_rows(ws, ws.x, pos, S, k_cache, v_cache, rope, nsplit, rope_local, norm=self.norm_1)  # -> ws.y
            _linear(self.attn.proj, ws.y, ws.x, epilogue=EPI_RESIDUAL, residual=ws.x)  # x = x + h
            self.mlp.run_rows(ws, ws.x, residual=ws.x, out=ws.x, norm=self.norm_2)  # x = x + mlp(norm_2(x))

    def forward(self, x: torch.Tensor, rope: RoPECache, max_seq_length: int, mask: Optional[torch.Tensor] = None,
                input_pos: Optional[torch.Tensor] = None, kv_cache: Optional[KVCache] = None
                ) -> Tuple[torch.Tensor, Optional[KVCache]]:
        """Stand-alone block call with the reference's signature; ``rope`` rows are already indexed per token
        (model.py:88-89).  ``mask`` is accepted and ignored: causality comes from ``input_pos``."""
        B, T, C = x.size()
        c = self.config
        ws = Workspace(c, T, x.device, 1)
        out = torch.empty_like(x)
        cos, sin = (r.contiguous() for r in rope)
        for b in range(B):
            if kv_cache is not None:
                kc, vc = kv_cache[0][b], kv_cache[1][b]
                S = kc.size(1)
                pos = input_pos[:1].to(torch.int32)
            else:
                S = T
                kc, vc = (torch.empty((c.n_query_groups, S, c.head_size), dtype=torch.bfloat16, device=x.device) for _ in range(2))
                pos = ws.zero_pos
            ws.x.copy_(x[b])
            self.run_rows(ws, pos, S, kc, vc, (cos, sin), ops.attn_nsplit(c.n_query_groups, S, c.q_per_kv, T), rope_local=True)
            out[b].copy_(ws.x)
        return out, kv_cache


class CausalSelfAttention(nn.Module):
    def __init__(self, config: Config) -> None:
        super().__init__()
        # fused q/k/v projection (rows interleaved per query group: [q * q_per_kv, k, v] * n_query_groups) and output projection
        self.attn = nn.Linear(config.n_embd, config.qkv_size, bias=config.bias)
        self.proj = nn.Linear(config.n_embd, config.n_embd, bias=config.bias)
        self.config = config

    def run_rows(self, ws: Workspace, x: torch.Tensor, pos: torch.Tensor, S: int, k_cache: torch.Tensor,
                 v_cache: torch.Tensor, rope: RoPECache, nsplit: int, rope_local: bool = False,
                 norm: Optional[nn.Module] = None) -> torch.Tensor:
        """(norm +) qkv linear, split + RoPE + cache append, attention over the cache; leaves the heads in ``ws.y``
        (the output projection is fused with the residual add by the caller).  A single new token takes the fused
        kernel (one launch); several rows (prefill) take rope_kvappend + attn_decode over all rows."""
        c = self.config
        _linear(self.attn, x, ws.qkv, norm=norm)
        if ws.M == 1 and not rope_local and c.q_per_kv <= ops.FUSED_ATTN_MAX_Q_PER_KV:
            return ops.attn_fused_decode(ws.qkv, rope[0], rope[1], c.rope_n_elem, pos, k_cache, v_cache, c.n_query_groups,
                                         c.q_per_kv, c.head_size, S, nsplit, ws.attn_ws(c, nsplit), ws.tickets, ws.y)
        ops.rope_kvappend(ws.qkv, rope[0], rope[1], c.rope_n_elem, pos, c.n_query_groups, c.q_per_kv, c.head_size, S,
                          ws.q, k_cache, v_cache, rope_local)
        if (ATTN_PREFILL_MFMA and getattr(ws, "pos_is_zero", False) and ops.ATTN_PREFILL_MIN_ROWS <= ws.M <= S
                and c.head_size in (32, 64, 128)):
            # a prompt that starts at position 0 (no ring wrap inside the call): flash attention on the matrix cores
            n = c.n_query_groups * c.head_size * ((S + 63) // 64 * 64)
            if getattr(ws, "vt_scratch", None) is None or ws.vt_scratch.numel() < n:
                ws.vt_scratch = torch.empty((n,), dtype=torch.bfloat16, device=ws.x.device)
            return ops.attn_prefill(ws.q, pos, k_cache, v_cache, c.n_query_groups, c.q_per_kv, c.head_size, S, ws.y, ws.vt_scratch)
        return ops.attn_decode(ws.q, pos, k_cache, v_cache, c.n_query_groups, c.q_per_kv, c.head_size, S, nsplit,
                               ws.attn_ws(c, nsplit), ws.y)

    def forward(self, x: torch.Tensor, rope: RoPECache, max_seq_length: int, mask: Optional[torch.Tensor] = None,
                input_pos: Optional[torch.Tensor] = None, kv_cache: Optional[KVCache] = None
                ) -> Tuple[torch.Tensor, Optional[KVCache]]:
        B, T, C = x.size()
        c = self.config
        ws = Workspace(c, T, x.device, 1)
        out = torch.empty_like(x)
        cos, sin = (r.contiguous() for r in rope)
        for b in range(B):
            if kv_cache is not None:
                kc, vc = kv_cache[0][b], kv_cache[1][b]
                S = kc.size(1)
                pos = input_pos[:1].to(torch.int32)
            else:
                S = T
                kc, vc = (torch.empty((c.n_query_groups, S, c.head_size), dtype=torch.bfloat16, device=x.device) for _ in range(2))
                pos = ws.zero_pos
            self.run_rows(ws, x[b].contiguous(), pos, S, kc, vc, (cos, sin), ops.attn_nsplit(c.n_query_groups, S, c.q_per_kv, T), True)
            _linear(self.proj, ws.y, out[b])
        return out, kv_cache


class GptNeoxMLP(nn.Module):
    def __init__(self, config: Config) -> None:
        super().__init__()
        self.fc = nn.Linear(config.n_embd, config.intermediate_size, bias=config.bias)
        self.proj = nn.Linear(config.intermediate_size, config.n_embd, bias=config.bias)

    def run_rows(self, ws: Workspace, x: torch.Tensor, *, residual: Optional[torch.Tensor], out: torch.Tensor,
                 norm: Optional[nn.Module] = None) -> torch.Tensor:
        self.run_up(ws, x, norm=norm)
        return self.run_down(ws, residual=residual, out=out)

    def run_up(self, ws, x: torch.Tensor, *, norm: Optional[nn.Module] = None) -> torch.Tensor:
        return _linear(self.fc, x, ws.h, epilogue=EPI_GELU, norm=norm)  # exact-erf GELU fused (model.py:284-287)

    def run_down(self, ws, *, residual: Optional[torch.Tensor], out: torch.Tensor) -> torch.Tensor:
        return _linear(self.proj, ws.h, out, epilogue=EPI_RESIDUAL if residual is not None else EPI_NONE, residual=residual)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        rows = x.reshape(-1, x.shape[-1]).contiguous()
        ws = _MlpScratch(rows, self.fc.out_features)
        return self.run_rows(ws, rows, residual=None, out=torch.empty_like(rows)).view(x.shape)


class LLaMAMLP(nn.Module):
    def __init__(self, config: Config) -> None:
        super().__init__()
        self.fc_1 = nn.Linear(config.n_embd, config.intermediate_size, bias=config.bias)
        self.fc_2 = nn.Linear(config.n_embd, config.intermediate_size, bias=config.bias)
        self.proj = nn.Linear(config.intermediate_size, config.n_embd, bias=config.bias)

    def run_rows(self, ws: Workspace, x: torch.Tensor, *, residual: Optional[torch.Tensor], out: torch.Tensor,
                 norm: Optional[nn.Module] = None) -> torch.Tensor:
        self.run_up(ws, x, norm=norm)
        return self.run_down(ws, residual=residual, out=out)

    def run_up(self, ws, x: torch.Tensor, *, norm: Optional[nn.Module] = None) -> torch.Tensor:
        return _linear(self.fc_1, x, ws.h, epilogue=EPI_SWIGLU, partner=self.fc_2, norm=norm)  # silu(fc_1 x) * fc_2 x (model.py:297-301)

    def run_down(self, ws, *, residual: Optional[torch.Tensor], out: torch.Tensor) -> torch.Tensor:
        return _linear(self.proj, ws.h, out, epilogue=EPI_RESIDUAL if residual is not None else EPI_NONE, residual=residual)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        rows = x.reshape(-1, x.shape[-1]).contiguous()
        ws = _MlpScratch(rows, self.fc_1.out_features)
        return self.run_rows(ws, rows, residual=None, out=torch.empty_like(rows)).view(x.shape)


class _MlpScratch:
    def __init__(self, rows: torch.Tensor, hidden: int) -> None:
        self.h = torch.empty((rows.shape[0], hidden), dtype=rows.dtype, device=rows.device)


def build_rope_cache(seq_len: int, n_elem: int, dtype: torch.dtype, device: torch.device, base: int = 10000,
                     condense_ratio: int = 1) -> RoPECache:
    """cos/sin tables (seq_len, n_elem), halves duplicated; fp16 for 16-bit models (reference model.py:304-327).

    The reference builds theta, the position index and their outer product with int/int true divisions, i.e. in the
    ambient default dtype — bf16 under Fabric's ``bf16-true`` — before the final ``.half()``.  ``dtype`` plays that
    role here: a 16-bit ``dtype`` reproduces the bf16-session table (positions rounded to bf16 included), fp32 the
    fp32-session one.  Evaluated with torch on the host so the table is bit-identical to a CPU run, then uploaded.
    """
    math_dtype = dtype if dtype in (torch.float16, torch.bfloat16) else torch.float32
    theta = 1.0 / (base ** (torch.arange(0, n_elem, 2).to(math_dtype) / n_elem))
    positions = torch.arange(seq_len).to(math_dtype) / condense_ratio
    angles = torch.outer(positions, theta).repeat(1, 2)
    cos, sin = torch.cos(angles), torch.sin(angles)
    if dtype in (torch.float16, torch.bfloat16, torch.int8):
        cos, sin = cos.half(), sin.half()
    return cos.to(device), sin.to(device)


def apply_rope(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    """Rotate-half RoPE on a (..., T, n_elem) tensor (reference model.py:330-336).  Utility for callers and tests;
    the decode path applies RoPE inside ``parrot_qkv_rope_kvappend``."""
    half = x.size(-1) // 2
    rotated = torch.cat((-x[..., half:], x[..., :half]), dim=-1)
    return ((x * cos) + (rotated * sin)).type_as(x)
