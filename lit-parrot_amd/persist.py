"""Host side of the persistent decode step (``csrc/persist.hip``): builds the per-token op program of a model once,
keeps the device-side state, and launches one kernel per token through ``parrot_pk_step``.

Supported: every Linear is a ``ColBlockQuantizedLinear`` (int4), head size 64 or 128, ``256 % n_query_groups == 0``,
``q_per_kv <= 16`` and every K cut into a slab count that divides 12.  Anything else keeps the multi-launch step
(still HIP kernels) — ``PersistentStep.supported(model)`` says which.
"""
import ctypes as C
from typing import List, Optional

import torch

from . import _hip, ops
from ._hip import EPI_GELU, EPI_NONE, EPI_RESIDUAL, EPI_SWIGLU, ParrotHipError, check, ptr
from .quantize.gptq import ColBlockQuantizedLinear
from .rmsnorm import RMSNorm

from ._hip import PK_ARGMAX, PK_ATTN, PK_GEMV, PK_MAX_SLABS, PK_WGS, PkOp, PkSlab, PkState


def _norm_fields(mod) -> tuple:
    if isinstance(mod, RMSNorm):
        return 1, mod.weight.data, None, mod.eps
    if isinstance(mod, torch.nn.LayerNorm):
        return 2, mod.weight.data, None if mod.bias is None else mod.bias.data, mod.eps
    raise ParrotHipError(f"persistent step: unsupported norm {type(mod).__name__}")


class PersistentStep:
    """One-launch-per-token executor bound to a model's weights, KV caches and the loop state of a DecodeSession."""

    @staticmethod
    def supported(model) -> Optional[str]:
        """None if the model can run the persistent step, else the reason."""
        c = model.config
        linears = [m for m in model.modules() if isinstance(m, torch.nn.Linear) or hasattr(m, "hip_linear")]
        if not linears or not all(isinstance(m, ColBlockQuantizedLinear) for m in linears):
            return "not every Linear is an int4 ColBlockQuantizedLinear"
        if c.head_size not in (64, 128):
            return f"head size {c.head_size}"
        if PK_WGS % c.n_query_groups or c.q_per_kv > 16:
            return "query group count"
        lib = _hip.load()
        probe = PkOp()
        for m in linears:
            if lib.parrot_pk_fill_w4(C.byref(probe), m.out_features, m.in_features, m.tile_cols) != 0:
                return _hip.last_error()
        return None

    def __init__(self, model, tokens: torch.Tensor, pos: torch.Tensor, caches: List[tuple], S: int, greedy: bool) -> None:
        why = self.supported(model)
        if why is not None:
            raise ParrotHipError(f"persistent step not available: {why}")
        self.lib = _hip.load()
        c = model.config
        dev = tokens.device
        bf = dict(dtype=torch.bfloat16, device=dev)
        # exchanged activation vectors (written with write-through stores, read with agent-scope loads)
        self.X = torch.zeros((1, c.n_embd), **bf)
        self.T = torch.zeros((1, c.n_embd), **bf)
        self.QKV = torch.zeros((1, c.qkv_size), **bf)
        self.Y = torch.zeros((1, c.n_embd), **bf)
        self.H = torch.zeros((1, c.intermediate_size), **bf)
        self.logits = torch.zeros((1, c.padded_vocab_size), **bf)
        nsplit = PK_WGS // c.n_query_groups
        self.attn_ws = ops.attn_workspace(1, c.n_head, c.head_size, nsplit, dev)
        self.tickets = torch.zeros((c.n_query_groups,), dtype=torch.int32, device=dev)
        self.counters = torch.zeros((8 * 32,), dtype=torch.int32, device=dev)
        self.err = torch.zeros((1,), dtype=torch.int32, device=dev)
        self.argmax_val = torch.zeros((PK_WGS,), dtype=torch.float32, device=dev)
        self.argmax_idx = torch.zeros((PK_WGS,), dtype=torch.int32, device=dev)

        ops_list: List[PkOp] = []

        def gemv(lin: ColBlockQuantizedLinear, x, out, *, norm=None, epilogue=EPI_NONE, residual=None, partner=None,
                 x_emb=False, res_emb=False, track=False) -> None:
            op = PkOp()
            check(self.lib.parrot_pk_fill_w4(C.byref(op), lin.out_features, lin.in_features, lin.tile_cols), "parrot_pk_fill_w4")
            op.type, op.epilogue = PK_GEMV, epilogue
            op.W = ptr(lin.packed())
            op.W2 = ptr(partner.packed()) if partner is not None else None
            op.x, op.out = ptr(x), ptr(out)
            op.residual = ptr(residual) if residual is not None else None
            op.bias = ptr(lin.bias) if lin.bias is not None else None
            if norm is not None:
                kind, w, b, eps = _norm_fields(norm)
                op.norm_kind, op.norm_w, op.norm_b, op.norm_eps = kind, ptr(w), ptr(b), float(eps)
            op.x_from_embedding, op.res_from_embedding, op.track_argmax = int(x_emb), int(res_emb), int(track)
            ops_list.append(op)

        for i, (block, (kc, vc)) in enumerate(zip(model.transformer.h, caches)):
            first = i == 0
            gemv(block.attn.attn, self.X, self.QKV, norm=block.norm_1, x_emb=first)
            at = PkOp()
            at.type, at.x, at.out, at.k_cache, at.v_cache = PK_ATTN, ptr(self.QKV), ptr(self.Y), ptr(kc), ptr(vc)
            ops_list.append(at)
            mlp = block.mlp
            if c.parallel_residual:
                gemv(block.attn.proj, self.Y, self.T, epilogue=EPI_RESIDUAL, residual=self.X, res_emb=first)
                n2 = block.norm_1 if c.shared_attention_norm else block.norm_2
                mlp_in, mlp_emb, res, res_out = self.X, first, self.T, self.X
            else:
                gemv(block.attn.proj, self.Y, self.X, epilogue=EPI_RESIDUAL, residual=self.X, res_emb=first)
                n2, mlp_in, mlp_emb, res, res_out = block.norm_2, self.X, False, self.X, self.X
            if hasattr(mlp, "fc_1"):
                gemv(mlp.fc_1, mlp_in, self.H, norm=n2, epilogue=EPI_SWIGLU, partner=mlp.fc_2, x_emb=mlp_emb)
            else:
                gemv(mlp.fc, mlp_in, self.H, norm=n2, epilogue=EPI_GELU, x_emb=mlp_emb)
            gemv(mlp.proj, self.H, res_out, epilogue=EPI_RESIDUAL, residual=res)
        gemv(model.lm_head, self.X, self.logits, norm=model.transformer.ln_f, track=True)
        if greedy:
            am = PkOp()
            am.type = PK_ARGMAX
            ops_list.append(am)

        arr = (PkOp * len(ops_list))(*ops_list)
        host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
        self.ops_dev = host.to(dev)
        st = PkState()
        st.ops, st.nops, st.d = ptr(self.ops_dev), len(ops_list), c.n_embd
        st.tokens, st.pos, st.wte = ptr(tokens), ptr(pos), ptr(model.transformer.wte.weight.data)
        cos, sin = model.rope_cache
        st.rope_cos, st.rope_sin = ptr(cos), ptr(sin)
        st.n_elem, st.n_groups, st.q_per_kv, st.hs, st.S = c.rope_n_elem, c.n_query_groups, c.q_per_kv, c.head_size, S
        st.V, st.rsqrt_mode = c.padded_vocab_size, ops.RMSNORM_RSQRT_MODE
        kmax = max(o.K for o in ops_list if o.type == PK_GEMV)
        st.lds_x_bytes = (2 * kmax + 15) // 16 * 16
        st.attn_ws, st.tickets, st.counters, st.err = ptr(self.attn_ws), ptr(self.tickets), ptr(self.counters), ptr(self.err)
        st.argmax_val, st.argmax_idx = ptr(self.argmax_val), ptr(self.argmax_idx)
        self.state = st
        self.n_ops = len(ops_list)
        self.dbg = None

    def enable_stamps(self) -> torch.Tensor:
        """Diagnostic: let workgroup 0 record 100 MHz timestamps at its phase boundaries (8 per op)."""
        self.dbg = torch.zeros((self.n_ops * 8,), dtype=torch.int64, device=self.logits.device)
        self.state.dbg = ptr(self.dbg)
        return self.dbg

    def step(self) -> torch.Tensor:
        check(self.lib.parrot_pk_step(C.byref(self.state), _hip.stream()), "parrot_pk_step")
        return self.logits

    def check_error(self) -> None:
        """Host-side check of the kernel's timeout word (syncs)."""
        e = int(self.err.item())
        if e:
            raise ParrotHipError(f"persistent step: barrier timeout, error word {e & 0xffffffff:#x}")
