"""Host side of the stream engine (``csrc/engine.hip``): one launch per decode token.

Builds, once per (model, max_seq_length): the E4 copies of every int4 Linear, the per-layer granule buffers the CUs
hand activations over through, and the op table of the token — the launch sequence of ``GPT.run_rows`` for one row
(reference lit_gpt/model.py:63-111, :158-180) restated as a static program.  ``step()`` enqueues
``parrot_eng_step`` (graph-capturable).  Models outside what the engine is built for keep the multi-launch step —
``StreamEngine.supported(model)`` gives the reason.
"""
import ctypes as C
from typing import List, Optional

import torch

from . import _hip, ops
from ._hip import (ENG_ATTN, ENG_EPI_LOGITS, ENG_GEMV, ENG_WGS, EPI_NONE, EPI_RESIDUAL, EPI_SWIGLU, EngOp, EngState,
                   ParrotHipError, check, ptr)
from .quantize.gptq import ColBlockQuantizedLinear
from .rmsnorm import RMSNorm


def e4_image(lin: ColBlockQuantizedLinear, partner: Optional[ColBlockQuantizedLinear] = None) -> torch.Tensor:
    """E4 copy of an int4 Linear (of the SwiGLU pair ``lin`` = fc_1, ``partner`` = fc_2), built by the repack kernel
    from the reference-format buffers the module keeps (quantize/gptq.py:216-231)."""
    lib = _hip.load()
    N, K = lin.out_features, lin.in_features
    nbytes = lib.parrot_e4_bytes(N, K, int(partner is not None))
    if nbytes < 0:
        raise ParrotHipError(f"parrot_e4_bytes failed ({nbytes}): {_hip.last_error()}")
    out = torch.empty((nbytes,), dtype=torch.uint8, device=lin.quant_weight.device)
    keep = []

    def bufs(m):
        s = m.scales.to(torch.bfloat16).contiguous()
        z = m.zeros.to(torch.bfloat16).contiguous()
        keep.extend((s, z))
        if m.quant_weight.stride() != (1, m.out_features):
            raise ParrotHipError("ColBlockQuantizedLinear.quant_weight lost its column-major layout")
        return ptr(m.quant_weight), ptr(s), ptr(z)

    q1, s1, z1 = bufs(lin)
    q2, s2, z2 = bufs(partner) if partner is not None else (None, None, None)
    check(lib.parrot_e4_repack(q1, s1, z1, q2, s2, z2, N, K, ptr(out), _hip.stream()), "parrot_e4_repack")
    return out


class StreamEngine:
    """One-launch-per-token executor bound to a model's weights, KV caches and the loop state of a DecodeSession."""

    @staticmethod
    def supported(model) -> Optional[str]:
        """None if the model can run on the engine, else the reason."""
        c = model.config
        linears = [m for m in model.modules() if isinstance(m, torch.nn.Linear) or hasattr(m, "hip_linear")]
        if not linears or not all(isinstance(m, ColBlockQuantizedLinear) for m in linears):
            return "not every Linear is an int4 ColBlockQuantizedLinear"
        if any(m.tile_cols != 128 or m.bias is not None for m in linears):
            return "int4 group size other than 128, or a bias"
        if c.parallel_residual or c._norm_class != "RMSNorm" or c._mlp_class != "LLaMAMLP":
            return "not a sequential-residual RMSNorm / SwiGLU model"
        if c.head_size not in (64, 128) or c.q_per_kv not in (1, 2):
            return f"head size {c.head_size} / q_per_kv {c.q_per_kv}"
        if c.n_query_groups > ENG_WGS or c.rope_n_elem % 16 or c.rope_n_elem > 128:
            return "query group count / rotary width"
        if c.n_embd % 8 or c.qkv_size % 8 or c.padded_vocab_size % 8 or c.intermediate_size % 4 or c.n_embd > 16384:
            return "row counts that do not fill the 8-row blocks"
        dev = next(model.parameters()).device
        if dev.type != "cuda" or torch.cuda.get_device_properties(dev).multi_processor_count < ENG_WGS:
            return f"the engine keeps {ENG_WGS} workgroups resident, one per CU: the device has fewer CUs"
        lib = _hip.load()
        nsplit = min(8, ENG_WGS // c.n_query_groups)
        b0 = lib.parrot_eng_lds_bytes(max(c.intermediate_size, c.n_embd), c.head_size, c.q_per_kv, nsplit)
        b1 = lib.parrot_eng_lds_bytes(c.n_embd, 0, 0, 0)
        if b0 < 0 or b1 < 0:
            return _hip.last_error()
        if 7 * 17 * 1024 + b0 + b1 + 8500 > 160 * 1024:
            return "activation vectors do not fit the LDS beside the weight ring"
        return None

    def __init__(self, model, tokens: torch.Tensor, pos: torch.Tensor, caches: List[tuple], S: int, greedy: bool) -> None:
        why = self.supported(model)
        if why is not None:
            raise ParrotHipError(f"stream engine not available: {why}")
        self.lib = _hip.load()
        c = model.config
        dev = tokens.device
        L, d, hs, inter, V = c.n_layer, c.n_embd, c.head_size, c.intermediate_size, c.padded_vocab_size
        nsplit = min(8, ENG_WGS // c.n_query_groups)
        self.logits = torch.zeros((1, V), dtype=torch.bfloat16, device=dev)
        # granule buffers, per layer (written once per launch each): qkv, attention partials, heads, x after the
        # attention branch, MLP hidden, x after the MLP branch; zero = "never written" (the epoch starts at 1)
        sizes = dict(qkv=c.qkv_size // 2, part=c.n_head * nsplit * (hs + 2), y=d // 2, xa=d // 2, h=inter // 2, xb=d // 2)
        per_layer = sum(sizes.values())
        self.granules = torch.zeros((L * per_layer + 2 * ENG_WGS,), dtype=torch.int64, device=dev)

        def gran(layer: int, name: str) -> int:
            off = layer * per_layer
            for k, n in sizes.items():
                if k == name:
                    return self.granules.data_ptr() + 8 * off
                off += n
            raise KeyError(name)

        self.e4 = []  # keeps the E4 images alive

        def image(lin, partner=None) -> int:
            t = e4_image(lin, partner)
            self.e4.append(t)
            return t.data_ptr()

        ops_list: List[EngOp] = []

        def gemv(W: int, K: int, nblocks: int, epilogue: int, buf: int, inp: Optional[int], out: int, *, norm=None,
                 in_emb=False, res_emb=False) -> None:
            op = EngOp()
            op.type, op.epilogue, op.K, op.nblocks, op.nq, op.buf = ENG_GEMV, epilogue, K, nblocks, (K + 1023) // 1024, buf
            op.W, op.inp, op.out = W, inp, out
            if norm is not None:
                if not isinstance(norm, RMSNorm):
                    raise ParrotHipError(f"stream engine: unsupported norm {type(norm).__name__}")
                op.norm_kind, op.norm_w, op.norm_eps = 1, ptr(norm.weight.data), float(norm.eps)
            op.in_embedding, op.res_embedding = int(in_emb), int(res_emb)
            ops_list.append(op)

        for i, (block, (kc, vc)) in enumerate(zip(model.transformer.h, caches)):
            first = i == 0
            gemv(image(block.attn.attn), d, c.qkv_size // 8, EPI_NONE, 1, None if first else gran(i - 1, "xb"), gran(i, "qkv"),
                 norm=block.norm_1, in_emb=first)
            at = EngOp()
            at.type, at.inp, at.out, at.part = ENG_ATTN, gran(i, "qkv"), gran(i, "y"), gran(i, "part")
            at.k_cache, at.v_cache = ptr(kc), ptr(vc)
            ops_list.append(at)
            gemv(image(block.attn.proj), d, d // 8, EPI_RESIDUAL, 0, gran(i, "y"), gran(i, "xa"), res_emb=first)
            gemv(image(block.mlp.fc_1, block.mlp.fc_2), d, inter // 4, EPI_SWIGLU, 1, gran(i, "xa"), gran(i, "h"), norm=block.norm_2)
            gemv(image(block.mlp.proj), inter, d // 8, EPI_RESIDUAL, 0, gran(i, "h"), gran(i, "xb"))
        gemv(image(model.lm_head), d, V // 8, ENG_EPI_LOGITS, 1, gran(L - 1, "xb"), ptr(self.logits), norm=model.transformer.ln_f)

        arr = (EngOp * len(ops_list))(*ops_list)
        self.ops_dev = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
        self.epoch = torch.ones((1,), dtype=torch.int32, device=dev)
        self.err = torch.zeros((1,), dtype=torch.int32, device=dev)
        st = EngState()
        st.ops, st.nops, st.d = ptr(self.ops_dev), len(ops_list), d
        st.tokens, st.pos, st.epoch, st.err = ptr(tokens), ptr(pos), ptr(self.epoch), ptr(self.err)
        st.wte = ptr(model.transformer.wte.weight.data)
        cos, sin = model.rope_cache
        self.rope = (cos, sin)
        st.rope_cos, st.rope_sin = ptr(cos), ptr(sin)
        st.n_elem, st.n_groups, st.q_per_kv, st.hs, st.S = c.rope_n_elem, c.n_query_groups, c.q_per_kv, hs, S
        st.V, st.rsqrt_mode, st.nsplit, st.greedy = V, ops.RMSNORM_RSQRT_MODE, nsplit, int(greedy)
        st.lds_buf0_bytes = self.lib.parrot_eng_lds_bytes(max(inter, d), hs, c.q_per_kv, nsplit)
        st.lds_buf1_bytes = self.lib.parrot_eng_lds_bytes(d, 0, 0, 0)
        st.arg = self.granules.data_ptr() + 8 * L * per_layer
        self.state = st
        self.n_ops = len(ops_list)
        self.dbg = None
        self._keep = (tokens, pos, caches, model)

    def enable_stamps(self) -> torch.Tensor:
        """Diagnostic: let workgroup 0 record 100 MHz timestamps per op (enter, input ready, last unit done)."""
        self.dbg = torch.zeros((self.n_ops * 8,), dtype=torch.int64, device=self.logits.device)
        self.state.dbg = ptr(self.dbg)
        self.dbg_all = torch.zeros((self.n_ops * ENG_WGS * 2,), dtype=torch.int64, device=self.logits.device)
        self.state.dbg_all = ptr(self.dbg_all)
        return self.dbg

    def step(self) -> torch.Tensor:
        check(self.lib.parrot_eng_step(C.byref(self.state), _hip.stream()), "parrot_eng_step")
        return self.logits

    def check_error(self) -> None:
        """Host-side check of the kernel's time-out word (syncs)."""
        e = int(self.err.item()) & 0xFFFFFFFF
        if e:
            raise ParrotHipError(f"stream engine: a bounded wait gave up, error word {e:#x}")
