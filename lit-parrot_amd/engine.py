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
from ._hip import (ENG_ATTN, ENG_EPI_LOGITS, ENG_GEMV, ENG_W_E4, ENG_W_E8, ENG_W_E16, ENG_W_TWO_LOADERS, ENG_WGS, EPI_GELU, EPI_NONE,
                   EPI_RESIDUAL, EPI_SWIGLU, EngOp, EngState, ParrotHipError, check, ptr)
from .quantize.bnb import InferenceLinear8bitLt
from .quantize.gptq import ColBlockQuantizedLinear
from .rmsnorm import RMSNorm


def e4_image(lin: ColBlockQuantizedLinear, partner: Optional[ColBlockQuantizedLinear] = None, k0: int = 0,
             k1: Optional[int] = None) -> torch.Tensor:
    """E4 copy of an int4 Linear (of the SwiGLU pair ``lin`` = fc_1, ``partner`` = fc_2; of its input columns
    [k0, k1): one K-chunk), built by the repack kernel from the reference-format buffers the module keeps
    (quantize/gptq.py:216-231)."""
    lib = _hip.load()
    N, K = lin.out_features, lin.in_features
    k1 = K if k1 is None else k1
    if k0 % 128 or (k1 % 128 and k1 != K) or not 0 <= k0 < k1 <= K:
        raise ParrotHipError(f"e4_image: the chunk [{k0}, {k1}) must start and end on quantisation groups")
    nbytes = lib.parrot_e4_bytes(N, k1 - k0, int(partner is not None))
    if nbytes < 0:
        raise ParrotHipError(f"parrot_e4_bytes failed ({nbytes}): {_hip.last_error()}")
    keep = []

    def bufs(m):
        qw, scales, zeros = m.reference_buffers()  # (rebuilt from the W4K image when the module has released its own)
        if qw.stride() != (1, m.out_features):
            raise ParrotHipError("ColBlockQuantizedLinear.quant_weight lost its column-major layout")
        g0, g1 = k0 // 128, -(-k1 // 128)
        if m.tile_cols == 128:
            s, z = scales[:, g0:g1], zeros[:, g0:g1]
        elif m.tile_cols >= m.in_features:  # one scale per row (the reference's "gptq.int4"): the same for every group of 128
            s, z = scales[:, :1].expand(-1, g1 - g0), zeros[:, :1].expand(-1, g1 - g0)
        else:
            raise ParrotHipError(f"e4_image: int4 group size {m.tile_cols} (128, or one group per row)")
        s, z = s.to(torch.bfloat16).contiguous(), z.to(torch.bfloat16).contiguous()
        q = qw[:, k0 // 2:k1 // 2]  # storage rows k0/2 .. k1/2 of the (K/2, N) array: contiguous
        keep.extend((s, z, q))
        return ptr(q), ptr(s), ptr(z)

    q1, s1, z1 = bufs(lin)
    q2, s2, z2 = bufs(partner) if partner is not None else (None, None, None)
    out = torch.empty((nbytes,), dtype=torch.uint8, device=keep[0].device)
    check(lib.parrot_e4_repack(q1, s1, z1, q2, s2, z2, N, k1 - k0, ptr(out), _hip.stream()), "parrot_e4_repack")
    return out


def e16_image(lin: torch.nn.Linear, partner: Optional[torch.nn.Linear] = None, k0: int = 0, k1: Optional[int] = None) -> torch.Tensor:
    """E16 copy of a bf16 ``nn.Linear`` weight (of the SwiGLU pair; of its input columns [k0, k1)): per 8 rows and 1024
    columns one 16-KiB ring slot."""
    lib = _hip.load()
    N, K = lin.out_features, lin.in_features
    k1 = K if k1 is None else k1
    nbytes = lib.parrot_e16_bytes(N, k1 - k0, int(partner is not None))
    if nbytes < 0:
        raise ParrotHipError(f"parrot_e16_bytes failed ({nbytes}): {_hip.last_error()}")
    w1 = lin.weight.data[:, k0:k1].contiguous()
    w2 = partner.weight.data[:, k0:k1].contiguous() if partner is not None else None
    if w1.dtype != torch.bfloat16 or (w2 is not None and w2.dtype != torch.bfloat16):
        raise ParrotHipError("e16_image: the weights must be bf16")
    out = torch.empty((nbytes,), dtype=torch.uint8, device=w1.device)
    check(lib.parrot_e16_repack(ptr(w1), ptr(w2) if w2 is not None else None, N, k1 - k0, ptr(out), _hip.stream()), "parrot_e16_repack")
    return out


def e8_image(lin: InferenceLinear8bitLt, partner: Optional[InferenceLinear8bitLt] = None):
    """(E8 copy of an LLM.int8 Linear's CB - of the SwiGLU pair -, the rows' scales SCB in block order: 8 fp32 per block)."""
    lib = _hip.load()
    N, K = lin.out_features, lin.in_features
    nbytes = lib.parrot_e8_bytes(N, K, int(partner is not None))
    if nbytes < 0:
        raise ParrotHipError(f"parrot_e8_bytes failed ({nbytes}): {_hip.last_error()}")
    w1 = lin.weight.data.contiguous()
    w2 = partner.weight.data.contiguous() if partner is not None else None
    if w1.dtype != torch.int8 or (w2 is not None and w2.dtype != torch.int8):
        raise ParrotHipError("e8_image: the weights are not quantised yet")
    out = torch.empty((nbytes,), dtype=torch.uint8, device=w1.device)
    check(lib.parrot_e8_repack(ptr(w1), ptr(w2) if w2 is not None else None, N, K, ptr(out), _hip.stream()), "parrot_e8_repack")
    if partner is None:
        scb = lin.weight.SCB.float().contiguous()
    else:
        scb = torch.cat([lin.weight.SCB.float().view(-1, 4), partner.weight.SCB.float().view(-1, 4)], dim=1).contiguous()
    return out, scb


def _is_int8_linear(m) -> bool:
    return isinstance(m, InferenceLinear8bitLt) and m.is_quantized


def _wfmt_of(linears) -> Optional[int]:
    if all(isinstance(m, ColBlockQuantizedLinear) for m in linears):
        return ENG_W_E4
    if all(_is_int8_linear(m) for m in linears):
        return ENG_W_E8
    if all(_is_bf16_linear(m) for m in linears):
        return ENG_W_E16
    return None


def _is_bf16_linear(m) -> bool:
    return type(m) is torch.nn.Linear and m.weight.dtype == torch.bfloat16


class StreamEngine:
    """One-launch-per-token executor bound to a model's weights, KV caches and the loop state of a DecodeSession."""

    @staticmethod
    def supported(model) -> Optional[str]:
        """None if the model can run on the engine, else the reason."""
        c = model.config
        linears = [m for m in model.modules() if isinstance(m, torch.nn.Linear) or hasattr(m, "hip_linear")]
        if not linears:
            return "no Linear layers"
        wfmt = _wfmt_of(linears)
        if wfmt is None:
            return "the Linears are neither all int4 GPTQ, nor all LLM.int8, nor all plain bf16"
        if wfmt == ENG_W_E4 and any((m.tile_cols != 128 and m.tile_cols < m.in_features) or m.bias is not None for m in linears):
            return "int4 group size other than 128 / per row, or an int4 Linear with a bias"
        if wfmt == ENG_W_E8 and (any(m.bias is not None for m in linears) or len({m.threshold for m in linears}) != 1):
            return "LLM.int8 Linears with a bias, or with different thresholds"
        norms = [model.transformer.ln_f] + [n for b in model.transformer.h for n in (b.norm_1, getattr(b, "norm_2", None)) if n is not None]
        for n in norms:
            if not isinstance(n, (RMSNorm, torch.nn.LayerNorm)):
                return f"unsupported norm {type(n).__name__}"
        if c.n_embd > 8192:
            return "norm weights beyond one ring slot"
        if not c.parallel_residual and c.shared_attention_norm:
            return "sequential residual with a shared attention norm"
        if c.head_size not in (64, 128):
            return f"head size {c.head_size}"
        if StreamEngine._attn_shape(c)[2] < 1 or c.rope_n_elem % 16 or c.rope_n_elem > 128:
            return "more query-head pairs than CUs / rotary width"
        swiglu = c._mlp_class == "LLaMAMLP"
        if c.n_embd % 8 or c.qkv_size % 8 or c.padded_vocab_size % 8 or c.intermediate_size % (4 if swiglu else 8) or c.n_embd > 16384:
            return "row counts that do not fill the 8-row blocks"
        dev = next(model.parameters()).device
        if dev.type != "cuda" or torch.cuda.get_device_properties(dev).multi_processor_count < ENG_WGS:
            return f"the engine keeps {ENG_WGS} workgroups resident, one per CU: the device has fewer CUs"
        if wfmt == ENG_W_E8 and len(StreamEngine._down_chunks(c, wfmt)) > 1:
            return "an LLM.int8 down-projection wider than one LDS image (the row's absmax is over the whole input)"
        b0, b1, _ = StreamEngine._lds_buffers(c, wfmt)
        kmax = max([c.n_embd] + [k1 - k0 for k0, k1 in StreamEngine._down_chunks(c, wfmt)])
        if b0 < 0 or b1 < 0 or _hip.load().parrot_eng_lds_total(kmax, StreamEngine._state_wfmt(c, wfmt), b0, b1) < 0:
            return _hip.last_error()
        if wfmt == ENG_W_E4 and StreamEngine.CHUNK % 128:
            return "the K-chunks of an int4 down-projection must start on quantisation groups"
        if (c.n_embd // 8 + ENG_WGS - 1) // ENG_WGS > 8:
            return "more than 64 residual rows per CU"
        return None

    @staticmethod
    def _state_wfmt(c, wfmt: int) -> int:
        """The launch's weight format word: int4 models with a parallel-residual block run the two-loader build (weights stream
        across the hand-offs there: Falcon-40B 173 -> 190 tok/s; the sequential Llama block loses 5 % with it)."""
        return wfmt | ENG_W_TWO_LOADERS if wfmt == ENG_W_E4 and c.parallel_residual else wfmt

    @staticmethod
    def _attn_shape(c):
        """(query heads per virtual group, virtual groups per K/V group, CUs per virtual group).  The attention unit holds
        1 or 2 query heads; a K/V group with more (GQA, MQA) is attended as several virtual groups, each streaming the
        group's K/V rows for its own heads."""
        hq = 2 if c.q_per_kv % 2 == 0 else 1
        vper = c.q_per_kv // hq
        return hq, vper, min(8, ENG_WGS // (c.n_query_groups * vper))

    # parts of the MLP up-projection in front of, between and behind the two attention ops of a parallel-residual block.
    # Measured (tools/ab_up_split.sh): StableLM-3B 830 / 825 / 822 / 834 / 824 tok/s and Falcon-40B int4 191.2 / 191.3 /
    # 190.7 / 191.8 / 194.1 for (1,1,1) / (1,2,1) / (2,1,1) / (1,1,2) / (2,2,1): within 1.5 %, thirds kept
    UP_SPLIT = (1, 1, 1)
    CHUNK_ABOVE = 16384  # the widest input an LDS activation buffer holds beside the ring
    CHUNK = 8192         # input columns per K-chunk of a down-projection wider than that

    @staticmethod
    def _down_chunks(c, wfmt=ENG_W_E16):
        """[k0, k1) ranges of the MLP down-projection's input: one range, or chunks when it does not fit LDS (an int8 image
        is half the size: up to 11 units of 2048 columns)."""
        K = c.intermediate_size
        if K <= (22528 if wfmt == ENG_W_E8 else StreamEngine.CHUNK_ABOVE):
            return [(0, K)]
        return [(k, min(k + StreamEngine.CHUNK, K)) for k in range(0, K, StreamEngine.CHUNK)]

    @staticmethod
    def _lds_buffers(c, wfmt=ENG_W_E16):
        """Bytes of the two LDS activation buffers and which one the attention ops use as scratch.  Consecutive Linears
        alternate between the buffers.  Sequential residual: QKV 1, out-projection 0, MLP up 1, down-projection (chunks) 0,
        1, ...; the attention scratch shares buffer 0 (idle between the QKV Linear and the out-projection).  Parallel
        residual: QKV 1, MLP up 0 - its thirds run around the attention ops, so their scratch goes to buffer 1 -
        out-projection 1, down-projection (chunks) 0, 1, ..."""
        lib = _hip.load()
        hq, _, nsplit = StreamEngine._attn_shape(c)
        k = [c.n_embd, c.n_embd]
        for i, (k0, k1) in enumerate(StreamEngine._down_chunks(c, wfmt)):
            k[i & 1] = max(k[i & 1], k1 - k0)
        ab = 1 if c.parallel_residual else 0
        fn = lib.parrot_eng_lds_bytes_e8 if wfmt == ENG_W_E8 else lib.parrot_eng_lds_bytes
        sizes = [fn(k[b], *((c.head_size, hq, nsplit) if b == ab else (0, 0, 0))) for b in (0, 1)]
        return sizes[0], sizes[1], ab

    @staticmethod
    def faster_than_multi_launch(model, window: int, int4_min_window: int) -> bool:
        """The measured choice between the two executors: bf16 and LLM.int8 weights - the engine at every window; int4 -
        from ``int4_min_window`` KV slots on, or at every window for multi-query models with more heads per K/V head than the
        multi-launch step's fused attention kernel takes."""
        if any(_is_bf16_linear(m) or _is_int8_linear(m) for m in model.modules()):
            return True  # (LLM.int8: Llama-2-7B 475 vs 427 tok/s)
        if model.config.q_per_kv > ops.FUSED_ATTN_MAX_Q_PER_KV:
            return True  # the multi-launch step has no fused attention for that many heads per K/V head (Falcon-7B int4: 615 vs 249 tok/s)
        if model.config.parallel_residual:
            return True  # weights stream across the block's hand-offs (two-loader build): Falcon-40B int4 190 vs 178 tok/s
        return window >= int4_min_window

    def __init__(self, model, tokens: torch.Tensor, pos: torch.Tensor, caches: List[tuple], S: int, greedy: bool) -> None:
        why = self.supported(model)
        if why is not None:
            raise ParrotHipError(f"stream engine not available: {why}")
        self.lib = _hip.load()
        c = model.config
        dev = tokens.device
        L, d, hs, inter, V = c.n_layer, c.n_embd, c.head_size, c.intermediate_size, c.padded_vocab_size
        _, vper, nsplit = self._attn_shape(c)
        swiglu = c._mlp_class == "LLaMAMLP"
        wfmt = _wfmt_of([m for m in model.modules() if isinstance(m, torch.nn.Linear) or hasattr(m, "hip_linear")])
        attn_buf = self._lds_buffers(c, wfmt)[2]
        self.logits = torch.zeros((1, V), dtype=torch.bfloat16, device=dev)
        # granule buffers, per layer (written once per launch each): qkv, attention partials, heads, x after the
        # attention branch (sequential residual only), MLP hidden, x after the block; zero = "never written" (the epoch
        # starts at 1)
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

        self.images = []  # keeps the E4 / E16 / E8 images of this engine's ops alive
        ops_list: List[EngOp] = []

        # The kernel-layout images belong to the MODEL, not to the session: generate()'s caller resets the KV caches between
        # prompts (model.reset_cache(), generate/base.py:250 of the reference), the session and its engine are rebuilt, and a
        # rebuilt engine must not repack (and, for a moment, hold a second copy of) every weight.  Keyed by the Linear, its
        # partner, the K-chunk and the state of the weights the image was built from.
        model_images = model.__dict__.setdefault("_engine_images", {})
        images = {}
        self.kmax = 0
        chunks = self._down_chunks(c, wfmt)

        def down(i: int, res_in: int) -> None:
            """The MLP down-projection of block i: one op, or one per K-chunk with the rows' sums accumulated in the CU."""
            mlp = model.transformer.h[i].mlp
            for j, (k0, k1) in enumerate(chunks):
                last = j == len(chunks) - 1
                acc = 0 if len(chunks) == 1 else (1 if j == 0 else (3 if last else 2))
                gemv(mlp.proj, d // 8, EPI_RESIDUAL if last else EPI_NONE, j & 1, gran(i, "h") + 8 * (k0 // 2), gran(i, "xb") if last else None,
                     res_in=res_in, res_out=0, publish=last, cols=(k0, k1), acc=acc)

        def gemv(lin, nblocks: int, epilogue: int, buf: int, inp: Optional[int], out: Optional[int], *, partner=None, norm=None,
                 norm2=None, no_gather=False, in_emb=False, res_emb=False, res_in=0, res_out=0, publish=True, part=(0, 1),
                 cols=None, acc=0) -> None:
            op = EngOp()
            k0, k1 = cols if cols is not None else (0, lin.in_features)
            K = k1 - k0
            op.type, op.epilogue, op.K, op.nblocks, op.nq, op.buf = ENG_GEMV, epilogue, K, nblocks, (K + 1023) // 1024, buf
            op.acc = acc
            self.kmax = max(self.kmax, K)
            op.blk_part, op.blk_parts = part
            if part[1] > 1 and epilogue == EPI_RESIDUAL:
                raise ParrotHipError("stream engine: a Linear with a residual epilogue is not split into parts")
            e4 = isinstance(lin, ColBlockQuantizedLinear)
            e8 = _is_int8_linear(lin)
            op.wfmt = ENG_W_E4 if e4 else (ENG_W_E8 if e8 else ENG_W_E16)
            if not e4 and not e8 and lin.bias is not None:
                if partner is not None:
                    raise ParrotHipError("stream engine: a SwiGLU pair with biases is not built")
                op.bias = ptr(lin.bias.data)
            img = images.get((id(lin), k0, k1))
            if img is None:
                def stamp(m):  # what the image was built from: a later load / .to() / optimiser step moves it on
                    return None if m is None else ((id(m), m.image_epoch) if e4 else (id(m), m.weight.data_ptr(), m.weight._version))
                mkey = (stamp(lin), stamp(partner), k0, k1)
                img = model_images.get(mkey)
                if img is None:
                    for old in [k for k in model_images if k[0][0] == id(lin) and k[2:] == (k0, k1)]:
                        del model_images[old]  # an image of the same Linear from before its weights changed
                    if e8:
                        if (k0, k1) != (0, lin.in_features):
                            raise ParrotHipError("stream engine: an LLM.int8 Linear is not split into K-chunks")
                        img = e8_image(lin, partner)  # (image, rows' scales)
                    else:
                        img = e4_image(lin, partner, k0, k1) if e4 else e16_image(lin, partner, k0, k1)
                    model_images[mkey] = img
                images[(id(lin), k0, k1)] = img
                if e8:
                    self.images.append(img[1])
                self.images.append(img[0] if e8 else img)
            if e8:
                op.nq = (K + 2047) // 2048  # int8 units are 2048 columns
                op.bias, op.threshold = img[1].data_ptr(), float(lin.threshold)
                img = img[0]
            op.W, op.inp, op.out = img.data_ptr(), inp, out
            if norm is not None:
                op.norm_w, op.norm_eps = ptr(norm.weight.data), float(norm.eps)
                if isinstance(norm, RMSNorm):
                    op.norm_kind = 1
                else:
                    op.norm_kind = 2
                    if norm.bias is not None:
                        op.norm_b = ptr(norm.bias.data)
            if norm2 is not None:  # a second norm of the same input, for the next op (which then gathers nothing)
                if type(norm2) is not type(norm) or float(norm2.eps) != float(norm.eps):
                    raise ParrotHipError("stream engine: the two norms of a parallel-residual block must be of one kind")
                op.norm2_w = ptr(norm2.weight.data)
                if getattr(norm2, "bias", None) is not None:
                    op.norm2_b = ptr(norm2.bias.data)
            op.in_embedding, op.res_embedding, op.no_gather = int(in_emb), int(res_emb), int(no_gather)
            op.res_in, op.res_out, op.publish = res_in, res_out, int(publish)
            ops_list.append(op)

        for i, (block, (kc, vc)) in enumerate(zip(model.transformer.h, caches)):
            first = i == 0
            x_in = None if first else gran(i - 1, "xb")
            mlp_norm = (block.norm_1 if c.shared_attention_norm else block.norm_2) if c.parallel_residual else None
            gemv(block.attn.attn, c.qkv_size // 8, EPI_NONE, 1, x_in, gran(i, "qkv"), norm=block.norm_1, norm2=mlp_norm, in_emb=first)
            at = EngOp()
            at.type, at.inp, at.out, at.part = ENG_ATTN, gran(i, "qkv"), gran(i, "y"), gran(i, "part")
            at.k_cache, at.v_cache = ptr(kc), ptr(vc)
            mlp = block.mlp
            up, partner = (mlp.fc_1, mlp.fc_2) if swiglu else (mlp.fc, None)
            up_blocks, up_epi = (inter // 4, EPI_SWIGLU) if swiglu else (inter // 8, EPI_GELU)
            if c.parallel_residual:
                # x + attn(norm_1(x)) + mlp(norm_2(x)) (model.py:166-171).  Both norms see the block's input: the QKV op's
                # gather applies both (same row statistics) and leaves norm_2(x) in the other LDS buffer, so the MLP's
                # up-projection starts without a gather and runs between QKV and attention - by the time its weights
                # have streamed the QKV vector has long arrived everywhere.  The out-projection keeps x + attn in the CU
                # that owns the rows, the down-projection adds its rows and hands the block's output over.
                # The attention op runs as two ops with thirds of the up-projection around them: while the partial states
                # of a head travel to its leader CU, and the heads to everybody, weights keep streaming.
                a, b, cc = self.UP_SPLIT  # parts of the up-projection in front of, between and behind the attention halves
                n_parts = a + b + cc
                for j in range(a):
                    gemv(up, up_blocks, up_epi, 0, None, gran(i, "h"), partner=partner, no_gather=True, part=(j, n_parts))
                at.epilogue, at.buf, at.no_gather = 1, attn_buf, 1
                ops_list.append(at)
                for j in range(a, a + b):
                    gemv(up, up_blocks, up_epi, 0, None, gran(i, "h"), partner=partner, no_gather=True, part=(j, n_parts))
                at2 = EngOp()
                at2.type, at2.epilogue, at2.buf = ENG_ATTN, 2, attn_buf
                at2.inp, at2.out, at2.part, at2.k_cache, at2.v_cache = at.inp, at.out, at.part, at.k_cache, at.v_cache
                ops_list.append(at2)
                for j in range(a + b, n_parts):
                    gemv(up, up_blocks, up_epi, 0, None, gran(i, "h"), partner=partner, no_gather=True, part=(j, n_parts))
                gemv(block.attn.proj, d // 8, EPI_RESIDUAL, 1, gran(i, "y"), None, res_emb=first, res_in=0, res_out=1, publish=False)
                down(i, 1)
            else:
                at.buf = attn_buf
                ops_list.append(at)
                gemv(block.attn.proj, d // 8, EPI_RESIDUAL, 0, gran(i, "y"), gran(i, "xa"), res_emb=first)
                gemv(up, up_blocks, up_epi, 1, gran(i, "xa"), gran(i, "h"), partner=partner, norm=block.norm_2)
                down(i, 0)
        gemv(model.lm_head, V // 8, ENG_EPI_LOGITS, 1, gran(L - 1, "xb"), ptr(self.logits), norm=model.transformer.ln_f)

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
        st.vper = vper
        st.kmax = self.kmax
        st.wfmt = self._state_wfmt(c, ops_list[0].wfmt)
        st.lds_buf0_bytes, st.lds_buf1_bytes, st.attn_buf = self._lds_buffers(c, wfmt)
        st.arg = self.granules.data_ptr() + 8 * L * per_layer
        # two words of pinned host memory the kernel can write: [0] the first error code, [1] the epoch of the last launch that
        # ran to its end.  A host watchdog (bench.py) reads them while a launch is stuck, without any HIP call.
        self.host_words = torch.zeros((2,), dtype=torch.int32).pin_memory()
        st.host_words = self.host_words.data_ptr()
        self.state = st
        self.n_ops = len(ops_list)
        self.dbg = None
        self._keep = (tokens, pos, caches, model)

    def enable_stamps(self) -> torch.Tensor:
        """Diagnostic: let workgroup 0 record 100 MHz timestamps per op (enter, input ready, last unit done)."""
        self.dbg = torch.zeros((self.n_ops * 16,), dtype=torch.int64, device=self.logits.device)
        self.state.dbg = ptr(self.dbg)
        self.dbg_all = torch.zeros((self.n_ops * ENG_WGS * 2,), dtype=torch.int64, device=self.logits.device)
        self.state.dbg_all = ptr(self.dbg_all)
        return self.dbg

    def step(self) -> torch.Tensor:
        check(self.lib.parrot_eng_step(C.byref(self.state), _hip.stream()), "parrot_eng_step")
        return self.logits

    ERROR_KINDS = {1: "loader: a ring slot was never released", 2: "consumer barrier", 3: "a ring slot never landed",
                   4: "hand-off of a Linear's input (0x40: gate, 0x41: sweep, 0x42: waves behind the gate)",
                   5: "attention hand-off (0x50 QKV rows, 0x51 partial states)", 6: "arg-max hand-off",
                   7: "table check (0x70 more LLM.int8 outliers than the list holds, 0x71 K-chunk op outside its limits)"}

    def check_error(self) -> None:
        """Host-side check of the kernel's error word (syncs): the code of the FIRST wait that gave up or check that failed.
        The word is sticky - every later launch falls through its waits and decodes garbage - so it is cleared here and
        the caller must start over."""
        e = int(self.err.item()) & 0xFFFFFFFF
        if e:
            self.err.zero_()
            self.host_words.zero_()
            kind = self.ERROR_KINDS.get(e >> 28, "?")
            raise ParrotHipError(f"stream engine: a bounded wait gave up, first error word {e:#x} ({kind}; low bits: op index or ring sequence number)")

    def progress(self) -> tuple:
        """(first error code, epoch of the last completed launch) from the pinned host words: no HIP call, safe from a watchdog thread."""
        return int(self.host_words[0]) & 0xFFFFFFFF, int(self.host_words[1]) & 0xFFFFFFFF
