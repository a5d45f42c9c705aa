"""LLM.int8 Linear replacement (reference quantize/bnb.py:18-60) without bitsandbytes.

``InferenceLinear8bitLt`` keeps the reference class's surface: an ``nn.Linear`` subclass constructed with
``has_fp16_weights=False, threshold=6.0`` (bnb.py:26-33) that quantises its weight when a float ``weight`` arrives
through ``load_state_dict`` (bnb.py:38-50) and afterwards holds ``weight`` = int8 ``CB`` with attributes ``CB`` and
``SCB`` (bnb.py:52-60).  The arithmetic — bitsandbytes' ``double_quant`` / ``MatMul8bitLt`` in the reference — is
done by the W8 HIP kernels (csrc/w8.hip), which restate the published LLM.int8 algorithm.
``Linear4bit`` (nf4 / fp4, with or without double quantisation; bnb.py:62-75) follows further down.
"""
from typing import Optional

import torch

from .. import ops
from .._hip import EPI_NONE, ParrotHipError


FUSED_DECODE = True  # single rows take parrot_w8_gemv_fused (tests switch it off to compare with the two-launch path)


class InferenceLinear8bitLt(torch.nn.Linear):
    def __init__(self, in_features: int, out_features: int, bias: bool = True, *, has_fp16_weights: bool = False,
                 threshold: float = 6.0, device=None, dtype=None) -> None:
        if has_fp16_weights:
            raise NotImplementedError("inference only: has_fp16_weights=False (quantize/bnb.py:30)")
        super().__init__(in_features, out_features, bias, device=device, dtype=dtype)
        self.threshold = float(threshold)
        self.weight.requires_grad_(False)
        if self.bias is not None:
            self.bias.requires_grad_(False)
        self._act: Optional[ops.W8Act] = None
        self._pair = None  # (CB, SCB) of [self; partner] concatenated for the SwiGLU epilogue
        # the reference quantises the freshly initialised weight right away (bnb.py:34-36); that needs the GPU, so
        # here it happens when the module reaches the device (see _apply) or a checkpoint is loaded

    @property
    def is_quantized(self) -> bool:
        return self.weight.dtype == torch.int8

    def _quantize_weight(self, weight: torch.Tensor) -> None:
        """Row-wise absmax int8 of ``weight`` on the current GPU (double_quant semantics, bnb.py:52-60)."""
        if not torch.cuda.is_available():
            raise ParrotHipError("InferenceLinear8bitLt quantises on the GPU: no HIP device is visible")
        dev = weight.device if weight.is_cuda else torch.device("cuda", torch.cuda.current_device())
        # the checkpoint's own precision goes to the kernel, which rounds to fp16 like the reference's `weight.contiguous().half()`
        # (bnb.py:54): casting an fp16 / fp32 checkpoint to bf16 first would drop three mantissa bits before the quantiser
        w = weight.detach().to(device=dev)
        if w.dtype not in (torch.bfloat16, torch.float16, torch.float32):
            w = w.float()
        w = w.contiguous()
        CB = torch.empty(w.shape, dtype=torch.int8, device=dev)
        SCB = torch.empty((w.shape[0],), dtype=torch.float32, device=dev)
        ops.w8_quantize_rows(w, CB, SCB)
        self.weight = torch.nn.Parameter(CB, requires_grad=False)
        setattr(self.weight, "CB", CB)
        setattr(self.weight, "SCB", SCB)
        self._pair = None

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs) -> None:
        key = prefix + "weight"
        if key in state_dict:
            self._quantize_weight(state_dict.pop(key))
            # nn.Module would now report the popped key as missing; the reference loads with strict=False
            # (generate/base.py:222).  Put a placeholder so that strict loading also works.
            state_dict[key] = self.weight.data
            super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)
            state_dict.pop(key)
            return
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def _apply(self, fn, *args, **kwargs):
        scb = getattr(self.weight, "SCB", None)
        out = super()._apply(fn, *args, **kwargs)
        if self.is_quantized and scb is not None:
            # nn.Module._apply may have rebuilt the Parameter: carry CB / SCB over, on the same device as CB
            setattr(self.weight, "CB", self.weight.data)
            setattr(self.weight, "SCB", scb.to(self.weight.device))
            self._pair = None
        elif not self.is_quantized and self.weight.is_cuda:
            self._quantize_weight(self.weight.data)
        return out

    # -------------------------------------------------------------------------------------- HIP path
    def prep(self, x: torch.Tensor, norm=None) -> ops.W8Act:
        M, K = x.shape
        if self._act is None or (self._act.M, self._act.K) != (M, K) or self._act.xq.device != x.device:
            self._act = ops.W8Act(M, K, x.device)
        return ops.w8_prep_act(x, self.threshold, self._act, norm)

    def hip_linear(self, x: torch.Tensor, out: torch.Tensor, *, epilogue: int = EPI_NONE, residual=None,
                   partner: Optional["InferenceLinear8bitLt"] = None, act: Optional[ops.W8Act] = None,
                   norm=None) -> torch.Tensor:
        if not self.is_quantized:
            raise ParrotHipError("InferenceLinear8bitLt: weight not quantised yet (move the module to the GPU)")
        CB, SCB = self.weight.data, self.weight.SCB
        if partner is not None:
            if self._pair is None:
                self._pair = (torch.cat([CB, partner.weight.data]).contiguous(), torch.cat([SCB, partner.weight.SCB]).contiguous())
            CB, SCB = self._pair
        if act is None and x.shape[0] == 1 and self.in_features <= ops.W8_FUSED_MAX_K and FUSED_DECODE:
            # decode: activation quantiser (with the fused norm) + GEMV in one launch
            return ops.w8_linear_fused(CB, SCB, self.out_features, self.in_features, x, self.threshold, out, bias=self.bias,
                                       epilogue=epilogue, residual=residual, norm=norm)
        act = act if act is not None else self.prep(x, norm)  # the norm is fused into the activation quantiser
        return ops.w8_linear(CB, SCB, self.out_features, self.in_features, act, out, bias=self.bias, epilogue=epilogue,
                             residual=residual)

    def forward(self, inp: torch.Tensor) -> torch.Tensor:
        x = inp.reshape(-1, self.in_features).contiguous()
        out = torch.empty((x.shape[0], self.out_features), dtype=inp.dtype, device=inp.device)
        self.hip_linear(x, out)
        return out.view(*inp.shape[:-1], self.out_features)


# ======================================================================================================================
# bitsandbytes 4-bit (NF4 / FP4, optional double quantisation): reference quantize/bnb.py:62-75, lit_gpt/utils.py:36-68
# ======================================================================================================================
# The codebooks and decision thresholds are bitsandbytes' published constants (functional.py `get_4bit_type`,
# csrc/kernels.cu `dQuantizeNF4` / `dQuantizeFP4` / `dDequantizeFP4Tree`); bitsandbytes itself is not a dependency.
NF4_CODE = (-1.0, -0.6961928009986877, -0.5250730514526367, -0.39491748809814453, -0.28444138169288635, -0.18477343022823334,
            -0.09105003625154495, 0.0, 0.07958029955625534, 0.16093020141124725, 0.24611230194568634, 0.33791524171829224,
            0.44070982933044434, 0.5626170039176941, 0.7229568362236023, 1.0)
# x > threshold[i] for i < c  <=>  code index c
NF4_THRESHOLDS = (-0.8480964004993439, -0.6106329262256622, -0.4599952697753906, -0.33967943489551544, -0.23460740596055984,
                  -0.13791173323988914, -0.045525018125772476, 0.03979014977812767, 0.1202552504837513, 0.2035212516784668,
                  0.2920137718319893, 0.3893125355243683, 0.5016634166240692, 0.6427869200706482, 0.8614784181118011)
# index = sign bit (8) | 3-bit pattern; value of each pattern as dDequantizeFP4Tree returns it
FP4_CODE = (0.0, 0.005208333333, 0.66666667, 1.0, 0.33333333, 0.5, 0.16666667, 0.25,
            -0.0, -0.005208333333, -0.66666667, -1.0, -0.33333333, -0.5, -0.16666667, -0.25)
FP4_THRESHOLDS = (0.00260417, 0.0859375, 0.20833333, 0.29166667, 0.4166667, 0.583333, 0.8333333)  # on |x|, ascending
FP4_RANK_TO_PATTERN = (0b000, 0b001, 0b110, 0b111, 0b100, 0b101, 0b010, 0b011)
BLOCK_4BIT = 64     # weights per absmax (bitsandbytes' default for 4-bit)
BLOCK_NESTED = 256  # absmax values per second-level absmax with compress_statistics


def dynamic_map_8bit() -> torch.Tensor:
    """bitsandbytes' signed "dynamic" 8-bit codebook (functional.create_dynamic_map, 7 exponent bits): the code the absmax
    values are stored in with compress_statistics.  256 sorted fp32 values."""
    data = []
    for i in range(7):
        boundaries = torch.linspace(0.1, 1, 2 ** i + 1)
        means = (boundaries[:-1] + boundaries[1:]) / 2.0
        data += ((10 ** (-6 + i)) * means).tolist()
        data += (-(10 ** (-6 + i)) * means).tolist()
    data += [0.0, 1.0]
    assert len(data) == 256
    data.sort()
    return torch.tensor(data, dtype=torch.float32)


def _nearest_dynamic(code: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """Index of the code value nearest to x in [-1, 1] (kernels.cu dQuantize<0>: bisection to a pivot, then the midpoint
    towards the neighbour on x's side decides)."""
    pivot = torch.full_like(x, 127, dtype=torch.int64)
    upper_pivot = torch.full_like(pivot, 255)
    lower_pivot = torch.zeros_like(pivot)
    lower = torch.full_like(x, -1.0)
    upper = torch.full_like(x, 1.0)
    val = code[pivot]
    step = 64
    while step > 0:
        gt = x > val
        lower_pivot = torch.where(gt, pivot, lower_pivot)
        lower = torch.where(gt, val, lower)
        upper_pivot = torch.where(gt, upper_pivot, pivot)
        upper = torch.where(gt, upper, val)
        pivot = torch.where(gt, pivot + step, pivot - step)
        val = code[pivot]
        step >>= 1
    upper = torch.where(upper_pivot == 255, code[upper_pivot], upper)
    lower = torch.where(lower_pivot == 0, code[lower_pivot], lower)
    gt = x > val
    up = torch.where(x > (upper + val) * 0.5, upper_pivot, pivot)
    down = torch.where(x < (lower + val) * 0.5, lower_pivot, pivot)
    return torch.where(gt, up, down)


def quantize_4bit(weight: torch.Tensor, quant_type: str = "fp4", compress_statistics: bool = False):
    """bitsandbytes.functional.quantize_4bit on any device with torch ops (load-time work).  Returns
    (packed uint8 (n/2, 1), quant_state) with quant_state laid out like bitsandbytes 0.40/0.41:
    [absmax, shape, dtype, blocksize, compressed_stats, quant_type, code]; compressed_stats = [offset, [absmax2, code8]]
    and absmax = the uint8 indices when compress_statistics, else None and absmax = fp32."""
    if quant_type not in ("nf4", "fp4"):
        raise NotImplementedError(f"4-bit quantization data type {quant_type} is not implemented.")
    n = weight.numel()
    if n % BLOCK_4BIT:
        raise ValueError(f"number of weights {n} is not a multiple of the block size {BLOCK_4BIT}")
    dev = weight.device
    blk = weight.detach().reshape(-1, BLOCK_4BIT).float()
    absmax = blk.abs().amax(dim=1)
    inv = torch.where(absmax > 0, 1.0 / absmax, torch.zeros_like(absmax))  # the kernel multiplies by the reciprocal
    xn = blk * inv[:, None]
    if quant_type == "nf4":
        q = torch.bucketize(xn, torch.tensor(NF4_THRESHOLDS, dtype=torch.float32, device=dev))  # number of thresholds < x
    else:
        rank = torch.bucketize(xn.abs(), torch.tensor(FP4_THRESHOLDS, dtype=torch.float32, device=dev))
        q = torch.tensor(FP4_RANK_TO_PATTERN, dtype=torch.int64, device=dev)[rank] + 8 * (xn < 0)
    q = q.to(torch.uint8).reshape(-1)
    packed = ((q[0::2] << 4) | q[1::2]).reshape(-1, 1)  # first weight in the HIGH nibble
    code = torch.tensor(NF4_CODE if quant_type == "nf4" else FP4_CODE, dtype=torch.float32, device=dev)
    if compress_statistics:
        offset = absmax.mean()
        centred = absmax - offset
        code8 = dynamic_map_8bit().to(dev)
        nb = -(-centred.numel() // BLOCK_NESTED)
        pad = nb * BLOCK_NESTED - centred.numel()
        c2 = torch.cat([centred, centred.new_zeros(pad)]).view(nb, BLOCK_NESTED)
        absmax2 = c2.abs().amax(dim=1)
        inv2 = torch.where(absmax2 > 0, 1.0 / absmax2, torch.zeros_like(absmax2))
        qabs = _nearest_dynamic(code8, c2 * inv2[:, None]).to(torch.uint8).reshape(-1)[: centred.numel()]
        state = [qabs, tuple(weight.shape), weight.dtype, BLOCK_4BIT, [offset, [absmax2, code8]], quant_type, code]
    else:
        state = [absmax, tuple(weight.shape), weight.dtype, BLOCK_4BIT, None, quant_type, code]
    return packed, state


def absmax_of(quant_state) -> torch.Tensor:
    """The fp32 absmax per block of a quant_state (de-nested when the statistics are compressed)."""
    absmax, _, _, _, compressed, _, _ = quant_state
    if compressed is None:
        return absmax
    offset, (absmax2, code8) = compressed
    blocks = torch.arange(absmax.numel(), device=absmax.device) // BLOCK_NESTED
    return code8[absmax.long()] * absmax2[blocks] + offset


def dequantize_4bit(packed: torch.Tensor, quant_state) -> torch.Tensor:
    """bitsandbytes.functional.dequantize_4bit with torch ops (format checks, state-dict round trips; the HIP path has its
    own kernel): code[q] * absmax in fp32, rounded once to the dtype the weight had."""
    _, shape, dtype, blocksize, _, _, code = quant_state
    b = packed.reshape(-1)
    q = torch.stack([b >> 4, b & 0xF], dim=1).reshape(-1, blocksize).long()
    return (code[q] * absmax_of(quant_state)[:, None]).to(dtype).reshape(shape)


class Linear4bit(torch.nn.Linear):
    """``bnb.nn.Linear4bit`` as the reference wraps it (quantize/bnb.py:62-75): an ``nn.Linear`` whose ``weight`` becomes the
    packed uint8 tensor (n/2, 1) with a ``quant_state`` attribute once it is on the GPU.  A float ``weight`` arriving through
    ``load_state_dict`` (or present when the module is moved to the GPU) is quantised there; the matmul is the codebook GEMV
    kernel (decode) or dequantise + MFMA GEMM (prefill)."""

    def __init__(self, in_features: int, out_features: int, bias: bool = True, compute_dtype=None, compress_statistics: bool = True,
                 quant_type: str = "fp4", device=None, dtype=None) -> None:
        if quant_type not in ("nf4", "fp4"):
            raise NotImplementedError(f"4-bit quantization data type {quant_type} is not implemented.")
        if in_features % BLOCK_4BIT:
            raise NotImplementedError(f"Linear4bit on the HIP path needs in_features to be a multiple of {BLOCK_4BIT}")
        super().__init__(in_features, out_features, bias, device=device, dtype=dtype)
        self.compute_dtype, self.compress_statistics, self.quant_type = compute_dtype, compress_statistics, quant_type
        self.weight.requires_grad_(False)
        if self.bias is not None:
            self.bias.requires_grad_(False)
        self._packed = None  # W4K records (device)
        self._codes = None   # (bf16 codebook as int32 words, fp32 codebook) on the device

    @property
    def is_quantized(self) -> bool:
        return self.weight.dtype == torch.uint8

    def _quantize_weight(self, weight: torch.Tensor) -> None:
        if not torch.cuda.is_available():
            raise ParrotHipError("Linear4bit quantises on the GPU: no HIP device is visible")
        if tuple(weight.shape) != (self.out_features, self.in_features):
            raise ValueError(f"weight shape {tuple(weight.shape)} != ({self.out_features}, {self.in_features})")
        dev = weight.device if weight.is_cuda else torch.device("cuda", torch.cuda.current_device())
        packed, state = quantize_4bit(weight.detach().to(dev), self.quant_type, self.compress_statistics)
        self.weight = torch.nn.Parameter(packed, requires_grad=False)
        setattr(self.weight, "quant_state", state)
        self._packed = None

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs) -> None:
        key = prefix + "weight"
        if key in state_dict and state_dict[key].is_floating_point():
            self._quantize_weight(state_dict.pop(key))
            state_dict[key] = self.weight.data  # placeholder so that strict loading also works (cf. InferenceLinear8bitLt)
            super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)
            state_dict.pop(key)
            return
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def _apply(self, fn, *args, **kwargs):
        state = getattr(self.weight, "quant_state", None)
        out = super()._apply(fn, *args, **kwargs)
        self._packed = None
        if self.is_quantized and state is not None:
            dev = self.weight.device
            moved = [t.to(dev) if torch.is_tensor(t) else t for t in state]
            if moved[4] is not None:
                moved[4] = [moved[4][0].to(dev), [moved[4][1][0].to(dev), moved[4][1][1].to(dev)]]
            setattr(self.weight, "quant_state", moved)
        elif not self.is_quantized and self.weight.is_cuda:
            self._quantize_weight(self.weight.data)
        return out

    def dequantized_weight(self) -> torch.Tensor:
        return dequantize_4bit(self.weight.data, self.weight.quant_state)

    # -------------------------------------------------------------------------------------- HIP path
    def packed(self) -> torch.Tensor:
        """W4K records with group = 64 and the block's fp32 absmax as the group word (include/parrot_hip.h), built once."""
        if self._packed is None:
            if not self.is_quantized or not self.weight.is_cuda:
                raise ParrotHipError("Linear4bit: weight not quantised yet (move the module to the GPU)")
            from .gptq import pack_nibbles

            N, K, dev = self.out_features, self.in_features, self.weight.device
            b = self.weight.data.reshape(N, K // 2)
            q = torch.stack([b >> 4, b & 0xF], dim=2).reshape(N, K)  # bitsandbytes: first weight in the high nibble
            halves = absmax_of(self.weight.quant_state).float().reshape(N, K // BLOCK_4BIT).contiguous().view(torch.int16)
            lo = halves[:, 0::2].contiguous().view(torch.bfloat16)  # raw bit halves of the fp32 words
            hi = halves[:, 1::2].contiguous().view(torch.bfloat16)
            buf = torch.empty((ops.w4_packed_bytes(N, K, BLOCK_4BIT),), dtype=torch.uint8, device=dev)
            ops.w4_repack(pack_nibbles(q), lo, hi, N, K, BLOCK_4BIT, buf, 0)
            code = self.weight.quant_state[6].float()
            words = code.to(torch.bfloat16).view(torch.int16).to(torch.int32) & 0xFFFF
            self._codes = (words.contiguous(), code.contiguous())
            self._packed = buf
        return self._packed

    def hip_linear(self, x: torch.Tensor, out: torch.Tensor, *, epilogue: int = EPI_NONE, residual=None,
                   partner: Optional["Linear4bit"] = None, norm=None) -> torch.Tensor:
        if partner is not None and (partner.quant_type != self.quant_type or partner.in_features != self.in_features
                                    or partner.out_features != self.out_features):
            raise ParrotHipError("SwiGLU partner must have the same shape and 4-bit type")
        packed = self.packed()
        return ops.w4c_linear(packed, self._codes[0], self._codes[1], self.out_features, self.in_features, BLOCK_4BIT, x, out,
                              bias=self.bias, epilogue=epilogue, residual=residual,
                              packed2=partner.packed() if partner is not None else None, norm=norm)

    def forward(self, inp: torch.Tensor) -> torch.Tensor:
        x = inp.reshape(-1, self.in_features).contiguous()
        out = torch.empty((x.shape[0], self.out_features), dtype=inp.dtype, device=inp.device)
        self.hip_linear(x, out)
        return out.view(*inp.shape[:-1], self.out_features)
