"""GPTQ int4 Linear replacement with the reference's state-dict format, computed by the W4 HIP kernels.

Interface mirrors ``ColBlockQuantizedLinear`` (reference quantize/gptq.py:205-264): same constructor, same buffers
(``quant_weight`` uint8 logical (out, in/2) stored column-major, ``scales``/``zeros`` (out, ceil(in/tile_cols)),
optional ``bias``), ``pack_weight`` / ``get_weight`` / ``forward``.  Differences, on purpose:
  * ``forward`` runs grouped (tile_cols != -1) and biased layers on the fused kernel too (the reference's Triton path
    is per-channel only and drops the bias, gptq.py:255-264);
  * ``pack_weight`` rounds to nearest before the uint8 cast (the reference truncates, gptq.py:239, which is only
    right for inputs already on the grid);
  * the kernel reads a repacked copy of the weights ("W4K") built once after loading.  A decode session
    (generate/base.py) then RELEASES the reference-layout buffers (``release_reference``): W4K is the one resident image
    of the weights, and ``state_dict()`` / ``get_weight()`` / a later ``.to()`` rebuild ``quant_weight`` / ``scales`` /
    ``zeros`` from it with the exact inverse repack (bit for bit: tests/test_full_size_gpu.py).
``rtn_quantize`` is the round-to-nearest quantiser used for synthetic weights (find_params_weight semantics,
gptq.py:317-347).
"""
from typing import Optional, Tuple

import torch

from .. import ops
from .._hip import EPI_NONE, ParrotHipError


class ColBlockQuantizedLinear(torch.nn.Module):
    def __init__(self, in_features: int, out_features: int, bias: bool, *, bits: int = 4, tile_cols: int = -1) -> None:
        super().__init__()
        if bits != 4:
            raise NotImplementedError("only 4-bit weights are built for the HIP path")
        if in_features % 2:
            raise ValueError("in_features must be even")
        assert isinstance(bias, bool)
        self.in_features, self.out_features, self.bits = in_features, out_features, bits
        self.tile_cols = in_features if tile_cols == -1 else tile_cols
        self.entries_per_byte = 2
        n_groups = -(-in_features // self.tile_cols)
        # (out, in/2) view of an (in/2, out) row-major buffer: the reference's on-disk layout
        self.register_buffer("quant_weight", torch.empty((in_features // 2, out_features), dtype=torch.uint8).t())
        self.register_buffer("scales", torch.empty((out_features, n_groups)))
        self.register_buffer("zeros", torch.empty((out_features, n_groups)))
        self.register_buffer("bias", torch.empty((out_features,)) if bias else None)
        self._packed: Optional[torch.Tensor] = None  # W4K copy (device), rebuilt when the buffers change
        self._released = False     # the reference-layout buffers are empty placeholders: W4K is the only image
        self._ref_dtype = None     # dtype of scales / zeros while released
        self.image_epoch = 0       # moves on whenever the weights may have changed (derived images are keyed by it)

    # -------------------------------------------------------------------------------------- format (torch ops)
    def pack_weight(self, weight: torch.Tensor) -> None:
        """Quantise ``weight`` (out, in) onto the grid given by ``scales``/``zeros`` and store the nibbles."""
        self.restore_reference()
        self.image_epoch += 1
        w = weight.to(device=self.quant_weight.device, dtype=torch.float32)
        g = torch.arange(self.in_features, device=w.device) // self.tile_cols
        q = torch.round(w / self.scales.float()[:, g] + self.zeros.float()[:, g]).clamp_(0, 15).to(torch.uint8)
        self.quant_weight.copy_(q[:, 0::2] | (q[:, 1::2] << 4))
        self._packed = None

    def get_weight(self, dtype: torch.dtype = torch.float) -> torch.Tensor:
        """Dequantised (out, in) weight: (q - zero) * scale computed in ``dtype`` like the reference (:243-252)."""
        qw, scales, zeros = self.reference_buffers()
        q = torch.empty((self.out_features, self.in_features), dtype=dtype, device=qw.device)
        q[:, 0::2] = (qw & 0xF).to(dtype)
        q[:, 1::2] = (qw >> 4).to(dtype)
        g = torch.arange(self.in_features, device=qw.device) // self.tile_cols
        q -= zeros.to(dtype)[:, g]
        q *= scales.to(dtype)[:, g]
        return q

    # -------------------------------------------------------------------------------------- one resident image
    def release_reference(self) -> bool:
        """Free ``quant_weight`` / ``scales`` / ``zeros`` once the W4K image exists: it holds the same information (nibbles,
        and bf16 {scale, zero} per group) and ``parrot_w4_repack`` direction 1 inverts it exactly.  Only when the stored
        scales are bf16 (the image's precision); returns whether the buffers were released."""
        if self._released:
            return True
        if not self.quant_weight.is_cuda or self.scales.dtype != torch.bfloat16 or self.zeros.dtype != torch.bfloat16:
            return False
        self.packed()
        self._ref_dtype = self.scales.dtype
        dev = self.quant_weight.device
        self.quant_weight = torch.empty((0,), dtype=torch.uint8, device=dev)
        self.scales = torch.empty((0,), dtype=self._ref_dtype, device=dev)
        self.zeros = torch.empty((0,), dtype=self._ref_dtype, device=dev)
        self._released = True
        return True

    def reference_buffers(self) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """(quant_weight, scales, zeros) in the reference's layout (quantize/gptq.py:216-226): the module's own buffers, or -
        after ``release_reference`` - fresh tensors rebuilt from the W4K image (the caller drops them when done)."""
        if not self._released:
            return self.quant_weight, self.scales, self.zeros
        dev = self._packed.device
        n_groups = -(-self.in_features // self.tile_cols)
        qw = torch.empty((self.in_features // 2, self.out_features), dtype=torch.uint8, device=dev).t()
        scales = torch.empty((self.out_features, n_groups), dtype=torch.bfloat16, device=dev)
        zeros = torch.empty((self.out_features, n_groups), dtype=torch.bfloat16, device=dev)
        ops.w4_repack(qw, scales, zeros, self.out_features, self.in_features, self.tile_cols, self._packed, 1)
        return qw, scales.to(self._ref_dtype), zeros.to(self._ref_dtype)

    def restore_reference(self) -> None:
        """Make the reference-layout buffers resident again (before they are modified, moved or loaded into)."""
        if self._released:
            qw, scales, zeros = self.reference_buffers()
            self._released = False
            self.quant_weight, self.scales, self.zeros = qw, scales, zeros

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        # the state-dict contract (<name>.quant_weight / .scales / .zeros / .bias) holds whether or not the buffers are resident
        if not self._released:
            return super()._save_to_state_dict(destination, prefix, keep_vars)
        qw, scales, zeros = self.reference_buffers()
        destination[prefix + "quant_weight"], destination[prefix + "scales"], destination[prefix + "zeros"] = qw, scales, zeros
        if self.bias is not None:
            destination[prefix + "bias"] = self.bias if keep_vars else self.bias.detach()

    def _load_from_state_dict(self, *args, **kwargs) -> None:
        self.restore_reference()
        super()._load_from_state_dict(*args, **kwargs)
        self._packed = None
        self.image_epoch += 1

    def _apply(self, fn, *args, **kwargs):
        # .to(device) / .to(dtype): keep quant_weight's column-major layout and drop the derived copy
        self.restore_reference()
        self._packed = None
        self.image_epoch += 1
        qw = self.quant_weight
        out = super()._apply(fn, *args, **kwargs)
        if self.quant_weight.stride() != (1, self.out_features):
            self.quant_weight = self.quant_weight.t().contiguous().t()
        del qw
        return out

    # -------------------------------------------------------------------------------------- HIP path
    def packed(self) -> torch.Tensor:
        """The kernel-native W4K buffer, built on first use with the repack kernel."""
        if self._packed is None:
            assert not self._released
            if not self.quant_weight.is_cuda:
                raise ParrotHipError("ColBlockQuantizedLinear: move the module to the GPU before running it")
            nbytes = ops.w4_packed_bytes(self.out_features, self.in_features, self.tile_cols)
            buf = torch.empty((nbytes,), dtype=torch.uint8, device=self.quant_weight.device)
            ops.w4_repack(self.quant_weight, self.scales.to(torch.bfloat16).contiguous(),
                          self.zeros.to(torch.bfloat16).contiguous(), self.out_features, self.in_features,
                          self.tile_cols, buf, 0)
            self._packed = buf
        return self._packed

    def hip_linear(self, x: torch.Tensor, out: torch.Tensor, *, epilogue: int = EPI_NONE, residual=None,
                   partner: Optional["ColBlockQuantizedLinear"] = None, norm=None) -> torch.Tensor:
        """rows (M, in) -> (M, out) with a fused epilogue; ``partner`` is fc_2 for the SwiGLU epilogue."""
        if partner is not None and (partner.tile_cols != self.tile_cols or partner.in_features != self.in_features
                                    or partner.out_features != self.out_features):
            raise ParrotHipError("SwiGLU partner must have the same shape and group size")
        return ops.w4_linear(self.packed(), self.out_features, self.in_features, self.tile_cols, x, out, bias=self.bias,
                             epilogue=epilogue, residual=residual,
                             packed2=partner.packed() if partner is not None else None, norm=norm)

    def forward(self, inp: torch.Tensor) -> torch.Tensor:
        x = inp.reshape(-1, self.in_features)
        out = torch.empty((x.shape[0], self.out_features), dtype=inp.dtype, device=inp.device)
        self.hip_linear(x.contiguous(), out)
        return out.view(*inp.shape[:-1], self.out_features)

    def extra_repr(self) -> str:
        return f"in_features={self.in_features}, out_features={self.out_features}, bits=4, tile_cols={self.tile_cols}"


def rtn_quantize(weight: torch.Tensor, tile_cols: int = 128, bits: int = 4) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Round-to-nearest asymmetric quantisation per (row, group of ``tile_cols`` columns).

    Grid parameters follow ``GPTQQuantizer.find_params_weight`` (quantize/gptq.py:317-347): the range always
    contains 0, scale = (max - min) / 15, zero = round(-min / scale).  Returns (q uint8 (out, in), scales, zeros)
    with scales/zeros (out, groups) in ``weight.dtype`` — the values the reference would store.
    """
    out_f, in_f = weight.shape
    if tile_cols == -1:
        tile_cols = in_f
    maxq = 2 ** bits - 1
    n_groups = -(-in_f // tile_cols)
    pad = n_groups * tile_cols - in_f
    w = weight.float()
    if pad:  # ragged last group: pad with a value inside the group's range so that min/max are unchanged
        w = torch.cat([w, w[:, -1:].expand(out_f, pad)], dim=1)
    blk = w.view(out_f, n_groups, tile_cols)
    lo = blk.amin(dim=2).clamp(max=0)
    hi = blk.amax(dim=2).clamp(min=0)
    flat = (lo == 0) & (hi == 0)
    lo = torch.where(flat, torch.full_like(lo, -1), lo)
    hi = torch.where(flat, torch.full_like(hi, 1), hi)
    s = (hi - lo) / maxq
    z = torch.round(-lo / s)
    # the stored parameters are rounded to the model dtype first, then used for the grid (as on load)
    scales, zeros = s.to(weight.dtype), z.to(weight.dtype)
    # q = clamp(round(x / scale) + zero, 0, maxq)  (GPTQQuantizer.quantize_weight, gptq.py:313-315)
    q = (torch.round(blk / scales.float()[:, :, None]) + zeros.float()[:, :, None]).clamp_(0, maxq).to(torch.uint8)
    return q.view(out_f, n_groups * tile_cols)[:, :in_f].contiguous(), scales, zeros


def pack_nibbles(q: torch.Tensor) -> torch.Tensor:
    """(out, in) uint8 values 0..15 -> the reference's quant_weight tensor: (out, in/2) with strides (1, out)."""
    return (q[:, 0::2] | (q[:, 1::2] << 4)).t().contiguous().t()


# ------------------------------------------------------------------------------------------------ the GPTQ quantiser
class GPTQQuantizer:
    """Post-training int4 quantiser of one Linear with the reference's interface (quantize/gptq.py:267-444; the algorithm is
    Frantar et al., arXiv:2210.17323): ``collect_input_stats`` as a forward hook accumulates the Hessian of the layer inputs,
    ``quantize()`` returns ``(ColBlockQuantizedLinear, loss)``.

    Device side: the Hessian update and the trailing update of every 128-column block are torch matmuls, the Cholesky
    factorisations are torch.linalg; the serial column loop of a block - the part that does not map onto library calls - is
    the HIP kernel ``parrot_gptq_block`` (one wave per output row, lanes hold the block's columns).
    Differences from the reference, on purpose:
      * grouped quantisation (``groupsize`` 32 / 64 / 128) works: the reference writes the per-group parameters with the wrong
        shape and reads them from not yet compensated weights (:407-412); here they come from the current columns of the group;
      * the grid parameters are rounded to the checkpoint dtype BEFORE the weights are put on the grid, so that the stored
        (scales, zeros, nibbles) reproduce the quantiser's result exactly (the reference quantises with fp32 parameters and
        stores them in the weight dtype).
    """

    def __init__(self, linear_module, *, bits, perchannel=True, sym=False, blocksize=128, percdamp=0.01, groupsize=-1,
                 actorder=False) -> None:
        assert isinstance(linear_module, torch.nn.Linear)
        if bits != 4 or not perchannel or sym or blocksize != 128:
            raise NotImplementedError("the HIP quantiser builds 4-bit asymmetric per-row grids in blocks of 128 columns")
        if groupsize != -1 and (groupsize <= 0 or 128 % groupsize):
            raise NotImplementedError("groupsize must be -1 or a divisor of 128")
        assert not (actorder and groupsize != -1), "The permutation trick does not work for grouped quantization"
        self.linear_module = linear_module
        self.dev = linear_module.weight.device
        if self.dev.type != "cuda":
            raise ParrotHipError("GPTQQuantizer runs on the HIP device (no CPU fallback): move the module to cuda")
        self.rows, self.columns = linear_module.weight.shape
        self.H = torch.zeros((self.columns, self.columns), device=self.dev)
        self.nsamples = 0
        self.bits, self.maxq = bits, 2 ** bits - 1
        self.blocksize, self.percdamp, self.groupsize, self.actorder = blocksize, percdamp, groupsize, actorder
        self.tile_cols = self.columns if groupsize == -1 else groupsize

    def collect_input_stats(self, _1, inp, _2) -> None:
        """Running mean of 2 X^T X over the calibration batches (:349-362)."""
        inp = inp[0].detach()
        if inp.dim() == 2:
            inp = inp.unsqueeze(0)
        b = inp.shape[0]
        rows = inp.reshape(-1, inp.shape[-1]).float()
        self.H *= self.nsamples / (self.nsamples + b)
        self.nsamples += b
        rows = rows * (2 / self.nsamples) ** 0.5
        self.H.addmm_(rows.t(), rows)

    @staticmethod
    def find_params_weight(x: torch.Tensor, maxq: int = 15) -> Tuple[torch.Tensor, torch.Tensor]:
        """Per-row asymmetric grid whose range contains 0 (:317-347)."""
        zero_ = torch.zeros(x.shape[0], device=x.device)
        lo = torch.minimum(x.min(1)[0], zero_)
        hi = torch.maximum(x.max(1)[0], zero_)
        flat = (lo == 0) & (hi == 0)
        lo[flat], hi[flat] = -1.0, 1.0
        # IEEE division (torch's device division by a constant multiplies by the reciprocal: 1 ulp off the reference's CPU
        # result): divide in float64 and round once
        scale = ((hi - lo).double() / maxq).float()
        zero = torch.round((-lo).double() / scale.double()).float()
        return scale.unsqueeze(1), zero.unsqueeze(1)

    @torch.no_grad()
    def quantize(self):
        from .. import _hip
        from .._hip import check, ptr

        lin = self.linear_module
        wdtype = lin.weight.dtype
        W = lin.weight.detach().to(dtype=torch.float32, copy=True).contiguous()
        rows, cols = W.shape
        ngroups = -(-cols // self.tile_cols)
        scales = torch.zeros((rows, ngroups), dtype=torch.float32, device=self.dev)
        zeros = torch.zeros_like(scales)
        if self.groupsize == -1:
            s, z = self.find_params_weight(W, self.maxq)
            scales[:] = s.to(wdtype).float()  # the precision it will be stored with
            zeros[:] = z
        H = self.H
        del self.H
        dead = torch.diag(H) == 0
        H[dead, dead] = 1
        W[:, dead] = 0
        perm = None
        if self.actorder:
            perm = torch.argsort(torch.diag(H), descending=True)
            W = W[:, perm].contiguous()
            H = H[perm][:, perm]
        idx = torch.arange(cols, device=self.dev)
        H[idx, idx] += self.percdamp * torch.mean(torch.diag(H))
        Hinv = torch.linalg.cholesky(torch.cholesky_inverse(torch.linalg.cholesky(H)), upper=True).contiguous()
        del H
        Q = torch.zeros_like(W)
        err = torch.zeros((rows, 128), dtype=torch.float32, device=self.dev)
        loss_rows = torch.zeros((rows,), dtype=torch.float32, device=self.dev)
        lib = _hip.load()
        for i1 in range(0, cols, 128):
            i2 = min(i1 + 128, cols)
            check(lib.parrot_gptq_block(ptr(W), cols, rows, i1, i2 - i1, ptr(Hinv), cols, ptr(Q), cols, ptr(err), ptr(scales),
                                        ptr(zeros), ngroups, 0 if self.groupsize == -1 else self.groupsize, self.maxq,
                                        int(wdtype == torch.bfloat16), ptr(loss_rows), _hip.stream()), "parrot_gptq_block")
            if i2 < cols:
                W[:, i2:].addmm_(err[:, : i2 - i1], Hinv[i1:i2, i2:], alpha=-1.0)
        if perm is not None:
            Q = Q[:, torch.argsort(perm)]
        error = float(loss_rows.sum())
        q_module = ColBlockQuantizedLinear(lin.in_features, lin.out_features, lin.bias is not None, bits=self.bits,
                                           tile_cols=self.groupsize).to(self.dev)
        q_module.scales = scales.to(wdtype)
        q_module.zeros = zeros.to(wdtype)
        q_module.pack_weight(Q)
        if lin.bias is not None:
            q_module.bias = lin.bias.detach().clone()
        return q_module, error


@torch.no_grad()
def blockwise_quantization(model, sample_inputs: torch.Tensor, working_device=None, *, bits: int = 4, groupsize: int = -1,
                           verbose: bool = False):
    """Classic post-training quantisation of every Linear of ``model`` in order (quantize/gptq.py:461-548): each Linear sees
    the outputs of the already quantised layers in front of it.  ``sample_inputs``: (n_samples, T) token ids.  The model is a
    ``lit_parrot_amd.GPT`` with dense bf16 Linears on the HIP device (``working_device`` is accepted for interface parity; the
    whole model stays resident: 288 GB of HBM).  Returns {layer name: GPTQ loss}."""
    from .. import model as model_mod

    dev = model.transformer.wte.weight.device
    if dev.type != "cuda":
        raise ParrotHipError("blockwise_quantization runs on the HIP device: move the model to cuda")
    sample_inputs = sample_inputs.to(dev)
    n, T = sample_inputs.shape
    inps = model.transformer.wte(sample_inputs)
    outs = torch.zeros_like(inps)
    cos, sin = model.build_rope_cache(sample_inputs)
    rope = (cos[:T].contiguous(), sin[:T].contiguous())
    # fc_1 and fc_2 of a SwiGLU MLP read the same rows (neither sees the other's output), so they are observed in one pass
    # and replaced together - the fused SwiGLU launch needs both weights in the same format
    names = [("attn.attn",), ("attn.proj",)]
    names += [("mlp.fc",)] if model.config._mlp_class == "GptNeoxMLP" else [("mlp.fc_1", "mlp.fc_2")]
    names += [("mlp.proj",)]
    losses = {}

    def run_block(block):
        for j in range(n):
            outs[j: j + 1], _ = block(inps[j: j + 1], rope, model.config.block_size)

    def quantise(targets, run):
        """targets: [(parent module, attribute, label)] observed in ONE run of ``run`` and then replaced."""
        quantizers = []
        for parent, attr, label in targets:
            module = getattr(parent, attr)
            gptq = GPTQQuantizer(module, bits=bits, groupsize=groupsize, actorder=(groupsize == -1))
            model_mod.LINEAR_OBSERVERS[id(module)] = (lambda q: lambda rows: q.collect_input_stats(None, (rows.unsqueeze(0),), None))(gptq)
            quantizers.append((parent, attr, label, module, gptq))
        try:
            run()
        finally:
            for _, _, _, module, _ in quantizers:
                model_mod.LINEAR_OBSERVERS.pop(id(module), None)
        for parent, attr, label, _, gptq in quantizers:
            q_module, error = gptq.quantize()
            setattr(parent, attr, q_module)
            losses[label] = error
            if verbose:
                print(f"{label}: quantization error {error:.1f}", flush=True)

    for i, block in enumerate(model.transformer.h):
        for group in names:
            targets = []
            for name in group:
                pname, dname = name.rsplit(".", 1)
                targets.append((block.get_submodule(pname), dname, f"transformer.h.{i}.{name}"))
            quantise(targets, lambda: run_block(block))
        run_block(block)  # the quantised block's outputs are the next block's inputs
        inps, outs = outs, inps
    # lm_head sees the final norm's output (the fused pipeline applies ln_f inside the lm_head launch)
    hidden = inps

    def run_head():
        for j in range(n):
            x = hidden[j]
            ws_out = torch.empty((x.shape[0], model.config.padded_vocab_size), dtype=x.dtype, device=dev)
            model_mod._linear(model.lm_head, x.contiguous(), ws_out, norm=model.transformer.ln_f)

    quantise([(model, "lm_head", "lm_head")], run_head)
    return losses
