"""bitsandbytes 4-bit Linear (NF4 / FP4, +-double quantisation).

CPU part: the oracle restatement (oracle/nf4.py, the kernels' decision trees) against the product's loader
(quantize/bnb.py, bucketize formulation) and against the one independent anchor there is offline: the NF4 codebook
re-derived from its definition.  GPU part: the codebook GEMV / dequantise kernels against the oracle.
Tolerances: dequantise is exact (one rounding, same operands).  The GEMV keeps the codebook in bf16 and applies absmax to
the block sum (sum_k x_k * bf16(code[q_k])) * absmax, while bitsandbytes rounds code * absmax to bf16 per weight: both are
2^-9-relative per weight; the kernel is held to <= 1 bf16 ulp of ITS definition evaluated in float64, and to the int4 bound
of north_star (1e-2 of the output scale) against the oracle's MatMul4Bit.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import assert_bf16_close, rbf
from lit_parrot_amd.quantize import bnb as P
from oracle import nf4 as O

BF = torch.bfloat16
MODES = [("nf4", False), ("nf4", True), ("fp4", False), ("fp4", True)]


def gen(seed):
    return torch.Generator().manual_seed(seed)


# ------------------------------------------------------------------------------------------------------------ CPU
def test_nf4_codebook_is_the_normal_quantile_map():
    """functional.create_normal_map(offset=0.9677083, use_extra_value=True): 8 positive + 0 + 7 negative quantiles of
    N(0, 1), normalised to [-1, 1] (QLoRA, arXiv:2305.14314 app. E)."""
    from scipy.stats import norm

    offset = 0.9677083
    v1 = norm.ppf(torch.linspace(offset, 0.5, 9)[:-1]).tolist()
    v3 = (-norm.ppf(torch.linspace(offset, 0.5, 8)[:-1])).tolist()
    values = np.sort(np.array(v1 + [0.0] + v3))
    values /= values.max()
    assert np.allclose(values, O.NF4, atol=2e-7)  # the published table is this map in fp32 (table digits: 1e-7)
    assert np.array_equal(np.array(P.NF4_CODE, dtype=np.float32), O.NF4)
    mid = (O.NF4[1:].astype(np.float64) + O.NF4[:-1].astype(np.float64)) / 2
    assert np.allclose(mid, np.array(P.NF4_THRESHOLDS), atol=1e-7)  # the tree's thresholds are the midpoints


def test_fp4_tree_is_self_consistent():
    assert np.array_equal(np.array(P.FP4_CODE, dtype=np.float32), O.FP4)
    pos = O.FP4[:8]
    assert np.array_equal(O.quantize_fp4(pos)[1:], np.arange(1, 8))  # every positive code value maps to itself
    assert np.array_equal(O.quantize_fp4(-pos)[1:], np.arange(9, 16))
    order = np.argsort(pos)
    mids = (pos[order][1:] + pos[order][:-1]) / 2
    assert np.allclose(mids, np.array(P.FP4_THRESHOLDS), atol=2e-7)
    assert [int(i) for i in order] == list(P.FP4_RANK_TO_PATTERN)


def test_dynamic_map():
    code = O.create_dynamic_map()
    assert code.size == 256 and np.all(np.diff(code) > 0) and code[-1] == 1.0 and code[127] == 0.0
    assert np.array_equal(code, P.dynamic_map_8bit().numpy())
    x = torch.rand(2000, generator=gen(1)) * 2 - 1
    fast = P._nearest_dynamic(torch.from_numpy(code), x).numpy()
    slow = np.array([O.quantize_dynamic_scalar(code, np.float32(v)) for v in x.numpy()])
    assert np.array_equal(fast, slow)
    assert np.all(np.abs(code[fast] - x.numpy()) <= np.abs(code[None, :] - x.numpy()[:, None]).min(axis=1) + 1e-7)  # nearest value


@pytest.mark.parametrize("quant_type,dq", MODES)
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
def test_loader_matches_oracle(quant_type, dq, dtype):
    w = (torch.randn(24, 320, generator=gen(2)) * 0.02).to(dtype)
    w[3, 64:128] = 0  # an all-zero block
    po, so = O.quantize_4bit(w, quant_type, dq)
    pp, sp = P.quantize_4bit(w, quant_type, dq)
    assert pp.dtype == torch.uint8 and tuple(pp.shape) == (24 * 320 // 2, 1)
    assert torch.equal(po, pp)
    assert np.array_equal(O.absmax_of(so), P.absmax_of(sp).numpy())
    d = P.dequantize_4bit(pp, sp)
    assert d.dtype == dtype and torch.equal(d, O.dequantize_4bit(po, so))
    assert torch.all(d[3, 64:128] == 0)
    # quantisation error of a normal block: NF4 ~ 9 %, FP4 ~ 12 % relative rms (QLoRA table 2 ordering)
    rel = float((d.float() - w.float()).pow(2).mean().sqrt() / w.float().std())
    assert rel < (0.11 if quant_type == "nf4" else 0.14)
    # first weight of a pair in the HIGH nibble
    first = int(pp[0, 0]) >> 4
    code = O.NF4 if quant_type == "nf4" else O.FP4
    assert abs(code[first] * O.absmax_of(so)[0] - float(w[0, 0])) <= 0.2 * O.absmax_of(so)[0]


def test_requantising_the_dequantised_weights_is_idempotent():
    w = torch.randn(8, 128, generator=gen(3)) * 0.02
    for quant_type in ("nf4", "fp4"):
        p1, s1 = O.quantize_4bit(w, quant_type, False)
        d1 = O.dequantize_4bit(p1, s1)
        p2, s2 = O.quantize_4bit(d1, quant_type, False)
        assert torch.equal(O.dequantize_4bit(p2, s2), d1)


def test_mode_names_select_the_class():
    from lit_parrot_amd import quantization

    for mode, (qt, dq) in {"bnb.nf4": ("nf4", False), "bnb.nf4-dq": ("nf4", True), "bnb.fp4": ("fp4", False), "bnb.fp4-dq": ("fp4", True)}.items():
        with quantization(mode):
            lin = torch.nn.Linear(128, 64, bias=False)
        assert isinstance(lin, P.Linear4bit) and isinstance(lin, torch.nn.Linear)
        assert (lin.quant_type, lin.compress_statistics) == (qt, dq) and lin.bias is None
    assert torch.nn.Linear is not P.Linear4bit
    with pytest.raises(NotImplementedError):
        P.Linear4bit(128, 64, quant_type="int4")


# ------------------------------------------------------------------------------------------------------------ GPU
def _module(N, K, quant_type, dq, seed, bias=False):
    g = gen(seed)
    w = (torch.randn(N, K, generator=g) * 0.02).to(BF)
    b = (torch.randn(N, generator=g) * 0.1).to(BF) if bias else None
    lin = P.Linear4bit(K, N, bias, quant_type=quant_type, compress_statistics=dq)
    lin.load_state_dict({"weight": w, **({"bias": b} if bias else {})})
    lin = lin.to("cuda").to(BF)  # float parameters (the bias) in the model dtype; the packed weight stays uint8
    po, so = O.quantize_4bit(w, quant_type, dq)
    if dq:
        # the offset is a mean over all absmax values: the GPU sums in another order than the CPU (bitsandbytes computes it on
        # the GPU as well), so it may differ in the last bit, and with it a few 8-bit indices.  Hold the module's statistics
        # to the oracle's within that, then let the oracle continue from the module's own statistics.
        qabs, _, _, _, (offset, (absmax2, code8)), _, _ = lin.weight.quant_state
        assert abs(float(offset) - float(so["offset"])) <= 2e-7 * abs(float(so["offset"]))
        assert np.allclose(absmax2.cpu().numpy(), so["absmax2"], rtol=1e-5, atol=1e-9)
        assert np.array_equal(code8.cpu().numpy(), so["code8"])
        assert float(np.mean(qabs.cpu().numpy() == so["qabsmax"])) >= 0.98
        so.update(qabsmax=qabs.cpu().numpy(), offset=np.float32(offset.item()), absmax2=absmax2.cpu().numpy())
    return lin, w, b, po, so


def _kernel_weights(po, so):
    """The GEMV's definition: bf16 codebook, absmax applied per block: float64 (N, K)."""
    b = po.numpy().reshape(-1)
    q = np.stack([b >> 4, b & 0xF], axis=1).reshape(-1, 64)
    code = torch.from_numpy(O.NF4 if so["quant_type"] == "nf4" else O.FP4).to(BF).double().numpy()
    return torch.from_numpy(code[q] * O.absmax_of(so).astype(np.float64)[:, None]).reshape(so["shape"])


SHAPES = [(64, 256), (40, 320), (16, 64), (256, 4096), (96, 11008), (24, 8192), (8, 32768)]


@pytest.mark.gpu
@pytest.mark.parametrize("quant_type,dq", MODES)
@pytest.mark.parametrize("N,K", SHAPES)
def test_state_and_dequant_kernel_are_exact(hip_lib, N, K, quant_type, dq):
    from lit_parrot_amd import ops

    lin, w, _, po, so = _module(N, K, quant_type, dq, 11)
    assert lin.weight.dtype == torch.uint8 and torch.equal(lin.weight.data.cpu(), po)
    assert np.array_equal(P.absmax_of(lin.weight.quant_state).cpu().numpy(), O.absmax_of(so))
    out = torch.empty((N, K), dtype=BF, device="cuda")
    lin.packed()
    ops.w4c_dequant(lin._packed, lin._codes[1], N, K, 64, out)
    assert torch.equal(out.cpu(), O.dequantize_4bit(po, so))


@pytest.mark.gpu
@pytest.mark.parametrize("quant_type,dq", [("nf4", False), ("fp4", True)])
@pytest.mark.parametrize("N,K", SHAPES)
@pytest.mark.parametrize("M", [1, 3, 8])
def test_gemv_matches_oracle(hip_lib, N, K, quant_type, dq, M):
    lin, w, b, po, so = _module(N, K, quant_type, dq, 12, bias=True)
    x = torch.randn(M, K, generator=gen(13)).to(BF)
    out = lin(x.cuda())
    want = rbf(x.double() @ _kernel_weights(po, so).t() + b.double())
    assert_bf16_close(out, want, ulps=1, atol=2e-3, what=f"w4c_gemv {quant_type} N={N} K={K} M={M}")
    ref = O.linear(x, po, so, b)  # bitsandbytes' definition
    assert float((out.cpu().float() - ref.float()).abs().max()) <= 1e-2 * max(1.0, float(ref.float().abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize("quant_type", ["nf4", "fp4"])
def test_gemv_epilogues_norm_and_swiglu(hip_lib, quant_type):
    from lit_parrot_amd import ops
    from lit_parrot_amd._hip import EPI_GELU, EPI_RESIDUAL, EPI_SWIGLU

    N, K, M = 520, 1024, 2
    lin, _, _, po, so = _module(N, K, quant_type, False, 21)
    lin2, _, _, po2, so2 = _module(N, K, quant_type, False, 22)
    Wk, Wk2 = _kernel_weights(po, so), _kernel_weights(po2, so2)
    g = gen(23)
    x = torch.randn(M, K, generator=g).to(BF)
    res = torch.randn(M, N, generator=g).to(BF)
    nw = (1 + 0.1 * torch.randn(K, generator=g)).to(BF)
    xd = x.cuda()
    out = torch.empty((M, N), dtype=BF, device="cuda")
    lin.hip_linear(xd, out, epilogue=EPI_RESIDUAL, residual=res.cuda())
    assert_bf16_close(out, res.double() + rbf(x.double() @ Wk.t()), ulps=1, atol=2e-3, what="residual")
    lin.hip_linear(xd, out, epilogue=EPI_GELU)
    assert_bf16_close(out, F.gelu(rbf(x.double() @ Wk.t())), ulps=1, atol=2e-3, what="gelu")
    lin.hip_linear(xd, out, epilogue=EPI_SWIGLU, partner=lin2)
    assert_bf16_close(out, rbf(F.silu(rbf(x.double() @ Wk.t()))) * rbf(x.double() @ Wk2.t()), ulps=1, atol=2e-3, what="swiglu")
    # fused RMSNorm == stand-alone norm kernel followed by the plain launch, bit for bit
    xn = torch.empty_like(xd)
    ops.rmsnorm(xd, nw.cuda(), 1e-5, xn)
    a, b = torch.empty_like(out), torch.empty_like(out)
    lin.hip_linear(xd, a, norm=ops.Norm(1, nw.cuda(), None, 1e-5))
    lin.hip_linear(xn, b)
    assert torch.equal(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("quant_type,dq", [("nf4", True), ("fp4", False)])
@pytest.mark.parametrize("M,N,K", [(48, 384, 1024), (200, 130, 4096), (1500, 2048, 256)])
def test_prefill_rows_on_the_matrix_cores(hip_lib, quant_type, dq, M, N, K):
    """More than 8 rows: the codebook GEMM (parrot_w4c_gemm: split-K and unsplit launches, ragged tiles) has the GEMV's numerics;
    with ops.W4C_PREFILL_FUSED = False the rows take bitsandbytes' own order of operations (dequantise + dense GEMM)."""
    from lit_parrot_amd import ops

    lin, _, b, po, so = _module(N, K, quant_type, dq, 31, bias=True)
    x = torch.randn(M, K, generator=gen(32)).to(BF)
    out = lin(x.cuda())
    want = rbf(x.double() @ _kernel_weights(po, so).t() + b.double())
    assert_bf16_close(out, want, ulps=1, atol=3e-3, what="codebook gemm")
    ref = O.linear(x, po, so, b)  # bitsandbytes' definition
    assert float((out.cpu().float() - ref.float()).abs().max()) <= 1e-2 * max(1.0, float(ref.float().abs().max()))
    if M <= 200:
        ops.W4C_PREFILL_FUSED = False
        try:
            out2 = lin(x.cuda())
        finally:
            ops.W4C_PREFILL_FUSED = True
        want2 = rbf(x.double() @ O.dequantize_4bit(po, so).double().t() + b.double())
        assert_bf16_close(out2, want2, ulps=1, atol=2e-3, what="dequantise + gemm")


@pytest.mark.gpu
@pytest.mark.parametrize("quant_type", ["nf4", "fp4"])
def test_prefill_swiglu_pair(hip_lib, quant_type):
    from lit_parrot_amd._hip import EPI_SWIGLU

    N, K = 384, 512
    lin, _, _, po, so = _module(N, K, quant_type, False, 41)
    lin2, _, _, po2, so2 = _module(N, K, quant_type, False, 42)
    Wk, Wk2 = _kernel_weights(po, so), _kernel_weights(po2, so2)
    for M in (40, 700):  # split-K partials + second stage / (700 x 384: 18 tiles -> still split) the two-pass kernel
        x = torch.randn(M, K, generator=gen(43)).to(BF)
        out = torch.empty((M, N), dtype=BF, device="cuda")
        lin.hip_linear(x.cuda(), out, epilogue=EPI_SWIGLU, partner=lin2)
        want = rbf(F.silu(rbf(x.double() @ Wk.t()))) * rbf(x.double() @ Wk2.t())
        assert_bf16_close(out, want, ulps=1, atol=3e-3, what=f"codebook gemm swiglu M={M}")


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["bnb.nf4", "bnb.nf4-dq", "bnb.fp4", "bnb.fp4-dq"])
def test_model_with_4bit_linears_generates_like_its_dequantised_twin(hip_lib, mode):
    """GPT built under quantization(mode), fed a dense checkpoint, against a bf16 GPT holding the dequantised weights:
    logits within the int4 bound, greedy tokens tie-aware equal."""
    from lit_parrot_amd import GPT, Config, generate, quantization
    from lit_parrot_amd.checkpoint import stream_load
    from lit_parrot_amd.config import name_to_config
    from lit_parrot_amd.synth import synthetic_state_dict

    cfg = Config(**{**name_to_config["tiny-llama-hs128"], "intermediate_size": 384})  # every in_features a multiple of 64
    sd = synthetic_state_dict(cfg, seed=7, dtype=BF)
    with torch.device("cuda"), quantization(mode):
        qmodel = GPT(cfg)
    qmodel = qmodel.to(BF).eval()
    assert stream_load(qmodel, sd) == []
    with torch.device("cuda"):
        twin = GPT(cfg).to(BF).eval()
    dense = dict(sd)
    n4 = 0
    for name, mod in qmodel.named_modules():
        if isinstance(mod, P.Linear4bit):
            assert mod.is_quantized
            dense[name + ".weight"] = mod.dequantized_weight()
            n4 += 1
    assert n4 == 1 + 5 * cfg.n_layer
    twin.load_state_dict(dense)
    idx = torch.randint(0, cfg.padded_vocab_size, (1, 20), generator=gen(5)).cuda()
    lq, lt = qmodel(idx).float(), twin(idx).float()
    # the int4 bound of north_star (1e-2 of the logit scale) plus one bf16 ulp at that scale: the logits themselves are bf16, and
    # the two models round differently at every Linear (absmax applied per block vs per weight)
    scale = max(1.0, float(lt.abs().max()))
    assert float((lq - lt).abs().max()) <= (1e-2 + 2 ** -7) * scale
    assert float((lq - lt).abs().mean()) <= 3e-3 * scale
    prompt = idx[0, :8]
    a = generate(qmodel, prompt, 24, max_seq_length=24, temperature=1.0, top_k=1)
    qmodel.reset_cache()
    assert a.shape == (24,) and torch.equal(a[:8], prompt)
    # every generated token is an arg-max of the twin's logits for the same prefix, up to the logit bound above
    lt_all = twin(a[None, :-1]).float()[0]
    top = lt_all.max(dim=-1).values
    chosen = lt_all.gather(1, a[1:, None])[:, 0]
    assert torch.all(top[7:] - chosen[7:] <= 2e-2 * max(1.0, float(lt_all.abs().max())))
