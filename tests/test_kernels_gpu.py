"""Every C-ABI entry point against the CPU oracle on seeded inputs (runs on the MI355X box: ``-m gpu``).

Expected values are computed on the CPU in float64 from exactly the same bf16/int inputs, with the reference's
rounding points (bf16 after the Linear, after the activation, after the residual add ...).  The GPU accumulates in
fp32 in a different order, so a result may land on the neighbouring bf16 value: the bar is <= 1 bf16 ulp unless a
test says otherwise; integer / byte / index work is compared exactly.
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import assert_bf16_close, rbf

pytestmark = pytest.mark.gpu

from lit_parrot_amd import ops  # noqa: E402
from lit_parrot_amd._hip import EPI_GELU, EPI_NONE, EPI_RESIDUAL, EPI_SWIGLU, ParrotHipError  # noqa: E402
from lit_parrot_amd.quantize.gptq import ColBlockQuantizedLinear  # noqa: E402
from oracle import int4 as o4  # noqa: E402
from oracle import int8 as o8  # noqa: E402
from oracle import model as om  # noqa: E402

DEV = torch.device("cuda", 0)
BF = torch.bfloat16


def gen(seed):
    return torch.Generator().manual_seed(seed)


def expected_epilogue(acc, acc2, bias, residual, epi):
    """acc: float64 dot products.  Mirrors parrot_common.h::apply_epilogue / the reference's bf16 rounding points."""
    v = acc + (bias.double() if bias is not None else 0)
    v = rbf(v)
    if epi == EPI_RESIDUAL:
        v = residual.double() + v
    elif epi == EPI_GELU:
        v = F.gelu(v)
    elif epi == EPI_SWIGLU:
        v = rbf(F.silu(v)) * rbf(acc2)
    return v


# ------------------------------------------------------------------------------------------------ int4
W4_SHAPES = [  # N, K, group
    (64, 256, 128), (48, 256, -1), (48, 256, 64), (40, 352, 128), (16, 64, 32), (32, 768, 128),
    (256, 4096, 128), (128, 11008, 128), (24, 8192, 128), (8, 32768, 128), (100, 4096, -1), (36, 2048, 256),
]


def make_w4(N, K, group, seed):
    g = gen(seed)
    w = (torch.randn(N, K, generator=g) * 0.02).to(BF)
    qw, s, z = o4.rtn_quantize(w, group, BF)
    tc = K if group == -1 else group
    return qw, s, z, tc, o4.get_weight(qw, s.float(), z.float(), tc, torch.float32).double()  # exact (q-z)*s


def w4_module(qw, s, z, N, K, group, bias=None):
    lin = ColBlockQuantizedLinear(K, N, bias is not None, bits=4, tile_cols=group)
    lin.quant_weight.copy_(qw)
    lin.scales = s.clone()
    lin.zeros = z.clone()
    if bias is not None:
        lin.bias = bias.clone()
    return lin.to(DEV)


@pytest.mark.parametrize("N,K,group", W4_SHAPES)
def test_w4_repack_round_trip_is_exact(N, K, group):
    qw, s, z, tc, _ = make_w4(N, K, group, 1)
    lin = w4_module(qw, s, z, N, K, group)
    packed = lin.packed()
    qw2 = torch.zeros_like(lin.quant_weight)
    assert qw2.stride() == (1, N)
    s2, z2 = torch.zeros_like(lin.scales), torch.zeros_like(lin.zeros)
    ops.w4_repack(qw2, s2, z2, N, K, tc, packed, 1)
    assert torch.equal(qw2.cpu(), qw) and torch.equal(s2.cpu(), s) and torch.equal(z2.cpu(), z)


@pytest.mark.parametrize("N,K,group", W4_SHAPES)
@pytest.mark.parametrize("M", [1, 3])
def test_w4_gemv_matches_oracle(N, K, group, M):
    qw, s, z, tc, Wd = make_w4(N, K, group, 2)
    g = gen(3)
    x = torch.randn(M, K, generator=g).to(BF)
    bias = (torch.randn(N, generator=g) * 0.1).to(BF)
    lin = w4_module(qw, s, z, N, K, group, bias)
    out = torch.empty((M, N), dtype=BF, device=DEV)
    lin.hip_linear(x.to(DEV), out)
    want = expected_epilogue(x.double() @ Wd.t(), None, bias, None, EPI_NONE)
    assert_bf16_close(out, want, ulps=1, atol=2e-3, what=f"w4_gemv N={N} K={K} g={group} M={M}")
    # the reference's definition (bf16 get_weight + F.linear, gptq.py:263-264) within the int4 bound of north_star
    ref = F.linear(x, o4.get_weight(qw, s, z, tc, BF), bias)
    assert float((out.cpu().float() - ref.float()).abs().max()) <= 1e-2 * max(1.0, float(ref.float().abs().max()))


@pytest.mark.parametrize("kind", ["w4", "bf16"])
@pytest.mark.parametrize("N,K,epi,M", [(40000, 128, EPI_NONE, 1), (20003, 256, EPI_SWIGLU, 1), (33001, 128, EPI_RESIDUAL, 3)])
def test_gemv_workgroups_walk_several_row_batches(kind, N, K, epi, M):
    """More row batches than resident workgroups: every workgroup walks several batches (lm_head-sized launches), with the
    rolling window of weight requests crossing the batch boundary, a ragged last batch and a fused norm computed once."""
    g = gen(41)
    x = torch.randn(M, K, generator=g).to(BF)
    res = torch.randn(M, N, generator=g).to(BF) if epi == EPI_RESIDUAL else None
    nw = (1 + 0.1 * torch.randn(K, generator=g)).to(BF)
    norm = ops.Norm(1, nw.to(DEV), None, 1e-5)
    out = torch.empty((M, N), dtype=BF, device=DEV)
    out2 = torch.empty((M, N), dtype=BF, device=DEV)
    xnd = torch.empty((M, K), dtype=BF, device=DEV)
    ops.rmsnorm(x.to(DEV), nw.to(DEV), 1e-5, xnd)
    if kind == "w4":
        qw, s, z, tc, Wd = make_w4(N, K, 128, 7)
        lin = w4_module(qw, s, z, N, K, 128, None)
        lin2 = None
        Wd2 = None
        if epi == EPI_SWIGLU:
            qw2, s2, z2, _, Wd2 = make_w4(N, K, 128, 8)
            lin2 = w4_module(qw2, s2, z2, N, K, 128)
        lin.hip_linear(x.to(DEV), out, epilogue=epi, residual=res.to(DEV) if res is not None else None, partner=lin2, norm=norm)
        lin.hip_linear(xnd, out2, epilogue=epi, residual=res.to(DEV) if res is not None else None, partner=lin2)
    else:
        W = (torch.randn(N, K, generator=g) * 0.05).to(BF)
        W2 = (torch.randn(N, K, generator=g) * 0.05).to(BF) if epi == EPI_SWIGLU else None
        Wd, Wd2 = W.double(), (W2.double() if W2 is not None else None)
        kw = dict(epilogue=epi, residual=res.to(DEV) if res is not None else None, weight2=W2.to(DEV) if W2 is not None else None)
        ops.bf16_linear(W.to(DEV), x.to(DEV), out, norm=norm, **kw)
        ops.bf16_linear(W.to(DEV), xnd, out2, **kw)
    # fused norm == stand-alone norm kernel followed by the same GEMV (bit for bit up to the statistic's summation order)
    assert float((out.float() - out2.float()).abs().max()) <= 2 ** -6 and float((out == out2).float().mean()) > 0.97
    # and against float64 on the normalised rows
    xd = xnd.cpu().double()
    want = expected_epilogue(xd @ Wd.t(), xd @ Wd2.t() if Wd2 is not None else None, None, res, epi)
    assert_bf16_close(out2, want, ulps=1, atol=3e-3, what=f"{kind} multi-batch N={N} K={K} epi={epi} M={M}")


@pytest.mark.parametrize("M", [1, 2, 4, 5, 9])
@pytest.mark.parametrize("epi", [EPI_NONE, EPI_RESIDUAL, EPI_GELU, EPI_SWIGLU])
def test_w4_epilogues_and_row_counts(M, epi):
    N, K, group = 96, 512, 128
    qw, s, z, tc, Wd = make_w4(N, K, group, 4)
    qw2, s2, z2, _, Wd2 = make_w4(N, K, group, 5)
    g = gen(6)
    x = torch.randn(M, K, generator=g).to(BF)
    res = torch.randn(M, N, generator=g).to(BF)
    bias = None if epi == EPI_SWIGLU else (torch.randn(N, generator=g) * 0.1).to(BF)
    lin, lin2 = w4_module(qw, s, z, N, K, group, bias), w4_module(qw2, s2, z2, N, K, group)
    out = torch.empty((M, N), dtype=BF, device=DEV)
    lin.hip_linear(x.to(DEV), out, epilogue=epi, residual=res.to(DEV) if epi == EPI_RESIDUAL else None,
                   partner=lin2 if epi == EPI_SWIGLU else None)
    want = expected_epilogue(x.double() @ Wd.t(), x.double() @ Wd2.t(), bias, res, epi)
    assert_bf16_close(out, want, ulps=1, atol=2e-3, what=f"w4 epilogue {epi} M={M}")


def test_w4_residual_in_place_and_linearity():
    """x <- x + W y in place (the decode step does this), and the kernel is linear in x up to bf16 rounding."""
    N, K, group = 128, 4096, 128
    qw, s, z, tc, Wd = make_w4(N, K, group, 7)
    lin = w4_module(qw, s, z, N, K, group)
    g = gen(8)
    y = torch.randn(1, K, generator=g).to(BF)
    xres = torch.randn(1, N, generator=g).to(BF)
    buf = xres.to(DEV).clone()
    lin.hip_linear(y.to(DEV), buf, epilogue=EPI_RESIDUAL, residual=buf)
    assert_bf16_close(buf, xres.double() + rbf(y.double() @ Wd.t()), ulps=1, atol=2e-3, what="in-place residual")
    # scaling x by 2 (exact in bf16) scales the output by exactly 2
    o1, o2 = torch.empty((1, N), dtype=BF, device=DEV), torch.empty((1, N), dtype=BF, device=DEV)
    lin.hip_linear(y.to(DEV), o1)
    lin.hip_linear((y * 2).to(DEV), o2)
    assert torch.equal(o2, o1 * 2)


def test_w4_module_forward_shape_and_bad_inputs():
    N, K, group = 64, 256, 128
    qw, s, z, tc, Wd = make_w4(N, K, group, 9)
    lin = w4_module(qw, s, z, N, K, group)
    x = torch.randn(2, 3, K, generator=gen(1)).to(BF)
    out = lin(x.to(DEV))
    assert out.shape == (2, 3, N) and out.dtype == BF
    assert_bf16_close(out.view(6, N), rbf(x.view(6, K).double() @ Wd.t()), ulps=1, atol=2e-3, what="module forward")
    with pytest.raises(ParrotHipError):
        lin(x.to(DEV).float())  # fp32 activations: not built


# ------------------------------------------------------------------------------------------------ MFMA prefill GEMMs
@pytest.mark.parametrize("N,K,group", [(96, 512, 128), (256, 4096, 128), (200, 352, 128), (128, 256, -1), (64, 256, 32), (130, 11008, 128),
                                       (72, 4544, -1)])  # (Falcon-7B's width, per-channel: three slabs, whose starts must be even slices)
@pytest.mark.parametrize("M", [9, 37, 128, 200])
def test_w4_gemm_matches_oracle_and_the_gemv(N, K, group, M):
    qw, s, z, tc, Wd = make_w4(N, K, group, 30)
    g = gen(31)
    x = torch.randn(M, K, generator=g).to(BF)
    bias = (torch.randn(N, generator=g) * 0.1).to(BF)
    lin = w4_module(qw, s, z, N, K, group, bias)
    out = torch.empty((M, N), dtype=BF, device=DEV)
    lin.hip_linear(x.to(DEV), out)  # M > 8 -> parrot_w4_gemm (matrix cores)
    want = expected_epilogue(x.double() @ Wd.t(), None, bias, None, EPI_NONE)
    assert_bf16_close(out, want, ulps=1, atol=3e-3, what=f"w4_gemm N={N} K={K} g={group} M={M}")
    # same numerics as the decode GEMV (row blocks of <= 8): identical up to fp32 summation order
    ref = torch.empty_like(out)
    for m0 in range(0, M, 8):
        lin.hip_linear(x[m0:m0 + 8].to(DEV), ref[m0:m0 + 8])
    assert_bf16_close(out, ref.float(), ulps=1, atol=3e-3, what="w4_gemm vs gemv")


@pytest.mark.parametrize("group,epi", [(128, EPI_SWIGLU), (64, EPI_RESIDUAL), (-1, EPI_GELU), (64, EPI_SWIGLU)])
def test_w4_gemm_lds_dma_kernel_unsplit_launch_and_single_step_groups(group, epi):
    """gemm2_w4_kernel without split-K (>= 192 tiles: the in-kernel epilogues, the two-pass SwiGLU with the gate kept in
    registers), groups of one K-step (64), of two (128) and per-channel; last row tile ragged."""
    M, N, K = 1500, 2048, 256
    g = gen(37)
    x, res = torch.randn(M, K, generator=g).to(BF), torch.randn(M, N, generator=g).to(BF)
    bias = None if epi == EPI_SWIGLU else (torch.randn(N, generator=g) * 0.1).to(BF)
    qw, s, z, tc, Wd = make_w4(N, K, group, 38)
    qw2, s2, z2, _, Wd2 = make_w4(N, K, group, 39)
    lin, lin2 = w4_module(qw, s, z, N, K, group, bias), w4_module(qw2, s2, z2, N, K, group)
    out = torch.full((M + 1, N), 7.0, dtype=BF, device=DEV)
    lin.hip_linear(x.to(DEV), out[:M], epilogue=epi, residual=res.to(DEV) if epi == EPI_RESIDUAL else None,
                   partner=lin2 if epi == EPI_SWIGLU else None)
    want = expected_epilogue(x.double() @ Wd.t(), x.double() @ Wd2.t(), bias, res, epi)
    # every epilogue here applies a function to the Linear's result AFTER its rounding to bf16 (a reference rounding point): where the
    # fp32 sum sits on a rounding tie, one ulp of that intermediate (2^-8 |v|, |v| <~ 2) moves the output by up to 8e-3 - with 3 M
    # outputs a handful do
    assert_bf16_close(out[:M], want, ulps=1, atol=8e-3, what=f"w4 gemm2 unsplit g={group} epi {epi}")
    assert float((out[:M].cpu().double() - want).abs().mean()) < 2e-3
    assert torch.all(out[M] == 7.0), "wrote past the last row"


@pytest.mark.parametrize("kind", [1, 2])
@pytest.mark.parametrize("M,N,K,group", [(200, 256, 4096, 128), (129, 96, 512, 64), (640, 2048, 11008, 128)])
def test_w4_gemm_with_the_norm_fused_in_front(kind, M, N, K, group):
    """parrot_w4_gemm with a norm argument (one launch: normalised rows + their per-group sums) returns the bits of the stand-alone
    norm kernel followed by the GEMM's own activation-sum pre-pass - RMSNorm and LayerNorm, split and unsplit launches, a ragged
    last row tile, groups of 64 and 128."""
    from lit_parrot_amd import ops

    g = gen(51)
    x = (torch.randn(M, K, generator=g) * 1.5 + 0.1).to(BF).to(DEV)
    w = (1 + 0.1 * torch.randn(K, generator=g)).to(BF).to(DEV)
    b = (0.1 * torch.randn(K, generator=g)).to(BF).to(DEV) if kind == 2 else None
    qw, s, z, tc, Wd = make_w4(N, K, group, 52)
    lin = w4_module(qw, s, z, N, K, group)
    norm = ops.Norm(kind, w, b, 1e-5)
    outs = []
    for fused in (True, False):
        ops.W4_GEMM_FUSED_NORM = fused
        try:
            out = torch.empty((M + 1, N), dtype=BF, device=DEV).fill_(3.0)
            lin.hip_linear(x, out[:M], norm=norm)
            outs.append(out)
        finally:
            ops.W4_GEMM_FUSED_NORM = True
    assert torch.equal(outs[0], outs[1]), f"{int((outs[0] != outs[1]).sum())} elements differ between the fused and the stand-alone norm"
    assert torch.all(outs[0][M] == 3.0)
    xn = torch.empty_like(x)
    (ops.rmsnorm(x, w, 1e-5, xn) if kind == 1 else ops.layernorm(x, w, b, 1e-5, xn))
    want = rbf(xn.cpu().double() @ Wd.t())
    assert_bf16_close(outs[0][:M], want, ulps=1, atol=2e-2, what="w4 gemm behind a fused norm")


@pytest.mark.parametrize("epi", [EPI_NONE, EPI_SWIGLU])
def test_w4_gemm_long_unsplit_loop_repeats_bit_identically(epi):
    """The unsplit int4 prompt kernel over a long K loop (32 steps, a slab crossing), called four times on the same operands:
    the same bits every time, and the right ones.  (An expansion issued from inline asm one wait state in front of the MFMA
    that reads it gave run-to-run different results in one 32 x 32 sub-tile per wave - only in loops long enough for the
    matrix pipe to run dry, which the 4-step launches of the test above never are.)"""
    M, N, K, group = 640, 8192, 2048, 128
    g = gen(41)
    x = torch.randn(M, K, generator=g).to(BF)
    qw, s, z, tc, Wd = make_w4(N, K, group, 42)
    qw2, s2, z2, _, Wd2 = make_w4(N, K, group, 43)
    lin, lin2 = w4_module(qw, s, z, N, K, group), w4_module(qw2, s2, z2, N, K, group)
    outs = []
    for rep in range(4):
        out = torch.empty((M, N), dtype=BF, device=DEV)
        lin.hip_linear(x.to(DEV), out, epilogue=epi, partner=lin2 if epi == EPI_SWIGLU else None)
        outs.append(out)
    for o in outs[1:]:
        assert torch.equal(outs[0], o), f"{int((outs[0] != o).sum())} elements differ between two calls"
    want = expected_epilogue(x.double() @ Wd.t(), x.double() @ Wd2.t() if epi == EPI_SWIGLU else None, None, None, epi)
    assert_bf16_close(outs[0], want, ulps=1, atol=6e-2 if epi == EPI_SWIGLU else 1.5e-2, what=f"w4 gemm2 unsplit long K epi {epi}")
    assert float((outs[0].cpu().double() - want).abs().mean()) < 3e-3


@pytest.mark.parametrize("epi", [EPI_RESIDUAL, EPI_GELU, EPI_SWIGLU])
@pytest.mark.parametrize("kind", ["w4", "bf16"])
def test_gemm_epilogues(epi, kind):
    N, K, M = 160, 512, 70
    g = gen(32)
    x, res = torch.randn(M, K, generator=g).to(BF), torch.randn(M, N, generator=g).to(BF)
    bias = None if epi == EPI_SWIGLU else (torch.randn(N, generator=g) * 0.1).to(BF)
    out = torch.empty((M, N), dtype=BF, device=DEV)
    if kind == "w4":
        qw, s, z, tc, Wd = make_w4(N, K, 128, 33)
        qw2, s2, z2, _, Wd2 = make_w4(N, K, 128, 34)
        lin, lin2 = w4_module(qw, s, z, N, K, 128, bias), w4_module(qw2, s2, z2, N, K, 128)
        lin.hip_linear(x.to(DEV), out, epilogue=epi, residual=res.to(DEV) if epi == EPI_RESIDUAL else None,
                       partner=lin2 if epi == EPI_SWIGLU else None)
    else:
        W, W2 = ((torch.randn(N, K, generator=g) * 0.05).to(BF) for _ in range(2))
        Wd, Wd2 = W.double(), W2.double()
        ops.bf16_linear(W.to(DEV), x.to(DEV), out, bias=bias.to(DEV) if bias is not None else None, epilogue=epi,
                        residual=res.to(DEV) if epi == EPI_RESIDUAL else None, weight2=W2.to(DEV) if epi == EPI_SWIGLU else None)
    want = expected_epilogue(x.double() @ Wd.t(), x.double() @ Wd2.t(), bias, res, epi)
    assert_bf16_close(out, want, ulps=1, atol=3e-3, what=f"{kind} gemm epilogue {epi}")


@pytest.mark.parametrize("N,K", [(64, 128), (300, 768), (128, 4096), (1000, 352)])
@pytest.mark.parametrize("M", [9, 64, 130])
def test_bf16_gemm_matches_oracle(N, K, M):
    g = gen(35)
    W = (torch.randn(N, K, generator=g) * 0.02).to(BF)
    x = torch.randn(M, K, generator=g).to(BF)
    bias = (torch.randn(N, generator=g) * 0.1).to(BF)
    out = torch.empty((M, N), dtype=BF, device=DEV)
    ops.bf16_linear(W.to(DEV), x.to(DEV), out, bias=bias.to(DEV))
    assert_bf16_close(out, rbf(x.double() @ W.double().t() + bias.double()), ulps=1, atol=1e-3, what=f"bf16 gemm {M}x{N}x{K}")
    norm = ops.Norm(2, (1 + 0.1 * torch.randn(K, generator=g)).to(BF).to(DEV), (0.1 * torch.randn(K, generator=g)).to(BF).to(DEV), 1e-5)
    a, b = torch.empty_like(out), torch.empty_like(out)
    ops.bf16_linear(W.to(DEV), x.to(DEV), a, norm=norm)  # prefill: stand-alone norm kernel, then the GEMM
    for m0 in range(0, M, 8):
        ops.bf16_linear(W.to(DEV), x[m0:m0 + 8].to(DEV), b[m0:m0 + 8], norm=norm)  # decode: fused prologue
    assert_bf16_close(a, b.float(), ulps=1, atol=2e-3, what="norm + gemm vs fused gemv")


@pytest.mark.parametrize("M,N,K,epi", [(512, 1024, 4096, EPI_NONE), (300, 389, 1024, EPI_RESIDUAL), (129, 3200, 512, EPI_GELU),
                                       (1000, 256, 2048, EPI_NONE), (40, 128, 64, EPI_RESIDUAL)])
def test_bf16_gemm_lds_dma_kernel_tiles_splits_and_ragged_edges(M, N, K, epi):
    """The 128 x 128 x 64 LDS-DMA kernel (gemm2.hip): K split (few tiles), no split (many tiles), rows / columns that do not
    fill the last tile, a single K-step; every epilogue it takes."""
    g = gen(36)
    W = (torch.randn(N, K, generator=g) * 0.02).to(BF)
    x = torch.randn(M, K, generator=g).to(BF)
    bias = (torch.randn(N, generator=g) * 0.1).to(BF)
    res = torch.randn(M, N, generator=g).to(BF)
    out = torch.full((M + 1, N), 7.0, dtype=BF, device=DEV)  # one guard row behind the result
    ops.bf16_linear(W.to(DEV), x.to(DEV), out[:M], bias=bias.to(DEV), epilogue=epi, residual=res.to(DEV) if epi == EPI_RESIDUAL else None)
    want = expected_epilogue(x.double() @ W.double().t(), None, bias, res, epi)
    # residual: the Linear's result is rounded to bf16 BEFORE the add (a reference rounding point); where the sum cancels, one
    # ulp of that intermediate (2^-8 of |acc| <~ 2) is many ulps of the small result
    assert_bf16_close(out[:M], want, ulps=1, atol=8e-3 if epi == EPI_RESIDUAL else 2e-3, what=f"bf16 gemm2 {M}x{N}x{K} epi {epi}")
    assert torch.all(out[M] == 7.0), "wrote past the last row"


@pytest.mark.parametrize("kind", ["bf16", "w4"])
def test_prompt_gemm_tile_order_covers_every_tile_exactly_once(kind):
    """The XCD-aware workgroup -> (m tile, n tile, K split) map of gemm2.hip must be a bijection for every tile count: a sweep of
    row / column counts (n-tile counts that are and are not multiples of 8, m-tile counts with and without small divisors, split and
    unsplit launches) against a float64 product computed with torch on the device.  A tile computed twice is harmless, a tile never
    computed shows up as the buffer's fill value."""
    g = gen(40)
    K = 128
    for M in (129, 384, 640, 896, 1300, 2048):
        for N in (128, 1024, 1152, 2944, 9216):
            x = torch.randn(M, K, generator=g).to(BF).to(DEV)
            out = torch.full((M, N), 777.0, dtype=BF, device=DEV)
            if kind == "bf16":
                W = (torch.randn(N, K, generator=g) * 0.05).to(BF).to(DEV)
                ops.bf16_linear(W, x, out)
                want = x.double() @ W.double().t()
            else:
                qw, s_, z_, tc, Wd = make_w4(N, K, 128, 41)
                lin = w4_module(qw, s_, z_, N, K, 128)
                lin.hip_linear(x, out)
                want = x.double() @ Wd.to(DEV).t()
            err = (out.double() - want).abs()
            assert float(err.max()) <= 2 ** -7 * max(1.0, float(want.abs().max())), (kind, M, N, float(err.max()))


# ------------------------------------------------------------------------------------------------ dense bf16
@pytest.mark.parametrize("N,K", [(64, 128), (100, 768), (64, 4096), (32, 16384), (16, 3072), (8, 32768), (40, 352)])
@pytest.mark.parametrize("M", [1, 2, 3])
def test_bf16_gemv_matches_oracle(N, K, M):
    g = gen(10)
    W = (torch.randn(N, K, generator=g) * 0.02).to(BF)
    x = torch.randn(M, K, generator=g).to(BF)
    bias = (torch.randn(N, generator=g) * 0.1).to(BF)
    out = torch.empty((M, N), dtype=BF, device=DEV)
    ops.bf16_linear(W.to(DEV), x.to(DEV), out, bias=bias.to(DEV))
    assert_bf16_close(out, rbf(x.double() @ W.double().t() + bias.double()), ulps=1, atol=1e-3, what=f"bf16 gemv {N}x{K}")
    # and torch's own bf16 Linear on the CPU (what the reference runs)
    assert float((out.cpu().float() - F.linear(x, W, bias).float()).abs().max()) <= 1e-2


@pytest.mark.parametrize("epi", [EPI_RESIDUAL, EPI_GELU, EPI_SWIGLU])
def test_bf16_epilogues(epi):
    N, K, M = 72, 1024, 2
    g = gen(11)
    W, W2 = ((torch.randn(N, K, generator=g) * 0.05).to(BF) for _ in range(2))
    x, res = torch.randn(M, K, generator=g).to(BF), torch.randn(M, N, generator=g).to(BF)
    out = torch.empty((M, N), dtype=BF, device=DEV)
    ops.bf16_linear(W.to(DEV), x.to(DEV), out, epilogue=epi, residual=res.to(DEV) if epi == EPI_RESIDUAL else None,
                    weight2=W2.to(DEV) if epi == EPI_SWIGLU else None)
    want = expected_epilogue(x.double() @ W.double().t(), x.double() @ W2.double().t(), None, res, epi)
    assert_bf16_close(out, want, ulps=1, atol=1e-3, what=f"bf16 epilogue {epi}")


# ------------------------------------------------------------------------------------------------ LLM.int8
def test_w8_weight_quantisation_is_exact():
    g = gen(12)
    W = (torch.randn(50, 352, generator=g) * 0.02).to(BF)
    W[7] = 0
    CB = torch.empty((50, 352), dtype=torch.int8, device=DEV)
    SCB = torch.empty((50,), dtype=torch.float32, device=DEV)
    ops.w8_quantize_rows(W.to(DEV), CB, SCB)
    cb, scb = o8.quantize_weight_rows(W)
    assert torch.equal(CB.cpu(), cb) and torch.equal(SCB.cpu(), scb)
    # fp16 and fp32 checkpoints (Llama-2 HF weights are fp16): the reference quantises `weight.half()` (quantize/bnb.py:54), so
    # values that bf16 cannot hold must reach the quantiser with their fp16 mantissa
    for dt in (torch.float16, torch.float32):
        Wd = (torch.randn(50, 352, generator=g) * 0.02).to(dt)
        assert not torch.equal(Wd.to(BF).to(dt), Wd)
        ops.w8_quantize_rows(Wd.to(DEV), CB, SCB)
        cb, scb = o8.quantize_weight_rows(Wd)
        assert torch.equal(CB.cpu(), cb) and torch.equal(SCB.cpu(), scb), dt
        cb_bf, _ = o8.quantize_weight_rows(Wd.to(BF))
        assert not torch.equal(cb_bf, cb), "the bf16 detour would have changed the int8 weights"
    from lit_parrot_amd.quantize.bnb import InferenceLinear8bitLt

    lin = InferenceLinear8bitLt(352, 50, bias=False)
    Wd = (torch.randn(50, 352, generator=g) * 0.02)
    lin.load_state_dict({"weight": Wd})
    cb, scb = o8.quantize_weight_rows(Wd)
    assert torch.equal(lin.weight.data.cpu(), cb) and torch.equal(lin.weight.SCB.cpu(), scb)


@pytest.mark.parametrize("N,K", [(64, 256), (96, 4096), (40, 352), (16, 11008)])
@pytest.mark.parametrize("outliers", [False, True])
@pytest.mark.parametrize("M", [1, 3, 9, 70])  # 9 and 70 rows: the int8 MFMA GEMM (when K is a multiple of 64), else the GEMV per row
def test_w8_linear_matches_oracle(N, K, outliers, M):
    g = gen(13)
    W = (torch.randn(N, K, generator=g) * 0.02).to(BF)
    x = torch.randn(M, K, generator=g).to(BF)
    if outliers:  # force the mixed-precision decomposition: |x| >= 6 in a few columns, in different rows
        x[0, 3], x[0, K - 1], x[M - 1, 17], x[M // 2, 40] = 9.5, -7.25, 6.0, -11.0
    bias = (torch.randn(N, generator=g) * 0.1).to(BF)
    cb, scb = o8.quantize_weight_rows(W)
    act = ops.w8_prep_act(x.to(DEV), 6.0, ops.W8Act(M, K, DEV))
    ca, sca, sub_a = o8.quantize_act_rows(x, 6.0)
    # outlier COLUMNS of the whole call (LLM.int8's feature dimensions): cleared in every row's int8 copy, listed for every row
    cols = (x.half().float().abs() >= 6.0).any(dim=0)
    assert int(cols.sum()) == (4 if outliers else 0) or M == 1
    assert torch.equal(act.xq.cpu(), ca) and torch.equal(act.sca.cpu(), sca)
    assert torch.equal(act.xout.cpu()[:, cols], sub_a[:, cols]) and torch.equal(act.xout.cpu(), x.half().float())
    listed = cols.nonzero().flatten().tolist()
    for m in range(M):
        n = int(act.nout[m])
        assert sorted(act.oidx[m, :n].cpu().tolist()) == listed, m
        if M > 1:
            assert act.oidx[m, :n].cpu().tolist() == listed
    out = torch.empty((M, N), dtype=BF, device=DEV)
    ops.w8_linear(cb.to(DEV), scb.to(DEV), N, K, act, out, bias=bias.to(DEV))
    want = o8.linear(x, cb, scb, bias, 6.0)
    # integer accumulate is exact; the fp16 roundings make it bit-exact up to fp32 summation order in the outlier part
    assert_bf16_close(out, want.float(), ulps=1 if outliers else 0, atol=0.0, what=f"w8 {N}x{K} outliers={outliers}")


@pytest.mark.parametrize("M,N,K,epi", [(130, 300, 512, EPI_NONE), (64, 520, 11008, EPI_RESIDUAL), (300, 2300, 256, EPI_SWIGLU), (1100, 3000, 128, EPI_GELU)])
def test_w8_prompt_gemm_on_the_lds_dma_structure_equals_the_first_generation(M, N, K, epi):
    """parrot_w8_gemm (128 x 128 tiles by LDS-DMA, exact int32 sums to a workspace, element-wise dequantise / outlier / epilogue pass;
    split and unsplit K, ragged tiles, SwiGLU over [fc_1; fc_2]) against the first-generation int8 GEMM and the oracle: the integer
    part is order-free and the element-wise arithmetic is the same code, so the two agree bit for bit."""
    g = gen(14)
    swi = epi == EPI_SWIGLU
    W = (torch.randn(N * (2 if swi else 1), K, generator=g) * 0.02).to(BF)
    x = torch.randn(M, K, generator=g).to(BF)
    x[0, 3], x[M // 2, K - 1], x[M - 1, 17] = 9.5, -7.25, 6.0
    res = torch.randn(M, N, generator=g).to(BF)
    bias = None if swi else (torch.randn(N, generator=g) * 0.1).to(BF)
    cb, scb = o8.quantize_weight_rows(W)
    act = ops.w8_prep_act(x.to(DEV), 6.0, ops.W8Act(M, K, DEV))
    outs = []
    for new in (True, False):
        ops.W8_PREFILL_GEMM2 = new
        try:
            out = torch.full((M + 1, N), 7.0, dtype=BF, device=DEV)
            ops.w8_linear(cb.to(DEV), scb.to(DEV), N, K, act, out[:M], bias=bias.to(DEV) if bias is not None else None, epilogue=epi,
                          residual=res.to(DEV) if epi == EPI_RESIDUAL else None)
        finally:
            ops.W8_PREFILL_GEMM2 = True
        assert torch.all(out[M] == 7.0)
        outs.append(out[:M].clone())
    # the integer part is order-free and the element-wise arithmetic is the same code; the mixed-precision part is summed by an fp16
    # MFMA GEMM over the call's outlier columns in one path and serially in the other: one bf16 ulp at most
    assert_bf16_close(outs[0], outs[1].float(), ulps=1, atol=0.0, what="w8 gemm2 vs first generation")
    if epi == EPI_NONE:
        assert_bf16_close(outs[0], o8.linear(x, cb, scb, bias, 6.0).float(), ulps=1, atol=0.0, what="w8 gemm2 vs oracle")


@pytest.mark.parametrize("N,K", [(64, 256), (96, 4096), (40, 352), (16, 11008), (20000, 512), (24, 16384)])
@pytest.mark.parametrize("outliers", [0, 3, 400])
def test_w8_fused_single_row_matches_oracle_and_the_two_launch_path(N, K, outliers):
    """parrot_w8_gemv_fused (activation quantiser + GEMV in one launch, decode) against the oracle and against
    parrot_w8_prep_act + parrot_w8_gemv; 400 outlier columns overflow the in-LDS list (the epilogue walks the row)."""
    g = gen(23)
    W = (torch.randn(N, K, generator=g) * 0.02).to(BF)
    x = torch.randn(1, K, generator=g).to(BF)
    if outliers:
        idx = torch.randperm(K, generator=g)[: min(outliers, K // 2)]
        x[0, idx] = (6.0 + torch.rand(idx.numel(), generator=g) * 4).to(BF) * torch.where(torch.rand(idx.numel(), generator=g) < 0.5, -1.0, 1.0).to(BF)
    bias = (torch.randn(N, generator=g) * 0.1).to(BF)
    cb, scb = o8.quantize_weight_rows(W)
    out = torch.empty((1, N), dtype=BF, device=DEV)
    ops.w8_linear_fused(cb.to(DEV), scb.to(DEV), N, K, x.to(DEV), 6.0, out, bias=bias.to(DEV))
    want = o8.linear(x, cb, scb, bias, 6.0)
    assert_bf16_close(out, want.float(), ulps=1 if outliers else 0, atol=0.0, what=f"w8 fused {N}x{K} outliers={outliers}")
    act = ops.w8_prep_act(x.to(DEV), 6.0, ops.W8Act(1, K, DEV))
    out2 = torch.empty_like(out)
    ops.w8_linear(cb.to(DEV), scb.to(DEV), N, K, act, out2, bias=bias.to(DEV))
    assert_bf16_close(out, out2.float(), ulps=1 if outliers else 0, atol=0.0, what="w8 fused vs two launches")


@pytest.mark.parametrize("epi", [EPI_RESIDUAL, EPI_GELU, EPI_SWIGLU])
@pytest.mark.parametrize("kind", [0, 1, 2])
def test_w8_fused_epilogues_and_norms(epi, kind):
    N, K = 200, 768
    g = gen(24)
    W = (torch.randn(N, K, generator=g) * 0.02).to(BF)
    W2 = (torch.randn(N, K, generator=g) * 0.02).to(BF)
    x = (torch.randn(1, K, generator=g) * 2).to(BF)
    x[0, 5] = 30.0  # an outlier survives the norm
    res = torch.randn(1, N, generator=g).to(BF)
    cb, scb = o8.quantize_weight_rows(W)
    cb2, scb2 = o8.quantize_weight_rows(W2)
    CB = torch.cat([cb, cb2]).contiguous().to(DEV) if epi == EPI_SWIGLU else cb.to(DEV)
    SCB = torch.cat([scb, scb2]).contiguous().to(DEV) if epi == EPI_SWIGLU else scb.to(DEV)
    norm = None
    if kind:
        norm = ops.Norm(kind, (1 + 0.1 * torch.randn(K, generator=g)).to(BF).to(DEV),
                        (0.1 * torch.randn(K, generator=g)).to(BF).to(DEV) if kind == 2 else None, 1e-5)
    kw = dict(epilogue=epi, residual=res.to(DEV) if epi == EPI_RESIDUAL else None)
    out = torch.empty((1, N), dtype=BF, device=DEV)
    ops.w8_linear_fused(CB, SCB, N, K, x.to(DEV), 6.0, out, norm=norm, **kw)
    act = ops.w8_prep_act(x.to(DEV), 6.0, ops.W8Act(1, K, DEV), norm)
    out2 = torch.empty_like(out)
    ops.w8_linear(CB, SCB, N, K, act, out2, **kw)
    # same arithmetic; only the summation order of the norm statistic / the outlier part may differ
    d = (out.float() - out2.float()).abs()
    assert float(d.max()) <= 2 ** -5 * max(1.0, float(out2.float().abs().max())) and float((d == 0).float().mean()) > 0.9


# ------------------------------------------------------------------------------------------------ norms
@pytest.mark.parametrize("d", [128, 768, 4096, 8192])
def test_rmsnorm_matches_reference_choreography(d):
    g = gen(14)
    x = (torch.randn(3, d, generator=g) * 2).to(BF)
    w = (1 + 0.1 * torch.randn(d, generator=g)).to(BF)
    out = torch.empty((3, d), dtype=BF, device=DEV)
    ops.rmsnorm(x.to(DEV), w.to(DEV), 1e-5, out)
    # The reference's op sequence with torch's *GPU* semantics for rsqrt: computed in fp32 and rounded once.  torch's CPU
    # kernel takes a scalar path for tensors of < 16 elements (one value per token row here) that rounds sqrt() to bf16
    # before the reciprocal, so the CPU oracle can sit 1-2 bf16 ulp away on whole rows; see DESIGN.md §6.
    xf = x.float()
    ms = rbf((rbf(xf * xf)).sum(-1, keepdim=True) / d)
    r = rbf(1.0 / torch.sqrt(rbf(ms + 1e-5)))
    want = w.float() * rbf(xf * r)
    assert_bf16_close(out, want, ulps=1, what="rmsnorm")  # <= 1 ulp: the fp32 sum of squares is ordered differently
    assert float((out.cpu().float() == want.to(BF).float()).float().mean()) > 0.98
    assert_bf16_close(out, om.rms_norm(x, w, 1e-5).float(), ulps=3, what="rmsnorm vs CPU oracle")
    # rsqrt_mode 1 reproduces the CPU scalar path: now the CPU oracle itself is matched to the ulp
    try:
        ops.RMSNORM_RSQRT_MODE = 1
        ops.rmsnorm(x.to(DEV), w.to(DEV), 1e-5, out)
    finally:
        ops.RMSNORM_RSQRT_MODE = 0
    assert_bf16_close(out, om.rms_norm(x, w, 1e-5).float(), ulps=1, what="rmsnorm (cpu rsqrt mode) vs CPU oracle")
    assert float((out.cpu() == om.rms_norm(x, w, 1e-5)).float().mean()) > 0.98


@pytest.mark.parametrize("d", [128, 768, 4096, 8192])
def test_layernorm_matches_torch(d):
    g = gen(15)
    x = (torch.randn(3, d, generator=g) * 2 + 0.3).to(BF)
    w, b = (1 + 0.1 * torch.randn(d, generator=g)).to(BF), (0.1 * torch.randn(d, generator=g)).to(BF)
    out = torch.empty((3, d), dtype=BF, device=DEV)
    ops.layernorm(x.to(DEV), w.to(DEV), b.to(DEV), 1e-5, out)
    want = F.layer_norm(x.double(), (d,), w.double(), b.double(), 1e-5)
    assert_bf16_close(out, want, ulps=1, atol=1e-3, what="layernorm")
    assert float((out.cpu().float() - F.layer_norm(x, (d,), w, b, 1e-5).float()).abs().max()) <= 2 ** -6


# ------------------------------------------------------------------------------------------------ attention
ATTN_SHAPES = [  # n_groups, q_per_kv, hs, n_elem
    (4, 1, 32, 8), (2, 1, 64, 64), (2, 2, 64, 64), (2, 1, 128, 128), (2, 4, 32, 32), (1, 4, 32, 32), (2, 16, 64, 64), (3, 1, 128, 32),
]


def ref_attention(qkv, cos, sin, pos0, n_groups, q_per_kv, hs, n_elem, S, kc, vc):
    """The reference's split / rope / cache append / SDPA (model.py:208-247) in float64 on bf16 inputs; kc, vc are the
    (n_groups, S, hs) caches BEFORE the call (updated in place, slot = position)."""
    T = qkv.shape[0]
    v5 = qkv.view(T, n_groups, q_per_kv + 2, hs)
    q, k, v = v5[:, :, :q_per_kv], v5[:, :, q_per_kv], v5[:, :, q_per_kv + 1]
    pos = torch.arange(pos0, pos0 + T)
    c, s = cos.index_select(0, pos), sin.index_select(0, pos)  # (T, n_elem) fp16
    qr = torch.cat((om.apply_rope(q[..., :n_elem].permute(1, 2, 0, 3), c, s), q[..., n_elem:].permute(1, 2, 0, 3)), dim=-1)  # (G, qpk, T, hs)
    kr = torch.cat((om.apply_rope(k[..., :n_elem].permute(1, 0, 2), c, s), k[..., n_elem:].permute(1, 0, 2)), dim=-1)  # (G, T, hs)
    kc[:, pos] = kr
    vc[:, pos] = v.permute(1, 0, 2)
    y = torch.empty((T, n_groups * q_per_kv * hs), dtype=torch.float64)
    for t in range(T):
        n_valid = pos0 + t + 1
        K_, V_ = kc[:, :n_valid].double(), vc[:, :n_valid].double()
        att = torch.einsum("gqd,gsd->gqs", qr[:, :, t].double(), K_) / math.sqrt(hs)
        y[t] = torch.einsum("gqs,gsd->gqd", att.softmax(-1), V_).reshape(-1)
    return qr, y


@pytest.mark.parametrize("n_groups,q_per_kv,hs,n_elem", ATTN_SHAPES)
@pytest.mark.parametrize("S,nsplit", [(16, 1), (96, 1), (96, 3), (700, 8)])
def test_rope_append_and_attention(n_groups, q_per_kv, hs, n_elem, S, nsplit):
    g = gen(16)
    n_head = n_groups * q_per_kv
    width = n_groups * (q_per_kv + 2) * hs
    cos, sin = om.rope_tables(1024, n_elem, BF, math_dtype=BF)
    T = min(9, S)
    kc = torch.zeros((n_groups, S, hs), dtype=BF)
    vc = torch.zeros((n_groups, S, hs), dtype=BF)
    kc_d, vc_d = kc.to(DEV), vc.to(DEV)
    q_d = torch.empty((T, n_head * hs), dtype=BF, device=DEV)
    y_d = torch.empty((T, n_head * hs), dtype=BF, device=DEV)
    ws = ops.attn_workspace(T, n_head, hs, nsplit, DEV)
    cos_d, sin_d = cos.to(DEV), sin.to(DEV)
    pos0 = 0
    for step, rows in enumerate([T, 1, 1, 1]):  # a prefill of T rows, then single-token steps
        if pos0 + rows > S:
            break
        qkv = torch.randn(rows, width, generator=g).to(BF)
        qr, want = ref_attention(qkv, cos, sin, pos0, n_groups, q_per_kv, hs, n_elem, S, kc, vc)
        pos_d = torch.tensor([pos0], dtype=torch.int32, device=DEV)
        ops.rope_kvappend(qkv.to(DEV), cos_d, sin_d, n_elem, pos_d, n_groups, q_per_kv, hs, S, q_d[:rows], kc_d, vc_d)
        # roped q and the cache contents are bit-exact (same fp32 ops, no FMA)
        assert torch.equal(q_d[:rows].cpu().view(rows, n_groups, q_per_kv, hs), qr.permute(2, 0, 1, 3))
        assert torch.equal(kc_d.cpu(), kc) and torch.equal(vc_d.cpu(), vc)
        ops.attn_decode(q_d[:rows], pos_d, kc_d, vc_d, n_groups, q_per_kv, hs, S, nsplit, ws, y_d[:rows])
        assert_bf16_close(y_d[:rows], want, ulps=1, atol=2e-3, what=f"attention step {step}")
        pos0 += rows


@pytest.mark.parametrize("n_groups,q_per_kv,hs,n_elem", [(2, 1, 64, 64), (2, 2, 64, 64), (2, 1, 128, 128), (1, 4, 32, 32), (3, 1, 128, 32)])
@pytest.mark.parametrize("S", [16, 300, 512])
def test_softmax_mode_1_is_the_reference_kernels_arithmetic(n_groups, q_per_kv, hs, n_elem, S):
    """softmax_mode 1 of parrot_attn_decode / parrot_attn_fused_decode against a float64 restatement of what torch's CPU flash
    kernel does with bf16 operands: p = exp(s - rowmax) rounded to bf16 before P.V, the denominator from the unrounded p.
    Heads within 1 bf16 ulp (the sums run in another order) and mostly bit-identical; both kernels agree with each other;
    windows beyond one 512-key block are refused."""
    g = gen(51)
    n_head, width = n_groups * q_per_kv, n_groups * (q_per_kv + 2) * hs
    cos, sin = (t.to(DEV) for t in om.rope_tables(2048, n_elem, BF, math_dtype=BF))
    kc = torch.randn((n_groups, S, hs), generator=g).to(BF).to(DEV); vc = torch.randn((n_groups, S, hs), generator=g).to(BF).to(DEV)
    kc2, vc2 = kc.clone(), vc.clone()
    q = torch.empty((1, n_head * hs), dtype=BF, device=DEV)
    y1, y2 = torch.empty_like(q), torch.empty_like(q)
    ws = ops.attn_workspace(1, n_head, hs, 1, DEV)
    tickets = torch.zeros((n_head,), dtype=torch.int32, device=DEV)
    ops.ATTN_SOFTMAX_MODE = 1
    try:
        for pos in (0, 3, S - 2, S - 1):
            qkv = torch.randn(1, width, generator=g).to(BF).to(DEV)
            pos_d = torch.tensor([pos], dtype=torch.int32, device=DEV)
            ops.rope_kvappend(qkv, cos, sin, n_elem, pos_d, n_groups, q_per_kv, hs, S, q, kc, vc)
            ops.attn_decode(q, pos_d, kc, vc, n_groups, q_per_kv, hs, S, 1, ws, y1)
            ops.attn_fused_decode(qkv, cos, sin, n_elem, pos_d, kc2, vc2, n_groups, q_per_kv, hs, S, 1, ws, tickets, y2)
            assert torch.equal(kc, kc2) and torch.equal(vc, vc2)
            n = pos + 1
            qd = q.cpu().double().view(n_groups, q_per_kv, hs)
            K, V = kc.cpu().double()[:, :n], vc.cpu().double()[:, :n]
            sc = torch.einsum("gqd,gsd->gqs", qd, K).float() * (1.0 / math.sqrt(hs))  # fp32 scores, scaled after the dot
            p = torch.exp(sc - sc.amax(dim=-1, keepdim=True))
            want = torch.einsum("gqs,gsd->gqd", p.to(BF).double(), V) / p.double().sum(-1, keepdim=True)
            for name_, got in (("attn_decode", y1), ("attn_fused_decode", y2)):
                assert_bf16_close(got.view(-1), want.reshape(-1), ulps=1, atol=2e-3, what=f"{name_} softmax_mode 1, pos {pos}")
                same = float((got.cpu().view(-1) == want.reshape(-1).to(BF)).float().mean())
                assert same > 0.9, f"{name_} pos {pos}: only {same:.3f} bit-identical to the restated reference arithmetic"
            assert float((y1 == y2).float().mean()) > 0.95
        big = torch.zeros((n_groups, 513, hs), dtype=BF, device=DEV)
        with pytest.raises(ParrotHipError, match="softmax_mode 1"):
            ops.attn_decode(q, pos_d, big, big, n_groups, q_per_kv, hs, 513, 1, ws, y1)
    finally:
        ops.ATTN_SOFTMAX_MODE = 0


def test_attention_ring_window_equals_rolled_cache():
    """pos >= S: slot = pos % S replaces the oldest key — the reference's roll-left + write-last (model.py:238-245)."""
    n_groups, q_per_kv, hs, n_elem, S = 2, 2, 64, 64, 10
    g = gen(17)
    cos, sin = om.rope_tables(64, n_elem, BF, math_dtype=BF)
    kc_d = torch.zeros((n_groups, S, hs), dtype=BF, device=DEV)
    vc_d = torch.zeros_like(kc_d)
    n_head, width = n_groups * q_per_kv, n_groups * (q_per_kv + 2) * hs
    q_d = torch.empty((1, n_head * hs), dtype=BF, device=DEV)
    y_d = torch.empty_like(q_d)
    ks, vs = [], []
    for pos in range(25):
        qkv = torch.randn(1, width, generator=g).to(BF)
        # oracle: keep the last S (roped) keys/values explicitly
        kfull = torch.zeros((n_groups, pos + 1, hs), dtype=BF)
        vfull = torch.zeros_like(kfull)
        qr, _ = ref_attention(qkv, cos, sin, pos, n_groups, q_per_kv, hs, n_elem, pos + 1, kfull, vfull)
        ks.append(kfull[:, pos]); vs.append(vfull[:, pos])
        K_ = torch.stack(ks[-S:], 1).double(); V_ = torch.stack(vs[-S:], 1).double()
        att = torch.einsum("gqd,gsd->gqs", qr[:, :, 0].double(), K_) / math.sqrt(hs)
        want = torch.einsum("gqs,gsd->gqd", att.softmax(-1), V_).reshape(1, -1)
        pos_d = torch.tensor([pos], dtype=torch.int32, device=DEV)
        ops.rope_kvappend(qkv.to(DEV), cos.to(DEV), sin.to(DEV), n_elem, pos_d, n_groups, q_per_kv, hs, S, q_d, kc_d, vc_d)
        ops.attn_decode(q_d, pos_d, kc_d, vc_d, n_groups, q_per_kv, hs, S, 1, None, y_d)
        assert_bf16_close(y_d, want, ulps=1, atol=2e-3, what=f"ring pos {pos}")


@pytest.mark.parametrize("kind", ["rms", "ln"])
@pytest.mark.parametrize("M", [1, 3])
@pytest.mark.parametrize("K", [256, 4096, 11008])
def test_norm_fused_into_linear_equals_norm_then_linear(kind, M, K):
    """The norm prologue of the GEMV kernels (norm_1 / norm_2 / ln_f fused into the next Linear) reproduces the
    stand-alone norm kernel followed by the plain Linear (same rounding points; only the fp32 sums are re-ordered)."""
    N = 64
    g = gen(20)
    x = (torch.randn(M, K, generator=g) * 1.5 + 0.1).to(BF).to(DEV)
    nw = (1 + 0.1 * torch.randn(K, generator=g)).to(BF).to(DEV)
    nb = (0.1 * torch.randn(K, generator=g)).to(BF).to(DEV) if kind == "ln" else None
    norm = ops.Norm(1 if kind == "rms" else 2, nw, nb, 1e-5)
    xn = torch.empty_like(x)
    ops.rmsnorm(x, nw, 1e-5, xn) if kind == "rms" else ops.layernorm(x, nw, nb, 1e-5, xn)
    # int4
    qw, s, z, tc, Wd = make_w4(N, K, 128, 21)
    lin = w4_module(qw, s, z, N, K, 128)
    a, b = torch.empty((M, N), dtype=BF, device=DEV), torch.empty((M, N), dtype=BF, device=DEV)
    lin.hip_linear(xn, a)
    lin.hip_linear(x, b, norm=norm)
    assert_bf16_close(b, a.float(), ulps=1, atol=2e-3, what=f"w4 fused {kind}")
    assert float((a == b).float().mean()) > 0.97
    # dense bf16
    if K % 8 == 0 and K <= 32768:
        W = (torch.randn(N, K, generator=g) * 0.02).to(BF).to(DEV)
        ops.bf16_linear(W, xn, a)
        ops.bf16_linear(W, x, b, norm=norm)
        assert_bf16_close(b, a.float(), ulps=1, atol=2e-3, what=f"bf16 fused {kind}")
        assert float((a == b).float().mean()) > 0.97
    # LLM.int8: the norm is fused into the activation quantiser
    act1 = ops.w8_prep_act(xn, 6.0, ops.W8Act(M, K, DEV))
    act2 = ops.w8_prep_act(x, 6.0, ops.W8Act(M, K, DEV), norm)
    assert float((act1.xq == act2.xq).float().mean()) > 0.995 and torch.allclose(act1.sca, act2.sca, rtol=1e-2)


@pytest.mark.parametrize("n_groups,q_per_kv,hs,n_elem", ATTN_SHAPES)
@pytest.mark.parametrize("S,nsplit", [(40, 1), (96, 3), (300, 8), (300, 1), (700, 2), (1500, 1)])
def test_fused_decode_attention_equals_the_three_kernel_path(n_groups, q_per_kv, hs, n_elem, S, nsplit):
    """parrot_attn_fused_decode == rope_kvappend + attn_decode (+ combine): the cache contents bit for bit, the heads up
    to the grouping of the softmax merge (the fused kernel merges row slots inside a wave with shuffles, and runs 16 waves
    per workgroup when a split is long), over a prefill and a run of single-token steps that wraps around the ring."""
    g = gen(22)
    n_head, width = n_groups * q_per_kv, n_groups * (q_per_kv + 2) * hs
    cos, sin = (t.to(DEV) for t in om.rope_tables(2048, n_elem, BF, math_dtype=BF))
    kc1 = torch.zeros((n_groups, S, hs), dtype=BF, device=DEV); vc1 = torch.zeros_like(kc1)
    kc2 = torch.zeros_like(kc1); vc2 = torch.zeros_like(kc1)
    q = torch.empty((1, n_head * hs), dtype=BF, device=DEV)
    y1, y2 = torch.empty_like(q), torch.empty_like(q)
    ws1, ws2 = ops.attn_workspace(1, n_head, hs, nsplit, DEV), ops.attn_workspace(1, n_head, hs, nsplit, DEV)
    tickets = torch.zeros((n_head,), dtype=torch.int32, device=DEV)
    for pos in list(range(0, 12)) + list(range(S - 3, S + 9)):
        qkv = torch.randn(1, width, generator=g).to(BF).to(DEV)
        pos_d = torch.tensor([pos], dtype=torch.int32, device=DEV)
        if pos == S - 3:  # jump ahead: fill both caches identically so that the slots in between are defined
            fill = torch.randn((n_groups, S, hs), generator=g).to(BF).to(DEV)
            for c_ in (kc1, vc1, kc2, vc2):
                c_.copy_(fill)
        ops.rope_kvappend(qkv, cos, sin, n_elem, pos_d, n_groups, q_per_kv, hs, S, q, kc1, vc1)
        ops.attn_decode(q, pos_d, kc1, vc1, n_groups, q_per_kv, hs, S, nsplit, ws1, y1)
        ops.attn_fused_decode(qkv, cos, sin, n_elem, pos_d, kc2, vc2, n_groups, q_per_kv, hs, S, nsplit, ws2, tickets, y2)
        assert torch.equal(kc1, kc2) and torch.equal(vc1, vc2), f"cache differs at pos {pos}"
        d = (y1.float() - y2.float()).abs()
        assert float(d.max()) <= 2 ** -7 * max(1.0, float(y1.float().abs().max())), f"fused attention differs at pos {pos}: max {float(d.max())}"
        assert float((d == 0).float().mean()) > 0.9, f"pos {pos}: only {float((d == 0).float().mean()):.3f} of the outputs identical"
        assert int(tickets.abs().sum()) == 0, "arrival tickets must be re-armed"


@pytest.mark.parametrize("n_groups,q_per_kv,hs", [(4, 1, 128), (2, 4, 64), (1, 3, 32), (8, 16, 64)])
@pytest.mark.parametrize("M,S,pos0", [(32, 32, 0), (100, 128, 0), (257, 300, 0), (70, 200, 37), (512, 512, 0)])
def test_prefill_attention_on_the_matrix_cores_equals_the_row_by_row_path(n_groups, q_per_kv, hs, M, S, pos0):
    """parrot_attn_prefill (flash attention, MFMA, P rounded to bf16) against parrot_attn_decode over the same rows (fp32 P) and
    against float64 softmax attention: ragged last query block, a prompt that starts behind cached context (pos0 > 0, no wrap),
    GQA / MQA head sharing, every head size."""
    if n_groups * q_per_kv * hs * M > 16 * 128 * 512:
        pytest.skip("kept small")
    g = gen(24)
    n_head = n_groups * q_per_kv
    q = torch.randn(M, n_head * hs, generator=g).to(BF)
    kc = torch.randn(n_groups, S, hs, generator=g).to(BF)
    vc = torch.randn(n_groups, S, hs, generator=g).to(BF)
    vc[:, pos0 + M:] = float("nan")  # slots behind the last key are never read as numbers
    kc[:, pos0 + M:] = float("nan")
    pos_d = torch.tensor([pos0], dtype=torch.int32, device=DEV)
    y1 = torch.empty((M, n_head * hs), dtype=BF, device=DEV)
    y2 = torch.empty_like(y1)
    nsplit = 1
    ops.attn_decode(q.to(DEV), pos_d, kc.to(DEV), vc.to(DEV), n_groups, q_per_kv, hs, S, nsplit, ops.attn_workspace(M, n_head, hs, nsplit, DEV), y1)
    ops.attn_prefill(q.to(DEV), pos_d, kc.to(DEV), vc.to(DEV), n_groups, q_per_kv, hs, S, y2)
    # float64 reference
    qd = q.double().view(M, n_groups, q_per_kv, hs)
    kd, vd = torch.nan_to_num(kc.double()), torch.nan_to_num(vc.double())
    s = torch.einsum("mgqd,gsd->mgqs", qd, kd) / math.sqrt(hs)
    mask = torch.arange(S)[None, :] <= (pos0 + torch.arange(M))[:, None]
    s = s.masked_fill(~mask[:, None, None, :], float("-inf"))
    want = torch.einsum("mgqs,gsd->mgqd", torch.softmax(s, dim=-1), vd).reshape(M, n_head * hs)
    assert torch.isfinite(y2.float()).all()
    err2 = float((y2.cpu().double() - want).abs().max())
    err1 = float((y1.cpu().double() - want).abs().max())
    assert err2 <= 2 ** -6, (err2, err1)  # values ~N(0,1): P and the output are rounded to bf16 (2^-8 relative each)
    assert float((y2.cpu().double() - want).abs().mean()) <= 2e-3
    assert float((y1.float() - y2.float()).abs().max()) <= 2 ** -5


# ------------------------------------------------------------------------------------------------ step glue
def test_embedding_and_argmax_advance():
    g = gen(18)
    wte = torch.randn(300, 128, generator=g).to(BF)
    tokens = torch.tensor([5, 299, 0, 17, 42, 0, 0, 0], dtype=torch.int64)
    out = torch.empty((3, 128), dtype=BF, device=DEV)
    ops.embedding(wte.to(DEV), tokens.to(DEV), None, 3, out)
    assert torch.equal(out.cpu(), wte[tokens[:3]])
    pos = torch.tensor([3], dtype=torch.int32, device=DEV)
    ops.embedding(wte.to(DEV), tokens.to(DEV), pos, 2, out[:2])
    assert torch.equal(out[:2].cpu(), wte[tokens[3:5]])
    # argmax: unique maximum, then a tie (lowest index wins), then a NaN that must not win
    tok_d = tokens.to(DEV)
    # ragged tail (V % 8 != 0), fewer than 8 logits, an unaligned row, an all-NaN row
    for V, off in ((5, 0), (13, 0), (301, 0), (4099, 0), (64, 1)):
        buf = torch.randn(V + off, generator=g).to(BF)
        buf[off + V - 1] = 7.0
        pos = torch.tensor([4], dtype=torch.int32, device=DEV)
        ops.argmax_advance(buf.to(DEV)[off:], tok_d, pos)
        assert int(tok_d[5]) == V - 1, (V, off, int(tok_d[5]))
        if V > 2:
            buf[off + 1] = 7.0  # tie: lowest index
            pos = torch.tensor([4], dtype=torch.int32, device=DEV)
            ops.argmax_advance(buf.to(DEV)[off:], tok_d, pos)
            assert int(tok_d[5]) == 1, (V, off, int(tok_d[5]))
    pos = torch.tensor([4], dtype=torch.int32, device=DEV)
    ops.argmax_advance(torch.full((40,), float("nan")).to(BF).to(DEV), tok_d, pos)
    assert int(tok_d[5]) == 0
    tok_d[5:7] = 0
    for V in (300, 32000, 50304):
        logits = torch.randn(V, generator=g).to(BF)
        logits[V - 7] = 9.0
        pos = torch.tensor([4], dtype=torch.int32, device=DEV)
        ops.argmax_advance(logits.to(DEV), tok_d, pos)
        assert int(pos) == 5 and int(tok_d[5]) == V - 7
        logits[11] = 9.0
        logits[3] = float("nan")
        ops.argmax_advance(logits.to(DEV), tok_d, pos)
        assert int(pos) == 6 and int(tok_d[6]) == 11
        tok_d[5:7] = 0


@pytest.mark.parametrize("V", [512, 1003, 32000, 50688, 65024])
@pytest.mark.parametrize("top_k,temperature", [(200, 0.8), (5, 0.8), (None, 1.0), (1, 0.7), (100000, 1.3)])
def test_topk_sample_draws_what_the_torch_ops_draw(V, top_k, temperature):
    """parrot_topk_sample against the device ops the reference's sampling step runs (generate/base.py:136-144): the
    probabilities bit for bit, and with the same torch seed the token torch.multinomial draws - the noise handed to the kernel
    is the very draw multinomial makes internally (empty_like(probs).exponential_(1))."""
    g = gen(40)
    tokens = torch.zeros((64,), dtype=torch.int64, device=DEV)
    pos = torch.zeros((1,), dtype=torch.int32, device=DEV)
    noise = torch.empty((V,), dtype=BF, device=DEV)
    probs = torch.empty((V,), dtype=BF, device=DEV)
    for trial in range(12):
        lg = (torch.randn(V, generator=g) * (3.0 if trial % 3 else 0.5)).to(BF)
        if trial == 3:
            lg[: V // 2] = lg[V // 2: V // 2 * 2]  # many exact ties, also across the k-th value
        if trial == 4:
            lg[::7] = float("-inf")
        if trial == 5:
            lg[:] = 0.25  # one value: everything is kept whatever k is
        lg = lg.to(DEV)
        t = lg / temperature
        if top_k is not None:
            v, _ = torch.topk(t, min(top_k, V))
            t = torch.where(t < v[[-1]], -float("Inf"), t)
        want_p = torch.nn.functional.softmax(t, dim=-1)
        torch.manual_seed(100 + trial)
        want = int(torch.multinomial(want_p, num_samples=1))
        torch.manual_seed(100 + trial)
        noise.exponential_(1)
        pos.fill_(trial)
        ops.topk_sample(lg, temperature, top_k, noise, tokens, pos, probs_out=probs)
        diff = int((probs != want_p).sum())
        # a crop leaves <= a few hundred terms in the softmax sum: the same bits.  Over the whole vocabulary the fp32 sum depends
        # on the order of ~V additions (torch's block reduction vs this kernel's; they differ by ~1e-6 relative): the few
        # probabilities that sit that close to a bf16 rounding boundary land on the neighbouring value (measured: 6 of 32000,
        # 58 of 50688), never further
        cropped = top_k is not None and top_k < V
        assert diff <= (0 if cropped else V // 500), f"trial {trial}: {diff} of {V} probabilities differ from torch's softmax bits"
        if diff:
            from helpers import bf16_ulp_distance
            assert int(bf16_ulp_distance(probs, want_p).max()) <= 1
        assert int(pos) == trial + 1 and int(tokens[trial + 1]) == want, f"trial {trial}: drew {int(tokens[trial + 1])}, torch.multinomial {want}"


def test_topk_sample_bad_arguments():
    tokens = torch.zeros((4,), dtype=torch.int64, device=DEV)
    pos = torch.zeros((1,), dtype=torch.int32, device=DEV)
    lg = torch.zeros((64,), dtype=BF, device=DEV)
    with pytest.raises(ParrotHipError):
        ops.topk_sample(lg, 0.0, 5, torch.ones_like(lg), tokens, pos)
    with pytest.raises(ParrotHipError):
        ops.topk_sample(lg, 1.0, 5, torch.ones((32,), dtype=BF, device=DEV), tokens, pos)
    with pytest.raises(ParrotHipError):
        ops.topk_sample(lg.float(), 1.0, 5, torch.ones_like(lg), tokens, pos)


def test_profiling_sink_reports_kernels():
    from lit_parrot_amd import _hip

    x = torch.randn(2, 256).to(BF).to(DEV)
    w = torch.ones(256, dtype=BF, device=DEV)
    out = torch.empty_like(x)
    _hip.prof_begin()
    for _ in range(5):
        ops.rmsnorm(x, w, 1e-5, out)
    stats = _hip.prof_end()
    assert set(stats) == {"rmsnorm"} and stats["rmsnorm"][1] == 5 and 0 < stats["rmsnorm"][0] < 50
