"""Shared helpers of the GPU parity tests."""
import torch


def bf16_ulp_distance(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """Distance in bf16 units-in-the-last-place between two bf16 tensors (monotone integer mapping of the bits)."""
    def key(t):
        i = t.contiguous().view(torch.int16).to(torch.int32)
        return torch.where(i < 0, -(i & 0x7FFF), i)
    return (key(a.to(torch.bfloat16).cpu()) - key(b.to(torch.bfloat16).cpu())).abs()


def assert_bf16_close(got: torch.Tensor, want: torch.Tensor, ulps: int = 1, atol: float = 0.0, what: str = ""):
    """``got`` (bf16, from the GPU) equals ``want`` rounded to bf16 up to ``ulps`` bf16 ulps, or ``atol`` absolute
    (for results that cancel to ~0, where an ulp is meaningless)."""
    got = got.detach().cpu().to(torch.bfloat16)
    want_bf = want.detach().cpu().to(torch.bfloat16)
    assert got.shape == want_bf.shape, (got.shape, want_bf.shape)
    assert torch.isfinite(got.float()).all(), f"{what}: non-finite output"
    d = bf16_ulp_distance(got, want_bf)
    bad = (d > ulps) & ((got.float() - want.detach().cpu().float()).abs() > atol)
    if bad.any():
        i = bad.nonzero()[0].tolist()
        raise AssertionError(f"{what}: {int(bad.sum())}/{bad.numel()} elements differ by more than {ulps} bf16 ulp "
                             f"(first at {i}: got {got[tuple(i)].item()} want {want[tuple(i)].item()}, max ulp {int(d.max())})")


def rbf(t: torch.Tensor) -> torch.Tensor:
    """round to bf16 precision, keep float64/float32 container"""
    return t.to(torch.bfloat16).to(t.dtype)
