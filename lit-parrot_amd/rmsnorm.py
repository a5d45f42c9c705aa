"""RMSNorm module (reference lit_gpt/rmsnorm.py:4-21) running the HIP kernel."""
import torch

from . import ops


class RMSNorm(torch.nn.Module):
    """``weight * x * rsqrt(mean(x*x, -1) + eps)`` evaluated in the input dtype, as the reference does."""

    def __init__(self, size: int, dim: int = -1, eps: float = 1e-5) -> None:
        super().__init__()
        if dim != -1:
            raise NotImplementedError("RMSNorm over the last dimension only")
        self.weight = torch.nn.Parameter(torch.ones(size))
        self.eps = eps
        self.dim = dim

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        rows = x.reshape(-1, x.shape[-1]).contiguous()
        out = torch.empty_like(rows)
        ops.rmsnorm(rows, self.weight.data, self.eps, out)
        return out.view(x.shape)
