"""TEST INFRASTRUCTURE ONLY — CPU restatement of the reference's GPTQ quantiser (quantize/gptq.py:267-444; the algorithm is
Frantar et al., arXiv:2210.17323).  Only tests/ may import this.  Pinned against tests/golden/gptq_quantizer.npz (the
reference's own ``GPTQQuantizer`` run in the build container, tests/golden/make_golden.py::golden_gptq_quantizer).

Per Linear (weight W of shape (rows, cols), fp32):
  * Hessian of the layer inputs, running mean: H <- H * n/(n+b) + (2/(n+b)) X^T X for every calibration batch of b rows
    (:349-362);
  * grid parameters per row (per-channel) from the ORIGINAL weights: find_params_weight (:317-347);
  * dead inputs (H_ii = 0): H_ii = 1, W[:, i] = 0 (:377-380); optional activation order: columns by decreasing H_ii (:381-384);
  * damping 0.01 * mean(diag H) on the diagonal, Hinv = chol(inv(H), upper) (:389-395);
  * column by column, in blocks of 128: q = grid(w); err = (w - q) / Hinv_ii; the not yet quantised columns of the block get
    -err x Hinv[i, i:], the columns behind the block -Err @ Hinv[block, behind] (:397-431);
  * loss = sum (w - q)^2 / Hinv_ii^2 / 2 (:421, :426).
Grouped quantisation (groupsize != -1) does not run in the reference (the per-group parameters are written with the wrong
shape, :409-412, and would be taken from not yet compensated weights); here the group's parameters come from the current,
error-compensated columns of the group, as in the paper's implementation.
"""
import math
from typing import Tuple

import torch


def find_params(x: torch.Tensor, maxq: int = 15) -> Tuple[torch.Tensor, torch.Tensor]:
    """Asymmetric per-row grid whose range always contains 0 (:317-347, sym=False, perchannel=True)."""
    zero_ = torch.zeros(x.shape[0])
    lo = torch.minimum(x.min(1)[0], zero_)
    hi = torch.maximum(x.max(1)[0], zero_)
    flat = (lo == 0) & (hi == 0)
    lo[flat], hi[flat] = -1.0, 1.0
    scale = (hi - lo) / maxq
    zero = torch.round(-lo / scale)
    return scale.unsqueeze(1), zero.unsqueeze(1)


def on_grid(w: torch.Tensor, scale: torch.Tensor, zero: torch.Tensor, maxq: int = 15) -> torch.Tensor:
    return scale * (torch.clamp(torch.round(w / scale) + zero, 0, maxq) - zero)


def hessian_from(batches_3d) -> torch.Tensor:
    """batches_3d: iterable of tensors shaped (b, T, cols) or (T, cols) as the forward hook receives them."""
    H, n = None, 0
    for x in batches_3d:
        b = 1 if x.dim() == 2 else x.shape[0]
        rows = x.reshape(-1, x.shape[-1]).float()
        if H is None:
            H = torch.zeros((rows.shape[1], rows.shape[1]))
        H *= n / (n + b)
        n += b
        xs = math.sqrt(2 / n) * rows.t()
        H += xs.matmul(xs.t())
    return H


@torch.no_grad()
def quantize(W: torch.Tensor, H: torch.Tensor, *, groupsize: int = -1, actorder: bool = False, blocksize: int = 128,
             percdamp: float = 0.01, maxq: int = 15):
    """Returns (Q dequantised (rows, cols) fp32, scales (rows, groups), zeros (rows, groups), loss)."""
    W = W.detach().float().clone()
    H = H.detach().float().clone()
    rows, cols = W.shape
    tile = cols if groupsize == -1 else groupsize
    ngroups = -(-cols // tile)
    scales = torch.zeros((rows, ngroups))
    zeros = torch.zeros((rows, ngroups))
    scale, zero = find_params(W, maxq)
    scales[:] = scale
    zeros[:] = zero
    dead = torch.diag(H) == 0
    H[dead, dead] = 1
    W[:, dead] = 0
    if actorder:
        assert groupsize == -1
        perm = torch.argsort(torch.diag(H), descending=True)
        W = W[:, perm]
        H = H[perm][:, perm]
    damp = percdamp * torch.mean(torch.diag(H))
    idx = torch.arange(cols)
    H[idx, idx] += damp
    Hinv = torch.linalg.cholesky(torch.cholesky_inverse(torch.linalg.cholesky(H)), upper=True)
    Q = torch.zeros_like(W)
    loss = 0.0
    for i1 in range(0, cols, blocksize):
        i2 = min(i1 + blocksize, cols)
        W1 = W[:, i1:i2].clone()
        Err = torch.zeros_like(W1)
        Hb = Hinv[i1:i2, i1:i2]
        for i in range(i2 - i1):
            c = i1 + i
            if groupsize != -1 and c % groupsize == 0:
                # current (compensated) values: columns of this block from W1, columns behind it from W
                cur = torch.cat([W1[:, i:], W[:, i2:]], dim=1)[:, :groupsize]
                scale, zero = find_params(cur, maxq)
                scales[:, c // groupsize] = scale[:, 0]
                zeros[:, c // groupsize] = zero[:, 0]
            w, d = W1[:, i], Hb[i, i]
            q = on_grid(w.unsqueeze(1), scale, zero, maxq).squeeze(1)
            Q[:, c] = q
            loss += float(((w - q) ** 2 / d ** 2).sum()) / 2
            e = (w - q) / d
            W1[:, i:] -= e.unsqueeze(1).matmul(Hb[i, i:].unsqueeze(0))
            Err[:, i] = e
        W[:, i2:] -= Err.matmul(Hinv[i1:i2, i2:])
    if actorder:
        Q = Q[:, torch.argsort(perm)]
    return Q, scales, zeros, loss
