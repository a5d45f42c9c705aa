"""Diagnostic: how much faster is the int4 GEMV when its weights were touched (prefetched) right before the launch?
Measures per-kernel durations with the profiling sink (dispatch-packet events), cold vs prefetched vs repeated."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from lit_parrot_amd import _hip, ops  # noqa: E402

DEV = torch.device("cuda", 0)


def run(lib, N, K, dual, mode, copies, bufs, x, out, iters=60, pf_frac=1.0, norm=None):
    st = _hip.stream()
    nbytes = bufs[0].numel()

    def gemv(i):
        w = bufs[i % copies]
        w2 = bufs[(i + 1) % copies] if dual else None
        rc = lib.parrot_w4_gemv(w.data_ptr(), w2.data_ptr() if dual else None, x.data_ptr(), K, 1, None, None, 0,
                                out.data_ptr(), N, N, K, 128, 3 if dual else 0, norm, st)
        assert rc == 0, _hip.last_error()

    for i in range(10):
        gemv(i)
    torch.cuda.synchronize()
    _hip.prof_begin()
    for i in range(iters):
        if mode == "cold":
            gemv(i)
        elif mode == "same":
            gemv(0)
        elif mode == "prefetched":
            n = int(nbytes * pf_frac) // 16 * 16
            lib.parrot_prefetch(bufs[i % copies].data_ptr(), n, 512, st)
            if dual:
                lib.parrot_prefetch(bufs[(i + 1) % copies].data_ptr(), n, 512, st)
            gemv(i)
    stats = _hip.prof_end()
    return {k: v[0] / v[1] * 1e3 for k, v in stats.items()}


def main():
    lib = _hip.load()
    for name, N, K, dual in (("qkv", 12288, 4096, False), ("proj", 4096, 4096, False), ("fc", 11008, 4096, True), ("down", 4096, 11008, False)):
        nbytes = ops.w4_packed_bytes(N, K, 128)
        copies = max(3, int(1.2e9 // nbytes))
        bufs = [torch.randint(0, 255, (nbytes,), dtype=torch.uint8, device=DEV) for _ in range(copies)]
        x = torch.randn(1, K, device=DEV).to(torch.bfloat16)
        out = torch.empty((1, N), dtype=torch.bfloat16, device=DEV)
        line = f"{name:5s} {nbytes * (2 if dual else 1) / 1e6:6.1f} MB:"
        for mode, frac in (("cold", 1.0), ("same", 1.0), ("prefetched", 1.0), ("prefetched", 0.5)):
            r = run(lib, N, K, dual, mode, copies, bufs, x, out, pf_frac=frac)
            g = [v for k, v in r.items() if k.startswith("w4_gemv")][0]
            p = r.get("prefetch")
            line += f"  {mode}{'' if frac == 1.0 else frac}: gemv {g:6.2f} us" + (f" (pf {p:5.2f})" if p else "")
        import ctypes as C

        nw = torch.ones(K, dtype=torch.bfloat16, device=DEV)
        for kind in (1, 2):
            nrm = C.byref(_hip.ParrotNorm(kind, nw.data_ptr(), nw.data_ptr() if kind == 2 else None, 1e-5, 0))
            r = run(lib, N, K, dual, "cold", copies, bufs, x, out, norm=nrm)
            g = [v for k, v in r.items() if k.startswith("w4_gemv")][0]
            line += f"  cold+{'rms' if kind == 1 else 'ln'}norm: {g:6.2f} us"
        print(line, flush=True)
        del bufs


if __name__ == "__main__":
    main()
