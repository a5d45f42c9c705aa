#!/usr/bin/env python3
"""Kernel microbenchmarks on the GPU box (not part of the product): times the weight-streaming GEMV entry points on the
Linear shapes of a config with rotating weight buffers (footprint > the 256 MiB Infinity Cache, so every launch streams
from HBM like the real decode step does), and prints achieved GB/s per shape and tuning setting.

    python tools/microbench.py [--config Llama-2-7b-hf] [--mode w4|bf16] [--iters 200]

The tuning sweeps and stamps need the diagnostic build of the library (`python lit-parrot_amd/_build.py --diag`): the hooks
(`parrot_tune_w4_stamps` ...) are not compiled into the shipped one.
"""
import argparse
import ctypes
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from lit_parrot_amd import _hip, ops  # noqa: E402
from lit_parrot_amd.config import Config  # noqa: E402

DEV = torch.device("cuda", 0)


def time_loop(fn, n_warm, n_iter):
    for i in range(n_warm):
        fn(i)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n_iter):
        fn(i)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n_iter * 1e3  # us per call (includes the launch gap between back-to-back kernels)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="Llama-2-7b-hf")
    ap.add_argument("--mode", default="w4")
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--group", type=int, default=128)
    ap.add_argument("--norm", type=int, default=0, help="1: fuse an RMSNorm prologue")
    ap.add_argument("--burst-only", type=int, default=0, help="1: sweep only the burst kernel's row groups per workgroup")
    ap.add_argument("--stamps", type=int, default=0, help="1: print in-kernel phase stamps of 3 workgroups (diagnostic)")
    args = ap.parse_args()
    cfg = Config.from_name(args.config)
    lib = _hip.load()
    tune = getattr(lib, "parrot_tune_w4_rows_per_wg", None)
    shapes = cfg.linear_shapes()
    for name, (N, K) in shapes.items():
        if args.mode == "w4":
            nbytes = ops.w4_packed_bytes(N, K, args.group)
            algo = N * K // 2 + N * (-(-K // args.group)) * 4
        else:
            nbytes = N * K * 2
            algo = nbytes
        copies = max(2, int(600e6 // nbytes) + 1)
        bufs = [torch.randint(0, 255, (nbytes,), dtype=torch.uint8, device=DEV) for _ in range(copies)]
        if args.mode == "w4":  # sane scales: bf16 1.0 / zero 8 patterns are not needed for timing; random bytes are fine
            pass
        x = torch.randn(1, K, device=DEV).to(torch.bfloat16)
        out = torch.empty((1, N), dtype=torch.bfloat16, device=DEV)
        dual = name == "mlp.fc_1"
        if name == "mlp.fc_2":
            continue
        st = _hip.stream()
        import ctypes as C
        nw = torch.ones(K, dtype=torch.bfloat16, device=DEV)
        norm = C.byref(_hip.ParrotNorm(1, nw.data_ptr(), None, 1e-5, 0)) if args.norm else None
        tune_stream = None  # the pipelined "stream" GEMV variant was retired in round 2 (kept in the history)
        variants = [(0, 0, 0), (1, 2, 1), (1, 2, 2), (1, 2, 4), (1, 4, 1), (1, 4, 2), (1, 3, 2), (1, 6, 2), (1, 8, 1)] if args.mode == "w4" and tune_stream is not None else [(0, 0, 0)]
        if args.burst_only:
            variants = [(0, 0, 0), (0, 0, 1), (0, 0, 2), (0, 0, 3), (0, 0, 4)]
        for use_stream, gx, rows in variants:
            if tune is not None:
                tune(rows)
            if tune_stream is not None:
                tune_stream(use_stream, gx)

            def call(i):
                w = bufs[i % copies]
                w2 = bufs[(i + 1) % copies] if dual else None
                if args.mode == "w4":
                    rc = lib.parrot_w4_gemv(w.data_ptr(), w2.data_ptr() if dual else None, x.data_ptr(), K, 1, None, None, 0,
                                            out.data_ptr(), N, N, K, args.group, 3 if dual else 0, norm, st)
                else:
                    rc = lib.parrot_bf16_gemv(w.view(torch.bfloat16).data_ptr(), w2.view(torch.bfloat16).data_ptr() if dual else None,
                                              x.data_ptr(), K, 1, None, None, 0, out.data_ptr(), N, N, K, 3 if dual else 0, norm, st)
                assert rc == 0, _hip.last_error()

            us = time_loop(call, 20, args.iters)
            if args.stamps and use_stream == 0:
                lib.parrot_tune_w4_stamps.argtypes = [C.c_void_p]
                # gaps between consecutive launches of a stream: every launch stamps into its own buffer
                seq = [torch.zeros(24, dtype=torch.int64, device=DEV) for _ in range(6)]
                for i, b_ in enumerate(seq):
                    lib.parrot_tune_w4_stamps(b_.data_ptr())
                    call(i)
                torch.cuda.synchronize()
                lib.parrot_tune_w4_stamps(None)
                sq = [b_.cpu().view(3, 8).double() * 0.01 for b_ in seq]
                line = "    consecutive launches (3 stamped workgroups each): "
                for a, b_ in zip(sq[:-1], sq[1:]):
                    line += f"[span {float(a[:, 4].max() - a[:, 0].min()):5.2f} | gap to next entry {float(b_[:, 0].min() - a[:, 4].max()):5.2f}] "
                print(line)
                dbg = torch.zeros(24, dtype=torch.int64, device=DEV)
                lib.parrot_tune_w4_stamps(dbg.data_ptr())
                for i in range(3):
                    call(i)
                torch.cuda.synchronize()
                lib.parrot_tune_w4_stamps(None)
                d = dbg.cpu().view(3, 8).double() * 0.01
                t0 = float(d[:, 0].min())
                for b, nm in enumerate(("first wg", "middle wg", "last wg")):
                    r = d[b]
                    print(f"    {nm:9s} entry +{float(r[0]) - t0:5.2f}us | x/norm ready {float(r[1] - r[0]):5.2f} | dots {float(r[2] - r[1]):5.2f} | barrier {float(r[3] - r[2]):5.2f} | epilogue {float(r[4] - r[3]):5.2f} | exit at +{float(r[4]) - t0:5.2f}")
            b = algo * (2 if dual else 1)
            print(f"{name:10s}{'+fc_2' if dual else '     '} N={N:6d} K={K:6d} stream={use_stream} G={128 * gx:4d} wps={rows}  {us:7.2f} us/launch  {b / us / 1e3:7.1f} GB/s  ({b / 1e6:.1f} MB)")
        if tune is not None:
            tune(0)
        if tune_stream is not None:
            tune_stream(1, 0)
        del bufs


if __name__ == "__main__":
    main()
