#!/bin/bash
# bf16 prefill GEMM: 128 x 128 / 4-wave tiles against 256 x 128 / 8-wave tiles (diagnostic build, switch from the environment)
set -o pipefail
OUT=gpurun_out/$1; shift; mkdir -p $OUT
python lit-parrot_amd/_build.py --diag > $OUT/build.log 2>&1 || { echo build failed; tail -20 $OUT/build.log; exit 1; }
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "gemm" > $OUT/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $OUT/pytest.log
for T in "$@"; do
  for wm in 2 4; do
    echo "== stablelm-base-alpha-3b bf16 T=$T PARROT_GEMM2_WM=$wm" | tee -a $OUT/progress.txt
    PARROT_GEMM2_WM=$wm timeout -k 10 300 python tools/prefill_breakdown.py stablelm-base-alpha-3b bf16 $T 2>$OUT/err_${T}_$wm.log | tee -a $OUT/progress.txt
  done
done
python lit-parrot_amd/_build.py > $OUT/build_restore.log 2>&1
echo done
