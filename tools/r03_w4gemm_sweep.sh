#!/bin/bash
# int4 prompt GEMM at M rows: split / tile-shape switches of the diagnostic build, one shape at a time (tools/gemm_probe.py)
set -o pipefail
OUT=gpurun_out/$1; M=$2; shift 2; mkdir -p $OUT
python lit-parrot_amd/_build.py --diag > $OUT/build.log 2>&1 || { echo build failed; tail -20 $OUT/build.log; exit 1; }
for spec in "$@"; do
  echo "== M=$M $spec" | tee -a $OUT/progress.txt
  env $spec timeout -k 10 300 python tools/gemm_probe.py Llama-2-7b-hf w4 $M 2>$OUT/err.log | tee -a $OUT/progress.txt
done
python lit-parrot_amd/_build.py > $OUT/build_restore.log 2>&1
echo done
