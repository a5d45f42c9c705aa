#!/bin/bash
# bf16 prompt GEMM: build-constant A/B, one shape at a time (tools/gemm_probe.py)
set -o pipefail
OUT=gpurun_out/$1; M=$2; shift 2; mkdir -p $OUT
for d in "$@"; do
  PARROT_BUILD_DEFINES="$d" python lit-parrot_amd/_build.py > $OUT/build.log 2>&1 || { echo "build [$d] failed"; tail -5 $OUT/build.log; continue; }
  echo "== M=$M [$d]" | tee -a $OUT/progress.txt
  timeout -k 10 200 python tools/gemm_probe.py stablelm-base-alpha-3b bf16 $M 2>&1 | grep -v amdgpu.ids | tee -a $OUT/progress.txt
done
PARROT_BUILD_DEFINES="" python lit-parrot_amd/_build.py > $OUT/build_restore.log 2>&1
echo done
