#!/bin/bash
# A/B of the prefill GEMM's tile / split-K selection (env hooks read once per process): StableLM-3B bf16, 512-token prompt
for cfg in "512 -2" "512 0" "512 2" "512 4" "512 8" "100000 4" "100000 8"; do
  set -- $cfg
  echo "== PARROT_GEMM_BIG_MIN=$1 PARROT_GEMM_KSPLIT=$2"
  PARROT_GEMM_BIG_MIN=$1 PARROT_GEMM_KSPLIT=$2 python tools/prefill_breakdown.py stablelm-base-alpha-3b bf16 512 2>/dev/null | grep -E "bf16_gemm|total"
done
