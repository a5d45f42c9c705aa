#!/bin/bash
# A/B: first-generation bf16 GEMM (PARROT_GEMM2=0) vs the LDS-DMA kernel, StableLM-3B 512-token prefill, alternating
for rep in 1 2; do
for v in 0 1; do
  echo "== PARROT_GEMM2=$v (rep $rep)"
  PARROT_GEMM2=$v python tools/prefill_breakdown.py stablelm-base-alpha-3b bf16 512 2>/dev/null | grep -E "bf16_gemm|total"
done
done
