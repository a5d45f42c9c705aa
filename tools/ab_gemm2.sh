#!/bin/bash
# A/B of the LDS-DMA bf16 GEMM's split-K target; StableLM-3B prefill of 512 tokens
for v in 256 384 512 768 1024; do
  echo "== T=512 PARROT_GEMM2_SPLIT_TARGET=$v"
  PARROT_GEMM2_SPLIT_TARGET=$v python tools/prefill_breakdown.py stablelm-base-alpha-3b bf16 512 2>/dev/null | grep -E "bf16_gemm|total"
done
