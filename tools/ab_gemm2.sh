#!/bin/bash
# A/B: first-generation GEMMs (PARROT_GEMM2=0) vs the LDS-DMA kernels; Llama-2-7B int4 g128 prefill of 128 / 512 / 2048 tokens
for T in 128 512 2048; do
for v in 0 1; do
  echo "== T=$T PARROT_GEMM2=$v"
  PARROT_GEMM2=$v python tools/prefill_breakdown.py Llama-2-7b-hf gptq.int4-g128 $T 2>/dev/null | grep -E "w4_gemm|total"
done
done
