#!/bin/bash
for T in 128 32; do
for v in 2 4; do
  echo "== T=$T int4 PARROT_GEMM2_W4_WN=$v"
  PARROT_GEMM2_W4_WN=$v python tools/prefill_breakdown.py Llama-2-7b-hf gptq.int4-g128 $T 2>/dev/null | grep -E "w4_gemm|splitk|total"
done
done
