#!/bin/bash
for T in 128 512; do
  echo "== T=$T int4"
  python tools/prefill_breakdown.py Llama-2-7b-hf gptq.int4-g128 $T 2>/dev/null | grep -E "w4_gemm|splitk|total"
done
echo "== T=512 stablelm bf16"
python tools/prefill_breakdown.py stablelm-base-alpha-3b bf16 512 2>/dev/null | grep -E "bf16_gemm|splitk|total"
echo "== T=128 falcon-40b int4"
python tools/prefill_breakdown.py falcon-40b gptq.int4-g128 128 2>/dev/null | grep -E "w4_gemm|splitk|total"
