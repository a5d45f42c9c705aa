#!/bin/bash
# A/B: prompt attention through the row-by-row decode kernels (PARROT_ATTN_PREFILL_MFMA=0) vs parrot_attn_prefill
for cfg in "stablelm-base-alpha-3b bf16 512" "stablelm-base-alpha-3b bf16 2048" "Llama-2-7b-hf gptq.int4-g128 128" "Llama-2-7b-hf gptq.int4-g128 2048" "falcon-40b gptq.int4-g128 512"; do
  set -- $cfg
  for v in 0 1; do
    echo "== $1 $2 T=$3 PARROT_ATTN_PREFILL_MFMA=$v"
    PARROT_ATTN_PREFILL_MFMA=$v python tools/prefill_breakdown.py $1 $2 $3 2>/dev/null | grep -E "attn|total"
  done
done
