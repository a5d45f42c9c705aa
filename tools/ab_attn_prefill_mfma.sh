#!/bin/bash
# A/B: prompt attention: row-by-row decode kernels (MFMA=0) / flash kernel with per-wave streams (LDS=0) / with K, V^T blocks shared through LDS
for cfg in "stablelm-base-alpha-3b bf16 512" "stablelm-base-alpha-3b bf16 2048" "Llama-2-7b-hf gptq.int4-g128 2048" "falcon-40b gptq.int4-g128 512"; do
  set -- $cfg
  for v in "1 0" "1 1"; do
    set -- $cfg $v
    echo "== $1 $2 T=$3 MFMA=$4 LDS=$5"
    PARROT_ATTN_PREFILL_MFMA=$4 PARROT_ATTN_PREFILL_LDS=$5 python tools/prefill_breakdown.py $1 $2 $3 2>/dev/null | grep -E "attn_prefill |total"
  done
done
