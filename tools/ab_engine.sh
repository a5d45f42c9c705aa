#!/bin/bash
# A/B inside one gpurun session: rebuild the library with different engine constants and time the headline decode.
# usage: [WORKLOAD=stablelm-3b-bf16] tools/ab_engine.sh "ENG_THIN_PIECES_V=4" "ENG_THIN_PIECES_V=64 ENG_MAXFLY_V=2" ...
set -e
for defs in "$@"; do
  PARROT_BUILD_DEFINES="$defs" python lit-parrot_amd/_build.py > /dev/null 2>&1
  for rep in 1 2; do
    python bench.py --workload ${WORKLOAD:-llama2-7b-int4} --steps 256 --warmup 16 --no-cpu-baseline --engine 1 2>/dev/null | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('$defs', 'rep $rep', round(r['value'],1), 'tok/s', round(r['ms_per_step']*1000,1), 'us')"
  done
done
PARROT_BUILD_DEFINES="" python lit-parrot_amd/_build.py > /dev/null 2>&1
