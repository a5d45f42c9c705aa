#!/bin/bash
# A/B inside one gpurun session: rebuild the library with different -D constants and time the headline decode (multi-launch).
# usage: tools/ab_build.sh "W4_PRIME_SMALL=2" "W4_PRIME_SMALL=4" ...
set -e
for defs in "$@"; do
  PARROT_BUILD_DEFINES="$defs" python lit-parrot_amd/_build.py > /dev/null 2>&1
  for rep in 1 2; do
    python bench.py --steps 256 --warmup 16 --no-cpu-baseline --engine 0 2>/dev/null | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('$defs', 'rep $rep', round(r['value'],1), 'tok/s', {n:round(v['avg_us'],2) for n,v in list(r['kernels'].items())[:3]})"
  done
done
PARROT_BUILD_DEFINES="" python lit-parrot_amd/_build.py > /dev/null 2>&1
