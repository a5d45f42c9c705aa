#!/bin/bash
# A/B inside one session: attention-hosted weight prefetch off / levels 1-3, alternating
for rep in 1 2; do
for lvl in 0 1 2 3; do
  PARROT_ATTN_PREFETCH=$lvl python bench.py --steps 192 --warmup 16 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json
d=json.loads(sys.stdin.readline())
k=d['kernels']
print('lvl $lvl rep $rep tok/s %.1f ms %.4f' % (d['value'], d['ms_per_step']), {n:round(v['avg_us'],2) for n,v in k.items()}, flush=True)
"
done
done
