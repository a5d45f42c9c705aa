#!/bin/bash
# cache-residency probe: the headline model cut to N blocks (weights of N <= 1 stay in the 256 MiB Infinity Cache between tokens)
set -o pipefail
OUT=gpurun_out/$1; shift; mkdir -p $OUT
for n in "$@"; do
  for e in 0 1; do
    timeout -k 10 300 python bench.py --steps 256 --warmup 16 --no-cpu-baseline --no-sampled --engine $e --layers $n > $OUT/bench_L${n}_e$e.json 2> $OUT/bench_L${n}_e$e.err
    echo "layers $n engine $e rc $? $(python -c "import json;r=json.load(open('$OUT/bench_L${n}_e$e.json'));print(round(r['ms_per_step']*1000,1),'us/token', {k:round(v['avg_us'],2) for k,v in r['kernels'].items()})" 2>&1)" | tee -a $OUT/progress.txt
  done
done
echo done
