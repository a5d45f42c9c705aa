#!/bin/bash
# quick GPU check: selected tests + an A/B of bench variants.  usage: tools/r03_quick.sh <outdir> "<pytest -k expr>" "tag|args" ...
set -o pipefail
OUT=gpurun_out/$1; shift
K="$1"; shift
mkdir -p $OUT
if [ -n "$K" ]; then
  timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "$K" > $OUT/pytest.log 2>&1; echo "pytest rc $?" | tee $OUT/progress.txt; tail -4 $OUT/pytest.log
fi
for spec in "$@"; do
  tag=${spec%%|*}; args=${spec#*|}
  timeout -k 10 600 python bench.py --steps 256 --warmup 16 --no-cpu-baseline $args > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err
  echo "bench $tag rc $? $(python -c "import json;r=json.load(open('$OUT/bench_$tag.json'));print(round(r['value'],1),'tok/s',round(r['ms_per_step']*1000,1),'us', 'engine',r['engine'], {k:round(v['avg_us'],2) for k,v in r['kernels'].items()})" 2>&1)" | tee -a $OUT/progress.txt
done
