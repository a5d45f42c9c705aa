#!/bin/bash
# build-constant sweep on the multi-launch headline: usage <outdir> "tag|defines" ...
set -o pipefail
OUT=gpurun_out/$1; shift; mkdir -p $OUT
for spec in "$@"; do
  tag=${spec%%|*}; defs=${spec#*|}
  PARROT_BUILD_DEFINES="$defs" python lit-parrot_amd/_build.py > $OUT/build_$tag.log 2>&1 || { echo "build $tag failed" | tee -a $OUT/progress.txt; continue; }
  for rep in 1 2; do
    timeout -k 10 600 python bench.py --steps 256 --warmup 16 --no-cpu-baseline --no-sampled --engine 0 > $OUT/bench_${tag}_$rep.json 2> $OUT/bench_${tag}_$rep.err
    echo "$tag [$defs] rep $rep rc $? $(python -c "import json;r=json.load(open('$OUT/bench_${tag}_$rep.json'));print(round(r['value'],1),'tok/s',round(r['ms_per_step']*1000,1),'us', {k:round(v['avg_us'],2) for k,v in r['kernels'].items()})" 2>&1)" | tee -a $OUT/progress.txt
  done
done
PARROT_BUILD_DEFINES="" python lit-parrot_amd/_build.py > $OUT/build_restore.log 2>&1
echo done
