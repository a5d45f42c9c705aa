#!/bin/bash
# engine build-constant sweep: usage <outdir> <workload> "<bench flags>" "tag|defines" ...
set -o pipefail
OUT=gpurun_out/$1; W=$2; FLAGS="$3"; shift 3; mkdir -p $OUT
for spec in "$@"; do
  tag=${spec%%|*}; defs=${spec#*|}
  PARROT_BUILD_DEFINES="$defs" python lit-parrot_amd/_build.py > $OUT/build_$tag.log 2>&1 || { echo "build $tag failed" | tee -a $OUT/progress.txt; continue; }
  for rep in 1 2; do
    timeout -k 10 600 python bench.py --workload $W --steps 256 --warmup 16 --no-cpu-baseline --no-sampled $FLAGS > $OUT/bench_${tag}_$rep.json 2> $OUT/bench_${tag}_$rep.err
    echo "$W $tag [$defs] rep $rep rc $? $(python -c "import json;r=json.load(open('$OUT/bench_${tag}_$rep.json'));print(round(r['value'],1),'tok/s',round(r['ms_per_step']*1000,1),'us')" 2>&1)" | tee -a $OUT/progress.txt
  done
done
PARROT_BUILD_DEFINES="" python lit-parrot_amd/_build.py > $OUT/build_restore.log 2>&1
echo done
