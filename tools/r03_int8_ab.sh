#!/bin/bash
# LLM.int8 on the engine: what the activation quantiser / the units / the hand-offs cost (diagnostic builds).  usage: <outdir> "tag|defines" ...
set -o pipefail
OUT=gpurun_out/$1; shift; mkdir -p $OUT
for spec in "$@"; do
  tag=${spec%%|*}; defs=${spec#*|}
  PARROT_BUILD_DEFINES="$defs" python lit-parrot_amd/_build.py > $OUT/build_$tag.log 2>&1 || { echo "build $tag failed" | tee -a $OUT/progress.txt; continue; }
  timeout -k 10 600 python bench.py --workload ${WORKLOAD:-llama2-7b-int8} --steps 128 --warmup 16 --no-cpu-baseline --no-sampled > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err
  echo "bench $tag [$defs] rc $? $(python -c "import json;r=json.load(open('$OUT/bench_$tag.json'));print(round(r['value'],1),'tok/s',round(r['ms_per_step']*1000,1),'us', 'engine', r['engine'])" 2>&1)" | tee -a $OUT/progress.txt
done
PARROT_BUILD_DEFINES="" python lit-parrot_amd/_build.py > $OUT/build_restore.log 2>&1
echo done
