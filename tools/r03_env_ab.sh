#!/bin/bash
# A/B of HIP runtime knobs on the headline bench (multi-launch step, hipGraph replay).  usage: tools/r03_env_ab.sh <outdir> "tag|ENV=V ENV2=V2" ...
set -o pipefail
OUT=gpurun_out/$1; shift; mkdir -p $OUT
for spec in "$@"; do
  tag=${spec%%|*}; envs=${spec#*|}
  env $envs timeout -k 10 600 python bench.py --steps 256 --warmup 16 --no-cpu-baseline --engine 0 > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err
  echo "bench $tag [$envs] rc $? $(python -c "import json;r=json.load(open('$OUT/bench_$tag.json'));print(round(r['value'],1),'tok/s',round(r['ms_per_step']*1000,1),'us')" 2>&1)" | tee -a $OUT/progress.txt
done
