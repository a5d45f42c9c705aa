#!/bin/bash
# The engine's skeleton (units and hand-offs stubbed) under the loupe: per-op stamps, and the loader unthrottled.
set -o pipefail
OUT=gpurun_out/$1; mkdir -p $OUT
run() {  # tag, defines, what (bench|stamps)
  PARROT_BUILD_DEFINES="$2" python lit-parrot_amd/_build.py > $OUT/build_$1.log 2>&1 || { echo "build $1 failed" | tee -a $OUT/progress.txt; return; }
  if [ "$3" = stamps ]; then
    timeout -k 10 300 python tools/eng_stamps.py llama2-7b-int4 8 > $OUT/stamps_$1.txt 2>&1; echo "stamps $1 rc $?" | tee -a $OUT/progress.txt
    grep -A8 "mean per op type" $OUT/stamps_$1.txt | tee -a $OUT/progress.txt
  else
    timeout -k 10 600 python bench.py --steps 256 --warmup 16 --no-cpu-baseline --engine 1 > $OUT/bench_$1.json 2> $OUT/bench_$1.err
    echo "bench $1 rc $? $(python -c "import json;r=json.load(open('$OUT/bench_$1.json'));print(round(r['value'],1),'tok/s',round(r['ms_per_step']*1000,1),'us')" 2>&1)" | tee -a $OUT/progress.txt
  fi
}
shift
for spec in "$@"; do
  tag=${spec%%|*}; rest=${spec#*|}; defs=${rest%%|*}; what=${rest#*|}
  run "$tag" "$defs" "$what"
done
PARROT_BUILD_DEFINES="" python lit-parrot_amd/_build.py > $OUT/build_restore.log 2>&1
echo done | tee -a $OUT/progress.txt
