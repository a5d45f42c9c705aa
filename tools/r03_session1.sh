#!/bin/bash
# round 3, GPU session 1: new fused attention + out-projection launch (tests, A/B), then the engine's measured ceiling
# (diagnostic builds: units stubbed / hand-offs stubbed / both).  Everything goes to gpurun_out/r03s1/.
set -o pipefail
OUT=gpurun_out/r03s1
mkdir -p $OUT
echo "== gpu tests" | tee $OUT/progress.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" | tee -a $OUT/progress.txt
tail -5 $OUT/pytest.log
b() {  # tag, extra args
  timeout -k 10 600 python bench.py --steps 256 --warmup 16 --no-cpu-baseline $2 > $OUT/bench_$1.json 2> $OUT/bench_$1.err
  echo "bench $1 rc $? $(python -c "import json;r=json.load(open('$OUT/bench_$1.json'));print(round(r['value'],1),'tok/s',r['ms_per_step'],'ms', 'engine',r['engine'], {k:round(v['avg_us'],2) for k,v in r['kernels'].items()})" 2>&1)" | tee -a $OUT/progress.txt
}
b fused1 "--fuse-attn-proj 1 --engine 0"
b fused0 "--fuse-attn-proj 0 --engine 0"
b fused1b "--fuse-attn-proj 1 --engine 0"
b fused0b "--fuse-attn-proj 0 --engine 0"
b engine "--engine 1"
for defs in "ENG_STUB_UNITS=1" "ENG_STUB_HANDOFF=1" "ENG_STUB_UNITS=1 ENG_STUB_HANDOFF=1"; do
  tag=$(echo $defs | tr ' =' '__')
  PARROT_BUILD_DEFINES="$defs" python lit-parrot_amd/_build.py > $OUT/build_$tag.log 2>&1 || { echo "build $tag failed" | tee -a $OUT/progress.txt; continue; }
  b eng_$tag "--engine 1"
done
PARROT_BUILD_DEFINES="" python lit-parrot_amd/_build.py > $OUT/build_restore.log 2>&1
echo "session done" | tee -a $OUT/progress.txt
