#!/bin/bash
# Measurement pass for the stream engine's bf16 path (run through gpurun): bench lines of both executors, rocprofv3 kernel
# stats, PMC traffic of the headline and of StableLM-3B.   Usage: tools/measure_bf16_engine.sh <tag>
set -o pipefail
TAG=${1:-r02b}
OUT=gpurun_out/measure_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
echo "== bench (no profiler)"
timeout -k 10 400 python bench.py --steps 256 > $OUT/llama2-7b-int4_bench.json 2> $OUT/llama2-7b-int4_bench.err || exit 1
timeout -k 10 400 python bench.py --steps 256 --engine 1 --no-cpu-baseline > $OUT/llama2-7b-int4-engine_bench.json 2> $OUT/llama2-7b-int4-engine_bench.err || exit 1
echo "headline lines done"
for w in stablelm-3b-bf16 pythia-160m-bf16 falcon-7b-bf16; do
  timeout -k 10 600 python bench.py --workload $w --steps 256 > $OUT/${w}_bench.json 2> $OUT/${w}_bench.err || exit 1
  timeout -k 10 600 python bench.py --workload $w --steps 256 --engine 0 --no-cpu-baseline > $OUT/${w}-multilaunch_bench.json 2> $OUT/${w}-multilaunch_bench.err || exit 1
  echo "$w done"
done
timeout -k 10 900 python bench.py --workload falcon-40b-int4 --steps 256 --engine 1 --no-cpu-baseline > $OUT/falcon-40b-int4-engine_bench.json 2> $OUT/falcon-40b-int4-engine_bench.err || exit 1
timeout -k 10 900 python bench.py --workload falcon-40b-int4 --steps 256 --no-cpu-baseline > $OUT/falcon-40b-int4_bench.json 2> $OUT/falcon-40b-int4_bench.err || exit 1
echo "falcon-40b done"
echo "== rocprofv3 kernel stats"
for w in llama2-7b-int4 stablelm-3b-bf16 pythia-160m-bf16 falcon-7b-bf16; do
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$w -o $w -- python3 bench.py --workload $w --steps 64 --no-cpu-baseline > $OUT/prof_${w}.log 2>&1 || exit 1
  echo "$w profiled"
done
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_llama2-7b-int4-engine -o llama2-7b-int4-engine -- python3 bench.py --steps 64 --engine 1 --no-cpu-baseline > $OUT/prof_llama2-7b-int4-engine.log 2>&1 || exit 1
echo "== PMC passes"
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o fetch -- python3 bench.py --steps 16 --warmup 4 --no-cpu-baseline > $OUT/pmc_fetch.log 2>&1 || exit 1
timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o write -- python3 bench.py --steps 16 --warmup 4 --no-cpu-baseline > $OUT/pmc_write.log 2>&1 || exit 1
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_stablelm -o fetch -- python3 bench.py --workload stablelm-3b-bf16 --steps 16 --warmup 4 --no-cpu-baseline > $OUT/pmc_fetch_stablelm.log 2>&1 || exit 1
timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_stablelm -o write -- python3 bench.py --workload stablelm-3b-bf16 --steps 16 --warmup 4 --no-cpu-baseline > $OUT/pmc_write_stablelm.log 2>&1 || exit 1
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_engine -o fetch -- python3 bench.py --steps 16 --warmup 4 --engine 1 --no-cpu-baseline > $OUT/pmc_fetch_engine.log 2>&1 || exit 1
find $OUT -type f \( -name "*.db" -o -name "*kernel_trace.csv" -o -name "*.pftrace" -o -name "*.json.gz" \) -delete
find $OUT -type f -size +8M -delete
du -sh $OUT
echo "measure done"
