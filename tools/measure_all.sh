#!/bin/bash
# Full measurement pass on the GPU box (run through gpurun): un-profiled bench lines, rocprofv3 kernel stats, PMC traffic.
# Usage: tools/measure_all.sh <tag>      outputs under gpurun_out/measure_<tag>/
set -o pipefail
TAG=${1:-r01}
OUT=gpurun_out/measure_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
SKIP_BENCH=${SKIP_BENCH:-0}
echo "== bench (no profiler)"
if [ "$SKIP_BENCH" != "1" ]; then
timeout -k 10 400 python bench.py --steps 256 > $OUT/llama2-7b-int4_bench.json 2> $OUT/llama2-7b-int4_bench.err || exit 1
tail -c 400 $OUT/llama2-7b-int4_bench.json; echo
# the stream engine on the same workload, and both executors over a 2k-token context
timeout -k 10 400 python bench.py --steps 256 --engine 1 --no-cpu-baseline > $OUT/llama2-7b-int4-engine_bench.json 2> $OUT/llama2-7b-int4-engine_bench.err || exit 1
timeout -k 10 400 python bench.py --steps 2048 --engine 0 --no-cpu-baseline > $OUT/llama2-7b-int4-2k_bench.json 2> $OUT/llama2-7b-int4-2k_bench.err || exit 1
timeout -k 10 400 python bench.py --steps 2048 --engine 1 --no-cpu-baseline > $OUT/llama2-7b-int4-2k-engine_bench.json 2> $OUT/llama2-7b-int4-2k-engine_bench.err || exit 1
echo "engine / long-context lines done"
for w in llama2-7b-int8 llama2-7b-nf4 stablelm-3b-bf16 falcon-40b-int4 pythia-160m-bf16; do
  timeout -k 10 600 python bench.py --workload $w --steps 128 --no-cpu-baseline > $OUT/${w}_bench.json 2> $OUT/${w}_bench.err || exit 1
  echo "$w done"
done
fi
echo "== rocprofv3 kernel stats"
for w in llama2-7b-int4 llama2-7b-int8 llama2-7b-nf4 stablelm-3b-bf16 falcon-40b-int4; do
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$w -o $w -- python3 bench.py --workload $w --steps 64 --no-cpu-baseline > $OUT/prof_${w}.log 2>&1 || exit 1
  echo "$w profiled"
done
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_llama2-7b-int4-engine -o llama2-7b-int4-engine -- python3 bench.py --steps 64 --engine 1 --no-cpu-baseline > $OUT/prof_llama2-7b-int4-engine.log 2>&1 || exit 1
echo "engine profiled"
echo "== PMC passes (headline)"
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o fetch -- python3 bench.py --steps 16 --warmup 4 --no-cpu-baseline > $OUT/pmc_fetch.log 2>&1 || exit 1
timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o write -- python3 bench.py --steps 16 --warmup 4 --no-cpu-baseline > $OUT/pmc_write.log 2>&1 || exit 1
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_engine -o fetch -- python3 bench.py --steps 16 --warmup 4 --engine 1 --no-cpu-baseline > $OUT/pmc_fetch_engine.log 2>&1 || exit 1
# keep what tools/summarize_profiles.py reads; the raw traces are too large to travel back
find $OUT -type f \( -name "*.db" -o -name "*kernel_trace.csv" -o -name "*.pftrace" -o -name "*.json.gz" \) -delete
find $OUT -type f -size +8M -delete
du -sh $OUT
echo "measure_all done"
