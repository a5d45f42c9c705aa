#!/bin/bash
# Measurement of the stream engine on LLM.int8 weights (run through gpurun).   Usage: tools/measure_int8_engine.sh <tag>
set -o pipefail
TAG=${1:-r02c}
OUT=gpurun_out/measure_${TAG}_int8
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
w=llama2-7b-int8
timeout -k 10 600 python bench.py --workload $w --steps 256 > $OUT/${w}_bench.json 2> $OUT/${w}_bench.err || exit 1
timeout -k 10 600 python bench.py --workload $w --steps 256 --engine 0 --no-cpu-baseline > $OUT/${w}-multilaunch_bench.json 2> $OUT/${w}-multilaunch_bench.err || exit 1
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$w -o $w -- python3 bench.py --workload $w --steps 64 --no-cpu-baseline > $OUT/prof_${w}.log 2>&1 || exit 1
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o fetch -- python3 bench.py --workload $w --steps 16 --warmup 4 --no-cpu-baseline > $OUT/pmc_fetch.log 2>&1 || exit 1
timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o write -- python3 bench.py --workload $w --steps 16 --warmup 4 --no-cpu-baseline > $OUT/pmc_write.log 2>&1 || exit 1
find $OUT -type f \( -name "*.db" -o -name "*kernel_trace.csv" -o -name "*.pftrace" -o -name "*.json.gz" \) -delete
find $OUT -type f -size +8M -delete
echo "measure done"
