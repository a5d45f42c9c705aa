#!/bin/bash
# bench line + rocprofv3 kernel stats + PMC traffic of ONE workload on its default executor.  Usage: tools/measure_one.sh <tag> <workload>
set -o pipefail
TAG=$1; w=$2
OUT=gpurun_out/measure_${TAG}_$w
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 900 python bench.py --workload $w --steps 256 --no-cpu-baseline > $OUT/${w}_bench.json 2> $OUT/${w}_bench.err || exit 1
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$w -o $w -- python3 bench.py --workload $w --steps 64 --no-cpu-baseline > $OUT/prof_${w}.log 2>&1 || exit 1
timeout -k 10 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -o fetch -- python3 bench.py --workload $w --steps 16 --warmup 4 --no-cpu-baseline > $OUT/pmc_fetch.log 2>&1 || exit 1
timeout -k 10 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -o write -- python3 bench.py --workload $w --steps 16 --warmup 4 --no-cpu-baseline > $OUT/pmc_write.log 2>&1 || exit 1
find $OUT -type f \( -name "*.db" -o -name "*kernel_trace.csv" -o -name "*.pftrace" -o -name "*.json.gz" \) -delete
find $OUT -type f -size +8M -delete
echo "measure done"
