#!/bin/bash
# Round-3 measurement pass of ONE (workload, executor): un-profiled bench line, rocprofv3 kernel stats, PMC FETCH_SIZE and
# WRITE_SIZE passes (separate runs, --kernel-trace only beside --pmc).  The PMC passes instrument the LIBRARY's kernels only
# (--kernel-include-regex parrot): round 2's silent Falcon-40B pass, caught by the watchdog in round 3 (phase 'building the
# synthetic model', profiles/r03b_falcon-40b-int4_pmc_hang.txt), sat in torch's own model-building kernels under counter
# collection - nothing of this library had run yet.  Every profiled run is bench.py itself behind `--`, carries
# the phase markers / watchdog of bench.py (--watchdog 90: a silent run says where it is and exits 3) and its own `timeout -k`.
# usage: tools/measure_r03.sh <tag> <workload> <run-name> [bench flags ...]     outputs: gpurun_out/measure_<tag>_<run-name>/
set -o pipefail
TAG=$1; W=$2; RUN=$3; shift 3
OUT=gpurun_out/measure_${TAG}_$RUN
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
B="bench.py --workload $W --no-cpu-baseline --no-sampled --watchdog 90 $*"
echo "== $RUN: bench" | tee $OUT/progress.txt
timeout -k 10 600 python $B --steps 256 > $OUT/${RUN}_bench.json 2> $OUT/${RUN}_bench.err || { echo "bench failed"; tail -3 $OUT/${RUN}_bench.err; exit 1; }
echo "== $RUN: kernel stats" | tee -a $OUT/progress.txt
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o $RUN -- python3 $B --steps 64 > $OUT/prof.log 2>&1 || { echo "stats pass failed"; tail -5 $OUT/prof.log; exit 1; }
echo "== $RUN: pmc FETCH_SIZE" | tee -a $OUT/progress.txt
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-include-regex parrot --kernel-trace --output-format csv -d $OUT/pmc_fetch -o fetch -- python3 $B --steps 16 --warmup 4 > $OUT/pmc_fetch.log 2>&1 || { echo "FETCH pass failed"; tail -5 $OUT/pmc_fetch.log; exit 1; }
echo "== $RUN: pmc WRITE_SIZE" | tee -a $OUT/progress.txt
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-include-regex parrot --kernel-trace --output-format csv -d $OUT/pmc_write -o write -- python3 $B --steps 16 --warmup 4 > $OUT/pmc_write.log 2>&1 || { echo "WRITE pass failed"; tail -8 $OUT/pmc_write.log; exit 1; }
find $OUT -type f \( -name "*.db" -o -name "*.pftrace" -o -name "*.json.gz" \) -delete
python tools/summarize_profiles.py ${TAG}_$RUN --stats $OUT/prof --fetch $OUT/pmc_fetch --write $OUT/pmc_write --bench $OUT/${RUN}_bench.json > $OUT/summary.log 2>&1 || { echo "summary failed"; tail -5 $OUT/summary.log; }
find $OUT -type f -name "*kernel_trace.csv" -delete
find $OUT -type f -size +8M -delete
mkdir -p gpurun_out/profiles_$TAG && cp profiles/${TAG}_${RUN}_* gpurun_out/profiles_$TAG/ 2>/dev/null
echo "== $RUN done" | tee -a $OUT/progress.txt
