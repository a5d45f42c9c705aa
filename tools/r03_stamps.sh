#!/bin/bash
# per-op stamps of the engine on a workload (diagnostic build with ENG_STAMPS=1 and optional extra defines)
OUT=gpurun_out/$1; W=$2; DEFS="$3"; mkdir -p $OUT
PARROT_BUILD_DEFINES="ENG_STAMPS=1 $DEFS" python lit-parrot_amd/_build.py > $OUT/build.log 2>&1 || { echo build failed; tail -5 $OUT/build.log; exit 1; }
timeout -k 10 300 python tools/eng_stamps.py $W 8 > $OUT/stamps_$W.txt 2>&1; echo "stamps rc $?"
sed -n 2,8p $OUT/stamps_$W.txt | cut -c1-460
grep -A8 "mean per op type" $OUT/stamps_$W.txt
PARROT_BUILD_DEFINES="" python lit-parrot_amd/_build.py > $OUT/build_restore.log 2>&1
