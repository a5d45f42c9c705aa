#!/bin/bash
OUT=gpurun_out/$1; mkdir -p $OUT
python lit-parrot_amd/_build.py --diag > $OUT/build_diag.log 2>&1 || { echo "diag build failed"; tail -5 $OUT/build_diag.log; exit 1; }
timeout -k 10 300 python tools/attn_stamps.py 2>&1 | tee $OUT/attn_stamps.txt
python lit-parrot_amd/_build.py > $OUT/build_restore.log 2>&1
