#!/bin/bash
# SQ counters of the engine kernel (separate rocprofv3 --pmc passes, kernel-trace only).  Usage: tools/engine_pmc.sh <workload> [engine flag]
set -o pipefail
W=${1:-llama2-7b-int4}
OUT=gpurun_out/engpmc_$W
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
i=0
for set in "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_IFETCH"; do
  i=$((i+1))
  timeout -k 10 600 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o p$i -- python3 bench.py --workload $W --steps 8 --warmup 2 --engine 1 --no-cpu-baseline > $OUT/p$i.log 2>&1 || { tail -5 $OUT/p$i.log; exit 1; }
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(list)
for f in glob.glob(out + "/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "eng_token" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    v = agg[k]
    print(f"{k:28s} mean {sum(v)/len(v):16.0f}  (n={len(v)})")
PY
find $OUT -type f \( -name "*.db" -o -name "*kernel_trace.csv" \) -delete
