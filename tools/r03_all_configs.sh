#!/bin/bash
# un-profiled bench lines of every BASELINE config (default executor), round 3
OUT=gpurun_out/$1; mkdir -p $OUT
for w in pythia-160m-bf16 stablelm-3b-bf16 llama2-7b-int8 llama2-7b-int4 falcon-40b-int4 falcon-7b-bf16 falcon-7b-int8 falcon-7b-int4 llama2-7b-nf4; do
  timeout -k 10 600 python bench.py --workload $w --steps 256 --warmup 16 --no-cpu-baseline > $OUT/${w}_bench.json 2> $OUT/${w}_bench.err
  echo "$w rc $? $(python -c "import json;r=json.load(open('$OUT/${w}_bench.json'));print(round(r['value'],1),'tok/s',round(r['ms_per_step']*1000,1),'us','engine',r['engine'],'step frac',round(r['step_roofline']['frac'],3),'sampled',round(r['sampled_tokens_per_s'] or 0,1),'prefill ms',round(r['prefill_ms'],2), 'prefill frac', round(r['prefill_roofline']['frac'],3))" 2>&1)" | tee -a $OUT/progress.txt
done
