for w in falcon-40b-int4 llama2-7b-int4; do
 for e in 0 1; do
  timeout -k 10 600 python bench.py --workload $w --steps 1800 --engine $e --no-cpu-baseline > gpurun_out/long_${w}_$e.json 2> gpurun_out/long_${w}_$e.err || exit 1
  python -c "
import json
r=json.loads(open('gpurun_out/long_${w}_$e.json').read().strip().splitlines()[-1]); print('$w engine=$e', round(r['value'],1), 'tok/s', round(r['ms_per_step']*1000,1), 'us', r['engine'])"
 done
done
