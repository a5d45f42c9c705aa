set -o pipefail
timeout -k 10 900 python -m pytest tests/test_engine_gpu.py -x -q > gpurun_out/eng_tests.log 2>&1 || { tail -30 gpurun_out/eng_tests.log; exit 1; }
tail -2 gpurun_out/eng_tests.log
for w in falcon-40b-int4 falcon-7b-int4; do
  timeout -k 10 900 python bench.py --workload $w --no-cpu-baseline > gpurun_out/b2_$w.json 2> gpurun_out/b2_$w.err || { tail -5 gpurun_out/b2_$w.err; exit 1; }
  python -c "
import json,sys
r=json.loads(open('gpurun_out/b2_$w.json').read().strip().splitlines()[-1]); print('$w (default executor)', round(r['value'],1), 'tok/s', round(r['ms_per_step']*1000,1), 'us engine', r['engine'], 'frac', round(r['step_roofline']['frac'],3))"
done
