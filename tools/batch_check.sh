set -o pipefail
if [ "$SKIP_TESTS" != "1" ]; then
timeout -k 10 900 python -m pytest tests/${TESTS:-} -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1 || { tail -30 gpurun_out/gpu_tests.log; exit 1; }
tail -2 gpurun_out/gpu_tests.log
fi
for w in "$@"; do
  timeout -k 10 600 python bench.py --workload $w --engine 1 --no-cpu-baseline > gpurun_out/b1_$w.json 2> gpurun_out/b1_$w.err || { tail -5 gpurun_out/b1_$w.err; exit 1; }
  python -c "
import json,sys
r=json.loads(open('gpurun_out/b1_$w.json').read().strip().splitlines()[-1]); print('$w', round(r['value'],1), 'tok/s', round(r['ms_per_step']*1000,1), 'us engine', r['engine'], 'frac', round(r['step_roofline']['frac'],3))"
done
