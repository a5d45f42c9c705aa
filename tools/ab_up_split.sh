# A/B of how the MLP up-projection is split around the attention halves (StreamEngine.UP_SPLIT), through gpurun
for w in stablelm-3b-bf16 falcon-40b-int4; do
 for sp in 1,1,1 1,2,1 2,1,1 1,1,2 2,2,1; do
  timeout -k 10 600 python -c "
import sys, runpy
sys.path.insert(0, '.')
from lit_parrot_amd.engine import StreamEngine
StreamEngine.UP_SPLIT = tuple(int(v) for v in '$sp'.split(','))
sys.argv = ['bench.py', '--workload', '$w', '--no-cpu-baseline']
runpy.run_path('bench.py', run_name='__main__')" 2>/dev/null | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('$w split $sp', round(r['value'],1), 'tok/s', r['engine'])"
 done
done
