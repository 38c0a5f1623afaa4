#!/bin/bash
mkdir -p gpurun_out/r3
for inv in host device; do
for cs in 600 1000; do
HX_AMG_COARSE_INV=$inv HX_AMG_COARSE_SIZE=$cs timeout -k 10 600 python bench.py --workload annulus-5M --no-cpu-baseline --warmup 1 --steps 3 2>gpurun_out/r3/coarse_$inv_$cs.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); st=d['config']['inner_stats']
print('5M inv=$inv coarse_size=$cs', d['ms_per_step'], 'levels', st['levels'], 'its', st['gmres_iters'], 'solve_s', st['solve_s'], 'setup_s', st['setup_s'])"
done; done
HX_AMG_COARSE_INV=device HX_AMG_COARSE_SIZE=4000 timeout -k 10 600 python bench.py --workload annulus-5M --no-cpu-baseline --warmup 1 --steps 2 2>&1 | tail -5 | cut -c1-600
