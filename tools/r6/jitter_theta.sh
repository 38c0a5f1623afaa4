#!/bin/bash
# strength threshold of the aggregation on a JITTERED mesh (1 M DOF, native numbering, nodes displaced by 0.2 h): inner iterations per solve
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"
for th in default 0.01 0.02 0.08 0.15 0.25; do
  if [ $th = default ]; then unset HX_AMG_THETA; else export HX_AMG_THETA=$th; fi
  python bench.py --workload annulus-1M --jitter 0.2 --no-cpu-baseline --warmup 0 --steps 1 2>/dev/null | python -c "
import json,sys,os
d=json.loads(sys.stdin.read()); i=d['config']['inner_stats']
print('theta', os.environ.get('HX_AMG_THETA','0.04 (default)'), 'ms', d['ms_per_step'], 'its/solve', i['iters_per_solve'], 'levels', i['levels'], 'complexity', i['operator_complexity'])"
done
unset HX_AMG_THETA
python bench.py --workload annulus-1M --jitter 0.0 --no-cpu-baseline --warmup 0 --steps 1 2>/dev/null | python -c "
import json,sys,os
d=json.loads(sys.stdin.read()); i=d['config']['inner_stats']
print('no jitter, default theta', 'ms', d['ms_per_step'], 'its/solve', i['iters_per_solve'], 'levels', i['levels'], 'complexity', i['operator_complexity'])"
