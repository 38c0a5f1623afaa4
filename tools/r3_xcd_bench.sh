#!/bin/bash
# XCD-contiguous schedule, automatic rule vs off: bench lines at 1 M and 5 M DOF
set -o pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
mkdir -p gpurun_out/xcd
for m in auto 0 auto 0; do
  if [ $m = auto ]; then unset HX_SELL_XCD; else export HX_SELL_XCD=$m; fi
  timeout -k 10 300 python bench.py --workload annulus-1M --no-cpu-baseline --warmup 1 --steps 4 2> /dev/null | tail -1 > gpurun_out/xcd/b1M_$m.json || exit 1
  timeout -k 10 300 python bench.py --no-cpu-baseline --warmup 1 --steps 3 2> /dev/null | tail -1 > gpurun_out/xcd/b5M_$m.json || exit 1
  python -c "
import json
for f in ('b1M','b5M'):
    d=json.load(open('gpurun_out/xcd/%s_$m.json'%f)); print('$m',f,d['value'],d['ms_per_step'],d['config']['inner_stats']['gmres_iters'],d['config']['omega'][0])"
done
