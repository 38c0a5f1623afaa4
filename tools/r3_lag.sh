#!/bin/bash
# Lagged residual-norm reads in hx_cocg (HX_COCG_LAG): kernel tests, then bench lines at 1 M and 5 M DOF with and without
set -o pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
mkdir -p gpurun_out/lag
for m in 1 0 1 0; do
  export HX_COCG_LAG=$m
  timeout -k 10 300 python bench.py --workload annulus-1M --no-cpu-baseline --warmup 1 --steps 8 2> /dev/null | tail -1 > gpurun_out/lag/b1M_$m.json || exit 1
  timeout -k 10 300 python bench.py --no-cpu-baseline --warmup 1 --steps 4 2> /dev/null | tail -1 > gpurun_out/lag/b5M_$m.json || exit 1
  python -c "
import json
for f in ('b1M','b5M'):
    d=json.load(open('gpurun_out/lag/%s_$m.json'%f)); print('lag=$m',f,d['value'],d["ms_per_step"],min(d["ms_each_step"]),d["ms_each_step"])"
done
