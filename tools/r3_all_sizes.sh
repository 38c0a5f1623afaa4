#!/bin/bash
# one default bench line per workload size (smoke over the whole range with the end-of-round defaults)
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/sizes
for wl in annulus-2k annulus-4k annulus-8k annulus-16k annulus-34k annulus-66k annulus-260k annulus-1M micca-real micca-real-r1; do
  timeout -k 10 300 python bench.py --workload $wl --no-cpu-baseline --warmup 1 --steps 3 2>gpurun_out/sizes/$wl.err | tail -1 > gpurun_out/sizes/$wl.json || { echo "$wl FAILED"; tail -3 gpurun_out/sizes/$wl.err; continue; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/sizes/$wl.json").read()); i=d["config"]["inner_stats"] or {}
print("%-14s %8d dof  %s ms  %s %s its/solve %s  levels %s  omega0 %s" % ("$wl", d["config"]["dofs"], d["ms_each_step"], d["config"]["inner_solver"], i.get("krylov"), i.get("iters_per_solve"), i.get("levels"), d["config"]["omega"][0]))
PY
done
