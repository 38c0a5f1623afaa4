#!/bin/bash
# small problems (COCG): with and without the recycled deflation vectors (default: from 1e5 rows on)
set -o pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
mkdir -p gpurun_out/small
for wl in annulus-16k annulus-34k annulus-66k micca-real micca-real-r1; do
  for k in default 8 4; do
    if [ $k = default ]; then unset HX_AMG_RECYCLE; else export HX_AMG_RECYCLE=$k; fi
    timeout -k 10 300 python bench.py --workload $wl --st-type amg --no-cpu-baseline --warmup 1 --steps 3 2>/dev/null | tail -1 > gpurun_out/small/${wl}_r$k.json || exit 1
    python - <<PY
import json
d=json.loads(open("gpurun_out/small/${wl}_r$k.json").read()); i=d["config"]["inner_stats"]
print("$wl recycle=$k", d["ms_each_step"], i.get("krylov"), i.get("gmres_iters"), i.get("iters_per_solve"), i.get("recycled_pairs"), d["config"]["krylov"]["krylov_ops"], d["config"]["omega"][0])
PY
  done
done
