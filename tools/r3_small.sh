#!/bin/bash
# small problems: flexible GMRES (default below cocg_min_rows) against COCG as the inner Krylov method
set -o pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
mkdir -p gpurun_out/small
for wl in annulus-16k annulus-34k annulus-66k annulus-260k micca-real; do
  for k in auto cocg; do
    HX_AMG_KRYLOV=$k timeout -k 10 300 python bench.py --workload $wl --st-type amg --no-cpu-baseline --warmup 1 --steps 3 2>/dev/null | tail -1 > gpurun_out/small/${wl}_$k.json || exit 1
    python - <<PY
import json
d=json.loads(open("gpurun_out/small/${wl}_$k.json").read()); i=d["config"]["inner_stats"]
print("$wl $k", d["ms_each_step"], i.get("krylov"), i.get("gmres_iters"), i.get("iters_per_solve"), i.get("recycled_pairs"), d["config"]["krylov"]["krylov_ops"], d["config"]["omega"][0])
PY
  done
done
