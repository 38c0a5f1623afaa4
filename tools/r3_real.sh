#!/bin/bash
# The reference's real MICCA mesh as it is and refined once / twice (unstructured, real geometry): one bench line each
set -o pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
mkdir -p gpurun_out/real
for wl in micca-real micca-real-r1 micca-real-r2; do
  timeout -k 10 900 python bench.py --workload $wl --no-cpu-baseline --warmup 1 --steps 2 > gpurun_out/real/$wl.json 2> gpurun_out/real/$wl.err || { tail -20 gpurun_out/real/$wl.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/real/$wl.json").read().strip().splitlines()[-1])
c=d["config"]; i=c["inner_stats"]
print("$wl", c["dofs"], c["tets"], d["ms_each_step"], d["value"], c["fpi_iterations"], i and {k:i.get(k) for k in ("levels","fine_format","stencil_slices","iters_per_solve","gmres_iters","recycled_pairs","warm_starts","krylov","setup_s")}, c["krylov"]["ops_per_solve"], c["omega"])
PY
done
