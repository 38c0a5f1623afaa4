#!/bin/bash
# AMG parameter probes on the twice-refined real mesh (1.87 M DOF, unstructured)
set -o pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
mkdir -p gpurun_out/real
run() {
  tag=$1; shift
  env "$@" timeout -k 10 600 python bench.py --workload micca-real-r2 --no-cpu-baseline --warmup 1 --steps 2 > gpurun_out/real/amg_$tag.json 2> gpurun_out/real/amg_$tag.err || { tail -5 gpurun_out/real/amg_$tag.err; return 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/real/amg_$tag.json").read().strip().splitlines()[-1])
i=d["config"]["inner_stats"]
print("$tag", d["ms_each_step"], i["gmres_iters"], i["iters_per_solve"], i["levels"], i["recycled_pairs"], d["config"]["omega"][0])
PY
}
run base HX_QUIET=1
run extra1 HX_AMG_COARSE_EXTRA=1
run theta02 HX_AMG_THETA=0.02
run theta08 HX_AMG_THETA=0.08
run theta16 HX_AMG_THETA=0.16
run pdamp1 HX_AMG_PDAMP=1.0
run recycle12 HX_AMG_RECYCLE=12
