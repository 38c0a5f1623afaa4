#!/bin/bash
# p^T K p from the operator product itself (HX_COCG_PQ_IN_PRODUCT): solver tests, then bench lines with and without
set -o pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
mkdir -p gpurun_out/pq
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "amg_gmres_inner or cocg_initial_guess or symmetry_probe or near_singular or routes_give or bitwise" > gpurun_out/pq/pytest.log 2>&1 || { tail -40 gpurun_out/pq/pytest.log; exit 1; }
tail -2 gpurun_out/pq/pytest.log
for m in 1 0 1 0; do
  export HX_COCG_PQ_IN_PRODUCT=$m
  timeout -k 10 300 python bench.py --workload annulus-1M --no-cpu-baseline --warmup 1 --steps 5 2> /dev/null | tail -1 > gpurun_out/pq/b1M_$m.json || exit 1
  timeout -k 10 300 python bench.py --no-cpu-baseline --warmup 1 --steps 3 2> /dev/null | tail -1 > gpurun_out/pq/b5M_$m.json || exit 1
  python - <<PY
import json
for f in ("b1M","b5M"):
    d=json.load(open("gpurun_out/pq/%s_$m.json"%f)); i=d["config"]["inner_stats"]
    print("pq_in_product=$m",f,d["ms_each_step"],i["gmres_iters"],i["solve_s"],d["roofline"]["avg_launch_ms"],d["config"]["omega"][:1])
PY
done
