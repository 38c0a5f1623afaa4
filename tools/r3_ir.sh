#!/bin/bash
# Iterative refinement with complex64 stages in hx_cocg (HX_COCG_IR): solver tests, then bench lines with and without
set -o pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
mkdir -p gpurun_out/ir
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "amg_gmres_inner or cocg_initial_guess or symmetry_probe or near_singular or routes_give" > gpurun_out/ir/pytest.log 2>&1 || { tail -40 gpurun_out/ir/pytest.log; exit 1; }
tail -2 gpurun_out/ir/pytest.log
for m in 1 0 1 0; do
  export HX_COCG_IR=$m
  timeout -k 10 300 python bench.py --workload annulus-1M --no-cpu-baseline --warmup 1 --steps 4 2> /dev/null | tail -1 > gpurun_out/ir/b1M_$m.json || exit 1
  timeout -k 10 300 python bench.py --no-cpu-baseline --warmup 1 --steps 3 2> /dev/null | tail -1 > gpurun_out/ir/b5M_$m.json || exit 1
  python - <<PY
import json
for f in ("b1M","b5M"):
    d=json.load(open("gpurun_out/ir/%s_$m.json"%f)); i=d["config"]["inner_stats"]
    print("ir=$m",f,d["ms_each_step"],i["gmres_iters"],i["iters_per_solve"],i["solve_s"],i["attainable_stops"],i["cocg_fallbacks"],d["config"]["krylov"]["krylov_ops"],d["config"]["omega"][:2])
PY
done
