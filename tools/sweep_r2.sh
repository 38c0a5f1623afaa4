#!/bin/bash
# Round-2 parameter sweep on one box (annulus-5M, one step each): deflation pairs x Chebyshev kind x spare Krylov columns.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
run() {  # label, env...
  local label=$1; shift
  env "$@" python bench.py --no-cpu-baseline > gpurun_out/sw_$label.json 2> /dev/null || { echo "$label FAILED"; return; }
  python3 - "$label" <<'PY'
import json, sys
d = json.load(open("gpurun_out/sw_%s.json" % sys.argv[1]))
s, k = d["config"]["inner_stats"], d["config"]["krylov"]
print("%-22s %7.1f ms  setup %.2f solve %.2f  solves %d its %d (%.1f/solve) fallbacks %d  ops %s  omega0 %.6f%+.6fj" % (
    sys.argv[1], d["ms_per_step"], s["setup_s"], s["solve_s"], s["solves"], s["gmres_iters"], s["iters_per_solve"],
    s["cocg_fallbacks"], k["ops_per_solve"], d["config"]["omega"][0][0], d["config"]["omega"][0][1]))
PY
}
for k in 4 6 8 10; do
  run k${k} HX_AMG_RECYCLE=$k
  run k${k}_cheb4 HX_AMG_RECYCLE=$k HX_CHEB_KIND=4
done
run k8_lo05 HX_AMG_RECYCLE=8 HX_CHEB_LO=0.05
run k8_lo2 HX_AMG_RECYCLE=8 HX_CHEB_LO=0.2
