#!/bin/bash
# Round-2 A/B runs on one box (annulus-5M unless WL is set, one step each): label, then environment overrides.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
WL=${WL:-annulus-5M}
run() {
  local label=$1; shift
  env "$@" python bench.py --no-cpu-baseline --workload $WL > gpurun_out/sw_$label.json 2> /dev/null || { echo "$label FAILED"; return; }
  python3 - "$label" <<'PY'
import json, sys
d = json.load(open("gpurun_out/sw_%s.json" % sys.argv[1]))
s, k = d["config"]["inner_stats"], d["config"]["krylov"]
print("%-22s %7.1f ms  setup %.2f solve %.2f  solves %d its %d (%.1f/solve) fallbacks %d  ops %s  omega0 %.6f%+.6fj" % (
    sys.argv[1], d["ms_per_step"], s["setup_s"], s["solve_s"], s["solves"], s["gmres_iters"], s["iters_per_solve"],
    s["cocg_fallbacks"], k["ops_per_solve"], d["config"]["omega"][0][0], d["config"]["omega"][0][1]))
PY
}
while read -r line; do
  [ -z "$line" ] && continue
  run $line
done
