#!/bin/bash
# Same-box comparison of this tree with the tree of round 5's end (a git worktree at _r5tree/, built with its own Makefile; not
# committed): alternating runs of bench.py, so that the drift of the device does not decide.  Usage: vs_round5.sh [workload ...]
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out/r6/vs5
for wl in "${@:-annulus-5M annulus-1M micca-real-r2}"; do for wl1 in $wl; do
  for i in 1 2 3; do for tree in r5 r6; do
    dir=$R; [ $tree = r5 ] && dir=$R/_r5tree
    (cd $dir && timeout -k 10 400 python bench.py --workload $wl1 --st-type amg --no-cpu-baseline --warmup 1 --steps 2 2>/dev/null) > gpurun_out/r6/vs5/${wl1}_${tree}_$i.json || { echo "FAILED $wl1 $tree"; continue; }
    python - gpurun_out/r6/vs5/${wl1}_${tree}_$i.json $wl1 $tree <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]; s = r.get("dominant_by_time") or {}
i = d["config"]["inner_stats"]
print("%-14s %s  %8.1f ms/call  product %.4f ms  smoother %s ms  solve_s %.3f setup_s %.3f its/solve %s" % (
    sys.argv[2], sys.argv[3], d["ms_per_step"], r.get("avg_launch_ms", 0), s.get("avg_launch_ms"), i["solve_s"], i["setup_s"], i["iters_per_solve"]), flush=True)
PY
  done; done
done; done
