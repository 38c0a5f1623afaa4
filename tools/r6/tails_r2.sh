#!/bin/bash
# ELL body + overflow tail (HX_SELL_TAIL=1, default) against slices as wide as their longest row (=0) on the
# unstructured workloads: kernels alone, then whole bench calls.
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out/r6/tails
for t in 0 1; do
  echo "== kernels alone, micca-real-r2, HX_SELL_TAIL=$t"
  HX_SELL_TAIL=$t timeout -k 10 300 python tools/sell_kernels_only.py micca-real-r2 20 2>&1 | grep "k_sell\|n="
done
for wl in "--workload micca-real-r2" "--numbering rcm --jitter 0.2"; do
  for t in 0 1; do
    tag=$(echo "$wl" | tr -d ' -.')_tail$t
    HX_SELL_TAIL=$t timeout -k 10 400 python bench.py $wl --no-cpu-baseline --warmup 1 --steps 3 > gpurun_out/r6/tails/$tag.json 2> gpurun_out/r6/tails/$tag.err || { echo "FAILED $tag"; tail -5 gpurun_out/r6/tails/$tag.err; exit 1; }
    python - $tag "$wl" $t <<'PY'
import json, sys
d = json.load(open("gpurun_out/r6/tails/%s.json" % sys.argv[1]))
r = d["roofline"]; s = r.get("dominant_by_time", {})
print("%-32s HX_SELL_TAIL=%s  %.1f ms/call, product %.4f ms (frac %.3f, by format bytes %.3f), smoother step %s ms (frac %s), inner its/solve %s, omega0 %s" % (
    sys.argv[2], sys.argv[3], d["ms_per_step"], r.get("avg_launch_ms", 0), r["frac"], r.get("frac_format_bytes", 0), s.get("avg_launch_ms"), s.get("frac"),
    d["config"]["inner_stats"]["iters_per_solve"], d["config"]["omega"][0]), flush=True)
print("    ", r["format"])
PY
  done
done
