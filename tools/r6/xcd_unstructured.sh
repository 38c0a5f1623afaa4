#!/bin/bash
# XCD-contiguous slice schedule on the refined real mesh (1.87 M DOF, unstructured): never / by footprint (default) / always
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out/r6/xcd
for x in 0 default 3; do
  if [ $x = default ]; then unset HX_SELL_XCD; else export HX_SELL_XCD=$x; fi
  timeout -k 10 500 python bench.py --workload micca-real-r2 --no-cpu-baseline --warmup 1 --steps 3 > gpurun_out/r6/xcd/r2_xcd_$x.json 2> gpurun_out/r6/xcd/r2_xcd_$x.err
  python - $x <<'PY'
import json, sys
d = json.load(open("gpurun_out/r6/xcd/r2_xcd_%s.json" % sys.argv[1]))
r = d["roofline"]; s = r.get("dominant_by_time", {})
print("HX_SELL_XCD=%-8s %.1f ms/call, product %.4f ms (frac %.3f, format %.3f), smoother step %s ms (frac %s), inner its/solve %s" % (
    sys.argv[1], d["ms_per_step"], r.get("ms_per_launch", 0), r["frac"], r.get("frac_format_bytes", 0), s.get("ms_per_launch"), s.get("frac"),
    d["config"]["inner_stats"]["iters_per_solve"]), flush=True)
PY
done
