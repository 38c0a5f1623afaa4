#!/bin/bash
# VERDICT r5 item 5, second half: XCD-contiguous schedule of the complex64 kernels at 5 M rows with the XCD's share cut into
# chunks (HX_SELL_XCD_CHUNK workgroups of 256 rows) instead of eighths of the operator.  One bench call per setting.
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out/r6/xcdc
run() {
  tag=$1
  timeout -k 10 300 python bench.py --no-cpu-baseline --warmup 1 --steps 2 > gpurun_out/r6/xcdc/$tag.json 2> gpurun_out/r6/xcdc/$tag.err || return 1
  python - $tag <<'PY'
import json, sys
d = json.load(open("gpurun_out/r6/xcdc/%s.json" % sys.argv[1]))
r = d["roofline"]; s = r.get("dominant_by_time", {})
print("%-22s %.1f ms/call, product %.4f ms (frac %.3f), smoother step %s ms (frac %s), inner its/solve %s" % (
    sys.argv[1], d["ms_per_step"], r.get("ms_per_launch", 0), r["frac"], s.get("ms_per_launch"), s.get("frac"),
    d["config"]["inner_stats"]["iters_per_solve"]), flush=True)
PY
}
unset HX_SELL_XCD HX_SELL_XCD_CHUNK
run default || exit 1
export HX_SELL_XCD=1
run xcd1_eighths || exit 1
for c in ${CHUNKS:-64 256 1024}; do HX_SELL_XCD_CHUNK=$c run xcd1_chunk$c || exit 1; done
export HX_SELL_XCD=3
[ -n "$CHUNKS" ] || HX_SELL_XCD_CHUNK=256 run xcd3_chunk256 || exit 1
