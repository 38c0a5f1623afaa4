#!/bin/bash
# XCD-contiguous chunks for the complex128 product at 5 M rows (HX_SELL_XCD=3: both value types), against the default
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out/r6/xcdp
run() {
  tag=$1
  timeout -k 10 300 python bench.py --no-cpu-baseline --warmup 1 --steps 2 > gpurun_out/r6/xcdp/$tag.json 2> gpurun_out/r6/xcdp/$tag.err || return 1
  python - $tag <<'PY'
import json, sys
d = json.load(open("gpurun_out/r6/xcdp/%s.json" % sys.argv[1]))
r = d["roofline"]; s = r.get("dominant_by_time", {})
print("%-22s %.1f ms/call, product %.4f ms (frac %.3f), smoother step %s ms (frac %s)" % (
    sys.argv[1], d["ms_per_step"], r.get("avg_launch_ms", 0), r["frac"], s.get("avg_launch_ms"), s.get("frac")), flush=True)
PY
}
unset HX_SELL_XCD HX_SELL_XCD_CHUNK
run default || exit 1
for c in 2 4 16 64; do HX_SELL_XCD=3 HX_SELL_XCD_CHUNK=$c run xcd3_chunk$c || exit 1; done
run default_again
