#!/bin/bash
# same-box A/B of the complex64 chunk schedule at 5 M rows (bench, in-situ smoother timing), every setting twice
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; mkdir -p gpurun_out/r6/xcdab
for i in 1 2; do for cfg in "default" "HX_SELL_XCD=0" "HX_SELL_XCD=1 HX_SELL_XCD_CHUNK=16" "HX_SELL_XCD_CHUNK=0" "HX_SELL_XCD=1 HX_SELL_XCD_CHUNK=4"; do
  tag=$(echo "$cfg" | tr -d ' =')_$i
  if [ "$cfg" = default ]; then timeout -k 10 300 python bench.py --no-cpu-baseline --warmup 1 --steps 2 > gpurun_out/r6/xcdab/$tag.json 2> /dev/null || exit 1
  else env $cfg timeout -k 10 300 python bench.py --no-cpu-baseline --warmup 1 --steps 2 > gpurun_out/r6/xcdab/$tag.json 2> /dev/null || exit 1; fi
  python - $tag "$cfg" <<'PY'
import json, sys
d = json.load(open("gpurun_out/r6/xcdab/%s.json" % sys.argv[1]))
r = d["roofline"]; s = r.get("dominant_by_time", {})
print("%-36s %.1f ms/call, product %.4f ms (frac %.3f), smoother step %s ms (frac %s)" % (
    sys.argv[2], d["ms_per_step"], r.get("avg_launch_ms", 0), r["frac"], s.get("avg_launch_ms"), s.get("frac")), flush=True)
PY
done; done
