#!/bin/bash
# Fused COCG reductions (single GPU): full GPU suite, then the 1M and 5M bench lines with and without the fusion.
set -o pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
mkdir -p gpurun_out/fuse
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/fuse/pytest.log 2>&1 || { tail -30 gpurun_out/fuse/pytest.log; exit 1; }
tail -3 gpurun_out/fuse/pytest.log
for f in 1 0; do
  HX_COCG_FUSE_REDUCTIONS=$f timeout -k 10 300 python bench.py --workload annulus-1M --no-cpu-baseline --warmup 1 --steps 4 \
      > gpurun_out/fuse/bench_1M_fuse$f.json 2> gpurun_out/fuse/bench_1M_fuse$f.err || exit 1
  HX_COCG_FUSE_REDUCTIONS=$f timeout -k 10 300 python bench.py --no-cpu-baseline --warmup 1 --steps 3 \
      > gpurun_out/fuse/bench_5M_fuse$f.json 2> gpurun_out/fuse/bench_5M_fuse$f.err || exit 1
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/fuse/bench_*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, d["value"], d["ms_per_step"], d.get("config", {}).get("inner_iterations"))
PY
