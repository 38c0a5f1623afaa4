#!/bin/bash
# Timing-only ablations of the fine-level smoother step (results wrong by construction): what a row-scaled operator D^-1 K
# (no D^-1 stream, unit diagonal not stored) could save at most.  5 M DOF, back-to-back launches.
set -o pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
mkdir -p gpurun_out/ablate
rm -f gpurun_out/ablate/times.txt
for m in 0 1 2 3 0 3; do
  echo "== HX_SELL_ABLATE=$m" >> gpurun_out/ablate/times.txt
  HX_SELL_ABLATE=$m timeout -k 10 300 python tools/sell_kernels_only.py annulus-5M 300 2>&1 | grep -E "us per" >> gpurun_out/ablate/times.txt || exit 1
done
cat gpurun_out/ablate/times.txt
