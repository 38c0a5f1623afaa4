#!/bin/bash
# XCD-contiguous slice schedule of the SELL-64 kernels (HX_SELL_XCD): back-to-back launch times over a size sweep
set -o pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
mkdir -p gpurun_out/xcd
rm -f gpurun_out/xcd/sweep.txt
for wl in annulus-260k 26x832x24 40x1280x36 44x1408x40 48x1536x42 52x1664x44; do
  for m in 0 3; do
    echo "== $wl HX_SELL_XCD=$m" >> gpurun_out/xcd/sweep.txt
    HX_SELL_XCD=$m timeout -k 10 300 python tools/sell_kernels_only.py $wl 200 2>&1 | grep -E "^n=|us per" >> gpurun_out/xcd/sweep.txt || exit 1
  done
done
cat gpurun_out/xcd/sweep.txt
