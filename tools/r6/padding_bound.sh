#!/bin/bash
# VERDICT r5 item 2b, upper bound: what the SELL kernels of an unstructured operator would gain if the 24 % padding of the slices
# cost nothing (an ELL body + overflow tail at best).  Timing-only ablation HX_SELL_ABLATE=4*q: every slice is cut to (16 - q)/16
# of its width (results are wrong; 13/16 = the share of true entries among the stored ones on the refined real mesh).
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"
for wl in "micca-real-r2" "annulus-5M 10 rcm 0.2"; do
  for q in 0 2 3 4; do
    echo "== $wl   HX_SELL_ABLATE=$((4 * q))  (width x $((16 - q))/16)"
    HX_SELL_ABLATE=$((4 * q)) timeout -k 10 300 python tools/sell_kernels_only.py ${wl/micca-real-r2/micca-real-r2 20} 2>&1 | grep -v amdgpu.ids
  done
done
