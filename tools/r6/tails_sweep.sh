#!/bin/bash
# parameters of the body / tail split on the refined real mesh, kernels alone
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"
for cfg in "0 128 8" "1 128 8" "1 64 8" "1 32 8" "1 256 8" "1 128 1" "1 64 1" "1 32 1" "1 16 1" "1 128 2" "1 128 4"; do
  set -- $cfg
  echo "== HX_SELL_TAIL=$1 HX_SELL_TAIL_STEP=$2 HX_SELL_TAIL_GRAN=$3"
  HX_SELL_TAIL=$1 HX_SELL_TAIL_STEP=$2 HX_SELL_TAIL_GRAN=$3 timeout -k 10 300 python tools/sell_kernels_only.py micca-real-r2 20 2>&1 | grep "k_sell\|tail"
done
