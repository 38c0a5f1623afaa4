#!/bin/bash
# HBM traffic of the operator-product kernels from the PMC counters: two separate rocprofv3 passes (FETCH_SIZE, WRITE_SIZE),
# each with --kernel-trace only, over tools/spmv_only.py.  Output: gpurun_out/pmc_spmv_summary.txt
set -o pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -- python3 "$R/tools/spmv_only.py" annulus-5M 10 > "$R/gpurun_out/pmc_$c.log" 2>&1 || exit 1
done
{
  for k in "k_sell<" "k_spmv<8"; do
    echo "== $k"
    python3 "$R/tools/summarize_pmc.py" "$k" $(find /tmp/pmc_FETCH_SIZE /tmp/pmc_WRITE_SIZE -name "*counter_collection.csv")
  done
} > "$R/gpurun_out/pmc_spmv_summary.txt"
cat "$R/gpurun_out/pmc_spmv_summary.txt"
