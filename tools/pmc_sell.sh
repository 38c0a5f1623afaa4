#!/bin/bash
# Memory-side counters of the SELL-64 kernels (operator product, V-cycle residual, smoother step): one rocprofv3 pass per counter
# group, each with --kernel-trace only, over tools/sell_kernels_only.py.  Output: gpurun_out/pmc_sell_summary.txt
set -o pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"
WL="${WL:-annulus-5M}"
export TMPDIR=/tmp
cd /tmp
i=0
GROUPS_DEFAULT=("FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum")
GROUPS_SQ=("SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES GRBM_GUI_ACTIVE" "SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR")
if [ "${PMC_SET:-mem}" = "sq" ]; then CSETS=("${GROUPS_SQ[@]}"); elif [ "${PMC_SET:-mem}" = "hbm" ]; then CSETS=("FETCH_SIZE" "WRITE_SIZE"); else CSETS=("${GROUPS_DEFAULT[@]}"); fi
for c in "${CSETS[@]}"; do
  i=$((i + 1))
  rm -rf /tmp/pmcs_$i
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmcs_$i -- python3 "$R/tools/sell_kernels_only.py" "$WL" 10 ${NUMBERING:-native} ${JITTER:-0} > "$R/gpurun_out/pmc_sell_$i.log" 2>&1 \
    || echo "counter group '$c' failed (see gpurun_out/pmc_sell_$i.log)"
done
python3 - $(find /tmp/pmcs_* -name "*counter_collection.csv") > "$R/gpurun_out/pmc_sell_summary.txt" <<'PY'
import csv, sys
from collections import defaultdict
acc = defaultdict(list)
for path in sys.argv[1:]:
    for row in csv.DictReader(open(path)):
        name = row.get("Kernel_Name", "")
        if "k_sell<" in name:
            acc[(name.split("(")[0][:90], row["Counter_Name"])].append(float(row["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    print("%-90s %-28s dispatches=%d mean=%.1f" % (k, c, len(v), sum(v) / len(v)))
PY
cat "$R/gpurun_out/pmc_sell_summary.txt"
