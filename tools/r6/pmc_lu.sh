#!/bin/bash
# Matrix-core and LDS counters of the factorisation's GEMM kernels at 249,497 DOF: one rocprofv3 pass per counter group (--pmc with
# --kernel-trace only), over tools/r6/splu_check.py.  Output: gpurun_out/r6/pmc_lu_summary.txt
set -o pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p "$R/gpurun_out/r6"
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z0-9_]*MFMA[A-Z0-9_]*\|SQ_LDS[A-Z0-9_]*\|SQ_INSTS_LDS\|SQ_ACTIVE_INST_LDS" | sort -u > "$R/gpurun_out/r6/pmc_lu_available.txt"
i=0
for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_INSTS_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_ACTIVE SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i + 1))
  rm -rf /tmp/pmclu_$i
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmclu_$i -- python3 "$R/tools/r6/splu_check.py" 1 --only 1 --no-scipy --no-trans > "$R/gpurun_out/r6/pmc_lu_$i.log" 2>&1 \
    || echo "counter group '$c' failed (see gpurun_out/r6/pmc_lu_$i.log)"
done
python3 - $(find /tmp/pmclu_* -name "*counter_collection.csv") > "$R/gpurun_out/r6/pmc_lu_summary.txt" <<'PY'
import csv, sys
from collections import defaultdict
acc = defaultdict(list)
for path in sys.argv[1:]:
    for row in csv.DictReader(open(path)):
        name = row.get("Kernel_Name", "")
        if "k_splu" in name:
            acc[(name.split("(")[0][:40], row["Counter_Name"])].append(float(row["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    print("%-40s %-30s dispatches=%5d sum=%.4e mean=%.1f" % (k, c, len(v), sum(v), sum(v) / len(v)))
PY
cat "$R/gpurun_out/r6/pmc_lu_summary.txt"
