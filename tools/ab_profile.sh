#!/bin/bash
# A/B on ONE box: bench annulus-5M with the round-1 inner solver (flexible GMRES, 6 recycled pairs) and with deflated COCG,
# then the rocprofv3 kernel statistics of the COCG run.  Summaries land in gpurun_out/.
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
HX_AMG_KRYLOV=fgmres HX_AMG_RECYCLE=6 python bench.py --no-cpu-baseline --steps 2 > gpurun_out/ab_fgmres.json 2> /dev/null || exit 1
python bench.py --no-cpu-baseline --steps 2 > gpurun_out/ab_cocg.json 2> /dev/null || exit 1
export TMPDIR=/tmp
rm -rf /tmp/prof_ab
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ab -- python3 "$OLDPWD/bench.py" --no-cpu-baseline \
    > "$OLDPWD/gpurun_out/ab_cocg_under_rocprof.json" 2> "$OLDPWD/gpurun_out/ab_prof.err") || exit 1
find /tmp/prof_ab -name "*kernel_stats.csv" -exec cp {} gpurun_out/ab_kernel_stats.csv \;
python3 - <<'PY'
import csv, json
rows = list(csv.DictReader(open("gpurun_out/ab_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.2f s" % (tot / 1e9))
for r in rows[:24]:
    print("%-60s calls=%7s avg=%9.1f us  %5.2f%%" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
for f in ("ab_fgmres", "ab_cocg"):
    d = json.load(open("gpurun_out/%s.json" % f))
    print(f, d["value"], d["ms_per_step"], d["config"]["inner_stats"], d["config"]["krylov"])
PY
