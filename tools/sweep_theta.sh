#!/bin/bash
# Third sweep (refined real mesh): weaker strength filter, stronger prolongator damping.
OUT=gpurun_out/r4/sweep_theta.txt
mkdir -p gpurun_out/r4
: > $OUT
run() {
  W=$1; name=$2; shift; shift
  env HX_QUIET=1 "$@" timeout -k 10 300 python bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline > /tmp/sweep.json 2> /tmp/sweep.err || { echo "$W $name FAILED: $(tail -1 /tmp/sweep.err)" >> $OUT; return 0; }
  python - "$W $name" >> $OUT <<'EOF'
import json, sys
d = json.load(open("/tmp/sweep.json")); c = d["config"]; s = c["inner_stats"]; k = c["krylov"]
print("%-36s ms/step %8.1f %s its/solve %5.1f  inner its %5d  ops %3d  levels %s" % (
    sys.argv[1], d["ms_per_step"], d["ms_each_step"], s["iters_per_solve"], s["gmres_iters"], k["krylov_ops"], s["levels"]))
EOF
}
run micca-real-r2 default_first_run_reads_high
run micca-real-r2 default
run micca-real-r2 theta0.01 HX_AMG_THETA=0.01
run micca-real-r2 theta0.005 HX_AMG_THETA=0.005
run micca-real-r2 theta0 HX_AMG_THETA=0.0
run micca-real-r2 pdamp1.5 HX_AMG_PDAMP=1.5
run micca-real-r2 pdamp1.7 HX_AMG_PDAMP=1.7
run micca-real-r2 theta0.01_pdamp1.5 HX_AMG_THETA=0.01 HX_AMG_PDAMP=1.5
run annulus-5M theta0.02 HX_AMG_THETA=0.02
run annulus-5M pdamp1.5 HX_AMG_PDAMP=1.5
cat $OUT
