#!/bin/bash
# Sweep of the V-cycle parameters on the refined real MICCA mesh (1.87 M DOF, unstructured): one bench line per variant.
# Usage (GPU box): bash tools/sweep_amg_real_mesh.sh [workload]   -> gpurun_out/r4/sweep_<workload>.txt
W=${1:-micca-real-r2}
OUT=gpurun_out/r4/sweep_${W}.txt
mkdir -p gpurun_out/r4
: > $OUT
run() {
  name=$1; shift
  env HX_QUIET=1 "$@" timeout -k 10 300 python bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline > /tmp/sweep.json 2> /tmp/sweep.err || { echo "$name FAILED" >> $OUT; return 0; }
  python - "$name" >> $OUT <<'EOF'
import json, sys
d = json.load(open("/tmp/sweep.json")); c = d["config"]; s = c["inner_stats"]; k = c["krylov"]
print("%-28s ms/step %8.1f  its/solve %5.1f  inner its %5d  ops %3d  levels %s  omega0 %.9f%+.9fj" % (
    sys.argv[1], d["ms_per_step"], s["iters_per_solve"], s["gmres_iters"], k["krylov_ops"], s["levels"], *c["omega"][0]))
EOF
}
run default
run pre3_post3 HX_AMG_PRE=3 HX_AMG_POST=3
run pre2_post3 HX_AMG_PRE=2 HX_AMG_POST=3
run pre1_post2 HX_AMG_PRE=1 HX_AMG_POST=2
run pre1_post1 HX_AMG_PRE=1 HX_AMG_POST=1
run theta0.08 HX_AMG_THETA=0.08
run theta0.02 HX_AMG_THETA=0.02
run theta0.16 HX_AMG_THETA=0.16
run pdamp1.0 HX_AMG_PDAMP=1.0
run recycle12 HX_AMG_RECYCLE=12
run recycle4 HX_AMG_RECYCLE=4
run cheb_lo0.05 HX_CHEB_LO=0.05
run cheb_lo0.2 HX_CHEB_LO=0.2
run cheb_kind4 HX_CHEB_KIND=4
cat $OUT
