#!/bin/bash
# Second sweep: lower bound of the Chebyshev interval (cheb_lo) / 4th-kind polynomials on the structured and the real mesh.
OUT=gpurun_out/r4/sweep_cheb.txt
mkdir -p gpurun_out/r4
: > $OUT
run() {
  W=$1; name=$2; shift; shift
  env HX_QUIET=1 "$@" timeout -k 10 300 python bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline > /tmp/sweep.json 2> /tmp/sweep.err || { echo "$W $name FAILED" >> $OUT; return 0; }
  python - "$W $name" >> $OUT <<'EOF'
import json, sys
d = json.load(open("/tmp/sweep.json")); c = d["config"]; s = c["inner_stats"]; k = c["krylov"]
print("%-36s ms/step %8.1f %s its/solve %5.1f  inner its %5d  ops %3d  omega0 %.9f%+.9fj" % (
    sys.argv[1], d["ms_per_step"], d["ms_each_step"], s["iters_per_solve"], s["gmres_iters"], k["krylov_ops"], *c["omega"][0]))
EOF
}
run annulus-5M default
run annulus-5M cheb_lo0.15 HX_CHEB_LO=0.15
run annulus-5M cheb_lo0.2 HX_CHEB_LO=0.2
run annulus-5M cheb_lo0.3 HX_CHEB_LO=0.3
run annulus-5M cheb_kind4 HX_CHEB_KIND=4
run micca-real-r2 default
run micca-real-r2 cheb_lo0.3 HX_CHEB_LO=0.3
run micca-real-r2 cheb_lo0.2_theta0.02 HX_CHEB_LO=0.2 HX_AMG_THETA=0.02
run micca-real-r2 kind4_theta0.02 HX_CHEB_KIND=4 HX_AMG_THETA=0.02
run annulus-1M default
run annulus-1M cheb_lo0.2 HX_CHEB_LO=0.2
run annulus-1M cheb_kind4 HX_CHEB_KIND=4
cat $OUT
