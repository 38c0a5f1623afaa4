#!/bin/bash
# LU / AMG crossover: one fixed-point call (bench.py step) per workload and inner solver.  usage: crossover.sh [workloads...]
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
mkdir -p gpurun_out/r6/crossover
WLS=("$@")
[ ${#WLS[@]} -eq 0 ] && WLS=(annulus-2k annulus-16k micca-real annulus-66k micca-real-r1 annulus-260k)
for wl in "${WLS[@]}"; do
  for st in lu amg; do
    timeout -k 10 600 python bench.py --workload "$wl" --st-type $st --no-cpu-baseline --warmup 1 --steps 3 \
      > gpurun_out/r6/crossover/${wl}_${st}.json 2> gpurun_out/r6/crossover/${wl}_${st}.err || echo "FAILED $wl $st"
    python - "$wl" "$st" <<'PY'
import json, sys
wl, st = sys.argv[1:3]
try:
    d = json.load(open("gpurun_out/r6/crossover/%s_%s.json" % (wl, st)))
    c = d["config"]; i = c["inner_stats"] or {}
    extra = ("factor %.1f ms analyse %.1f ms, %d solves, nnz(L) %.2e, %s TFLOP/s, %.2f GiB" % (1e3 * i.get("factor_device_s", 0), 1e3 * i.get("analyse_s", 0), i.get("solves", 0), i.get("nnz_L", 0), i.get("factor_tflops"), i.get("factor_gib", 0))
             if st == "lu" else "setup %.3f s, %s inner its/solve" % (i.get("setup_s", 0), i.get("iters_per_solve")))
    print("%-16s %-4s dofs %8d  %8.1f ms/call  %.3f eigenpairs/s  omega0 %s  | %s" % (wl, st, c["dofs"], d["ms_per_step"], d["value"], c["omega"][0], extra), flush=True)
except Exception as e:
    print(wl, st, "no result:", e, flush=True)
PY
  done
done
