set -x
mkdir -p gpurun_out/r5
for cfg in "last 0" "richest 0" "richest 1" "richest 2" "last 1"; do set -- $cfg
  HX_WIDE_START=$1 HX_WARM_EXTRA=$2 HX_KRYLOV_TRACE=1 timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r5/wide_$1_$2_5M.json 2> gpurun_out/r5/wide_$1_$2_5M.log
  HX_WIDE_START=$1 HX_WARM_EXTRA=$2 timeout -k 10 300 python bench.py --workload annulus-1M --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r5/wide_$1_$2_1M.json 2> /dev/null
  HX_WIDE_START=$1 HX_WARM_EXTRA=$2 timeout -k 10 300 python bench.py --workload micca-real-r2 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r5/wide_$1_$2_r2.json 2> /dev/null
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5/wide_*_*.json")):
    try:
        d=json.load(open(f)); k=d["config"]["krylov"]
        print(f.split("/")[-1], d["ms_per_step"], k["krylov_ops"]/ (d["steps"]+d["warmup"]) if False else k["krylov_ops"], k["ops_per_solve"][-8:], d["config"]["inner_stats"]["gmres_iters"], d["config"]["omega"])
    except Exception as e: print(f,"ERR",e)
PY
