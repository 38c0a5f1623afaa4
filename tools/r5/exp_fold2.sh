set -x
mkdir -p gpurun_out/r5
# (first bench of a fresh box reads high: a throw-away run first)
python bench.py --workload annulus-260k --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
for wl in micca-real annulus-66k annulus-260k annulus-1M annulus-5M; do
  for cfg in "1 1" "1 0" "0 0"; do set -- $cfg
    HX_AMG_FOLD_DIAG=$1 HX_AMG_FOLD_RESIDUAL=$2 timeout -k 10 300 python bench.py --workload $wl --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r5/fold2_d$1r$2_$wl.json 2> gpurun_out/r5/fold2_d$1r$2_$wl.log
  done
done
echo done
