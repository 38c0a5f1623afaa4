set -x
mkdir -p gpurun_out/r5
python -m pytest tests/test_gpu_parity.py -x -q -k "bitwise_unchanged or single_reduction_form or flagship_case or rijke3d_logged or initial_guess or annulus_fixed_point_amg" > gpurun_out/r5/test_fold3.log 2>&1; tail -3 gpurun_out/r5/test_fold3.log
python bench.py --workload annulus-260k --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
for wl in micca-real annulus-66k annulus-260k annulus-1M annulus-5M; do
  for cfg in "1 1" "0 1" "1 0"; do set -- $cfg
    HX_AMG_FOLD_DIAG=$1 HX_COCG_PQ_IN_PRODUCT=$2 timeout -k 10 300 python bench.py --workload $wl --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r5/fold3_d$1p$2_$wl.json 2> gpurun_out/r5/fold3_d$1p$2_$wl.log
  done
done
echo done
