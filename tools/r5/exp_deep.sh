set -x
mkdir -p gpurun_out/r5
timeout -k 10 900 python -m pytest tests/test_distributed_gloo.py -x -q -m gpu -k "matches_single_gpu_two_ranks" > gpurun_out/r5/test_deep.log 2>&1; tail -30 gpurun_out/r5/test_deep.log | cut -c1-300
for wl in micca-real annulus-66k; do
  timeout -k 10 400 python tools/ab_inprocess.py $wl "amg_fold_diag=1,cocg_pq_in_product=1" "amg_fold_diag=0,cocg_pq_in_product=1" "amg_fold_diag=1,cocg_pq_in_product=0" "amg_fold_diag=0,cocg_pq_in_product=0" --rounds 5 > gpurun_out/r5/ab_folds_$wl.txt 2>&1
  tail -6 gpurun_out/r5/ab_folds_$wl.txt
done
for wl in annulus-260k annulus-1M annulus-5M; do
  timeout -k 10 400 python tools/ab_inprocess.py $wl "amg_fold_diag=1" "amg_fold_diag=0" --rounds 5 > gpurun_out/r5/ab_folds_$wl.txt 2>&1
  tail -4 gpurun_out/r5/ab_folds_$wl.txt
done
echo done
