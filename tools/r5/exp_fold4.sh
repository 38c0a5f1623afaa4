set -x
mkdir -p gpurun_out/r5
for wl in micca-real annulus-66k annulus-260k annulus-1M annulus-5M; do
  timeout -k 10 400 python tools/ab_inprocess.py $wl "amg_fold_diag=1,cocg_pq_in_product=1" "amg_fold_diag=0,cocg_pq_in_product=1" "amg_fold_diag=1,cocg_pq_in_product=0" "amg_fold_diag=0,cocg_pq_in_product=0" --rounds 5 > gpurun_out/r5/ab_folds_$wl.txt 2>&1
  tail -6 gpurun_out/r5/ab_folds_$wl.txt
done
echo done
