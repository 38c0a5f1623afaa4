set -x
mkdir -p gpurun_out/r5
python -m pytest tests/test_gpu_parity.py -x -q -k "bitwise_unchanged or single_reduction_form" > gpurun_out/r5/test_fold.log 2>&1; tail -3 gpurun_out/r5/test_fold.log
for wl in annulus-1M annulus-5M micca-real; do
  for fold in 1 0; do
    HX_AMG_FOLD_DIAG=$fold timeout -k 10 300 python bench.py --workload $wl --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r5/fold${fold}_$wl.json 2> gpurun_out/r5/fold${fold}_$wl.log
  done
done
HX_AMG_GRAPH_FROM_A=1 timeout -k 10 300 python bench.py --workload annulus-5M --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r5/graphA_annulus-5M.json 2> gpurun_out/r5/graphA_annulus-5M.log
HX_AMG_GRAPH_FROM_A=1 timeout -k 10 300 python bench.py --workload micca-real-r2 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r5/graphA_r2.json 2> gpurun_out/r5/graphA_r2.log
HX_AMG_GRAPH_FROM_A=1 timeout -k 10 300 python bench.py --workload annulus-1M --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r5/graphA_annulus-1M.json 2> gpurun_out/r5/graphA_annulus-1M.log
echo done
