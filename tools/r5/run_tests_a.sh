set -x
mkdir -p gpurun_out/r5
python -m pytest tests/test_gpu_parity.py -x -q -k "six_reference_targets or refined_twice" > gpurun_out/r5/test_six2.log 2>&1; tail -4 gpurun_out/r5/test_six2.log
python -m pytest tests/test_gpu_kernels.py -x -q -k "pool_notes" > gpurun_out/r5/test_kern2.log 2>&1; tail -3 gpurun_out/r5/test_kern2.log
python -m pytest tests/test_distributed_gloo.py -x -q -m gpu -k "outstanding_request_guard or 200k_rows or partitions_of_the_refined" > gpurun_out/r5/test_shard.log 2>&1; tail -6 gpurun_out/r5/test_shard.log
for cfg in "0 3" "8 5"; do set -- $cfg
  HX_AMG_AGG_CAP=$1 HX_AMG_AGG_SEED_MIN=$2 timeout -k 10 300 python bench.py --workload micca-real-r2 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r5/agg2_r2_cap$1_s$2.json 2> gpurun_out/r5/agg2_r2_cap$1_s$2.log
done
echo done
