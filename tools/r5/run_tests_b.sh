set -x
mkdir -p gpurun_out/r5
python -m pytest tests/test_distributed_gloo.py -q -m gpu -k "outstanding_request_guard or 200k_rows or partitions_of_the_refined" > gpurun_out/r5/test_shard2.log 2>&1; tail -6 gpurun_out/r5/test_shard2.log
python -m pytest tests/test_gpu_parity.py -x -q -k "escalation_ladder" > gpurun_out/r5/test_ladder.log 2>&1; tail -6 gpurun_out/r5/test_ladder.log
echo done
