set -x
mkdir -p gpurun_out/r5
python -m pytest tests/test_distributed_gloo.py -q -m gpu -k "200k_rows" > gpurun_out/r5/test_shard3.log 2>&1; tail -4 gpurun_out/r5/test_shard3.log
python -m pytest tests/test_gpu_parity.py -x -q -k "escalation_ladder" > gpurun_out/r5/test_ladder2.log 2>&1; tail -4 gpurun_out/r5/test_ladder2.log
T=1000,2000,3225.12+481j,5000,9000,10000,11000,13000,15000,17500,20000,22000,25000
timeout -k 10 400 python tools/sweep_envelope.py annulus-1M --targets $T --out gpurun_out/r5/final_env_1M.jsonl > gpurun_out/r5/final_env_1M.log 2>&1
timeout -k 10 500 python tools/sweep_envelope.py micca-real-r2 --targets $T --out gpurun_out/r5/final_env_r2.jsonl > gpurun_out/r5/final_env_r2.log 2>&1
timeout -k 10 600 python tools/sweep_envelope.py annulus-5M --targets 1000,2000,3225.12+481j,5000,9000,10000,11000,13000,15000,17500,20000,22000 --out gpurun_out/r5/final_env_5M.jsonl > gpurun_out/r5/final_env_5M.log 2>&1
echo done
