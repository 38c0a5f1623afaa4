# envelope sweeps behind profiles/r05_envelope_with_the_ladder.txt (tools/summarize_envelope.py makes the table)
set -x
mkdir -p gpurun_out/r5
T=1000,2000,3225.12+481j,5000,9000,10000,11000,13000,15000,17500,20000,22000,25000
timeout -k 10 400 python tools/sweep_envelope.py annulus-1M --targets $T --out gpurun_out/r5/final_env_1M.jsonl > gpurun_out/r5/final_env_1M.log 2>&1
timeout -k 10 500 python tools/sweep_envelope.py micca-real-r2 --targets $T --out gpurun_out/r5/final_env_r2.jsonl > gpurun_out/r5/final_env_r2.log 2>&1
timeout -k 10 600 python tools/sweep_envelope.py annulus-5M --targets 1000,2000,3225.12+481j,5000,9000,10000,11000,13000,15000,17500,20000,22000 --out gpurun_out/r5/final_env_5M.jsonl > gpurun_out/r5/final_env_5M.log 2>&1
echo done
