set -x
mkdir -p gpurun_out/r5
timeout -k 10 500 python tools/sweep_envelope.py annulus-1M --targets 9000,11000,15000,17500,20000,22000,25000,30000,35000 --out gpurun_out/r5/lad2_1M.jsonl > gpurun_out/r5/lad2_1M.log 2>&1
HX_AMG_RUNG=1 timeout -k 10 300 python tools/sweep_envelope.py annulus-1M --legs passive --targets 22000,25000,30000,35000 --stop-after-failures 4 --out gpurun_out/r5/lad2_1M_rung1.jsonl > gpurun_out/r5/lad2_1M_rung1.log 2>&1
HX_AMG_RUNG=2 timeout -k 10 300 python tools/sweep_envelope.py annulus-1M --legs passive --targets 22000,25000,30000,35000 --stop-after-failures 4 --out gpurun_out/r5/lad2_1M_rung2.jsonl > gpurun_out/r5/lad2_1M_rung2.log 2>&1
timeout -k 10 500 python tools/sweep_envelope.py micca-real-r2 --targets 9000,11000,15000,17500,20000,22000,25000 --out gpurun_out/r5/lad2_r2.jsonl > gpurun_out/r5/lad2_r2.log 2>&1
timeout -k 10 600 python tools/sweep_envelope.py annulus-5M --targets 9000,11000,15000,17500,20000,22000 --out gpurun_out/r5/lad2_5M.jsonl > gpurun_out/r5/lad2_5M.log 2>&1
echo done
