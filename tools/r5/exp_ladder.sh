set -x
mkdir -p gpurun_out/r5
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r5/smoke_a.log 2>&1 || { tail -20 gpurun_out/r5/smoke_a.log; exit 1; }
T=15000,17500,20000,22000,25000,30000
timeout -k 10 400 python tools/sweep_envelope.py annulus-1M --targets $T --out gpurun_out/r5/lad_1M_default.jsonl > gpurun_out/r5/lad_1M_default.log 2>&1
T2=9000,11000,15000,20000,25000,30000
HX_AMG_RUNG=2 timeout -k 10 300 python tools/sweep_envelope.py annulus-1M --legs passive --targets $T2 --out gpurun_out/r5/lad_1M_rung2.jsonl > gpurun_out/r5/lad_1M_rung2.log 2>&1
HX_AMG_RUNG=2 HX_AMG_SHIFT_BETA=0.25 timeout -k 10 300 python tools/sweep_envelope.py annulus-1M --legs passive --targets $T2 --out gpurun_out/r5/lad_1M_rung2_b025.jsonl > gpurun_out/r5/lad_1M_rung2_b025.log 2>&1
HX_AMG_RUNG=2 HX_AMG_SHIFT_BETA=1.0 timeout -k 10 300 python tools/sweep_envelope.py annulus-1M --legs passive --targets $T2 --out gpurun_out/r5/lad_1M_rung2_b1.jsonl > gpurun_out/r5/lad_1M_rung2_b1.log 2>&1
HX_AMG_RUNG=2 HX_AMG_GRAPH_FROM_A=0 timeout -k 10 300 python tools/sweep_envelope.py annulus-1M --legs passive --targets $T2 --out gpurun_out/r5/lad_1M_rung2_graphK.jsonl > gpurun_out/r5/lad_1M_rung2_graphK.log 2>&1
HX_AMG_RUNG=0 HX_AMG_GRAPH_FROM_A=1 timeout -k 10 300 python tools/sweep_envelope.py annulus-1M --legs passive --targets $T2 --out gpurun_out/r5/lad_1M_rung0_graphA.jsonl > gpurun_out/r5/lad_1M_rung0_graphA.log 2>&1
HX_AMG_RUNG=1 timeout -k 10 300 python tools/sweep_envelope.py annulus-1M --legs passive --targets 17500,20000,22000,25000 --out gpurun_out/r5/lad_1M_rung1.jsonl > gpurun_out/r5/lad_1M_rung1.log 2>&1
echo done
