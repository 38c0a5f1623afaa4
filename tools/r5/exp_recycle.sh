set -x
mkdir -p gpurun_out/r5
T=9000,11000,15000,17500,20000,22000,25000
for cfg in "8 50" "16 50" "16 100" "32 100" "32 120"; do
  set -- $cfg
  HX_AMG_RECYCLE=$1 HX_AMG_RESTART=$2 timeout -k 10 300 python tools/sweep_envelope.py annulus-1M --legs passive --targets $T --stop-after-failures 3 --out gpurun_out/r5/rec_1M_k$1_m$2.jsonl > gpurun_out/r5/rec_1M_k$1_m$2.log 2>&1
done
HX_AMG_RECYCLE=32 HX_AMG_RESTART=100 HX_AMG_RECYCLE_THETA=0.6 timeout -k 10 300 python tools/sweep_envelope.py annulus-1M --legs passive --targets $T --stop-after-failures 3 --out gpurun_out/r5/rec_1M_k32_m100_th06.jsonl > gpurun_out/r5/rec_1M_k32_m100_th06.log 2>&1
echo done
