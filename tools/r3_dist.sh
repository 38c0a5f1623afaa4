#!/bin/bash
# round-3: the distributed path on one GPU (two ranks under gloo / the fake librccl), real RCCL on one rank, bench rehearsal
set -o pipefail
mkdir -p gpurun_out/r3
timeout -k 10 900 python -m pytest tests/test_distributed_gloo.py -m gpu -x -q -s > gpurun_out/r3/pytest_dist.log 2>&1; echo "pytest dist rc=$?" | tee gpurun_out/r3/summary_dist.txt
tail -25 gpurun_out/r3/pytest_dist.log
if [ "$1" == "bench" ]; then
timeout -k 10 500 env HX_DIST_BACKEND=gloo python bench.py --gpus 2 --workload annulus-1M --steps 1 --warmup 1 > gpurun_out/r3/bench_2ranks_1M.json 2> gpurun_out/r3/bench_2ranks_1M.err; echo "bench2 rc=$?" | tee -a gpurun_out/r3/summary_dist.txt
cut -c1-3000 gpurun_out/r3/bench_2ranks_1M.json; tail -5 gpurun_out/r3/bench_2ranks_1M.err
fi
