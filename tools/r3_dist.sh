#!/bin/bash
# round-3: the distributed path on one GPU (two ranks under gloo / the fake librccl), real RCCL on one rank, bench rehearsals
set -o pipefail
mkdir -p gpurun_out/r3
timeout -k 10 900 python -m pytest tests/test_distributed_gloo.py -m gpu -x -q -s > gpurun_out/r3/pytest_dist.log 2>&1; echo "pytest dist rc=$?" | tee gpurun_out/r3/summary_dist.txt
grep -v "Gloo\|socket.cpp\|amdgpu.ids" gpurun_out/r3/pytest_dist.log | tail -25
for wl in $@; do
timeout -k 10 600 env HX_DIST_BACKEND=gloo python bench.py --gpus 2 --workload $wl --steps 1 --warmup 1 > gpurun_out/r3/bench_2ranks_$wl.json 2> gpurun_out/r3/bench_2ranks_$wl.err; echo "bench2 $wl rc=$?" | tee -a gpurun_out/r3/summary_dist.txt
cut -c1-2600 gpurun_out/r3/bench_2ranks_$wl.json; tail -3 gpurun_out/r3/bench_2ranks_$wl.err
done
