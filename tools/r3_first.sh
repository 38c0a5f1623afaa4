#!/bin/bash
# round-3 first GPU session: test suite, default bench line, launcher rehearsal, 5 M DOF on unstructured numberings
set -o pipefail
mkdir -p gpurun_out/r3
python -m pytest tests -m gpu -x -q > gpurun_out/r3/pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/r3/summary.txt
tail -3 gpurun_out/r3/pytest.log
timeout -k 10 600 python bench.py --warmup 1 --steps 3 > gpurun_out/r3/bench_5M.json 2> gpurun_out/r3/bench_5M.err; echo "bench rc=$?" | tee -a gpurun_out/r3/summary.txt
timeout -k 10 500 env HX_DIST_BACKEND=gloo python bench.py --gpus 2 --workload annulus-1M --steps 1 --warmup 0 > gpurun_out/r3/bench_2ranks_1M.json 2> gpurun_out/r3/bench_2ranks_1M.err; echo "bench2 rc=$?" | tee -a gpurun_out/r3/summary.txt
timeout -k 10 600 python bench.py --no-cpu-baseline --warmup 1 --steps 2 --numbering morton --jitter 0.2 > gpurun_out/r3/bench_5M_morton.json 2> gpurun_out/r3/bench_5M_morton.err; echo "morton rc=$?" | tee -a gpurun_out/r3/summary.txt
timeout -k 10 900 python bench.py --no-cpu-baseline --warmup 0 --steps 1 --numbering random --jitter 0.2 > gpurun_out/r3/bench_5M_random.json 2> gpurun_out/r3/bench_5M_random.err; echo "random rc=$?" | tee -a gpurun_out/r3/summary.txt
for f in bench_5M bench_2ranks_1M bench_5M_morton bench_5M_random; do echo "== $f"; cut -c1-1500 gpurun_out/r3/$f.json; tail -3 gpurun_out/r3/$f.err; done
