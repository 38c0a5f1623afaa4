#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "5M or config2 or config5" > gpurun_out/r3/pytest_sizes.log 2>&1; echo "pytest sizes rc=$?" | tee gpurun_out/r3/summary_sizes.txt
tail -5 gpurun_out/r3/pytest_sizes.log
timeout -k 10 900 python bench.py --no-cpu-baseline --warmup 1 --steps 2 --numbering rcm --jitter 0.2 > gpurun_out/r3/bench_5M_rcm.json 2> gpurun_out/r3/bench_5M_rcm.err; echo "rcm rc=$?" | tee -a gpurun_out/r3/summary_sizes.txt
cut -c1-1800 gpurun_out/r3/bench_5M_rcm.json; tail -3 gpurun_out/r3/bench_5M_rcm.err
