#!/bin/bash
# Round-end measurement on the GPU box: default bench line (5M, with CPU baseline; one warm-up pass, as the driver's
# own run has -- the first pass of a fresh process pays code-object loading and first-touch allocation), 1M line, and the rocprofv3
# kernel-trace statistics of the same bench command.  Only the small summaries are kept (the trace CSV is > 64 MiB).
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
python bench.py --warmup 1 --steps 3 > gpurun_out/bench_final_5M.json 2> gpurun_out/bench_final_5M.err || exit 1
python bench.py --workload annulus-1M --no-cpu-baseline --warmup 1 --steps 3 > gpurun_out/bench_final_1M.json 2> /dev/null || exit 1
export TMPDIR=/tmp
rm -rf /tmp/prof_final
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_final -- python3 "$OLDPWD/bench.py" --no-cpu-baseline --warmup 1 --steps 1 \
    > "$OLDPWD/gpurun_out/bench_final_5M_under_rocprof.json" 2> "$OLDPWD/gpurun_out/bench_final_prof.err") || exit 1
find /tmp/prof_final -name "*kernel_stats.csv" -exec cp {} gpurun_out/final_kernel_stats.csv \;
find /tmp/prof_final -name "*kernel_trace.csv" -exec python3 tools/trace_by_grid.py {} 45 \; > gpurun_out/final_kernel_time_by_level.txt
ls -la gpurun_out/final_kernel_stats.csv gpurun_out/final_kernel_time_by_level.txt
