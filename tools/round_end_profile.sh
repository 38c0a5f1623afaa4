#!/bin/bash
# Round-end measurement on the GPU box (usage: round_end_profile.sh [round tag, default r05]): default bench line (5M, with CPU baseline; warm-up pass), 1M line, rocprofv3 kernel-trace
# statistics of the same bench command, PMC traffic of the SELL-64 kernels.  Only small summaries are kept.
set -o pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
TAG="${1:-r06}"
mkdir -p gpurun_out/final
python bench.py --warmup 1 --steps 3 > gpurun_out/final/bench_5M.json 2> gpurun_out/final/bench_5M.err || exit 1
python bench.py --workload annulus-1M --no-cpu-baseline --warmup 1 --steps 3 > gpurun_out/final/bench_1M.json 2> /dev/null || exit 1
export TMPDIR=/tmp
rm -rf /tmp/prof_final
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_final -- python3 "$R/bench.py" --no-cpu-baseline --warmup 1 --steps 1 \
    > "$R/gpurun_out/final/bench_5M_under_rocprof.json" 2> "$R/gpurun_out/final/prof.err") || exit 1
find /tmp/prof_final -name "*kernel_stats.csv" -exec cp {} gpurun_out/final/kernel_stats.csv \;
find /tmp/prof_final -name "*kernel_trace.csv" -exec python3 tools/trace_by_grid.py {} 45 \; > gpurun_out/final/kernel_time_by_level.txt
python3 tools/summarize_kernel_stats.py gpurun_out/final/kernel_stats.csv > gpurun_out/final/kernel_stats_summary.txt 2>/dev/null
ls -la gpurun_out/final/
WL=annulus-5M bash tools/pmc_sell.sh > /dev/null 2>&1
cp gpurun_out/pmc_sell_summary.txt gpurun_out/final/pmc_sell_summary.txt 2>/dev/null
head -30 gpurun_out/final/kernel_stats_summary.txt
# unstructured lines (the reference's real mesh refined twice; 5 M with jitter + RCM numbering)
python bench.py --workload micca-real-r2 --no-cpu-baseline --warmup 1 --steps 3 > gpurun_out/final/bench_micca_real_r2.json 2> /dev/null
python bench.py --numbering rcm --jitter 0.2 --no-cpu-baseline --warmup 1 --steps 2 > gpurun_out/final/bench_5M_jitter_rcm.json 2> /dev/null
echo "round tag $TAG: copy gpurun_out/final/* to profiles/${TAG}_*"
