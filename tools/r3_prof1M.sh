#!/bin/bash
set -o pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
WL="${WL:-annulus-1M}"
OUT=gpurun_out/prof_$WL
mkdir -p $OUT
export TMPDIR=/tmp
rm -rf /tmp/prof_x
(cd /tmp && rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d /tmp/prof_x -- python3 "$R/bench.py" --workload $WL --no-cpu-baseline --warmup 1 --steps 1 \
    > "$R/$OUT/bench.json" 2> "$R/$OUT/prof.err") || exit 1
find /tmp/prof_x -name "*kernel_trace.csv" -exec python3 tools/trace_by_grid.py {} 60 \; > $OUT/by_level.txt
find /tmp/prof_x -name "*kernel_trace.csv" -exec python3 tools/trace_gaps.py {} 40 \; > $OUT/gaps.txt
find /tmp/prof_x -name "*kernel_trace.csv" -exec sh -c 'cut -d, -f8-11,13- "$1" | gzip > '$OUT'/kernel_trace_cut.csv.gz' _ {} \;
find /tmp/prof_x -name "*kernel_trace.csv" -exec sh -c 'head -2 "$1"' _ {} \;
find /tmp/prof_x -name "*memory_copy_trace.csv" -exec sh -c 'gzip -c "$1" > '$OUT'/memory_copy_trace.csv.gz' _ {} \;
ls -la $OUT
