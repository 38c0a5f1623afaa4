#!/bin/bash
# Round-6 evidence for the device sparse LU (usage on the GPU box: bash tools/r6/lu_profiles.sh): f64 MFMA rate of the device,
# factorisation / solve timings and rocprofv3 kernel statistics at 34,787 and 249,497 DOF, LU / AMG crossover (analysis reused and
# redone), envelope at 260 k DOF, bench lines with st_type lu / amg.  Only small summaries are kept (gpurun_out/r6/final/).
set -o pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"
cd "$R"
O=gpurun_out/r6/final
mkdir -p $O
{ tools/r6/bin/mfma_f64_peak 1; tools/r6/bin/mfma_f64_peak 2; } > $O/mfma_f64_peak.txt 2>&1
timeout -k 10 300 python tools/r6/splu_check.py 1 --graph 2>&1 | grep -v amdgpu.ids > $O/splu_check.txt
export TMPDIR=/tmp
rm -rf /tmp/prof_lu
(cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_lu -- python3 "$R/tools/r6/splu_check.py" 1 --only 1 --no-scipy --no-trans > "$R/$O/splu_under_rocprof.txt" 2>&1)
find /tmp/prof_lu -name "*kernel_stats.csv" -exec cp {} $O/lu_kernel_stats_micca_real_r1.csv \;
find /tmp/prof_lu -name "*kernel_trace.csv" -exec python3 tools/trace_by_grid.py {} 60 \; > $O/lu_kernel_time_by_grid_micca_real_r1.txt
timeout -k 10 600 bash tools/r6/crossover.sh 2>&1 | tee $O/crossover_reuse.txt
HX_LU_REUSE_ANALYSIS=0 timeout -k 10 400 bash tools/r6/crossover.sh annulus-16k micca-real annulus-66k micca-real-r1 annulus-260k 2>&1 | grep " lu " | tee $O/crossover_cold.txt
for wl in micca-real micca-real-r1; do for st in lu amg; do
  cp gpurun_out/r6/crossover/${wl}_${st}.json $O/bench_${wl}_${st}.json 2>/dev/null
done; done
timeout -k 10 400 python tools/sweep_envelope.py annulus-260k --st-type amg --targets 22000,25000,30000 --legs passive --out $O/env_260k_amg.jsonl > /dev/null 2>&1
timeout -k 10 300 python tools/sweep_envelope.py annulus-260k --st-type lu --targets 3225.12+481j,11000,22000,25000,30000,40000 --out $O/env_260k_lu.jsonl > /dev/null 2>&1
python tools/summarize_envelope.py $O/env_260k_amg.jsonl $O/env_260k_lu.jsonl > $O/envelope_260k.txt 2>&1
ls -la $O
