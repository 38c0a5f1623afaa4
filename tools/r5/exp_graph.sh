# hipGraph replay of the V-cycle (off by default since round 2) against the folded eager cycle of round 5, in one process per case
set -x
mkdir -p gpurun_out/r5
for w in micca-real annulus-66k annulus-1M; do
  timeout -k 10 500 python tools/ab_inprocess.py $w "HX_AMG_GRAPH_MAX_ROWS=0" "HX_AMG_GRAPH_MAX_ROWS=3000000" --rounds 4 > gpurun_out/r5/graph_$w.txt 2>&1 || exit 1
  tail -4 gpurun_out/r5/graph_$w.txt
done
