# the multi-layer exchange of b split around owned-only work (amg_deep_overlap): sharded tests, then the 4-rank rehearsal with and without.
# Output: gpurun_out/r5/overlap_<0|1>_<case>.json
set -x
mkdir -p gpurun_out/r5
timeout -k 10 900 python -m pytest tests/test_distributed_gloo.py -x -q -m gpu > gpurun_out/r5/overlap_tests.log 2>&1; rc=$?; tail -3 gpurun_out/r5/overlap_tests.log
[ $rc -eq 0 ] || exit $rc
export HX_DIST_BACKEND=gloo HX_NATIVE_COMM=1 HX_RCCL_LIB=$PWD/tests/fake_rccl/libfakerccl.so
for ov in 1 0 1 0; do
  HX_AMG_DEEP_OVERLAP=$ov timeout -k 10 600 python bench.py --gpus 4 --workload annulus-5M --steps 2 --warmup 1 >> gpurun_out/r5/overlap_${ov}_4ranks_annulus5M.json 2>> gpurun_out/r5/overlap_${ov}_4ranks_annulus5M.err || exit 1
done
for ov in 1 0; do
  HX_AMG_DEEP_OVERLAP=$ov timeout -k 10 400 python bench.py --gpus 2 --workload annulus-1M --steps 2 --warmup 1 >> gpurun_out/r5/overlap_${ov}_2ranks_annulus1M.json 2>> gpurun_out/r5/overlap_${ov}_2ranks_annulus1M.err || exit 1
done
echo done
