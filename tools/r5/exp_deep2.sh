# deep-halo cycle against the classic schedule: rehearsals on the one-GPU box (ranks share the device; exchanges issued by the library
# through the asynchronous librccl stand-in).  Output: gpurun_out/r5/halo_<schedule>_<case>.json
set -x
mkdir -p gpurun_out/r5
export HX_DIST_BACKEND=gloo HX_NATIVE_COMM=1 HX_RCCL_LIB=$PWD/tests/fake_rccl/libfakerccl.so
for cfg in "classic 0 3" "deep2 1 2" "deep3 1 3"; do set -- $cfg
  HX_AMG_DEEP_HALO=$2 HX_AMG_DEEP_LAYERS=$3 timeout -k 10 400 python bench.py --gpus 2 --workload annulus-1M --steps 1 --warmup 1 > gpurun_out/r5/halo_$1_2ranks_annulus1M.json 2> gpurun_out/r5/halo_$1_2ranks_annulus1M.err; echo rc=$?
  HX_AMG_DEEP_HALO=$2 HX_AMG_DEEP_LAYERS=$3 timeout -k 10 600 python bench.py --gpus 4 --workload annulus-5M --steps 1 --warmup 1 > gpurun_out/r5/halo_$1_4ranks_annulus5M.json 2> gpurun_out/r5/halo_$1_4ranks_annulus5M.err; echo rc=$?
  HX_AMG_DEEP_HALO=$2 HX_AMG_DEEP_LAYERS=$3 timeout -k 10 400 python bench.py --gpus 3 --workload micca-real-r1 --steps 1 --warmup 1 > gpurun_out/r5/halo_$1_3ranks_micca_real_r1.json 2> gpurun_out/r5/halo_$1_3ranks_micca_real_r1.err; echo rc=$?
done
echo done
