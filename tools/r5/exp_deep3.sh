set -x
mkdir -p gpurun_out/r5
timeout -k 10 900 python -m pytest tests/test_distributed_gloo.py -x -q -m gpu -k "matches_single_gpu_two_ranks" > gpurun_out/r5/test_deep3.log 2>&1; tail -12 gpurun_out/r5/test_deep3.log | cut -c1-300
export HX_DIST_BACKEND=gloo HX_NATIVE_COMM=1 HX_RCCL_LIB=$PWD/tests/fake_rccl/libfakerccl.so
for layers in 3; do
  HX_AMG_DEEP_LAYERS=$layers timeout -k 10 400 python bench.py --gpus 2 --workload annulus-1M --steps 1 --warmup 1 > gpurun_out/r5/deepL${layers}_2ranks_annulus1M.json 2> gpurun_out/r5/deepL${layers}_2ranks_annulus1M.err; echo rc=$?
  HX_AMG_DEEP_LAYERS=$layers timeout -k 10 600 python bench.py --gpus 4 --workload annulus-5M --steps 1 --warmup 1 > gpurun_out/r5/deepL${layers}_4ranks_annulus5M.json 2> gpurun_out/r5/deepL${layers}_4ranks_annulus5M.err; echo rc=$?
  HX_AMG_DEEP_LAYERS=$layers timeout -k 10 400 python bench.py --gpus 3 --workload micca-real-r1 --steps 1 --warmup 1 > gpurun_out/r5/deepL${layers}_3ranks_micca_real_r1.json 2> gpurun_out/r5/deepL${layers}_3ranks_micca_real_r1.err; echo rc=$?
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5/deepL*_*ranks_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        c=d["config"]; st=c["inner_stats"]
        print(f.split("/")[-1], d.get("error"), d["ms_per_step"], st["gmres_iters"], st["sharded_levels"], st.get("deep_halo_levels"), st.get("deep_halo_layers"),
              "halos/it %.2f allred/it %.2f" % (st["cocg_halos"]/max(st["cocg_iters"],1), st["cocg_allreduces"]/max(st["cocg_iters"],1)), c["exchanges_per_step"]["halo_exchanges"], c["omega"][0])
    except Exception as e:
        print(f, "ERR", e)
PY
echo done
