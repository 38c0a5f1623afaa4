set -x
mkdir -p gpurun_out/r5
export HX_DIST_BACKEND=gloo HX_NATIVE_COMM=1 HX_RCCL_LIB=$PWD/tests/fake_rccl/libfakerccl.so
for deep in 0 1; do
  HX_AMG_DEEP_HALO=$deep timeout -k 10 400 python bench.py --gpus 2 --workload annulus-1M --steps 1 --warmup 1 > gpurun_out/r5/deep${deep}_2ranks_annulus1M.json 2> gpurun_out/r5/deep${deep}_2ranks_annulus1M.err; echo rc=$?
  HX_AMG_DEEP_HALO=$deep timeout -k 10 600 python bench.py --gpus 4 --workload annulus-5M --steps 1 --warmup 1 > gpurun_out/r5/deep${deep}_4ranks_annulus5M.json 2> gpurun_out/r5/deep${deep}_4ranks_annulus5M.err; echo rc=$?
  HX_AMG_DEEP_HALO=$deep timeout -k 10 400 python bench.py --gpus 3 --workload micca-real-r1 --steps 1 --warmup 1 > gpurun_out/r5/deep${deep}_3ranks_micca_real_r1.json 2> gpurun_out/r5/deep${deep}_3ranks_micca_real_r1.err; echo rc=$?
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r5/deep*_*ranks_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        c=d["config"]; st=c["inner_stats"]
        print(f.split("/")[-1], d.get("error"), d["ms_per_step"], st["gmres_iters"], st["sharded_levels"], st.get("deep_halo_levels"),
              "halos/it %.2f allred/it %.2f" % (st["cocg_halos"]/max(st["cocg_iters"],1), st["cocg_allreduces"]/max(st["cocg_iters"],1)), c["exchanges_per_step"], c["omega"][0])
    except Exception as e:
        print(f, "ERR", e)
PY
echo done
