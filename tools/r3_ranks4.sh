#!/bin/bash
# rehearsal of --gpus 4 on the one-GPU box: four ranks share the device (gloo, then library-issued exchanges through the librccl stand-in)
mkdir -p gpurun_out/r3
HX_DIST_BACKEND=gloo timeout -k 10 500 python bench.py --gpus 4 --workload annulus-1M --steps 1 --warmup 1 > gpurun_out/r3/bench_4ranks_1M_gloo.json 2> gpurun_out/r3/bench_4ranks_1M_gloo.err; echo "gloo rc=$?"
HX_DIST_BACKEND=gloo HX_NATIVE_COMM=1 HX_RCCL_LIB=$PWD/tests/fake_rccl/libfakerccl.so timeout -k 10 500 python bench.py --gpus 4 --workload annulus-1M --steps 1 --warmup 1 > gpurun_out/r3/bench_4ranks_1M_fake.json 2> gpurun_out/r3/bench_4ranks_1M_fake.err; echo "fake rc=$?"
for f in gloo fake; do python - <<PY
import json
d=json.loads(open("gpurun_out/r3/bench_4ranks_1M_$f.json").read().strip().splitlines()[-1])
c=d["config"]; print("$f", d.get("error"), d["ms_per_step"], c.get("partition"), c["inner_stats"]["gmres_iters"], c["inner_stats"]["sharded_levels"], c["exchanges_per_step"], c["omega"][0])
PY
done
