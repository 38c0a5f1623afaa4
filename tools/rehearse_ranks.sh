#!/bin/bash
# rehearsal of --gpus N on the one-GPU box: N ranks share the device (gloo, then library-issued exchanges through the librccl stand-in)
# usage: rehearse_ranks.sh [workload [ranks]]
WL=${1:-annulus-1M}
NR=${2:-4}
if [ "$NR" -gt 5 ]; then echo "at most 5 ranks on a one-GPU box (the pool allows 6 processes on the card, the launcher's agent counts)"; exit 2; fi
mkdir -p gpurun_out/rehearsal
HX_DIST_BACKEND=gloo timeout -k 10 700 python bench.py --gpus $NR --workload $WL --steps 1 --warmup 1 > gpurun_out/rehearsal/bench_${NR}ranks_${WL}_gloo.json 2> gpurun_out/rehearsal/bench_${NR}ranks_${WL}_gloo.err; echo "gloo rc=$?"
HX_DIST_BACKEND=gloo HX_NATIVE_COMM=1 HX_RCCL_LIB=$PWD/tests/fake_rccl/libfakerccl.so timeout -k 10 700 python bench.py --gpus $NR --workload $WL --steps 1 --warmup 1 > gpurun_out/rehearsal/bench_${NR}ranks_${WL}_fake.json 2> gpurun_out/rehearsal/bench_${NR}ranks_${WL}_fake.err; echo "fake rc=$?"
for f in gloo fake; do python - <<PY
import json
d=json.loads(open("gpurun_out/rehearsal/bench_${NR}ranks_${WL}_$f.json").read().strip().splitlines()[-1])
c=d["config"]; print("$f", d.get("error"), d["ms_per_step"], c.get("partition"), c["inner_stats"]["gmres_iters"], c["inner_stats"]["sharded_levels"], c["exchanges_per_step"], c["omega"])
PY
done
