# size-capped aggregation on the refined real mesh (needs tools/experiments/size_capped_aggregation.patch applied: HX_AMG_AGG_CAP / _SEED_MIN)
set -x
mkdir -p gpurun_out/r5
for cap in 0 6 8 10 12; do
  HX_AMG_AGG_CAP=$cap timeout -k 10 300 python bench.py --workload micca-real-r2 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r5/agg_r2_cap$cap.json 2> gpurun_out/r5/agg_r2_cap$cap.log
done
HX_AMG_AGG_CAP=8 HX_AMG_AGG_SEED_MIN=5 timeout -k 10 300 python bench.py --workload micca-real-r2 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r5/agg_r2_cap8_seed5.json 2> gpurun_out/r5/agg_r2_cap8_seed5.log
HX_AMG_AGG_CAP=8 HX_AMG_AGG_SEED_MIN=2 timeout -k 10 300 python bench.py --workload micca-real-r2 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r5/agg_r2_cap8_seed2.json 2> gpurun_out/r5/agg_r2_cap8_seed2.log
echo done
