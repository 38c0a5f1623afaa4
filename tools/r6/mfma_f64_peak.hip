// Bare v_mfma_f64_16x16x4_f64 rate on the device: every wave issues the instruction back to back on 8 independent accumulators;
// 4 waves per workgroup (one per SIMD), `wg_per_cu` workgroups per CU.  Prints TFLOP/s (2 * 16 * 16 * 4 flops per instruction).
// build: hipcc -O3 --offload-arch=gfx950 [-DNACC=16] mfma_f64_peak.hip -o bin/mfma_f64_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef double v4d __attribute__((ext_vector_type(4)));
#ifndef NACC
#define NACC 8          // independent accumulators per wave
#endif

__global__ void __launch_bounds__(256) k_mfma(double* out, int iters, double a0, double b0) {
  v4d acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (v4d){0, 0, 0, 0};
  double a = a0 + threadIdx.x * 1e-3, b = b0 - threadIdx.x * 1e-3;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main(int argc, char** argv) {
  const int wg_per_cu = argc > 1 ? atoi(argv[1]) : 1, iters = 20000;
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int grid = p.multiProcessorCount * wg_per_cu;
  double* out; hipMalloc(&out, (size_t)grid * 256 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k_mfma<<<grid, 256>>>(out, 100, 1.0, 2.0);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k_mfma<<<grid, 256>>>(out, iters, 1.0, 2.0);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)grid * 4 * iters * NACC * 2048.0;
  printf("%d CUs x %d workgroup(s) of 4 waves, %d accumulators per wave: %.1f TFLOP/s f64 MFMA (%.3f ms); per CU and clock at 2.4 GHz: %.1f flop\n", p.multiProcessorCount,
         wg_per_cu, NACC, flops / ms / 1e9, ms, flops / (ms * 1e-3) / p.multiProcessorCount / 2.4e9);
  return 0;
}
