// Cost of a hand-written grid barrier on gfx950 (one workgroup per CU or fewer): is a persistent "coarse tail" kernel for the
// small V-cycle levels worth building?  Build: hipcc --offload-arch=gfx950 -O3 grid_barrier_cost.hip -o grid_barrier_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ void grid_barrier(unsigned* bar, unsigned target, int* err) {
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    long spins = 0;
    while ((int)(__hip_atomic_load(bar, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1L << 24)) { *err = 1; break; }
    }
  }
  __syncthreads();
  __threadfence();
}

// every barrier is followed by a little dependent work on a vector another workgroup wrote (so that visibility is exercised)
__global__ void __launch_bounds__(256) k_barriers(unsigned* bar, unsigned base, int nb, float* v, int n, int* err) {
  const int gt = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
  for (int k = 0; k < nb; ++k) {
    for (int i = gt; i < n; i += nt) { const int j = (i + 4099) % n; v[(size_t)((k + 1) & 1) * n + i] = v[(size_t)(k & 1) * n + j] + 1.0f; }
    grid_barrier(bar, base + (unsigned)(k + 1) * gridDim.x, err);
  }
}
__global__ void __launch_bounds__(256) k_step(const float* in, float* out, int n) {
  const int gt = blockIdx.x * blockDim.x + threadIdx.x, nt = gridDim.x * blockDim.x;
  for (int i = gt; i < n; i += nt) { const int j = (i + 4099) % n; out[i] = in[j] + 1.0f; }
}

int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  printf("CUs %d\n", p.multiProcessorCount);
  unsigned* bar; int* err; float* v; const int NMAX = 1 << 20;
  hipMalloc(&bar, 4); hipMalloc(&err, 4); hipMalloc(&v, sizeof(float) * 2 * NMAX);
  hipMemset(bar, 0, 4); hipMemset(err, 0, 4); hipMemset(v, 0, sizeof(float) * 2 * NMAX);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int nb = 200;
  unsigned base = 0;
  for (int n : {4096, 65536, 524288}) for (int G : {16, 32, 64, 128, 256}) {
    if (G > p.multiProcessorCount) continue;
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
      hipMemsetAsync(v, 0, sizeof(float) * 2 * n, 0);
      hipEventRecord(e0, 0);
      k_barriers<<<G, 256, 0, 0>>>(bar, base, nb, v, n, err);
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      base += (unsigned)nb * G;
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    std::vector<float> h(n); hipMemcpy(h.data(), v + (size_t)(nb & 1) * n, sizeof(float) * n, hipMemcpyDeviceToHost);
    int herr; hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost);
    bool ok = true; for (int i = 0; i < n; ++i) if (h[i] != (float)nb) { ok = false; break; }
    // the same chain as separate launches
    float bestk = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
      hipEventRecord(e0, 0);
      for (int k = 0; k < nb; ++k) k_step<<<G, 256, 0, 0>>>(v + (size_t)(k & 1) * n, v + (size_t)((k + 1) & 1) * n, n);
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < bestk) bestk = ms;
    }
    printf("n %7d  G %3d : barrier kernel %.2f us/phase   separate launches %.2f us/phase   values %s  err %d\n", n, G, 1e3f * best / nb,
           1e3f * bestk / nb, ok ? "ok" : "WRONG", herr);
    fflush(stdout);
  }
  return 0;
}
