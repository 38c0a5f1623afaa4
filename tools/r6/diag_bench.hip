// Micro-benchmark of k_splu_diag (the kernel text is included from csrc/splu.hip): nf fronts of one 64 x 64 pivot block each.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I../../helmholtz-x_amd/csrc diag_bench.hip ../../helmholtz-x_amd/csrc/splu_analyse.hip ../../helmholtz-x_amd/csrc/pool.hip ../../helmholtz-x_amd/csrc/options.hip -o diag_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <complex>
thread_local char hx_errbuf[512];
#include "../../helmholtz-x_amd/csrc/splu.hip"

int main(int argc, char** argv) {
  const int nf = argc > 1 ? atoi(argv[1]) : 1, w = argc > 2 ? atoi(argv[2]) : 64, reps = 200;
  const int ns = w, m = w;
  std::vector<int32_t> sfirst(nf + 1), fronts(nf);
  std::vector<int64_t> sptr(nf + 1, 0), loff(nf), uoff(nf), ioff(nf);
  for (int f = 0; f <= nf; ++f) sfirst[f] = f * ns;
  for (int f = 0; f < nf; ++f) { fronts[f] = f; loff[f] = (int64_t)f * 2 * m * ns; uoff[f] = loff[f] + m * ns; ioff[f] = (int64_t)f * NB * NB; }
  std::vector<std::complex<double>> F((size_t)nf * 2 * m * ns);
  srand(1);
  for (auto& v : F) v = {rand() / (double)RAND_MAX - 0.5, rand() / (double)RAND_MAX - 0.5};
  for (int f = 0; f < nf; ++f) for (int i = 0; i < ns; ++i) F[loff[f] + i + (size_t)i * m] += 4.0;
  SpluDev d{};
  int32_t *dsf, *dfr, *dst; int64_t *dsp, *dlo, *duo, *dio; zc *dF, *dW;
  hipMalloc(&dsf, sfirst.size() * 4); hipMalloc(&dfr, nf * 4); hipMalloc(&dst, 16); hipMalloc(&dsp, sptr.size() * 8);
  hipMalloc(&dlo, nf * 8); hipMalloc(&duo, nf * 8); hipMalloc(&dio, nf * 8); hipMalloc(&dF, F.size() * 16); hipMalloc(&dW, (size_t)nf * NB * NB * 16);
  hipMemcpy(dsf, sfirst.data(), sfirst.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dfr, fronts.data(), nf * 4, hipMemcpyHostToDevice);
  hipMemcpy(dsp, sptr.data(), sptr.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dlo, loff.data(), nf * 8, hipMemcpyHostToDevice);
  hipMemcpy(duo, uoff.data(), nf * 8, hipMemcpyHostToDevice); hipMemcpy(dio, ioff.data(), nf * 8, hipMemcpyHostToDevice);
  hipMemcpy(dF, F.data(), F.size() * 16, hipMemcpyHostToDevice); hipMemset(dst, 0, 16);
  d.sfirst = dsf; d.sptr = dsp; d.loff = dlo; d.uoff = duo; d.ioff = dio; d.F = dF; d.W = dW; d.status = dst;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 10; ++i) k_splu_diag<<<nf, 512>>>(d, dfr, 0);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) k_splu_diag<<<nf, 512>>>(d, dfr, 0);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  // check: D * Dinv = I for front 0
  std::vector<std::complex<double>> W((size_t)NB * NB);
  hipMemcpy(W.data(), dW, W.size() * 16, hipMemcpyDeviceToHost);
  double err = 0;
  for (int i = 0; i < w; ++i) for (int j = 0; j < w; ++j) {
    std::complex<double> acc = 0;
    for (int k = 0; k < w; ++k) {
      std::complex<double> dik = (i >= k) ? F[loff[0] + i + (size_t)k * m] : F[uoff[0] + i + (size_t)k * ns];
      acc += dik * W[k + (size_t)j * NB];
    }
    err = fmax(err, std::abs(acc - (i == j ? 1.0 : 0.0)));
  }
  printf("nf %d w %d: %.2f us per launch, |D Dinv - I|max %.2e\n", nf, w, ms * 1e3 / reps, err);
  return 0;
}
