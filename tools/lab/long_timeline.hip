// long_timeline.hip -- where a lone wave's time goes in a long-chain dimension-in-lane kernel: shader-clock stamps per tile
// and phase (MTG_LAB_TIMELINE build of mtg_solve_dl_kernel; the product compiles the stamps out), B trajectories of
// LT_N coefficients / LT_K segments on a full grid, steady-state tiles only.
// build (tools/gpu_timeline.sh): hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -disable-machine-licm
//   -mllvm -amdgpu-kernarg-preload-count=14 -mllvm -pragma-unroll-threshold=1000000 -DMTG_LAB_TIMING -DMTG_LAB_TIMELINE=12
//   -DLT_H=6 -DLT_K=32 -DLT_WS=14 -DLT_LS=4 -DLT_RS=1
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../mav_trajectory_generation_amd/csrc/mtg_dimlane.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); std::exit(1); } } while (0)

constexpr int H = LT_H, N = 2 * H, K = LT_K, D = 3, NIT = MTG_LAB_TIMELINE;
using C = MtgCfg<H, 1, K, (1 << H) - 1, 1, (1 << H) - 1, H - 1, 0, LT_WS, 3, LT_LS, LT_RS>;
constexpr int NF = 2 * H + (K - 1);

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 100000;
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  std::vector<double> ht((size_t)K * B), hf((size_t)D * NF * B, 0.0);
  std::mt19937_64 rng(1234);
  std::uniform_real_distribution<double> U(-10.0, 10.0);
  for (int b = 0; b < B; ++b) {
    std::vector<double> pos((size_t)(K + 1) * D);
    for (auto& x : pos) x = U(rng);
    for (int k = 0; k < K; ++k) {
      double dist = 0;
      for (int d = 0; d < D; ++d) dist += (pos[(k + 1) * D + d] - pos[k * D + d]) * (pos[(k + 1) * D + d] - pos[k * D + d]);
      dist = std::sqrt(dist);
      ht[(size_t)k * B + b] = dist / 3.0 * 2 * (1.0 + 6.5 * 3.0 / 5.0 * std::exp(-dist / 3.0 * 2));
    }
    for (int d = 0; d < D; ++d) {
      hf[((size_t)d * NF + 0) * B + b] = pos[d];                                        // vertex 0: derivatives 0..H-1
      for (int v = 1; v < K; ++v) hf[((size_t)d * NF + H - 1 + v) * B + b] = pos[v * D + d];
      hf[((size_t)d * NF + H + K - 1) * B + b] = pos[K * D + d];                        // vertex K
    }
  }
  const size_t ncoef = (size_t)B * K * D * N;
  double *dt, *df, *dc, *ws;
  CK(hipMalloc(&dt, ht.size() * 8)); CK(hipMalloc(&df, hf.size() * 8)); CK(hipMalloc(&dc, ncoef * 8));
  CK(hipMemcpy(dt, ht.data(), ht.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(df, hf.data(), hf.size() * 8, hipMemcpyHostToDevice));
  const int ntiles = (B + 20) / 21, nwg = std::min(512, ntiles);   // NP = 1: one direction pair per workgroup, two workgroups per CU
  CK(hipMalloc(&ws, (size_t)nwg * 128 * std::max(1, C::WSJ * C::WSE) * 8));
  int* dstat; CK(hipMalloc(&dstat, 4)); CK(hipMemset(dstat, 0, 4));
  const size_t ndbg = (size_t)16 * 8192 + (size_t)nwg * 2 * NIT * 8;
  long long* dbg; CK(hipMalloc(&dbg, ndbg * 8)); CK(hipMemset(dbg, 0, ndbg * 8));
  auto kern = mtg_solve_dl_kernel<C, 3, 1, 0, 18>;
  const size_t lds = mtg_dl_lds_bytes<C, 3, 1>();
  CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  auto go = [&]() {
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(128), lds, st, (const double*)dt, (const double*)df, dc, dstat, (int*)nullptr, B, ntiles,
                       nwg, 0, ws, dbg);
  };
  for (int i = 0; i < 3; ++i) go();
  CK(hipStreamSynchronize(st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0, st));
  for (int i = 0; i < 10; ++i) go();
  CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  int hs = 0; CK(hipMemcpy(&hs, dstat, 4, hipMemcpyDeviceToHost));
  std::vector<double> out(std::min<size_t>(ncoef, 4096));
  CK(hipMemcpy(out.data(), dc, out.size() * 8, hipMemcpyDeviceToHost));
  double amax = 0; for (double x : out) amax = std::max(amax, std::fabs(x));
  std::printf("N %d K %d B %d: %d tiles on %d workgroups (%.2f tiles per wave pair), LDS %zu B, launch %.1f us (stamped build), status %d, max |c| %.3g\n",
              N, K, B, ntiles, nwg, (double)ntiles / nwg, lds, ms * 1e3 / 10, hs, amax);
  std::vector<long long> h(ndbg);
  CK(hipMemcpy(h.data(), dbg, ndbg * 8, hipMemcpyDeviceToHost));
  // phases of the steady-state tiles (tile iterations 1 .. : the first tile's inputs are loaded before the loop)
  const char* names[] = {"tile top -> step 0 done   ", "step 0 done -> step 1 done", "steps 2 .. -> forward done", "barrier (other direction) ",
                         "middle vertex             ", "backward + drains         ", "fetch next + end barrier  ", "whole tile                "};
  for (int dir = 0; dir < 2; ++dir) {
    std::vector<double> ph[8];
    for (int wv = dir; wv < nwg * 2; wv += 2) {
      for (int itn = 1; itn < NIT; ++itn) {
        const long long* r = &h[(size_t)16 * 8192 + ((size_t)wv * NIT + itn) * 8];
        if (r[0] == 0 || r[5] == 0) continue;
        const double v[8] = {(double)(r[1] - r[0]), (double)(r[7] - r[1]), (double)(r[2] - r[7]), (double)(r[3] - r[2]),
                             (double)(r[6] - r[3]), (double)(r[4] - r[6]), (double)(r[5] - r[4]), (double)(r[5] - r[0])};
        for (int k = 0; k < 8; ++k) ph[k].push_back(v[k]);
      }
    }
    std::printf("direction %c: %zu stamped tiles; shader cycles (share of the whole tile at the median)\n", dir ? 'B' : 'A', ph[7].size());
    double whole = 1;
    { auto v = ph[7]; std::sort(v.begin(), v.end()); whole = v[v.size() / 2]; }
    for (int k = 0; k < 8; ++k) {
      auto v = ph[k];
      if (v.empty()) continue;
      std::sort(v.begin(), v.end());
      std::printf("  %s p10 %8.0f  median %8.0f  p90 %8.0f   (%.3f)\n", names[k], v[v.size() / 10], v[v.size() / 2], v[v.size() * 9 / 10], v[v.size() / 2] / whole);
    }
  }
  {   // shader clock during the kernel: s_memtime ticks against the 100 MHz s_memrealtime, first tile top -> kernel end of wave 0 .. 
    std::vector<double> mhz;
    for (int wv = 0; wv < nwg * 2; ++wv) {
      const long long* r = &h[(size_t)wv * 16];
      if (r[15] > r[14]) mhz.push_back((double)(r[1] - r[0]) / ((double)(r[15] - r[14]) * 0.01));
    }
    if (!mhz.empty()) { std::sort(mhz.begin(), mhz.end()); std::printf("shader clock over a wave's life (s_memtime / s_memrealtime): median %.0f MHz (p10 %.0f, p90 %.0f)\n", mhz[mhz.size() / 2], mhz[mhz.size() / 10], mhz[mhz.size() * 9 / 10]); }
  }
  return 0;
}
