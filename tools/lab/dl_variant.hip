// dl_variant.hip -- A/B harness for ONE dimension-in-lane shape: the product kernel mtg_solve_dl_kernel compiled with whatever
// -D switches the variant under test needs, B trajectories on a full persistent grid, kernel time by HIP events and an FNV hash of
// the first coefficients (variants that must be bit-identical print the same hash).
// build (tools/gpu_r06_variants.sh): hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -disable-machine-licm
//   -mllvm -amdgpu-kernarg-preload-count=14 -mllvm -pragma-unroll-threshold=1000000 -DLT_H=6 -DLT_K=32 -DLT_WS=14 -DLT_LS=4 -DLT_RS=1 [-DLT_NP=1]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../../mav_trajectory_generation_amd/csrc/mtg_dimlane.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); std::exit(1); } } while (0)
#ifndef LT_NP
#define LT_NP 1
#endif
#ifndef LT_D
#define LT_D 3
#endif
#ifndef LT_OCC
#define LT_OCC 1      // waves per SIMD the register allocation of the lab kernel is held to (the product kernels: 1)
#endif
constexpr int H = LT_H, N = 2 * H, K = LT_K, D = LT_D, NP = LT_NP;
using C = MtgCfg<H, 1, K, (1 << H) - 1, 1, (1 << H) - 1, H - 1, 0, LT_WS, ((LT_WS > 0 || LT_RS) ? D : 0), LT_LS, LT_RS>;
constexpr int NF = 2 * H + (K - 1);

// the product body under the lab's own launch bounds (LT_OCC = 2: the register allocation held to 256, two workgroup sets per CU)
template <class CC, int DL_, int NP_>
__global__ __launch_bounds__(NP_ * 2 * kWave, LT_OCC) void lab_dl_kernel(const double* __restrict__ times, const double* __restrict__ dfix,
                                                                       double* __restrict__ coeffs, int* status, int* traj_status, int B,
                                                                       int ntiles, int nwg, int aos, double* ws) {
#ifdef LT_MASK21
  // power experiment: only the dimension-0 lanes of every wave run (the same instruction stream with a third of the lanes active;
  // the outputs are garbage) -- how much of a tile's time is the FP64 lanes' power?
  if ((threadIdx.x & 63) >= 21) return;
#endif
  mtg_solve_dl_body<CC, DL_, NP_, 0, 18, false>(times, dfix, coeffs, status, traj_status, B, ntiles, nwg, ws, aos, nullptr, nullptr);
}

// shader clock while the measured launches run: one wave on its own stream spins (s_sleep) until the host raises a flag in mapped
// memory, reading s_memtime (shader clock) and s_memrealtime (100 MHz) at both ends
__global__ __launch_bounds__(64) void lab_clock_probe(volatile int* stop, long long* out) {
  if (threadIdx.x != 0) return;
  const long long r0 = wall_clock64(), c0 = clock64();
  while (*stop == 0 && wall_clock64() - r0 < 2000000000LL) __builtin_amdgcn_s_sleep(32);     // (at most 20 s, whatever the host does)
  const long long c1 = clock64(), r1 = wall_clock64();
  out[0] = c1 - c0;
  out[1] = r1 - r0;
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 100000;
  const char* tag = argc > 2 ? argv[2] : "variant";
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
      hf[((size_t)d * NF + 0) * B + b] = pos[d];
      for (int v = 1; v < K; ++v) hf[((size_t)d * NF + H - 1 + v) * B + b] = pos[v * D + d];
      hf[((size_t)d * NF + H + K - 1) * B + b] = pos[K * D + d];
    }
  }
  const size_t ncoef = (size_t)B * K * D * N;
  double *dt, *df, *dc, *ws;
  CK(hipMalloc(&dt, ht.size() * 8)); CK(hipMalloc(&df, hf.size() * 8)); CK(hipMalloc(&dc, ncoef * 8));
  CK(hipMemcpy(dt, ht.data(), ht.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(df, hf.data(), hf.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemset(dc, 0xff, ncoef * 8));
  constexpr int TPW = 64 / D;
  const int ntiles = (B + TPW - 1) / TPW;
  const int nunits = (ntiles + NP - 1) / NP;
  // argv[3] (optional): number of persistent workgroups (default: one wave per SIMD of the whole chip) -- fewer workgroups = fewer busy
  // SIMDs per CU: does a wave get faster when its CU's other SIMDs are idle (a shared resource) or not (its own latency chain)?
  const int nwg = argc > 3 ? std::min(atoi(argv[3]), nunits) : std::min((NP == 1 ? 512 : 256) * LT_OCC, nunits);
  CK(hipMalloc(&ws, (size_t)nwg * (NP * 128) * std::max(1, C::WSJ * C::WSE) * 8));
  int* dstat; CK(hipMalloc(&dstat, 4)); CK(hipMemset(dstat, 0, 4));
  auto kern = lab_dl_kernel<C, D, NP>;
  const size_t lds = mtg_dl_lds_bytes<C, D, NP>();
  CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipFuncAttributes fa; CK(hipFuncGetAttributes(&fa, (const void*)kern));
  auto go = [&]() {
    hipLaunchKernelGGL(kern, dim3(nwg), dim3(NP * 128), lds, st, (const double*)dt, (const double*)df, dc, dstat, (int*)nullptr, B, ntiles,
                       nwg, 0, ws);
  };
  for (int i = 0; i < 3; ++i) go();
  CK(hipStreamSynchronize(st));
  hipStream_t pst; CK(hipStreamCreateWithFlags(&pst, hipStreamNonBlocking));
  int* stop; long long* pout;
  CK(hipHostMalloc((void**)&stop, sizeof(int), hipHostMallocMapped)); CK(hipHostMalloc((void**)&pout, 2 * sizeof(long long), hipHostMallocMapped));
  *stop = 0; pout[0] = pout[1] = 0;
  // (LT_NOPROBE: no clock probe.  A body that needs more than 504 registers cannot share its SIMD with the probe's wave: one workgroup
  // of a grid of exactly one wave per SIMD then starts after all the others have finished -- 1.4x the time, an artefact of the harness)
#ifndef LT_NOPROBE
  hipLaunchKernelGGL(lab_clock_probe, dim3(1), dim3(64), 0, pst, (volatile int*)stop, pout);
#endif
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  double best = 1e30, sum = 0;
  const int reps = 5, per = 6;
  for (int r = 0; r < reps; ++r) {
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < per; ++i) go();
    CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    best = std::min(best, (double)ms * 1e3 / per); sum += (double)ms * 1e3 / per;
  }
  *stop = 1;
  CK(hipStreamSynchronize(pst));
  const double mhz = pout[1] > 0 ? (double)pout[0] / ((double)pout[1] * 0.01) : 0.0;
  int hs = 0; CK(hipMemcpy(&hs, dstat, 4, hipMemcpyDeviceToHost));
  const size_t nh = std::min<size_t>(ncoef, (size_t)4096 * K * D * N);
  std::vector<double> out(nh);
  CK(hipMemcpy(out.data(), dc, nh * 8, hipMemcpyDeviceToHost));
  uint64_t hsh = 1469598103934665603ull;
  double amax = 0; bool finite = true;
  for (double x : out) {
    uint64_t u; std::memcpy(&u, &x, 8);
    hsh = (hsh ^ u) * 1099511628211ull;
    amax = std::max(amax, std::fabs(x)); finite = finite && std::isfinite(x);
  }
  const double bytes = 8.0 * (K + D * NF + K * D * N) * B;
  std::printf("{\"tag\": \"%s\", \"N\": %d, \"K\": %d, \"D\": %d, \"B\": %d, \"NP\": %d, \"occ\": %d, \"wg\": %d, \"lds\": %zu, \"vgprs\": %d, \"scratch\": %zu, "
              "\"shader_mhz\": %.0f, \"us_mean\": %.2f, \"us_best\": %.2f, \"frac_8TBps\": %.4f, \"status\": %d, \"finite\": %s, \"max_abs\": %.6g, \"hash\": \"%016llx\"}\n",
              tag, N, K, D, B, NP, LT_OCC, nwg, lds, fa.numRegs, (size_t)fa.localSizeBytes, mhz, sum / reps, best, bytes / (sum / reps * 1e-6) / 8e12, hs,
              finite ? "true" : "false", amax, (unsigned long long)hsh);
  return 0;
}
