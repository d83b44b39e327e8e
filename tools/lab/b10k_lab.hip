// b10k_lab.hip -- measurement harness for the small-batch launch forms of the solve kernel (BASELINE config 2:
// B = 10k, K = 8, N = 10, D = 3, snap).  Runs the production kernel and the experimental forms on the same synthetic
// SoA batch, checks the coefficients against each other, and times back-to-back launches on one stream with hipEvents
// (the protocol of mtg_time_last_solve / bench.py).  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17
//   -mllvm -disable-machine-licm -mllvm -amdgpu-kernarg-preload-count=14 [-DMTG_LAB_TIMING]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../../mav_trajectory_generation_amd/csrc/mtg_small.h"
#include "../../mav_trajectory_generation_amd/csrc/mtg_dimlane.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); std::exit(1); } } while (0)

// pure write of the coefficient buffer: every workgroup writes its tile's 64 x 1920-byte contiguous region in whole
// 16-byte-per-lane coalesced stores (AUX: 0 plain, 1 sc1 write-through, 2 nt)
template <int AUX>
__global__ __launch_bounds__(128) void k_fill(double* out, int B) {
  typedef unsigned int u4 __attribute__((ext_vector_type(4)));
  const u4 v = {1u, 2u, 3u, 4u};
  const long long n16 = (long long)B * 120;   // 16-byte chunks
  for (long long i = (long long)blockIdx.x * 7680 / 16 * 16 + threadIdx.x; i < std::min(n16, ((long long)blockIdx.x + 1) * 7680); i += 128) {
    u4* p = reinterpret_cast<u4*>(out) + i;
    if (AUX == 0) *p = v;
    else if (AUX == 1) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
  }
}
// pure read of the inputs (13 coalesced loads per lane as the kernel issues them)
__global__ __launch_bounds__(128) void k_read(const double* t, const double* f, double* sink, int B) {
  const int b = std::min(B - 1, (int)(blockIdx.x * 64 + (threadIdx.x & 63)));
  const int dir = threadIdx.x >> 6;
  double a = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) a += t[(size_t)(dir * 4 + k) * B + b];
#pragma unroll
  for (int c = 0; c < 9; ++c) a += f[((size_t)blockIdx.y * 17 + dir * 8 + c) * B + b];
  if (a == 1.2345e300) sink[0] = a;
}

using C1 = MtgCfg<5, 1, 8, 31, 1, 31, 4>;
using C3 = MtgCfg<5, 3, 8, 31, 1, 31, 4>;

template <class F>
static double time_us(hipStream_t st, F launch, int reps = 400) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 30; ++i) launch(i);
  CK(hipStreamSynchronize(st));
  CK(hipEventRecord(e0, st));
  for (int i = 0; i < reps; ++i) launch(i);
  CK(hipEventRecord(e1, st));
  CK(hipEventSynchronize(e1));
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipGetLastError());
  return ms * 1e3 / reps;
}

static double max_rel_poly_err(const std::vector<double>& a, const std::vector<double>& b, int n) {
  double worst = 0;
  for (size_t p = 0; p + n <= a.size(); p += n) {
    double num = 0, den = 0;
    for (int j = 0; j < n; ++j) { num = std::max(num, std::fabs(a[p + j] - b[p + j])); den = std::max(den, std::fabs(b[p + j])); }
    if (den == 0) den = 1;
    worst = std::max(worst, num / den);
  }
  return worst;
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 10000;
  const int NSETS = argc > 2 ? atoi(argv[2]) : 1;   // rotating input/output buffer sets (reads from HBM instead of cache)
  constexpr int K = 8, D = 3, N = 10, NF = 17;
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  // synthetic random-waypoint batch, canonical SoA
  std::vector<double> ht((size_t)K * B), hf((size_t)D * NF * B, 0.0);
  std::mt19937_64 rng(1234);
  std::uniform_real_distribution<double> U(-10.0, 10.0);
  for (int b = 0; b < B; ++b) {
    double pos[K + 1][D];
    for (int v = 0; v <= K; ++v) for (int d = 0; d < D; ++d) pos[v][d] = U(rng);
    for (int k = 0; k < K; ++k) {
      double dist = 0;
      for (int d = 0; d < D; ++d) dist += (pos[k + 1][d] - pos[k][d]) * (pos[k + 1][d] - pos[k][d]);
      dist = std::sqrt(dist);
      ht[(size_t)k * B + b] = dist / 3.0 * 2 * (1.0 + 6.5 * 3.0 / 5.0 * std::exp(-dist / 3.0 * 2));
    }
    for (int d = 0; d < D; ++d) {
      hf[((size_t)d * NF + 0) * B + b] = pos[0][d];                       // vertex 0: derivatives 0..4 (1..4 zero)
      for (int v = 1; v < K; ++v) hf[((size_t)d * NF + 4 + v) * B + b] = pos[v][d];
      hf[((size_t)d * NF + 12) * B + b] = pos[K][d];                     // vertex K: derivatives 0..4
    }
  }
  const size_t ncoef = (size_t)B * K * D * N;
  std::vector<double*> dt(NSETS), df(NSETS), dc(NSETS);
  for (int s = 0; s < NSETS; ++s) {
    CK(hipMalloc(&dt[s], ht.size() * 8)); CK(hipMalloc(&df[s], hf.size() * 8)); CK(hipMalloc(&dc[s], ncoef * 8));
    CK(hipMemcpy(dt[s], ht.data(), ht.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(df[s], hf.data(), hf.size() * 8, hipMemcpyHostToDevice));
  }
  int* dstat; CK(hipMalloc(&dstat, 4)); CK(hipMemset(dstat, 0, 4));
  long long* dbg; CK(hipMalloc(&dbg, sizeof(long long) * 16 * 8192)); CK(hipMemset(dbg, 0, sizeof(long long) * 16 * 8192));
  const int ntiles = (B + 63) / 64;

  auto params = [&](int s, int Dg) {
    MtgParams P; std::memset(&P, 0, sizeof(P));
    P.times = dt[s]; P.ts_b = 1; P.ts_k = B;
    P.dfix = df[s]; P.fs_b = 1; P.fs_c = B; P.fs_d = (long long)NF * B;
    P.coeffs = dc[s]; P.status = dstat; P.B = B; P.K = K; P.Dtot = D; P.deriv = 4;
    P.h1off = C1::H1OFF; P.ainvoff = C1::AINVOFF;
    (void)Dg;
    return P;
  };
  auto lds_prod = [&](int dc_) {
    const size_t stage = (size_t)64 * ((size_t)(dc_ * N / 2) | 1) * 2 * 8;
    return 2 * stage + (size_t)2 * (10 + dc_ * 4) * 64 * 8;
  };
  std::vector<double> ref(ncoef), out(ncoef);
  auto check = [&](const char* name) {
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(out.data(), dc[0], ncoef * 8, hipMemcpyDeviceToHost));
    int hs = 0; CK(hipMemcpy(&hs, dstat, 4, hipMemcpyDeviceToHost));
    std::printf("  %-28s max poly rel diff vs production split kernel %.3e  status %d\n", name, max_rel_poly_err(out, ref, N), hs);
  };
  // production kernels
  auto prod_split = [&](int i) {
    hipLaunchKernelGGL((mtg_solve_kernel<C1, 4>), dim3(std::min(ntiles, 256 * 8 / 3), 3), dim3(kBlock), lds_prod(1), st, params(i % NSETS, 1), ntiles);
  };
  auto prod_fused = [&](int i) {
    hipLaunchKernelGGL((mtg_solve_kernel<C3, 4>), dim3(std::min(ntiles, 256 * 8)), dim3(kBlock), lds_prod(3), st, params(i % NSETS, 3), ntiles);
  };
  prod_split(0);
  CK(hipStreamSynchronize(st));
  CK(hipMemcpy(ref.data(), dc[0], ncoef * 8, hipMemcpyDeviceToHost));
  CK(hipMemset(dc[0], 0, ncoef * 8));
  if (const char* rp = getenv("LAB_REF")) {   // cross-build comparison: first run writes the reference, later runs read it
    if (FILE* f = std::fopen(rp, "rb")) {
      std::vector<double> r2(ncoef);
      if (std::fread(r2.data(), 8, ncoef, f) == ncoef) {
        std::printf("production split kernel of THIS build vs reference file: max poly rel diff %.3e\n", max_rel_poly_err(ref, r2, N));
        ref = r2;
      }
      std::fclose(f);
    } else if (FILE* g = std::fopen(rp, "wb")) {
      std::fwrite(ref.data(), 8, ncoef, g);
      std::fclose(g);
    }
  }
  std::printf("B = %d, %d buffer set(s)\n", B, NSETS);
  std::printf("production split (471 x 128)      : %7.2f us\n", time_us(st, prod_split));
  std::printf("production fused (157 x 128)      : %7.2f us\n", time_us(st, prod_fused));
  prod_fused(0); check("production fused");
  {
    auto run_slab = [&](auto kern, const char* name) {
      CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mtg_slab_lds_bytes<C3>()));
      auto go = [=, &dt, &df, &dc](int i) {
        MtgParams P; std::memset(&P, 0, sizeof(P));
        const int s2 = i % NSETS;
        P.times = dt[s2]; P.ts_b = 1; P.ts_k = B; P.dfix = df[s2]; P.fs_b = 1; P.fs_c = B; P.fs_d = (long long)NF * B;
        P.coeffs = dc[s2]; P.status = dstat; P.B = B; P.K = K; P.Dtot = D; P.deriv = 4; P.h1off = C1::H1OFF; P.ainvoff = C1::AINVOFF;
        hipLaunchKernelGGL(kern, dim3(std::min(ntiles, getenv("LAB_SLAB_GRID") ? atoi(getenv("LAB_SLAB_GRID")) : 512)), dim3(kBlock), mtg_slab_lds_bytes<C3>(), st, P, ntiles);
      };
      CK(hipMemsetAsync(dc[0], 0, ncoef * 8, st));
      go(0); check(name);
      std::printf("%-34s: %7.2f %7.2f us\n", name, time_us(st, go), time_us(st, go));
    };
    run_slab((mtg_solve_slab_kernel<C3, 0>), "fused, slab output, write-back");
    run_slab((mtg_solve_slab_kernel<C3, 18>), "fused, slab output, nt sc1");
    run_slab((mtg_solve_slab_kernel<C3, 2>), "fused, slab output, nt");
    std::printf("production fused (again)          : %7.2f us\n", time_us(st, prod_fused));
  }

  {
    double* dcost; CK(hipMalloc(&dcost, (size_t)B * 8)); CK(hipMemset(dcost, 0, (size_t)B * 8));
    auto cost_only = [&](int i) {
      MtgParams P = params(i % NSETS, 1); P.cost = dcost;
      hipLaunchKernelGGL((mtg_solve_kernel<C1, 9>), dim3(std::min(ntiles, 256 * 8 / 3), 3), dim3(kBlock), lds_prod(1), st, P, ntiles);
    };
    std::printf("production split, cost only       : %7.2f us\n", time_us(st, cost_only));
    auto f0 = [&](int i) { hipLaunchKernelGGL(k_fill<0>, dim3((B * 120 + 7679) / 7680), dim3(128), 0, st, dc[i % NSETS], B); };
    auto f1 = [&](int i) { hipLaunchKernelGGL(k_fill<1>, dim3((B * 120 + 7679) / 7680), dim3(128), 0, st, dc[i % NSETS], B); };
    auto f2 = [&](int i) { hipLaunchKernelGGL(k_fill<2>, dim3((B * 120 + 7679) / 7680), dim3(128), 0, st, dc[i % NSETS], B); };
    std::printf("write 19.2 MB only: plain %.2f us, sc1 %.2f us, nt %.2f us\n", time_us(st, f0), time_us(st, f1), time_us(st, f2));
    auto r0 = [&](int i) { hipLaunchKernelGGL(k_read, dim3(ntiles, 3), dim3(128), 0, st, (const double*)dt[i % NSETS], (const double*)df[i % NSETS], dcost, B); };
    std::printf("read inputs only (471 x 128)      : %7.2f us\n", time_us(st, r0));
    prod_split(0);   // restore the coefficient buffer for the checks below
  }
  // small-launch form, one dimension per unit
  CK(hipFuncSetAttribute((const void*)mtg_solve_small_kernel<C1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)mtg_small_lds_bytes<C1>()));
  auto small1 = [&](int i) {
    const int s = i % NSETS;
    hipLaunchKernelGGL((mtg_solve_small_kernel<C1, 4>), dim3((ntiles + 1) / 2, 3), dim3(kSmallBlock), mtg_small_lds_bytes<C1>(), st,
                       (const double*)dt[s], (const double*)df[s], dc[s], dstat, B, ntiles, dbg);
  };
  CK(hipMemset(dc[0], 0, ncoef * 8));
  small1(0); check("small form, D = 1 units");
  std::printf("small form (%d x 256, D=1 units)  : %7.2f us\n", (ntiles + 1) / 2 * 3, time_us(st, small1));
  // dimension-in-lane form
  const int ntiles_dl = (B + 20) / 21, nwg_dl = (ntiles_dl + 1) / 2;
  const size_t lds_dl = mtg_dl_lds_bytes<C1, 3, 2>();
  auto run_dl = [&](auto kern, const char* name) {
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dl));
    const bool rot_in = !getenv("LAB_FIX_IN"), rot_out = !getenv("LAB_FIX_OUT");   // rotate inputs / outputs separately
    auto go = [=, &dt, &df, &dc](int i) {
      const int s = rot_in ? i % NSETS : 0, so = rot_out ? i % NSETS : 0;
      hipLaunchKernelGGL(kern, dim3(nwg_dl), dim3(256), lds_dl, st, (const double*)dt[s], (const double*)df[s], dc[so],
                         dstat, (int*)nullptr, B, ntiles_dl, nwg_dl, 0, (double*)nullptr
#if defined(MTG_LAB_TIMING)
                         , dbg
#endif
      );
    };
    CK(hipMemsetAsync(dc[0], 0, ncoef * 8, st));
    go(0); check(name);
    double a = time_us(st, go), b2 = time_us(st, go), c = time_us(st, go);
    std::printf("%-34s: %7.2f %7.2f %7.2f us\n", name, a, b2, c);
    return go;
  };
  auto dl_sc1 = run_dl((mtg_solve_dl_kernel<C1, 3, 2, 0, 16>), "dim-in-lane, sc1 stores");
#if !defined(MTG_LAB_TIMING)
  run_dl((mtg_solve_dl_kernel<C1, 3, 2, 0, 0>), "dim-in-lane, write-back stores");
  run_dl((mtg_solve_dl_kernel<C1, 3, 2, 0, 2>), "dim-in-lane, nt stores");
  run_dl((mtg_solve_dl_kernel<C1, 3, 2, 0, 1>), "dim-in-lane, sc0 stores");
  run_dl((mtg_solve_dl_kernel<C1, 3, 2, 0, 17>), "dim-in-lane, sc0 sc1 stores");
  run_dl((mtg_solve_dl_kernel<C1, 3, 2, 0, 18>), "dim-in-lane, nt sc1 stores");
  run_dl((mtg_solve_dl_kernel<C1, 3, 2, 0, 19>), "dim-in-lane, sc0 sc1 nt stores");
#endif
#if defined(MTG_LAB_TIMING)
  {
    CK(hipMemsetAsync(dbg, 0, sizeof(long long) * 16 * 8192, st));
    dl_sc1(0);
    CK(hipStreamSynchronize(st));
    const int nw = nwg_dl * 4;
    std::vector<long long> h((size_t)nw * 16);
    CK(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
    long long w0 = h[14];
    for (int w = 0; w < nw; ++w) w0 = std::min(w0, h[(size_t)w * 16 + 14]);
    auto pct = [](std::vector<double> v, const char* name) {
      std::sort(v.begin(), v.end());
      std::printf("  %-26s min %6.2f p25 %6.2f med %6.2f p75 %6.2f max %6.2f\n", name, v.front(), v[v.size() / 4], v[v.size() / 2], v[v.size() * 3 / 4], v.back());
    };
    std::vector<double> s0, s1, ph[6];
    for (int w = 0; w < nw; ++w) {
      const long long* r = &h[(size_t)w * 16];
      s0.push_back((r[14] - w0) * 0.01); s1.push_back((r[15] - w0) * 0.01);
      for (int k = 1; k <= 5; ++k) ph[k].push_back((double)(r[k] - r[k - 1]));
    }
    pct(s0, "wave start [us]"); pct(s1, "wave end [us]");
    {
      std::vector<double> ld, fw;
      for (int w = 0; w < nw; ++w) { const long long* r = &h[(size_t)w * 16]; ld.push_back((double)(r[6] - r[1])); fw.push_back((double)(r[2] - r[6])); }
      pct(ld, "loads issued -> arrived [cyc]"); pct(fw, "forward [cyc]");
    }
    pct(ph[1], "entry -> loads issued [cyc]"); pct(ph[2], "loads + forward [cyc]"); pct(ph[3], "barrier [cyc]");
    pct(ph[4], "mid + backward [cyc]"); pct(ph[5], "last stores acked [cyc]");
  }
#endif
  return 0;
}
