// dpp_row_bcast.hip -- the cross-lane primitive a ROW-cooperative (16 lanes per trajectory-half) formulation would be built
// on: v_fmac_f64 with a DPP row_newbcast source (gfx90a+: the only DPP control FP64 VALU ops accept).
//   D[lane] += S0[row's lane n] * S1[lane]   in ONE instruction: a rank-1 update of a block whose rows / columns live in the
// lanes of a 16-lane row costs no extra instruction for the broadcast.
// Measures (a) that it computes what the ISA text says, (b) its issue cost against a plain v_fma_f64, for 1 / 2 / 4 / 8
// waves per SIMD (whether co-resident waves add up to the 4-cycle FP64 rate, which the one-wave-per-SIMD solve kernels
// do not reach: profiles/r03f_pmc_stalls.txt).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(64) void k(double* out, long long* ticks, int iters) {
  double a[8], y = 1.0 + threadIdx.x * 1e-3, x = 1.0000001 + (threadIdx.x & 15) * 1e-9;
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = i * 0.125;
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (MODE == 0) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
        else if (MODE == 1) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(x), "v"(y));
        else asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y));
      }
    }
  }
  const long long t1 = clock64();
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
  double s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += a[i];
  if (s == 123.456) out[0] = s;
}

// NOP: wait states between the VALU write of the DPP source and the DPP read (the hazard recogniser does not see inside
// inline asm: "VALU writes VGPR -> DPP reads that VGPR" needs 2 wait states on gfx9)
template <int NOP>
__global__ void check(double* out) {
  double x = 100.0 + threadIdx.x, y = 2.0, acc = 0.5;
  if (NOP) asm volatile("s_nop 1\n v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(y));
  else asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(y));
  double m = 7.0;
  if (NOP) asm volatile("s_nop 1\n v_mov_b64_dpp %0, %1 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(m) : "v"(x));
  else asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(m) : "v"(x));
  // bank-masked form: lanes 0-7 of each row take lane 2, lanes 8-15 take lane 11 (two 8-lane groups per row)
  double acc2 = 0.25;
  asm volatile("s_nop 1\n v_fmac_f64_dpp %0, %1, %2 row_newbcast:2 row_mask:0xf bank_mask:0x3\n"
               "v_fmac_f64_dpp %0, %1, %2 row_newbcast:11 row_mask:0xf bank_mask:0xc" : "+v"(acc2) : "v"(x), "v"(y));
  out[threadIdx.x] = acc;
  out[64 + threadIdx.x] = m;
  out[128 + threadIdx.x] = acc2;
}

int main() {
  double* out; long long* ticks;
  hipMalloc(&out, 192 * 8); hipMalloc(&ticks, 8192 * 8);
  int bad = 0;
  for (int nop = 0; nop < 2; ++nop) {
    if (nop) check<1><<<1, 64>>>(out); else check<0><<<1, 64>>>(out);
    double h[192];
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
    int b0 = 0, b1 = 0, b2 = 0;
    for (int l = 0; l < 64; ++l) {
      const double want = 0.5 + (100.0 + (l & ~15) + 3) * 2.0, wm = 100.0 + (l & ~15) + 5;
      const double w2 = 0.25 + (100.0 + (l & ~15) + ((l & 8) ? 11 : 2)) * 2.0;
      b0 += h[l] != want; b1 += h[64 + l] != wm; b2 += h[128 + l] != w2;
    }
    std::printf("%s wait states: v_fmac_f64_dpp row_newbcast:3 %s (%d lanes off), v_mov_b64_dpp row_newbcast:5 %s (%d), bank-masked 8-lane pair %s (%d); lanes 0 / 20 / 45: fmac %.1f %.1f %.1f mov %.1f %.1f %.1f pair %.2f %.2f %.2f\n",
                nop ? "with 2" : "without", b0 ? "MISMATCH" : "ok", b0, b1 ? "MISMATCH" : "ok", b1, b2 ? "MISMATCH" : "ok", b2,
                h[0], h[20], h[45], h[64], h[84], h[109], h[128], h[148], h[173]);
    if (nop) bad = b0 + b1 + b2;
  }
  const int iters = 2000;
  const char* names[3] = {"v_fma_f64 (VOP3), 8 independent accumulators", "v_fmac_f64_dpp row_newbcast, 8 independent accumulators",
                          "v_fmac_f64 (VOP2), 8 independent accumulators"};
  for (int occ = 1; occ <= 8; occ *= 2) {
    for (int mode = 0; mode < 3; ++mode) {
      const int grid = 1024 * occ;
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      for (int rep = 0; rep < 2; ++rep) {
        if (rep == 1) hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(64), 0, 0, out, ticks, iters);
        if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(64), 0, 0, out, ticks, iters);
        if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(64), 0, 0, out, ticks, iters);
      }
      hipEventRecord(e1); hipDeviceSynchronize();
      float ms = 0; hipEventElapsedTime(&ms, e0, e1);
      std::vector<long long> hh(grid);
      hipMemcpy(hh.data(), ticks, grid * 8, hipMemcpyDeviceToHost);
      double s = 0;
      for (int i = 0; i < grid; ++i) s += (double)hh[i];
      const double n_inst = (double)iters * 64;
      // chip-wide: instructions per SIMD / (kernel time x clock) -> cycles of SIMD time per instruction
      std::printf("%d wave(s)/SIMD  %-58s: %6.2f clock64 ticks per instruction per wave; kernel %.1f us -> %.2f SIMD-cycles per instruction at 2.4 GHz\n",
                  occ, names[mode], s / grid / n_inst, ms * 1e3, ms * 1e-3 * 2.4e9 / (n_inst * occ));
    }
  }
  return bad ? 1 : 0;
}
