// dpp_quad.hip -- cost of the cross-lane traffic a lane-cooperative (4 lanes per trajectory-half) formulation needs:
// broadcasting one double from lane q of every quad to the quad's four lanes.
//   dpp  : two v_mov_b32 with quad_perm:[q,q,q,q] (DPP row/quad permutes run at VALU rate)
//   bperm: ds_bpermute_b32 x2 (goes through the LDS crossbar)
//   fma  : a dependent FP64 FMA per step, for reference
// One wave per SIMD-ish (1024 workgroups x 64), dependent chains, clock64 ticks per operation.
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(64) void k(double* out, long long* ticks, int iters) {
  double x = threadIdx.x * 1e-3 + 1.0;
  const double c = 1.0000001;
  const long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (MODE == 0) {
        x = __builtin_fma(x, c, 1e-9);
      } else if (MODE == 1) {
        int lo = __double2loint(x), hi = __double2hiint(x);
        lo = __builtin_amdgcn_update_dpp(lo, lo, 0x55 /*quad_perm [1,1,1,1]*/, 0xf, 0xf, false);
        hi = __builtin_amdgcn_update_dpp(hi, hi, 0x55, 0xf, 0xf, false);
        x = __builtin_fma(__hiloint2double(hi, lo), c, 1e-9);
      } else {
        int lo = __double2loint(x), hi = __double2hiint(x);
        const int src = ((threadIdx.x & ~3) | 1) << 2;
        lo = __builtin_amdgcn_ds_bpermute(src, lo);
        hi = __builtin_amdgcn_ds_bpermute(src, hi);
        x = __builtin_fma(__hiloint2double(hi, lo), c, 1e-9);
      }
    }
  }
  const long long t1 = clock64();
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
  if (x == 123.456) out[0] = x;
}

int main() {
  double* out; long long* ticks;
  hipMalloc(&out, 8); hipMalloc(&ticks, 1024 * 8);
  const int iters = 2000;
  const char* names[3] = {"dependent fma", "quad broadcast (2 x v_mov dpp) + fma", "quad broadcast (2 x ds_bpermute) + fma"};
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1024), dim3(64), 0, 0, out, ticks, iters);
      if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(1024), dim3(64), 0, 0, out, ticks, iters);
      if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(1024), dim3(64), 0, 0, out, ticks, iters);
    }
    hipDeviceSynchronize();
    long long h[1024];
    hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < 1024; ++i) s += (double)h[i];
    std::printf("%-42s: %6.2f clock64 ticks per step (step = the broadcast, if any, + one dependent FMA)\n", names[mode],
                s / 1024 / iters / 16);
  }
  return 0;
}
