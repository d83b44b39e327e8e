// Microbenchmark: what the NON-FMA instructions of the solve kernels cost a lone wave (one wave per SIMD): 8 independent
// v_fma_f64 chains per iteration (the 4.4-cycle baseline of fp64_issue.hip) with, per 8 FMAs, M extra instructions of one
// kind mixed in: v_accvgpr_write/read pairs (the register allocator's AGPR spills), v_mul_f64, s_mov literal pairs feeding
// an FMA operand, v_mov_b32.  Prints cycles per iteration and the marginal cycles per extra instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

enum { KIND_NONE = 0, KIND_ACC = 1, KIND_MUL = 2, KIND_MOV = 3, KIND_ACC_DEP = 4, KIND_SMOV = 5 };

template <int KIND, int M>
__global__ void k_mix(double* out, long long* cyc, int iters, double a, double b) {
  double x[8], y[8], z[8];
  unsigned t[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x * 1e-3 + i; y[i] = 1.0 + threadIdx.x * 1e-9 * (i + 1); z[i] = a * (threadIdx.x + i + 1) * 1e-12 + b; t[i] = threadIdx.x + i; }
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      x[i] = __builtin_fma(y[i], z[(i + 3) % 8], x[i]);
      if (i < M) {
        if constexpr (KIND == KIND_ACC) {            // independent AGPR round trip of a side value
          unsigned acc;
          asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(acc) : "v"(t[i]));
          asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(t[i]) : "a"(acc));
        } else if constexpr (KIND == KIND_ACC_DEP) {  // AGPR round trip of an FMA operand (low half of y[i])
          unsigned lo = __double2loint(y[i]), hi = __double2hiint(y[i]), acc;
          asm volatile("v_accvgpr_write_b32 %0, %1" : "=a"(acc) : "v"(lo));
          asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(lo) : "a"(acc));
          y[i] = __hiloint2double(hi, lo);
        } else if constexpr (KIND == KIND_MUL) {
          z[i] = z[i] * y[(i + 5) % 8];
          asm volatile("" : "+v"(z[i]));
        } else if constexpr (KIND == KIND_MOV) {
          asm volatile("v_mov_b32 %0, %1" : "=v"(t[i]) : "v"(t[(i + 1) % 8]));
        } else if constexpr (KIND == KIND_SMOV) {     // a 64-bit constant materialised with two s_mov_b32 and used by an FMA
          unsigned clo, chi;
          asm volatile("s_mov_b32 %0, 0x9abcdef0" : "=s"(clo));
          asm volatile("s_mov_b32 %0, 0x3ff12345" : "=s"(chi));
          x[i] = __builtin_fma(y[i], __hiloint2double(chi, clo), x[i]);
        }
      }
    }
  }
  long long t1 = clock64();
  double s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += x[i] + y[i] + z[i] + t[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <class K>
double run(K kern, int iters) {
  const int blocks = 1024, threads = 64;
  double* out; long long* cyc;
  (void)hipMalloc(&out, sizeof(double) * blocks * threads);
  (void)hipMalloc(&cyc, sizeof(long long) * blocks);
  for (int r = 0; r < 2; ++r) {
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters, 1.0000001, 1e-9);
    (void)hipDeviceSynchronize();
  }
  std::vector<long long> h(blocks);
  (void)hipMemcpy(h.data(), cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
  double mean = 0; for (auto v : h) mean += v; mean /= blocks;
  (void)hipFree(out); (void)hipFree(cyc);
  return mean / iters;
}

int main() {
  const int iters = 4000;
  const double base = run(k_mix<KIND_NONE, 0>, iters);
  printf("8 independent v_fma_f64 per iteration: %.2f cycles per iteration (%.2f per FMA)\n", base, base / 8);
  auto report = [&](const char* name, double c, int extra) {
    printf("%-64s %7.2f cycles per iteration, %+6.2f per extra instruction\n", name, c, (c - base) / extra);
  };
  report("+ 4 independent v_accvgpr_write/read pairs (8 instr)", run(k_mix<KIND_ACC, 4>, iters), 8);
  report("+ 8 independent v_accvgpr_write/read pairs (16 instr)", run(k_mix<KIND_ACC, 8>, iters), 16);
  report("+ 4 AGPR round trips of an FMA operand (8 instr)", run(k_mix<KIND_ACC_DEP, 4>, iters), 8);
  report("+ 4 v_mul_f64 (4 instr)", run(k_mix<KIND_MUL, 4>, iters), 4);
  report("+ 8 v_mul_f64 (8 instr)", run(k_mix<KIND_MUL, 8>, iters), 8);
  report("+ 8 v_mov_b32 (8 instr)", run(k_mix<KIND_MOV, 8>, iters), 8);
  report("+ 4 x (2 s_mov_b32 + 1 v_fma_f64 with that SGPR pair) (12 instr)", run(k_mix<KIND_SMOV, 4>, iters), 12);
  return 0;
}
