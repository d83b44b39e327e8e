// global_load_lds_dword semantics check (gfx950): every lane loads one dword from saddr + voffset into LDS at
// M0 + lane * 4.  Prints mismatches for a few (LDS offset, global offset) combinations.
// build: hipcc --offload-arch=gfx950 -O2 tools/micro/lds_direct_load.hip -o tools/micro/lds_direct_load.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const int* p, int* out, unsigned lds_off, unsigned stride_bytes) {
  extern __shared__ char lds[];
  const unsigned off = threadIdx.x * stride_bytes;
  const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)(size_t)lds + lds_off);
  // the offset in v10 with a non-zero v11 behind it: is the offset really read as 32 bits?
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tv_mov_b32 v10, %1\n\tv_mov_b32 v11, 0x7fffffff\n\tglobal_load_lds_dword v10, %2\n\ts_waitcnt vmcnt(0)" ::"s"(base), "v"(off), "s"(p) : "memory", "v10", "v11");
  __syncthreads();
  out[threadIdx.x] = ((const int*)(lds + lds_off))[threadIdx.x];
  if (threadIdx.x == 0) out[64] = (int)base;
}
int main() {
  const int n = 1 << 16;
  std::vector<int> h(n);
  for (int i = 0; i < n; ++i) h[i] = i * 7 + 1;
  int *d, *o;
  hipMalloc(&d, n * 4); hipMalloc(&o, 65 * 4);
  hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
  for (unsigned lds_off : {0u, 256u, 40000u}) {
    for (unsigned stride : {4u, 8u}) {
      hipMemset(o, 0xff, 65 * 4);
      hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
      hipLaunchKernelGGL(k, dim3(1), dim3(64), 65536, 0, d, o, lds_off, stride);
      hipError_t e = hipDeviceSynchronize();
      std::vector<int> r(65);
      hipMemcpy(r.data(), o, 65 * 4, hipMemcpyDeviceToHost);
      int bad = 0;
      for (int t = 0; t < 64; ++t) bad += r[t] != h[t * stride / 4];
      printf("lds_off %5u stride %u: %s, mismatches %d, m0 = %d, lane1 got %d expected %d\n", lds_off, stride, hipGetErrorString(e), bad, r[64], r[1], h[stride / 4]);
    }
  }
  return 0;
}
