// Probe of the direct-to-LDS load the ring form of the long chains relies on (mtg_lane.h: mtg_ws_prefetch_lds):
// global_load_lds_dwordx4 must write lane l's 16 bytes to M0 + 16 * l, and s_waitcnt vmcnt(0) must cover it.
// build: hipcc --offload-arch=gfx950 -O2 tools/micro/lds_dma_probe.hip -o gpurun_out/lds_dma_probe ; prints "lds dma layout ok".
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void lvoid;
typedef __attribute__((address_space(1))) const void gvoid;
__global__ void probe(const double* __restrict__ src, long long stride, double* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63;
  // two rows of 64 doubles, `stride` doubles apart in global memory -> 1024 consecutive bytes at lds + 2048
  const double* s = src + 2 * (lane & 31) + (lane >> 5) * stride;
  const unsigned slot = (unsigned)(size_t)(lds + 2048);
  __builtin_amdgcn_global_load_lds((gvoid*)s, (lvoid*)(size_t)__builtin_amdgcn_readfirstlane((int)slot), 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const double* l = (const double*)(lds + 2048);
  out[lane] = l[lane];
  out[64 + lane] = l[64 + lane];
}
int main() {
  const long long stride = 4096;
  std::vector<double> h(2 * stride);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (double)i;
  double *d, *o;
  if (hipMalloc(&d, h.size() * 8) != hipSuccess || hipMalloc(&o, 128 * 8) != hipSuccess) return 2;
  hipMemcpy(d, h.data(), h.size() * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 4096, 0, d, stride, o);
  double r[128];
  if (hipMemcpy(r, o, sizeof r, hipMemcpyDeviceToHost) != hipSuccess) return 2;
  int bad = 0;
  for (int i = 0; i < 64; ++i) bad += (r[i] != (double)i) + (r[64 + i] != (double)(stride + i));
  printf(bad ? "lds dma layout MISMATCH (%d)\n" : "lds dma layout ok\n", bad);
  return bad != 0;
}
