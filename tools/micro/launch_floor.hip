// launch_floor.hip -- how much of a ~10 us kernel on MI355X is dispatch + completion?  Back-to-back launches on one stream,
// hipEvent-bracketed (the same protocol as mtg_time_last_solve), with the bench kernel's launch geometry
// (471 workgroups x 128 threads, 52 KB dynamic LDS):  empty / LDS only / one coalesced load round trip /
// the bench workload's 19.2 MB written with plain and with sc1 stores.
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void k_empty() {}
__global__ void k_lds(int* sink) {
  extern __shared__ double lds[];
  if (threadIdx.x == 9999) sink[0] = (int)lds[0];
}
__global__ __launch_bounds__(128) void k_load(const double* in, double* out, long long n) {
  const long long i = (long long)blockIdx.x * 128 + threadIdx.x;
  double a = 0;
#pragma unroll
  for (int c = 0; c < 13; ++c) a += in[(i + (long long)c * 30000) % n];   // 13 independent coalesced loads
  if (a == 1.2345e300) out[0] = a;
}
template <int AUX>
__global__ __launch_bounds__(128) void k_write(double* out, long long n16) {
  typedef unsigned int u4 __attribute__((ext_vector_type(4)));
  const u4 v = {1u, 2u, 3u, 4u};
  for (long long i = (long long)blockIdx.x * 128 + threadIdx.x; i < n16; i += (long long)gridDim.x * 128) {
    u4* p = reinterpret_cast<u4*>(out) + i;
    if (AUX == 0) *p = v;
    else asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  }
}

template <class F>
static double us(F launch, int reps = 300) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 20; ++i) launch();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) launch();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3 / reps;
}

int main() {
  const long long n = 10000ll * 240;   // doubles of the bench workload's output
  double *in, *out; int* sink;
  hipMalloc(&in, n * 8); hipMalloc(&out, n * 8); hipMalloc(&sink, 4);
  hipMemset(in, 0, n * 8);
  std::printf("empty, 471 x 128                      : %6.2f us\n", us([&] { hipLaunchKernelGGL(k_empty, dim3(471), dim3(128), 0, 0); }));
  std::printf("empty, 471 x 128, 52 KB LDS           : %6.2f us\n", us([&] { hipLaunchKernelGGL(k_lds, dim3(471), dim3(128), 53248, 0, sink); }));
  for (int kb : {4, 8, 12, 16, 24, 32, 40, 64})
    std::printf("empty, 471 x 128, %2d KB LDS           : %6.2f us\n", kb, us([&] { hipLaunchKernelGGL(k_lds, dim3(471), dim3(128), kb * 1024, 0, sink); }));
  for (int g : {157, 314, 942})
    std::printf("empty, %3d x 128, 24 KB LDS           : %6.2f us\n", g, us([&] { hipLaunchKernelGGL(k_lds, dim3(g), dim3(128), 24 * 1024, 0, sink); }));
  std::printf("13 coalesced loads per lane, 471 x 128: %6.2f us\n", us([&] { hipLaunchKernelGGL(k_load, dim3(471), dim3(128), 0, 0, in, out, n); }));
  std::printf("write 19.2 MB, plain stores           : %6.2f us\n", us([&] { hipLaunchKernelGGL(k_write<0>, dim3(471), dim3(128), 0, 0, out, n / 2); }));
  std::printf("write 19.2 MB, sc1 stores             : %6.2f us\n", us([&] { hipLaunchKernelGGL(k_write<1>, dim3(471), dim3(128), 0, 0, out, n / 2); }));
  std::printf("write 19.2 MB, plain, 2048 x 128      : %6.2f us\n", us([&] { hipLaunchKernelGGL(k_write<0>, dim3(2048), dim3(128), 0, 0, out, n / 2); }));
  std::printf("write 19.2 MB, sc1, 2048 x 128        : %6.2f us\n", us([&] { hipLaunchKernelGGL(k_write<1>, dim3(2048), dim3(128), 0, 0, out, n / 2); }));
  return 0;
}
