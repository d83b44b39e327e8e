// fetch_calib.hip -- calibration of the rocprofv3 FETCH_SIZE / WRITE_SIZE counters on gfx950 for the access shapes
// the solve kernels use: coalesced 8-byte-per-lane loads (the kernels' input loads), 16-byte-per-lane loads (the shape
// the MI355X_MICROARCH.md "x2" correction was derived for) and 16-byte-per-lane stores, each over a 2 GiB buffer
// (8x the 256 MiB Infinity Cache) touched exactly once.  Run under
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -- tools/micro/fetch_calib      (and --pmc WRITE_SIZE)
// and divide the counter (KiB) by the bytes below: tools/gpu_profile2.sh does that and writes the factors.
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void k_read8(const double* in, double* sink, size_t n) {
  double a = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a += in[i];
  if (a == 1.2345e300) sink[0] = a;
}
__global__ void k_read16(const double2* in, double* sink, size_t n) {
  double a = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { double2 v = in[i]; a += v.x + v.y; }
  if (a == 1.2345e300) sink[0] = a;
}
__global__ void k_write16(double2* out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = make_double2(1.0, 2.0);
}

int main() {
  const size_t bytes = 2ull << 30;
  double *buf, *sink;
  if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 8) != hipSuccess) return 1;
  hipMemset(buf, 0, bytes);
  hipDeviceSynchronize();
  hipLaunchKernelGGL(k_read8, dim3(4096), dim3(256), 0, 0, buf, sink, bytes / 8);
  hipLaunchKernelGGL(k_read16, dim3(4096), dim3(256), 0, 0, (const double2*)buf, sink, bytes / 16);
  hipLaunchKernelGGL(k_write16, dim3(4096), dim3(256), 0, 0, (double2*)buf, bytes / 16);
  hipDeviceSynchronize();
  std::printf("bytes per kernel %zu\n", bytes);
  return 0;
}
