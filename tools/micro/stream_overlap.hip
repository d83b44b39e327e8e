// Do kernels launched on different HIP streams overlap on this device/runtime?  A spin kernel (64 workgroups x 256
// threads, ~40 us) is launched n times on ONE stream and once on each of n streams (fork/join with events on a main
// stream); wall time from events on the main stream + host time of the enqueue.
// build: hipcc --offload-arch=gfx950 -O2 tools/micro/stream_overlap.hip -o /tmp/stream_overlap
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void spin(long long cycles, int* sink) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < cycles) {}
  if (sink && threadIdx.x == 9999) *sink = 1;
}

int main() {
  hipStream_t main_s;
  CK(hipStreamCreateWithFlags(&main_s, hipStreamNonBlocking));
  const int NMAX = 12;
  std::vector<hipStream_t> side(NMAX);
  std::vector<hipEvent_t> join(NMAX);
  for (int i = 0; i < NMAX; ++i) {
    CK(hipStreamCreateWithFlags(&side[i], hipStreamNonBlocking));
    CK(hipEventCreateWithFlags(&join[i], hipEventDisableTiming));
  }
  hipEvent_t fork, e0, e1;
  CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const long long cyc = 4000;   // wall_clock64 ticks at 100 MHz: 40 us
  for (int wgs : {64, 256}) {
    for (int n : {1, 2, 4, 8, 12}) {
      for (int mode = 0; mode < 3; ++mode) {   // 0: one stream, 1: n streams, 2: one stream, hipExtAnyOrderLaunch
        double best = 1e30, host_best = 1e30;
        for (int rep = 0; rep < 6; ++rep) {
          CK(hipDeviceSynchronize());
          const auto h0 = std::chrono::steady_clock::now();
          CK(hipEventRecord(e0, main_s));
          if (mode == 0) {
            for (int i = 0; i < n; ++i) hipLaunchKernelGGL(spin, dim3(wgs), dim3(256), 0, main_s, cyc, (int*)nullptr);
          } else if (mode == 2) {
            for (int i = 0; i < n; ++i)
              hipExtLaunchKernelGGL(spin, dim3(wgs), dim3(256), 0, main_s, nullptr, nullptr, hipExtAnyOrderLaunch, cyc, (int*)nullptr);
          } else {
            CK(hipEventRecord(fork, main_s));
            for (int i = 0; i < n; ++i) CK(hipStreamWaitEvent(side[i], fork, 0));
            for (int i = 0; i < n; ++i) hipLaunchKernelGGL(spin, dim3(wgs), dim3(256), 0, side[i], cyc, (int*)nullptr);
            for (int i = 0; i < n; ++i) { CK(hipEventRecord(join[i], side[i])); CK(hipStreamWaitEvent(main_s, join[i], 0)); }
          }
          CK(hipEventRecord(e1, main_s));
          const auto h1 = std::chrono::steady_clock::now();
          CK(hipEventSynchronize(e1));
          float ms = 0;
          CK(hipEventElapsedTime(&ms, e0, e1));
          if (rep > 0) { best = std::min(best, (double)ms * 1e3); host_best = std::min(host_best, std::chrono::duration<double, std::micro>(h1 - h0).count()); }
        }
        printf("wgs=%3d n=%2d %-10s device %7.1f us   host enqueue %6.1f us\n", wgs, n, mode == 0 ? "one stream" : mode == 1 ? "n streams" : "any-order", best, host_best);
      }
    }
  }
  return 0;
}
