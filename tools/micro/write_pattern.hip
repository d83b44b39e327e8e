// write_pattern.hip -- what shape of store stream reaches the HBM write ceiling on MI355X?  Stand-alone microbenchmark
// (build: hipcc --offload-arch=gfx950 -O3 -o write_pattern write_pattern.hip).  Every variant writes the same buffer
// with 16-byte stores, 1 KiB contiguous per wave instruction; what varies is how the 1 KiB pieces are ordered in time:
//   oneshot : one wave = one contiguous slab of `slab` bytes, grid covers the buffer, workgroups retire as they finish
//   persist : `workers` resident waves, each walks slabs worker, worker + W, worker + 2W, ...   (the kernels' pattern)
//   rowmajor: `workers` resident waves, all of them advance together through the buffer 1 KiB at a time
//             (wave w writes piece step*W + w): the active window is W KiB wide instead of W slabs
// and the cache policy of the store (plain / nt / sc1 / sc0 sc1).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned int u4 __attribute__((ext_vector_type(4)));

template <int AUX>
__device__ __forceinline__ void store16(char* base, long long off, u4 v, long long limit) {
  if (off + 16 <= limit) {
    if (AUX == 0) *reinterpret_cast<u4*>(base + off) = v;
    else if (AUX == 1) __builtin_nontemporal_store(v, reinterpret_cast<u4*>(base + off));
    else {
      const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7fffffff, 0x00020000);
      (void)r;
      *reinterpret_cast<u4*>(base + off) = v;
    }
  }
}

template <int AUX>
__global__ __launch_bounds__(256) void k_oneshot(char* out, long long bytes, int slab) {
  const int lane = threadIdx.x & 63;
  const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long long base = wave * slab;
  u4 v = {(unsigned)wave, (unsigned)lane, 3u, 4u};
  for (int o = lane * 16; o < slab; o += 1024) store16<AUX>(out, base + o, v, bytes);
}

template <int AUX>
__global__ __launch_bounds__(256) void k_persist(char* out, long long bytes, int slab) {
  const int lane = threadIdx.x & 63;
  const long long W = (long long)gridDim.x * 4;
  u4 v = {(unsigned)blockIdx.x, (unsigned)lane, 3u, 4u};
  for (long long s = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); s * slab < bytes; s += W) {
    const long long base = s * slab;
    for (int o = lane * 16; o < slab; o += 1024) store16<AUX>(out, base + o, v, bytes);
    v.x += 1;
  }
}

template <int AUX>
__global__ __launch_bounds__(256) void k_rowmajor(char* out, long long bytes, int slab) {
  (void)slab;
  const int lane = threadIdx.x & 63;
  const long long W = (long long)gridDim.x * 4;
  u4 v = {(unsigned)blockIdx.x, (unsigned)lane, 3u, 4u};
  for (long long p = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); p * 1024 < bytes; p += W) {
    store16<AUX>(out, p * 1024 + lane * 16, v, bytes);
    v.x += 1;
  }
}

// persist, but with ~`work` dependent FP64 FMAs between slabs (a compute phase per slab, like the real kernels)
__global__ __launch_bounds__(256) void k_persist_work(char* out, long long bytes, int slab, int work, double* sink) {
  const int lane = threadIdx.x & 63;
  const long long W = (long long)gridDim.x * 4;
  double a = lane * 1e-3, b = 1.0000001;
  for (long long s = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); s * slab < bytes; s += W) {
    for (int i = 0; i < work; ++i) a = __builtin_fma(a, b, 1e-9);
    u4 v = {(unsigned)__double2loint(a), (unsigned)lane, 3u, 4u};
    const long long base = s * slab;
    for (int o = lane * 16; o < slab; o += 1024) store16<0>(out, base + o, v, bytes);
  }
  if (a == 123.456) *sink = a;
}

template <class F>
static double time_us(F launch, int reps = 5) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  launch(); hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) launch();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3 / reps;
}

int main(int argc, char** argv) {
  const long long bytes = (argc > 1 ? atoll(argv[1]) : 4096ll) << 20;   // MiB
  char* out; double* sink;
  if (hipMalloc(&out, bytes) != hipSuccess || hipMalloc(&sink, 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(out, 0, bytes);
  printf("buffer %lld MiB\n", bytes >> 20);
  for (int slab : {1024, 7680, 15360, 61440}) {
    const long long nslab = (bytes + slab - 1) / slab;
    const unsigned g1 = (unsigned)((nslab + 3) / 4);
    double t = time_us([&] { hipLaunchKernelGGL(k_oneshot<0>, dim3(g1), dim3(256), 0, 0, out, bytes, slab); });
    printf("oneshot  slab %6d            : %8.1f us  %5.2f TB/s\n", slab, t, bytes / t * 1e-6);
    for (int grid : {512, 1024, 2048, 4096}) {
      t = time_us([&] { hipLaunchKernelGGL(k_persist<0>, dim3(grid), dim3(256), 0, 0, out, bytes, slab); });
      printf("persist  slab %6d grid %5d : %8.1f us  %5.2f TB/s\n", slab, grid, t, bytes / t * 1e-6);
    }
  }
  for (int grid : {512, 1024, 2048, 4096}) {
    double t = time_us([&] { hipLaunchKernelGGL(k_rowmajor<0>, dim3(grid), dim3(256), 0, 0, out, bytes, 0); });
    printf("rowmajor             grid %5d : %8.1f us  %5.2f TB/s\n", grid, t, bytes / t * 1e-6);
  }
  {
    double t = time_us([&] { hipLaunchKernelGGL(k_persist<1>, dim3(2048), dim3(256), 0, 0, out, bytes, 7680); });
    printf("persist nt slab 7680 grid 2048  : %8.1f us  %5.2f TB/s\n", t, bytes / t * 1e-6);
    t = time_us([&] { hipLaunchKernelGGL(k_oneshot<1>, dim3((unsigned)(((bytes + 7679) / 7680 + 3) / 4)), dim3(256), 0, 0, out, bytes, 7680); });
    printf("oneshot nt slab 7680            : %8.1f us  %5.2f TB/s\n", t, bytes / t * 1e-6);
  }
  for (int work : {0, 200, 600, 1500}) {
    double t = time_us([&] { hipLaunchKernelGGL(k_persist_work, dim3(2048), dim3(256), 0, 0, out, bytes, 7680, work, sink); });
    printf("persist+work %4d slab 7680 grid 2048 : %8.1f us  %5.2f TB/s\n", work, t, bytes / t * 1e-6);
  }
  hipFree(out);
  return 0;
}
