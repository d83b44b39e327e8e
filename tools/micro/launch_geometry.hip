// launch_geometry.hip -- GPU-side cost of a small launch on MI355X as a function of its geometry.
//
// The round-1 launch-floor numbers were taken with host-issued back-to-back launches and turned out to be bounded by
// the host's enqueue rate for kernels with arguments.  Here a chain of R kernel nodes is captured into one hipGraph
// (no host in the loop) and every wave stamps the device-wide 100 MHz clock when it starts and when it ends:
//   period    = graph time / R                        (what a back-to-back stream pays per launch, GPU side)
//   ramp      = last wave start - first wave start     (dispatch skew inside one kernel)
//   span      = last wave end - first wave start
//   gap       = first wave start of kernel i+1 - last wave end of kernel i   (completion + next dispatch)
// Geometry = workgroups x waves per workgroup, dynamic LDS per workgroup, VGPR allocation (small / 256), and an
// optional body: SPIN shader cycles of s_sleep-free busy work (v_fma chain) to emulate a wave that lives a few us.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <bool BIGV>
__global__ __launch_bounds__(512) void k_probe(long long* stamps, int iter, int maxw, int spin, double* sink) {
  extern __shared__ double lds[];
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long long t0 = wall_clock64();
  double a = 1.0 + threadIdx.x;
  if (BIGV) asm volatile("v_mov_b32 v250, 0" ::: "v250");
  if (spin > 0) {
    const long long c0 = clock64();
    while (clock64() - c0 < spin) {
#pragma unroll
      for (int i = 0; i < 8; ++i) a = __builtin_fma(a, 1.0000001, 1e-9);
    }
  }
  if (a == 1.2345e300) { sink[0] = a + lds[threadIdx.x]; }
  const long long t1 = wall_clock64();
  if ((threadIdx.x & 63) == 0) {
    long long* s = stamps + ((long long)iter * maxw + wave) * 2;
    s[0] = t0;
    s[1] = t1;
  }
}

struct Geo { int wgs, threads, lds_kb; bool bigv; int spin; };

int main() {
  const int R = 24, MAXW = 4096;
  long long* d_st; double* sink;
  CK(hipMalloc(&d_st, sizeof(long long) * R * MAXW * 2));
  CK(hipMalloc(&sink, 8));
  hipStream_t st; CK(hipStreamCreate(&st));
  std::vector<Geo> geos = {
      {471, 128, 52, true, 0},  {471, 128, 0, true, 0},   {471, 128, 52, false, 0}, {471, 128, 0, false, 0},
      {157, 128, 52, true, 0},  {157, 384, 64, true, 0},  {157, 512, 64, true, 0},  {157, 256, 64, true, 0},
      {236, 256, 64, true, 0},  {118, 512, 64, true, 0},  {942, 64, 26, true, 0},   {314, 128, 52, true, 0},
      {314, 256, 64, true, 0},  {256, 256, 64, true, 0},  {256, 512, 64, true, 0},  {1024, 64, 16, true, 0},
      {471, 128, 52, true, 12000}, {157, 384, 64, true, 12000}, {157, 512, 64, true, 12000}, {314, 256, 64, true, 12000},
      {942, 64, 26, true, 12000},  {157, 128, 52, true, 12000},
  };
  std::vector<long long> h(R * MAXW * 2);
  std::printf("%-34s %8s %8s %8s %8s %8s\n", "geometry", "period", "ramp", "span", "gap", "ramp90");
  for (const Geo& g : geos) {
    const int nw = g.wgs * g.threads / 64;
    auto fn = g.bigv ? k_probe<true> : k_probe<false>;
    CK(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CK(hipMemsetAsync(d_st, 0, sizeof(long long) * R * MAXW * 2, st));
    hipGraph_t graph; hipGraphExec_t exec;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    for (int i = 0; i < R; ++i) hipLaunchKernelGGL(fn, dim3(g.wgs), dim3(g.threads), g.lds_kb * 1024, st, d_st, i, MAXW, g.spin, sink);
    CK(hipStreamEndCapture(st, &graph));
    CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    for (int w = 0; w < 3; ++w) CK(hipGraphLaunch(exec, st));
    CK(hipStreamSynchronize(st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int NG = 10;
    CK(hipEventRecord(e0, st));
    for (int w = 0; w < NG; ++w) CK(hipGraphLaunch(exec, st));
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(h.data(), d_st, sizeof(long long) * R * MAXW * 2, hipMemcpyDeviceToHost));
    std::vector<double> ramps, spans, gaps, ramp90;
    long long prev_end = 0;
    for (int i = 0; i < R; ++i) {
      std::vector<long long> s0(nw), s1(nw);
      for (int w = 0; w < nw; ++w) { s0[w] = h[((long long)i * MAXW + w) * 2]; s1[w] = h[((long long)i * MAXW + w) * 2 + 1]; }
      std::sort(s0.begin(), s0.end());
      const long long first = s0.front(), last = s0.back(), end = *std::max_element(s1.begin(), s1.end());
      ramps.push_back((last - first) * 0.01);
      ramp90.push_back((s0[(size_t)(nw * 0.9)] - first) * 0.01);
      spans.push_back((end - first) * 0.01);
      if (i > 0) gaps.push_back((first - prev_end) * 0.01);
      prev_end = end;
    }
    auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    char name[96];
    std::snprintf(name, sizeof name, "%4d x %3d lds %2dK %s spin %5d", g.wgs, g.threads, g.lds_kb, g.bigv ? "v256" : "v-sm", g.spin);
    std::printf("%-34s %8.2f %8.2f %8.2f %8.2f %8.2f\n", name, ms * 1e3 / (NG * R), med(ramps), med(spans), med(gaps), med(ramp90));
    CK(hipGraphExecDestroy(exec)); CK(hipGraphDestroy(graph));
  }
  // same chain, host-issued (what bench.py's stream sees): period only
  for (const Geo& g : {Geo{471, 128, 52, true, 0}, Geo{471, 128, 52, true, 12000}, Geo{157, 384, 64, true, 12000}}) {
    auto fn = k_probe<true>;
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(fn, dim3(g.wgs), dim3(g.threads), g.lds_kb * 1024, st, d_st, i % R, MAXW, g.spin, sink);
    CK(hipStreamSynchronize(st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < 240; ++i) hipLaunchKernelGGL(fn, dim3(g.wgs), dim3(g.threads), g.lds_kb * 1024, st, d_st, i % R, MAXW, g.spin, sink);
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    std::printf("host-issued %4d x %3d lds %2dK spin %5d: period %.2f us\n", g.wgs, g.threads, g.lds_kb, g.spin, ms * 1e3 / 240);
  }
  return 0;
}
