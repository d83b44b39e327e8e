// mtg_multi_device.hip -- one host process driving several GPUs through the C ABI.
//
// The path shards by independent trajectories: shard g of G is the contiguous range mtg_shard_range(batch, G, g) and is
// solved on device g by that device's own context / plan / stream -- no communication during the solve.  What a C++
// consumer of the reference (a single process) needs on top of the per-device entry points is only bookkeeping: one
// context + plan per device, fan-out of a call, a joint sync, and the optional final gather of the coefficient shards
// (peer copies over xGMI to one device, or DMA to host memory).  The multi-PROCESS form (one rank per GPU, RCCL
// all_gather) lives in mav_trajectory_generation_amd/dist.py and bench.py; both split the batch the same way.
#include <hip/hip_runtime.h>

#include <new>
#include <string>
#include <vector>

#include "../../include/mtg_hip.h"

extern "C" int mtg_context_stream_device(mtg_context* ctx, void** stream, int* device);

struct mtg_device_group {
  std::vector<int> devices;
  std::vector<mtg_context*> ctx;
  std::vector<mtg_plan*> plan;
  mtg_plan_info info;
  int N = 0, D = 0, K = 0;
};

extern "C" {

void mtg_shard_range(int64_t batch, int32_t n_shards, int32_t shard, int64_t* lo, int64_t* hi) {
  if (n_shards <= 0 || shard < 0 || shard >= n_shards || batch < 0) {   // no such shard: the empty range
    if (lo) *lo = 0;
    if (hi) *hi = 0;
    return;
  }
  const int64_t base = batch / n_shards, rem = batch % n_shards;
  const int64_t l = shard * base + (shard < rem ? shard : rem);
  if (lo) *lo = l;
  if (hi) *hi = l + base + (shard < rem ? 1 : 0);
}

int mtg_device_group_destroy(mtg_device_group* g) {
  if (!g) return MTG_OK;
  for (mtg_plan* p : g->plan) if (p) mtg_plan_destroy(p);
  for (mtg_context* c : g->ctx) if (c) mtg_context_destroy(c);
  delete g;
  return MTG_OK;
}

int mtg_device_group_create(int32_t n_devices, const int32_t* devices, const mtg_plan_desc* desc, mtg_device_group** out) {
  if (!out || n_devices < 1 || !desc) return MTG_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  mtg_device_group* g = new (std::nothrow) mtg_device_group();
  if (!g) return MTG_ERR_DEVICE;
  g->N = desc->n_coeffs; g->D = desc->dimension; g->K = desc->n_segments;
  for (int i = 0; i < n_devices; ++i) {
    const int dev = devices ? devices[i] : i;
    mtg_context* c = nullptr;
    int rc = mtg_context_create(dev, nullptr, &c);
    if (rc != MTG_OK) { mtg_device_group_destroy(g); return rc; }
    g->devices.push_back(dev);
    g->ctx.push_back(c);
    mtg_plan* p = nullptr;
    rc = mtg_plan_create(c, desc, &p);
    if (rc != MTG_OK) { mtg_device_group_destroy(g); return rc; }
    g->plan.push_back(p);
  }
  mtg_plan_get_info(g->plan[0], &g->info);
  // peer access for the gather (a device listed twice -- tests on a 1-GPU box -- needs none)
  for (size_t a = 0; a < g->devices.size(); ++a) {
    for (size_t b = 0; b < g->devices.size(); ++b) {
      if (g->devices[a] == g->devices[b]) continue;
      int can = 0;
      if (hipDeviceCanAccessPeer(&can, g->devices[a], g->devices[b]) == hipSuccess && can) {
        hipSetDevice(g->devices[a]);
        const hipError_t e = hipDeviceEnablePeerAccess(g->devices[b], 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { (void)hipGetLastError(); }
      }
    }
  }
  *out = g;
  return MTG_OK;
}

int mtg_device_group_size(const mtg_device_group* g) { return g ? (int)g->ctx.size() : 0; }
mtg_context* mtg_device_group_context(mtg_device_group* g, int32_t shard) {
  return (g && shard >= 0 && shard < (int)g->ctx.size()) ? g->ctx[shard] : nullptr;
}
mtg_plan* mtg_device_group_plan(mtg_device_group* g, int32_t shard) {
  return (g && shard >= 0 && shard < (int)g->plan.size()) ? g->plan[shard] : nullptr;
}

int mtg_device_group_solve_linear(mtg_device_group* g, int64_t batch, int32_t soa, const double* const* times,
                                  const double* const* d_fixed, double* const* coeffs, uint32_t flags) {
  if (!g || batch < 0 || !times || !coeffs || (flags & MTG_FLAG_HOST_POINTERS)) return MTG_ERR_INVALID_ARGUMENT;
  const int G = (int)g->ctx.size();
  for (int s = 0; s < G; ++s) {   // asynchronous on every device's own stream: the shards run concurrently
    int64_t lo, hi;
    mtg_shard_range(batch, G, s, &lo, &hi);
    if (hi == lo) continue;
    mtg_layout lay;
    if (soa) mtg_layout_soa(g->plan[s], hi - lo, &lay); else mtg_layout_aos(g->plan[s], hi - lo, &lay);
    const int rc = mtg_solve_linear(g->plan[s], hi - lo, &lay, times[s], d_fixed ? d_fixed[s] : nullptr, coeffs[s], nullptr,
                                    nullptr, flags);
    if (rc != MTG_OK) return rc;
  }
  return MTG_OK;
}

int mtg_device_group_sync(mtg_device_group* g) {
  if (!g) return MTG_ERR_INVALID_ARGUMENT;
  int first = MTG_OK;
  for (mtg_context* c : g->ctx) {
    const int rc = mtg_context_sync(c);
    if (rc != MTG_OK && first == MTG_OK) first = rc;
  }
  return first;
}

// Final gather: shard s (on device s) -> dst[lo_s .. hi_s) where dst is host memory (root < 0) or memory of device
// group[root] (peer copies over xGMI).  Each copy is ordered after its shard's solve on that shard's stream; returns
// after all of them completed.
int mtg_device_group_gather_coeffs(mtg_device_group* g, int64_t batch, const double* const* coeffs, int32_t root, double* dst) {
  if (!g || !coeffs || !dst || batch < 0 || root >= (int)g->ctx.size()) return MTG_ERR_INVALID_ARGUMENT;
  const int G = (int)g->ctx.size();
  const size_t per_traj = (size_t)g->K * g->D * g->N * sizeof(double);
  for (int s = 0; s < G; ++s) {
    int64_t lo, hi;
    mtg_shard_range(batch, G, s, &lo, &hi);
    if (hi == lo) continue;
    void* stream = nullptr;
    int dev = 0;
    if (mtg_context_stream_device(g->ctx[s], &stream, &dev) != MTG_OK) return MTG_ERR_DEVICE;
    if (hipSetDevice(dev) != hipSuccess) return MTG_ERR_DEVICE;
    char* out = reinterpret_cast<char*>(dst) + (size_t)lo * per_traj;
    hipError_t e;
    if (root < 0) e = hipMemcpyAsync(out, coeffs[s], (size_t)(hi - lo) * per_traj, hipMemcpyDeviceToHost, (hipStream_t)stream);
    else e = hipMemcpyPeerAsync(out, g->devices[root], coeffs[s], dev, (size_t)(hi - lo) * per_traj, (hipStream_t)stream);
    if (e != hipSuccess) return MTG_ERR_DEVICE;
  }
  for (int s = 0; s < G; ++s) {
    void* stream = nullptr;
    int dev = 0;
    mtg_context_stream_device(g->ctx[s], &stream, &dev);
    if (hipSetDevice(dev) != hipSuccess || hipStreamSynchronize((hipStream_t)stream) != hipSuccess) return MTG_ERR_DEVICE;
  }
  return MTG_OK;
}

}  // extern "C"
