// mtg_hip.hip -- gfx950 kernels and the C ABI of include/mtg_hip.h.
//
// Kernel inventory:
//   mtg_solve_kernel<Cfg, WITH_COST>  fused updateSegmentTimes + constructR + solve + coefficient
//       recovery (impl/polynomial_optimization_linear_impl.h:286-379, :263-283).  Workgroup =
//       two wavefronts (forward / backward chain direction) x 64 trajectories; persistent
//       grid-stride over 64-trajectory tiles; Schur complements of the middle vertex exchanged
//       through LDS.  Cfg::kStatic variants keep the back-substitution data in registers
//       (fully unrolled, compile-time masks); the generic variant streams it through a
//       lane-coalesced global workspace and takes K / masks at run time.
//   mtg_update_kernel<Cfg, WITH_COST> setFreeConstraints path (impl/...:500-508): recovery only.
//   mtg_rcp_selftest_kernel           accuracy probe of the pivot reciprocal.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <queue>
#include <functional>
#include <new>
#include <string>
#include <vector>

#include "../../include/mtg_hip.h"
#include "mtg_kernels.h"
#include "mtg_dimlane_rt.h"

int mtg_host_run(const MtgParams& P, int H, bool update);   // mtg_host.cpp: host build of the lane code
// mtg_coop.hip: the row-cooperative kernel (0 launched, 1 shape / size not covered, 2 runtime error) and its LDS need
int mtg_coop_launch(void* stream, int H, int D, int K, int deriv, long long B, const double* times, long long ts_b, long long ts_k,
                    const double* dfix, long long fs_b, long long fs_d, long long fs_c, double* coeffs, int* status, int* tstatus);
size_t mtg_coop_lds_bytes(int H, int D, int K);
extern "C" int mtg_basic_solution_one(int H, int K, int D, int deriv, const int* mask, const int* offF, const int* offP,
                                      const double* times, const double* dfix, double* dfree);   // mtg_basic.cpp (internal; exported for the CPU tests)

namespace {

__global__ void mtg_rcp_selftest_kernel(int n, double* out, int iters) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  double err = 0.0;
  if (i < n) {
    // deterministic pseudo-random positive doubles over ~24 binades
    unsigned long long z = 0x9E3779B97F4A7C15ull * (unsigned long long)(i + 1);
    z ^= z >> 31; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 29;
    const double m = 1.0 + (double)(z >> 11) * (1.0 / 9007199254740992.0);
    const int e = (int)((z & 0x3FF) % 49) - 24;
    const double x = ldexp(m, e);
    double r;
    if (iters == 2) {
      r = mtg_rcp(x);
    } else {
      r = __builtin_amdgcn_rcp(x);
      for (int it = 0; it < iters; ++it) r = mtg_fma(mtg_fma(-x, r, 1.0), r, r);
    }
    const double ref = 1.0 / x;
    err = fabs(r - ref) / ref;
  }
  // block max -> atomic max on the bit pattern (values are non-negative)
  __shared__ double sm[256];
  sm[threadIdx.x] = err;
  __syncthreads();
  for (int s = blockDim.x / 2; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) sm[threadIdx.x] = fmax(sm[threadIdx.x], sm[threadIdx.x + s]);
    __syncthreads();
  }
  if (threadIdx.x == 0) atomicMax((unsigned long long*)out, (unsigned long long)__double_as_longlong(sm[0]));
}


}  // namespace

#define MTG_DECL(H) SolveFn mtg_pick_generic_solve_h##H(int, int); UpdateFn mtg_pick_generic_update_h##H(int, bool);
MTG_DECL(1) MTG_DECL(2) MTG_DECL(3) MTG_DECL(4) MTG_DECL(5) MTG_DECL(6)
#undef MTG_DECL

SolveFn mtg_pick_generic_solve(int h, int d, int extra) {
  switch (h) {
    case 1: return mtg_pick_generic_solve_h1(d, extra);
    case 2: return mtg_pick_generic_solve_h2(d, extra);
    case 3: return mtg_pick_generic_solve_h3(d, extra);
    case 4: return mtg_pick_generic_solve_h4(d, extra);
    case 5: return mtg_pick_generic_solve_h5(d, extra);
    case 6: return mtg_pick_generic_solve_h6(d, extra);
  }
  return nullptr;
}
UpdateFn mtg_pick_generic_update(int h, int d, bool wc) {
  switch (h) {
    case 1: return mtg_pick_generic_update_h1(d, wc);
    case 2: return mtg_pick_generic_update_h2(d, wc);
    case 3: return mtg_pick_generic_update_h3(d, wc);
    case 4: return mtg_pick_generic_update_h4(d, wc);
    case 5: return mtg_pick_generic_update_h5(d, wc);
    case 6: return mtg_pick_generic_update_h6(d, wc);
  }
  return nullptr;
}

// ---------------------------------------------------------------------------------------------
struct mtg_context {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  int* d_status = nullptr;
  int* h_status = nullptr;  // pinned
  // small host-pointer calls (the single-trajectory drop-in path): one page-locked bounce buffer, so that a call is one
  // H2D DMA, the kernel, one D2H DMA and one synchronisation instead of five staged pageable copies
  double* h_bounce = nullptr;
  size_t h_bounce_bytes = 0;
  int n_cu = 256;
  // measurement knobs (A/B runs in tools/, form-forcing tests): set through mtg_context_set_option (include/mtg_hip_lab.h),
  // never read from the environment by the library; defaults = the shipped behaviour.  The names in the comments are the
  // environment variables the PYTHON layer forwards (mav_trajectory_generation_amd.Context) -- e.g. MTG_FORCE_DG -> "force_dg"
  int knob_force_dg = 0;        // MTG_FORCE_DG: dimension-group size of the specialised kernels
  bool knob_prefer_rolled = false;   // MTG_PREFER_ROLLED: rolled variant even where a static one exists
  bool knob_no_dimlane = false;      // MTG_NO_DIMLANE: never pick the dimension-in-lane form
  int dl_max_units_per_cu = -1;      // MTG_DL_MAX_UNITS: overrides the variants' upper limit (workgroups <= this x CUs; 0: none)
  bool knob_no_slab = false;         // MTG_NO_SLAB: fused form without the slab-output kernel
  bool knob_no_slab_extra = false;   // MTG_NO_SLAB_EXTRA: extra outputs (cost / d_P) through the older fused kernel
  bool knob_no_dl_extra = false;     // MTG_NO_DL_EXTRA: extra outputs never through the dimension-in-lane kernels
  bool knob_no_queue = false;        // MTG_NO_QUEUE: mtg_solve_linear_sequence as one launch per batch
  int knob_dl_grid_per_cu = 8;       // MTG_DL_GRID_PER_CU: workgroups per CU of a (non-workspace) dimension-in-lane launch
  int knob_dl_rt = -1;               // MTG_DL_RT: 1 = the run-time-K body even where a static variant exists, 0 = never (default: where none exists)
  bool knob_dl_any_rr = false;       // MTG_DL_ANY_SCHED=rr: round 2's unit schedule of the cross-structure launch
  bool knob_no_balance = false;      // MTG_NO_BALANCE: persistent grids are not evened out over their rounds
  int knob_slab_policy = -1;         // MTG_SLAB_POLICY: 0 write-back, 1 nt sc1
  int rolled_wg_per_cu = 4;          // MTG_ROLLED_WG_PER_CU: persistent workgroups per CU of the rolled (workspace) kernels
  bool knob_sample_generic = false;  // MTG_SAMPLE_GENERIC: mtg_sample_range never through its LDS-staged kernel
  int knob_extrema_split = -1;       // MTG_EXTREMA_SPLIT: lanes per root search of the extrema kernels (include/mtg_hip_lab.h; -1: default)
  int knob_coop = -1;                // MTG_COOP: 1 always / 0 never take the row-cooperative form where eligible (default: by size)
  // MTG_FLAG_CONCURRENT_ITEMS requests: side streams (created on first use) + fork / join events
  std::vector<hipStream_t> side_streams;
  hipEvent_t fork_event = nullptr;
  std::vector<hipEvent_t> join_events;
  // Cross-structure requests (mtg_multi_create -> mtg_solve_dl_any_kernel) are typically rebuilt with the SAME structure and new
  // buffers (a planner's mixed request per cycle; bench.py --config 4 rebuilds its 240-item request per timed region): the
  // per-workgroup unit lists depend only on the (kernel body, tile count) sequence of the items, so they are computed and
  // uploaded once per structure and shared read-only by every request of that structure; the workspace is one buffer per
  // context (requests of one context run in stream order), the small per-request item tables come from a free list.
  // Round 4: every create paid four hipMalloc, three synchronous copies and a 29k-unit heap schedule: 2.45 ms per 240 items,
  // three times the launch it prepared.
  struct DlAnySchedule {
    std::vector<long long> key;        // grid, schedule kind, then (body index, tiles) per item in launch order
    int grid = 0, nunits = 0;
    void* d_units = nullptr;           // MtgDlAnyUnit [nunits]
    int* d_wg_begin = nullptr;         // [grid + 1]
  };
  std::vector<DlAnySchedule> dl_any_schedules;      // never evicted while the context lives (bounded: kMaxDlAnySchedules)
  double* dl_any_ws = nullptr;
  size_t dl_any_ws_bytes = 0;
  std::vector<std::pair<void*, size_t>> dl_any_item_pool;   // free item tables (device)
  std::string last_error;
  std::mutex mu;
};

struct LaunchRecord {
  bool valid = false;
  SolveFn fn = nullptr;
  MtgParams params;
  int ntiles = 0, grid = 0, gridy = 1;
  size_t lds = 0;
  const MtgDimlaneEntry* dl = nullptr;   // dimension-in-lane launch (mtg_dimlane.h): uses params.{times,dfix,coeffs,status,tstatus,B}
  const MtgDimlaneRtEntry* rt = nullptr; // run-time-K dimension-in-lane launch (mtg_dimlane_rt.h)
  int dl_aos = 0;                        // input layout kind of a dimension-in-lane launch (dimlane_input_kind)
  bool coop = false;                     // row-cooperative launch (mtg_coop.hip)
  double* dl_ws = nullptr;
};

struct mtg_plan {
  mtg_context* ctx = nullptr;
  int N = 0, H = 0, D = 0, K = 0, deriv = 0;
  std::vector<int> mask;            // [K+1]
  std::vector<int> offF, offP;      // [K+2]
  int n_fixed = 0, n_free = 0;
  int null_dim = 0;                 // STRUCTURAL rank deficiency of the free system R_PP (structural_null_dim below)
  int* d_tables = nullptr;          // vmask | offF | offP
  const MtgStaticEntry* fast = nullptr;        // all dimensions in one workgroup
  const MtgStaticEntry* fast_split = nullptr;  // smallest dimension group that divides D
  const MtgDimlaneEntry* dimlane = nullptr;    // dimension-in-lane form (canonical SoA inputs, coefficient output only)
  const MtgDimlaneRtEntry* dimlane_rt = nullptr;   // run-time-K dimension-in-lane body (mtg_dimlane_rt.h): any chain length of the standard shapes
  bool slab_attr_set[2] = {false, false};      // LDS attribute of the slab-output kernels set
  bool slab_queue_attr_set = false;
  bool slab_extra_attr_set = false;
  double* ws = nullptr;
  size_t ws_bytes = 0;
  double* pert_cost = nullptr;      // [(K + 1)][batch] costs of mtg_mellinger_cost_gradient's virtual problems
  size_t pert_cost_bytes = 0;
  double* user_ws = nullptr;       // caller-owned workspace (mtg_plan_set_workspace)
  size_t user_ws_bytes = 0;
  // staging for MTG_FLAG_HOST_POINTERS
  double* stage = nullptr;
  size_t stage_bytes = 0;
  // MTG_FLAG_BASIC_SOLUTION with device pointers: [status word (8 bytes) | per-trajectory status int32 [batch]] of the call itself
  double* basic_status = nullptr;
  size_t basic_status_bytes = 0;
  // Structurally rank-deficient plans: the SHADOW plan = this pattern with null_dim additional slots fixed (to zero), chosen so
  // that the fixed functionals span the cost's null space -- a regular system whose solution is a basic solution of this one
  // (MTG_FLAG_BASIC_SOLUTION).  shadow_fixed_src[j]: column of this plan's d_fixed behind the shadow's fixed column j (-1: a
  // pinned slot, value 0); free_in_shadow[j]: the shadow's free column of this plan's free column j (-1: pinned, value 0).
  mtg_plan* shadow = nullptr;
  std::vector<int> shadow_fixed_src, free_in_shadow;
  int* d_shadow_maps = nullptr;      // device copy: shadow_fixed_src | free_in_shadow
  double* shadow_buf = nullptr;      // [batch][D][n_fixed of the shadow] | [batch][D][n_free of the shadow]
  size_t shadow_buf_bytes = 0;
  double* refine_buf = nullptr;      // MTG_FLAG_REFINE: x | residual | delta ([batch][D][n_free] each) | zeros ([batch][D][n_fixed])
  size_t refine_buf_bytes = 0;
  std::vector<LaunchRecord> last;
};

namespace {

int set_err(mtg_context* ctx, int code, const std::string& msg) {
  if (ctx) ctx->last_error = msg;
  return code;
}
#define MTG_HIP_TRY(ctx, expr)                                                                 \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess)                                                                      \
      return set_err(ctx, MTG_ERR_DEVICE, std::string(#expr) + ": " + hipGetErrorString(e_));  \
  } while (0)

int ensure_buffer(mtg_context* ctx, double** buf, size_t* cur, size_t need) {
  if (*cur >= need) return MTG_OK;
  if (*buf) MTG_HIP_TRY(ctx, hipFree(*buf));
  *buf = nullptr;
  *cur = 0;
  MTG_HIP_TRY(ctx, hipMalloc((void**)buf, need));
  *cur = need;
  return MTG_OK;
}

void fill_common(const mtg_plan* p, MtgParams& P, int64_t batch, const mtg_layout* L) {
  std::memset(&P, 0, sizeof(P));
  P.ts_b = L->times_stride_b; P.ts_k = L->times_stride_k;
  P.fs_b = L->fixed_stride_b; P.fs_d = L->fixed_stride_d; P.fs_c = L->fixed_stride_c;
  P.ps_b = L->free_stride_b; P.ps_d = L->free_stride_d; P.ps_c = L->free_stride_c;
  P.status = p->ctx->d_status;
  P.vmask = p->d_tables;
  P.offF = p->d_tables + (p->K + 1);
  P.offP = p->d_tables + (p->K + 1) + (p->K + 2);
  P.B = batch;
  P.K = p->K;
  P.Dtot = p->D;
  P.deriv = p->deriv;
  // host copies of the table offsets (same values as the __constant__ ones)
  static const int ainv_off[7] = {0, 0, 2, 10, 28, 60, 110};
  P.ainvoff = ainv_off[p->H];
  int off = 0;
  for (int n = 2; n < p->N; n += 2) off += (n / 2) * n * n;
  P.h1off = off + p->deriv * p->N * p->N;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
extern "C" {

const char* mtg_status_string(int status) {
  switch (status) {
    case MTG_OK: return "ok";
    case MTG_ERR_INVALID_ARGUMENT: return "invalid argument";
    case MTG_ERR_BAD_SEGMENT_TIME: return "segment times need to be greater than zero";
    case MTG_ERR_SINGULAR: return "non-positive pivot: free-constraint system is rank deficient";
    case MTG_ERR_DEVICE: return "HIP runtime error";
    case MTG_ERR_NO_DEVICE: return "no usable HIP device";
    case MTG_ERR_UNSUPPORTED: return "unsupported configuration";
  }
  return "unknown status";
}

const char* mtg_last_error_string(const mtg_context* ctx) { return ctx ? ctx->last_error.c_str() : ""; }

int mtg_context_create(int device, void* stream, mtg_context** out) {
  if (!out) return MTG_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return MTG_ERR_NO_DEVICE;
  mtg_context* ctx = new (std::nothrow) mtg_context();
  if (!ctx) return MTG_ERR_DEVICE;
  ctx->device = device;
  if (hipSetDevice(device) != hipSuccess) { delete ctx; return MTG_ERR_NO_DEVICE; }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) == hipSuccess) ctx->n_cu = prop.multiProcessorCount;
  if (stream) {
    ctx->stream = (hipStream_t)stream;
  } else {
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return MTG_ERR_DEVICE; }
    ctx->own_stream = true;
  }
  if (hipMalloc((void**)&ctx->d_status, sizeof(int)) != hipSuccess ||
      hipHostMalloc((void**)&ctx->h_status, sizeof(int), hipHostMallocDefault) != hipSuccess ||
      hipMemsetAsync(ctx->d_status, 0, sizeof(int), ctx->stream) != hipSuccess) {
    if (ctx->d_status) hipFree(ctx->d_status);
    if (ctx->h_status) hipHostFree(ctx->h_status);
    if (ctx->own_stream) hipStreamDestroy(ctx->stream);
    delete ctx;
    return MTG_ERR_DEVICE;
  }
  *ctx->h_status = 0;
  *out = ctx;
  return MTG_OK;
}

// include/mtg_hip_lab.h: measurement knobs by name (no environment reads inside the library)
int mtg_context_set_option(mtg_context* ctx, const char* name, int value) {
  if (!ctx || !name) return MTG_ERR_INVALID_ARGUMENT;
  std::lock_guard<std::mutex> lock(ctx->mu);
  const std::string n(name);
  if (n == "force_dg") ctx->knob_force_dg = value;
  else if (n == "prefer_rolled") ctx->knob_prefer_rolled = value != 0;
  else if (n == "no_dimlane") ctx->knob_no_dimlane = value != 0;
  else if (n == "no_slab") ctx->knob_no_slab = value != 0;
  else if (n == "no_queue") ctx->knob_no_queue = value != 0;
  else if (n == "no_slab_extra") ctx->knob_no_slab_extra = value != 0;
  else if (n == "no_dl_extra") ctx->knob_no_dl_extra = value != 0;
  else if (n == "no_balance") ctx->knob_no_balance = value != 0;
  else if (n == "dl_rt") ctx->knob_dl_rt = value;
  else if (n == "dl_grid_per_cu") ctx->knob_dl_grid_per_cu = std::max(1, value);
  else if (n == "dl_any_sched_rr") ctx->knob_dl_any_rr = value != 0;
  else if (n == "slab_policy") ctx->knob_slab_policy = value < 0 ? -1 : (value ? 1 : 0);
  else if (n == "rolled_wg_per_cu") ctx->rolled_wg_per_cu = std::max(1, value);
  else if (n == "dl_max_units") ctx->dl_max_units_per_cu = value;
  else if (n == "sample_generic") ctx->knob_sample_generic = value != 0;
  else if (n == "coop") ctx->knob_coop = value;
  else if (n == "extrema_split") ctx->knob_extrema_split = value;
  else return set_err(ctx, MTG_ERR_INVALID_ARGUMENT, "unknown option: " + n);
  return MTG_OK;
}

// (mtg_sample.hip, mtg_extrema.hip)
bool mtg_context_sample_generic(const mtg_context* ctx) { return ctx && ctx->knob_sample_generic; }
int mtg_context_extrema_split(const mtg_context* ctx) { return ctx ? ctx->knob_extrema_split : -1; }

int mtg_context_destroy(mtg_context* ctx) {
  if (!ctx) return MTG_OK;
  hipSetDevice(ctx->device);
  hipStreamSynchronize(ctx->stream);
  if (ctx->d_status) hipFree(ctx->d_status);
  if (ctx->h_status) hipHostFree(ctx->h_status);
  if (ctx->h_bounce) hipHostFree(ctx->h_bounce);
  if (ctx->own_stream) hipStreamDestroy(ctx->stream);
  for (hipStream_t q : ctx->side_streams) { hipStreamSynchronize(q); hipStreamDestroy(q); }
  for (hipEvent_t e : ctx->join_events) hipEventDestroy(e);
  if (ctx->fork_event) hipEventDestroy(ctx->fork_event);
  for (auto& sc : ctx->dl_any_schedules) { if (sc.d_units) hipFree(sc.d_units); if (sc.d_wg_begin) hipFree(sc.d_wg_begin); }
  if (ctx->dl_any_ws) hipFree(ctx->dl_any_ws);
  for (auto& pb : ctx->dl_any_item_pool) hipFree(pb.first);
  delete ctx;
  return MTG_OK;
}

// used by mtg_workload.hip: a plan's context, shape and device-resident mask table
int mtg_plan_context_tables(const mtg_plan* plan, mtg_context** ctx, int* n_coeffs, int* dimension, int* n_segments,
                            const int** device_masks) {
  if (!plan || !ctx || !n_coeffs || !dimension || !n_segments || !device_masks) return MTG_ERR_INVALID_ARGUMENT;
  *ctx = plan->ctx; *n_coeffs = plan->N; *dimension = plan->D; *n_segments = plan->K;
  *device_masks = plan->d_tables;     // [K + 1] fixed masks, then the offset tables
  return MTG_OK;
}

// used by the other translation units of the library (mtg_sample.hip): the context's stream and device
int mtg_context_stream_device(mtg_context* ctx, void** stream, int* device) {
  if (!ctx || !stream || !device) return MTG_ERR_INVALID_ARGUMENT;
  *stream = (void*)ctx->stream;
  *device = ctx->device;
  return MTG_OK;
}

// ---- device memory helpers: host code above the C ABI never sees HIP headers ---------------------------------
int mtg_device_malloc(mtg_context* ctx, size_t bytes, void** device_ptr) {
  if (!ctx || !device_ptr) return MTG_ERR_INVALID_ARGUMENT;
  *device_ptr = nullptr;
  if (bytes == 0) return MTG_OK;
  std::lock_guard<std::mutex> lock(ctx->mu);
  MTG_HIP_TRY(ctx, hipSetDevice(ctx->device));
  MTG_HIP_TRY(ctx, hipMalloc(device_ptr, bytes));
  return MTG_OK;
}

int mtg_device_free(mtg_context* ctx, void* device_ptr) {
  if (!ctx) return MTG_ERR_INVALID_ARGUMENT;
  if (!device_ptr) return MTG_OK;
  std::lock_guard<std::mutex> lock(ctx->mu);
  MTG_HIP_TRY(ctx, hipSetDevice(ctx->device));
  MTG_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));   // nothing queued on the context may still use it
  MTG_HIP_TRY(ctx, hipFree(device_ptr));
  return MTG_OK;
}

int mtg_copy_to_device(mtg_context* ctx, void* dst_device, const void* src_host, size_t bytes) {
  if (!ctx || (bytes && (!dst_device || !src_host))) return MTG_ERR_INVALID_ARGUMENT;
  if (bytes == 0) return MTG_OK;
  std::lock_guard<std::mutex> lock(ctx->mu);
  MTG_HIP_TRY(ctx, hipSetDevice(ctx->device));
  MTG_HIP_TRY(ctx, hipMemcpyAsync(dst_device, src_host, bytes, hipMemcpyHostToDevice, ctx->stream));
  MTG_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));   // the host buffer may be reused on return
  return MTG_OK;
}

int mtg_copy_to_host(mtg_context* ctx, void* dst_host, const void* src_device, size_t bytes) {
  if (!ctx || (bytes && (!dst_host || !src_device))) return MTG_ERR_INVALID_ARGUMENT;
  if (bytes == 0) return MTG_OK;
  std::lock_guard<std::mutex> lock(ctx->mu);
  MTG_HIP_TRY(ctx, hipSetDevice(ctx->device));
  MTG_HIP_TRY(ctx, hipMemcpyAsync(dst_host, src_device, bytes, hipMemcpyDeviceToHost, ctx->stream));
  MTG_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return MTG_OK;
}

static int status_code(mtg_context* ctx, int st) {
  if (st & MTG_FLAG_BAD_TIME) return set_err(ctx, MTG_ERR_BAD_SEGMENT_TIME, mtg_status_string(MTG_ERR_BAD_SEGMENT_TIME));
  if (st & MTG_FLAG_SINGULAR) return set_err(ctx, MTG_ERR_SINGULAR, mtg_status_string(MTG_ERR_SINGULAR));
  return MTG_OK;
}

// The device status word is fetched on EVERY sync (one 4-byte copy): kernels replayed from a captured hipGraph never
// pass through the library, so no host-side bookkeeping can know whether flags were raised since the last sync.
int mtg_context_sync(mtg_context* ctx) {
  if (!ctx) return MTG_ERR_INVALID_ARGUMENT;
  std::lock_guard<std::mutex> lock(ctx->mu);
  MTG_HIP_TRY(ctx, hipSetDevice(ctx->device));
  MTG_HIP_TRY(ctx, hipMemcpyAsync(ctx->h_status, ctx->d_status, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  MTG_HIP_TRY(ctx, hipMemsetAsync(ctx->d_status, 0, sizeof(int), ctx->stream));
  MTG_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return status_code(ctx, *ctx->h_status);
}

// STRUCTURAL rank deficiency of R_PP.  The cost 0.5 d^T R d = sum over segments of the integral of (p^(d))^2 (LIN:124-140) vanishes
// exactly on the trajectories whose every segment is a polynomial of degree < d; interior vertices share all h >= d + 1
// derivative slots (LIN:199-205), so those are ONE polynomial of degree < d over the whole trajectory, and the null space of R_PP
// is the subspace of it on which every FIXED slot vanishes: the functionals  p -> p^(q)(t_v)  for the fixed slots (v, q), q < d.
// Its dimension d - rank(those functionals on P_(d-1)) depends on the constraint PATTERN only (Hermite data at distinct
// instants are always independent; Birkhoff-type patterns -- a derivative fixed without the lower ones -- generically), not on
// the batch's segment times: it is computed here, once per plan, at two sets of generic vertex instants.
// Why not in the kernels: the reference decides "rank-deficient" with a rank-revealing QR (LIN:365-367); an LDL^T sweep sees a
// zero pivot as round-off x the conditioning of everything eliminated before it -- on chains of free vertices that is anything
// between 1e-12 and 1e0 of the diagonal, of either sign (tests/test_pivot_threshold.py), overlapping the legitimate pivots of
// regular ill-conditioned problems (1e-7 of the diagonal).  No pivot threshold separates the two; the structure does.
// pins (optional): null_dim free slots (vertex, derivative), lowest vertices / derivatives first, whose functionals complete the
// fixed ones to a basis of P_(d-1)'s dual -- fixing them (to zero) makes the system regular without changing the minimum cost
// (any minimiser differs from one that satisfies them by an element of the null space).
static int structural_null_dim(int H, int K, int d, const std::vector<int>& mask, std::vector<std::pair<int, int>>* pins = nullptr) {
  if (pins) pins->clear();
  if (d <= 0) return 0;
  int best_rank = 0;
  for (int trial = 0; trial < 2 && best_rank < d; ++trial) {
    // generic vertex instants in [0, 1]: increments from a fixed irrational rotation (the two trials share no ratio)
    std::vector<long double> tv((size_t)K + 1, 0.0L);
    for (int v = 1; v <= K; ++v) {
      const long double u = (v + 1) * (trial == 0 ? 0.6180339887498948482L : 0.4142135623730950488L);
      tv[v] = tv[v - 1] + (0.35L + (u - (long long)u)) / (long double)K;
    }
    auto functional = [&](int v, int q, std::vector<long double>& row) {      // p -> p^(q)(t_v) on the monomials 1, t, ..., t^(d-1)
      row.assign((size_t)d, 0.0L);
      for (int m = q; m < d; ++m) {
        long double c = 1.0L;
        for (int i = 0; i < q; ++i) c *= (long double)(m - i);
        for (int i = 0; i < m - q; ++i) c *= tv[v];
        row[m] = c;
      }
    };
    // incremental echelon basis: basis[i] has its pivot (largest entry at insertion) in column piv[i]
    std::vector<std::vector<long double>> basis;
    std::vector<int> piv;
    auto add_if_independent = [&](std::vector<long double> row) -> bool {
      long double scale = 0.0L;
      for (long double x : row) scale = std::max(scale, std::fabs(x));
      if (scale == 0.0L) return false;
      for (size_t i = 0; i < basis.size(); ++i) {
        const long double f = row[piv[i]] / basis[i][piv[i]];
        if (f != 0.0L) for (int c = 0; c < d; ++c) row[c] -= f * basis[i][c];
        row[piv[i]] = 0.0L;
      }
      int pc = -1;
      long double big = 1e-9L * scale;
      for (int c = 0; c < d; ++c) if (std::fabs(row[c]) > big) { big = std::fabs(row[c]); pc = c; }
      if (pc < 0) return false;
      basis.push_back(row);
      piv.push_back(pc);
      return true;
    };
    std::vector<long double> row;
    for (int v = 0; v <= K; ++v)
      for (int q = 0; q < H && q < d; ++q)
        if ((mask[v] >> q) & 1) { functional(v, q, row); add_if_independent(row); }
    const int rank = (int)basis.size();
    if (rank > best_rank) {
      best_rank = rank;
      if (pins) {
        pins->clear();
        for (int v = 0; v <= K && (int)basis.size() < d; ++v)
          for (int q = 0; q < H && q < d && (int)basis.size() < d; ++q) {
            if ((mask[v] >> q) & 1) continue;
            functional(v, q, row);
            if (add_if_independent(row)) pins->push_back({v, q});
          }
      }
    }
  }
  return d - best_rank;
}

namespace {
// Plans whose free system is structurally rank-deficient: every trajectory of a solve is flagged (context word and, when the
// caller asked for it, the per-trajectory status), whatever the sweep's pivots looked like.
__global__ void mtg_flag_all_kernel(int* status, int* tstatus, long long B, int flag) {
  const long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (b == 0 && status) atomicOr(status, flag);
  if (tstatus && b < B) atomicOr(tstatus + b, flag);
}
// shadow d_fixed [B][D][nfs] from the caller's (any strides): column j <- source column src[j], or 0 for a pinned slot
__global__ void mtg_pin_gather_kernel(const double* __restrict__ src, long long fs_b, long long fs_d, long long fs_c, const int* __restrict__ map,
                                      double* __restrict__ dst, long long B, int D, int nfs) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * D * nfs) return;
  const int j = (int)(i % nfs);
  const int dm = (int)((i / nfs) % D);
  const long long b = i / ((long long)nfs * D);
  const int c = map[j];
  dst[i] = c >= 0 ? src[b * fs_b + dm * fs_d + c * fs_c] : 0.0;
}
// the same into a canonical SoA destination [D][nfs][Bs] (b fastest; Bs: the row stride) -- the asynchronous shadow solves of the
// queue / merged entries keep the caller's layout KIND so that the shadow takes the same launch forms
__global__ void mtg_pin_gather_soa_kernel(const double* __restrict__ src, long long fs_b, long long fs_d, long long fs_c, const int* __restrict__ map,
                                          double* __restrict__ dst, long long B, long long Bs, int D, int nfs) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * D * nfs) return;
  const long long b = i % B;
  const int j = (int)((i / B) % nfs);
  const int dm = (int)(i / (B * nfs));
  const int c = map[j];
  dst[((long long)dm * nfs + j) * Bs + b] = c >= 0 ? src[b * fs_b + dm * fs_d + c * fs_c] : 0.0;
}
// the caller's d_free (any strides) from the shadow's [B][D][nps]: free column j <- shadow column map[j], or 0 for a pinned slot
__global__ void mtg_pin_scatter_kernel(const double* __restrict__ src, const int* __restrict__ map, double* __restrict__ dst, long long ps_b,
                                       long long ps_d, long long ps_c, long long B, int D, int np, int nps) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * D * np) return;
  const int j = (int)(i % np);
  const int dm = (int)((i / np) % D);
  const long long b = i / ((long long)np * D);
  const int c = map[j];
  dst[b * ps_b + dm * ps_d + j * ps_c] = c >= 0 ? src[(b * D + dm) * (long long)nps + c] : 0.0;
}
}  // namespace
static void flag_structurally_singular(const mtg_plan* p, hipStream_t st, int* status, int* tstatus, int64_t batch) {
  if (p->null_dim <= 0 || p->n_free == 0) return;
  const int64_t n = tstatus ? batch : 1;
  hipLaunchKernelGGL(mtg_flag_all_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, status, tstatus, (long long)batch, (int)MTG_FLAG_SINGULAR);
}

// MTG_FLAG_BASIC_SOLUTION in the asynchronous batched entries (mtg_solve_linear_sequence*, mtg_multi_*): a structurally
// rank-deficient plan is replaced by its SHADOW (a regular plan like any other); what the shadow needs that the caller does not
// hold is its d_fixed -- the caller's columns plus zeros at the pinned slots -- gathered on the device into `dst` in the caller's
// layout KIND (SoA stays SoA: same launch forms).  SL: the caller's layout with the fixed-value strides of that buffer.
// Nothing synchronises: the per-trajectory host fall-back of the synchronous entries (a trajectory on which the shadow's own
// factorisation breaks down) does not exist here -- such a trajectory stays flagged in the context's status word.
static int64_t padded16(int64_t batch);
static size_t shadow_fixed_elems(const mtg_plan* p, int64_t batch) { return (size_t)padded16(batch) * p->D * std::max(p->shadow->n_fixed, 1); }
static void shadow_gather_async(const mtg_plan* p, int64_t batch, const mtg_layout* L, const double* d_fixed, double* dst, mtg_layout* SL, hipStream_t st) {
  const int Dd = p->D, nfs = p->shadow->n_fixed;
  const bool soa = L->fixed_stride_b == 1 && L->times_stride_b == 1 && L->times_stride_k >= batch &&
                   L->times_stride_k <= padded16(batch);                     // canonical / padded SoA inputs
  *SL = *L;
  const long long n = (long long)batch * Dd * nfs;
  if (soa) {
    const long long Bs = L->times_stride_k;                                 // the caller's row stride (batch, or its padded value)
    SL->fixed_stride_b = 1; SL->fixed_stride_c = Bs; SL->fixed_stride_d = (int64_t)nfs * Bs;
    if (n > 0)
      hipLaunchKernelGGL(mtg_pin_gather_soa_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_fixed, (long long)L->fixed_stride_b,
                         (long long)L->fixed_stride_d, (long long)L->fixed_stride_c, (const int*)p->d_shadow_maps, dst, (long long)batch, Bs, Dd, nfs);
  } else {
    SL->fixed_stride_b = (int64_t)Dd * nfs; SL->fixed_stride_d = nfs; SL->fixed_stride_c = 1;
    if (n > 0)
      hipLaunchKernelGGL(mtg_pin_gather_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d_fixed, (long long)L->fixed_stride_b,
                         (long long)L->fixed_stride_d, (long long)L->fixed_stride_c, (const int*)p->d_shadow_maps, dst, (long long)batch, Dd, nfs);
  }
}

int mtg_plan_rank_deficiency(const mtg_plan* p) { return p ? p->null_dim : MTG_ERR_INVALID_ARGUMENT; }

int mtg_structural_rank_deficiency(int32_t n_coeffs, int32_t n_segments, int32_t derivative_to_optimize, const uint32_t* fixed_mask) {
  if (n_coeffs < 2 || n_coeffs > MTG_MAX_N || (n_coeffs & 1) || n_segments < 1 || !fixed_mask || derivative_to_optimize < 0 ||
      derivative_to_optimize > n_coeffs / 2 - 1)
    return MTG_ERR_INVALID_ARGUMENT;
  const int H = n_coeffs / 2;
  std::vector<int> mask((size_t)n_segments + 1);
  int n_free = 0;
  for (int v = 0; v <= n_segments; ++v) {
    mask[v] = (int)(fixed_mask[v] & (uint32_t)((1 << H) - 1));
    n_free += H - __builtin_popcount((unsigned)mask[v]);
  }
  return n_free > 0 ? structural_null_dim(H, n_segments, derivative_to_optimize, mask) : 0;
}

int mtg_plan_create(mtg_context* ctx, const mtg_plan_desc* desc, mtg_plan** out) {
  if (!ctx || !desc || !out || !desc->fixed_mask) return MTG_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  const int N = desc->n_coeffs, D = desc->dimension, K = desc->n_segments, d = desc->derivative_to_optimize;
  if (N < 2 || N > MTG_MAX_N || (N & 1)) return set_err(ctx, MTG_ERR_INVALID_ARGUMENT, "n_coeffs must be even in [2,12]");
  if (D < 1 || K < 1) return set_err(ctx, MTG_ERR_INVALID_ARGUMENT, "dimension and n_segments must be >= 1");
  if (d < 0 || d > N / 2 - 1) return set_err(ctx, MTG_ERR_INVALID_ARGUMENT, "derivative_to_optimize out of range");
  mtg_plan* p = new (std::nothrow) mtg_plan();
  if (!p) return MTG_ERR_DEVICE;
  p->ctx = ctx; p->N = N; p->H = N / 2; p->D = D; p->K = K; p->deriv = d;
  const int full = (1 << p->H) - 1;
  p->mask.resize(K + 1);
  p->offF.assign(K + 2, 0);
  p->offP.assign(K + 2, 0);
  for (int v = 0; v <= K; ++v) {
    p->mask[v] = (int)(desc->fixed_mask[v] & (uint32_t)full);
    const int nf = __builtin_popcount((unsigned)p->mask[v]);
    p->offF[v + 1] = p->offF[v] + nf;
    p->offP[v + 1] = p->offP[v] + (p->H - nf);
  }
  p->n_fixed = p->offF[K + 1];
  p->n_free = p->offP[K + 1];
  std::vector<std::pair<int, int>> pins;
  p->null_dim = p->n_free > 0 ? structural_null_dim(p->H, K, d, p->mask, &pins) : 0;
  p->fast = mtg_find_static(p->H, D, K, d, p->mask.data());
  for (int dg = 1; dg < D && !p->fast_split; ++dg) {
    if (D % dg == 0) p->fast_split = mtg_find_static(p->H, dg, K, d, p->mask.data());
  }
  p->dimlane = mtg_find_dimlane(p->H, D, K, d, p->mask.data());
  p->dimlane_rt = mtg_find_dimlane_rt(p->H, D, K, d, p->mask.data());
  std::vector<int> tab;
  tab.insert(tab.end(), p->mask.begin(), p->mask.end());
  tab.insert(tab.end(), p->offF.begin(), p->offF.end());
  tab.insert(tab.end(), p->offP.begin(), p->offP.end());
  if (hipSetDevice(ctx->device) != hipSuccess || hipMalloc((void**)&p->d_tables, tab.size() * sizeof(int)) != hipSuccess ||
      hipMemcpy(p->d_tables, tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) {
    delete p;
    return set_err(ctx, MTG_ERR_DEVICE, "plan table upload failed");
  }
  if (p->null_dim > 0 && (int)pins.size() == p->null_dim) {
    // the shadow plan: the same problem with the pinned slots fixed (a regular pattern: its own null_dim is 0, no recursion)
    std::vector<uint32_t> smask((size_t)K + 1);
    for (int v = 0; v <= K; ++v) smask[v] = (uint32_t)p->mask[v];
    for (auto& pq : pins) smask[pq.first] |= 1u << pq.second;
    mtg_plan_desc sd{N, D, K, d, smask.data()};
    const int rs = mtg_plan_create(ctx, &sd, &p->shadow);
    if (rs != MTG_OK || p->shadow->null_dim != 0) {
      if (p->shadow) mtg_plan_destroy(p->shadow);
      p->shadow = nullptr;       // (the per-trajectory pivoted QR of mtg_basic.cpp stays as the way to a basic solution)
    } else {
      int src = 0, sfree = 0;
      for (int v = 0; v <= K; ++v)
        for (int q = 0; q < p->H; ++q) {
          const bool fixed = (p->mask[v] >> q) & 1, sfixed = (smask[v] >> q) & 1;
          if (sfixed) p->shadow_fixed_src.push_back(fixed ? src : -1);
          if (fixed) ++src;
          if (!fixed) p->free_in_shadow.push_back(sfixed ? -1 : sfree);
          if (!sfixed) ++sfree;
        }
      std::vector<int> maps(p->shadow_fixed_src);
      maps.insert(maps.end(), p->free_in_shadow.begin(), p->free_in_shadow.end());
      if (hipMalloc((void**)&p->d_shadow_maps, maps.size() * sizeof(int)) != hipSuccess ||
          hipMemcpy(p->d_shadow_maps, maps.data(), maps.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) {
        mtg_plan_destroy(p);
        return set_err(ctx, MTG_ERR_DEVICE, "plan table upload failed");
      }
    }
  }
  *out = p;
  return MTG_OK;
}

int mtg_plan_destroy(mtg_plan* p) {
  if (!p) return MTG_OK;
  hipSetDevice(p->ctx->device);
  hipStreamSynchronize(p->ctx->stream);
  if (p->d_tables) hipFree(p->d_tables);
  if (p->ws) hipFree(p->ws);
  if (p->pert_cost) hipFree(p->pert_cost);
  if (p->stage) hipFree(p->stage);
  if (p->basic_status) hipFree(p->basic_status);
  if (p->d_shadow_maps) hipFree(p->d_shadow_maps);
  if (p->shadow_buf) hipFree(p->shadow_buf);
  if (p->refine_buf) hipFree(p->refine_buf);
  if (p->shadow) mtg_plan_destroy(p->shadow);
  delete p;
  return MTG_OK;
}

mtg_context* mtg_plan_context(const mtg_plan* p) { return p ? p->ctx : nullptr; }

int mtg_plan_get_shape(const mtg_plan* p, int32_t* n_coeffs, int32_t* dimension, int32_t* n_segments, int32_t* derivative_to_optimize) {
  if (!p) return MTG_ERR_INVALID_ARGUMENT;
  if (n_coeffs) *n_coeffs = p->N;
  if (dimension) *dimension = p->D;
  if (n_segments) *n_segments = p->K;
  if (derivative_to_optimize) *derivative_to_optimize = p->deriv;
  return MTG_OK;
}

int mtg_plan_get_info(const mtg_plan* p, mtg_plan_info* out) {
  if (!p || !out) return MTG_ERR_INVALID_ARGUMENT;
  out->n_all = p->N * p->K;
  out->n_fixed = p->n_fixed;
  out->n_free = p->n_free;
  out->kernel_variant = p->fast ? (p->fast->k < 0 ? 3 : 1) : (p->fast_split ? 2 : 0);
  out->algorithmic_bytes_per_trajectory = 8ll * (p->K + (int64_t)p->D * p->n_fixed + (int64_t)p->K * p->D * p->N);
  return MTG_OK;
}

void mtg_layout_aos(const mtg_plan* p, int64_t batch, mtg_layout* L) {
  (void)batch;
  L->times_stride_b = p->K; L->times_stride_k = 1;
  L->fixed_stride_b = (int64_t)p->D * p->n_fixed; L->fixed_stride_d = p->n_fixed; L->fixed_stride_c = 1;
  L->free_stride_b = (int64_t)p->D * p->n_free; L->free_stride_d = p->n_free; L->free_stride_c = 1;
}

void mtg_layout_soa_padded(const mtg_plan* p, int64_t batch, mtg_layout* L) {
  const int64_t bs = (batch + 15) & ~(int64_t)15;
  L->times_stride_b = 1; L->times_stride_k = bs;
  L->fixed_stride_b = 1; L->fixed_stride_d = (int64_t)p->n_fixed * bs; L->fixed_stride_c = bs;
  L->free_stride_b = 1; L->free_stride_d = (int64_t)p->n_free * bs; L->free_stride_c = bs;
}

void mtg_layout_soa(const mtg_plan* p, int64_t batch, mtg_layout* L) {
  L->times_stride_b = 1; L->times_stride_k = batch;
  L->fixed_stride_b = 1; L->fixed_stride_d = (int64_t)p->n_fixed * batch; L->fixed_stride_c = batch;
  L->free_stride_b = 1; L->free_stride_d = (int64_t)p->n_free * batch; L->free_stride_c = batch;
}

static int64_t span(int64_t batch, int64_t sb, int64_t n1, int64_t s1, int64_t n2, int64_t s2) {
  return (batch - 1) * sb + (n1 - 1) * s1 + (n2 - 1) * s2 + 1;
}

// The dimension-in-lane form applies to: a plan with a matching variant, canonical SoA inputs (times[K][B],
// d_fixed[D][n_fixed][B]), coefficient output only, sizes whose 32-bit byte offsets cannot overflow.  Chosen by default
// while the launch is at most a few workgroups per CU (measured cross-over against the fused form: DESIGN.md section 4);
// MTG_FLAG_DIMLANE forces it, MTG_FLAG_FUSED_DIMS / MTG_FLAG_SPLIT_DIMS / MTG_FLAG_GENERIC_KERNEL exclude it.
// Persistent grid over equal-cost tiles: with `cap` resident workgroups the launch takes ceil(ntiles / cap) rounds whatever
// the grid; spreading the tiles evenly over those rounds (grid = ceil(ntiles / rounds) <= cap) keeps the rounds, and
// every round runs with fewer workgroups competing for HBM (a 20 x 10k queue: 3140 tiles = 6.13 rounds of 512 -> 7 rounds
// of 449 instead of 6 full rounds and a 13 %-full one).  MTG_NO_BALANCE: the full grid (A/B runs).
static int balanced_grid(const mtg_context* ctx, int ntiles, int cap) {
  if (ntiles <= cap || ctx->knob_no_balance) return std::min(ntiles, cap);
  const int rounds = (ntiles + cap - 1) / cap;
  return (ntiles + rounds - 1) / rounds;
}

// default range of the dimension-in-lane form (mtg_dimlane_variants.inc): LO * CUs <= workgroups <= HI * CUs / 2
// (HI = 0: no upper limit; HI = 3 = 1.5 workgroups per CU, the measured cross-over against the slab-output fused kernel)
static bool dimlane_is_default(const mtg_plan* p, const MtgDimlaneEntry* dl, int64_t trajectories) {
  const int64_t units = ((trajectories + dl->tpw - 1) / dl->tpw + dl->np - 1) / dl->np;
  const int64_t cus = p->ctx->n_cu;
  const int hi = p->ctx->dl_max_units_per_cu >= 0 ? 2 * p->ctx->dl_max_units_per_cu : dl->hi_per_cu;
  if (units < (int64_t)dl->lo_per_cu * cus) return false;
  return hi == 0 || 2 * units <= (int64_t)hi * cus;
}

// Input layout kinds the dimension-in-lane kernels read: 0 = canonical SoA (times[K][B], d_fixed[D][n_fixed][B]), 1 = canonical
// AoS (times[B][K], d_fixed[B][D][n_fixed]: the reference's natural order), 2 = SoA with the row stride padded to the next
// multiple of 16 trajectories (mtg_layout_soa_padded; the static variants' single and queue launches only), -1 = anything else
// (fused / generic kernels).
static int64_t padded16(int64_t batch) { return (batch + 15) & ~(int64_t)15; }
static int dimlane_input_kind(const mtg_plan* p, const mtg_layout* L, int64_t batch) {
  if (L->times_stride_b == 1 && L->times_stride_k == batch && L->fixed_stride_b == 1 && L->fixed_stride_c == batch &&
      L->fixed_stride_d == (int64_t)p->n_fixed * batch)
    return 0;
  const int64_t bs = padded16(batch);
  if (bs != batch && L->times_stride_b == 1 && L->times_stride_k == bs && L->fixed_stride_b == 1 && L->fixed_stride_c == bs &&
      L->fixed_stride_d == (int64_t)p->n_fixed * bs)
    return 2;
  if (L->times_stride_b == p->K && L->times_stride_k == 1 && L->fixed_stride_b == (int64_t)p->D * p->n_fixed &&
      L->fixed_stride_c == 1 && L->fixed_stride_d == p->n_fixed)
    return 1;
  return -1;
}

static const MtgDimlaneEntry* pick_dimlane(const mtg_plan* p, int64_t batch, const mtg_layout* L, const MtgParams& P,
                                           uint32_t flags, bool cost_only) {
  const MtgDimlaneEntry* dl = p->dimlane;
  if (!dl || p->ctx->knob_no_dimlane || cost_only) return nullptr;
  // extra outputs (cost / d_P): the main-table variants have a kernel for them (round 3; MTG_NO_DL_EXTRA: as before, through
  // the fused kernels)
  if ((P.dfree || P.cost) && (!dl->launch_extra || p->ctx->knob_no_dl_extra)) return nullptr;
  // (N = 12 / K = 32 with extra outputs spills 544 registers: 110 vs 128 us at 10k, but 542 vs 443 us at 50k against the
  // rolled fused kernel -- profiles/r03r_k32_extra_outputs.jsonl)
  if ((P.dfree || P.cost) && dl->h == 6 && dl->k == 32 && batch > 20000 && !(flags & MTG_FLAG_DIMLANE)) return nullptr;
  if (flags & (MTG_FLAG_GENERIC_KERNEL | MTG_FLAG_FUSED_DIMS | MTG_FLAG_SPLIT_DIMS)) return nullptr;
  if (dimlane_input_kind(p, L, batch) < 0) return nullptr;
  if (padded16(batch) * 8 * (int64_t)std::max(p->K, p->n_fixed * p->D) >= (1ll << 32)) return nullptr;
  if (flags & MTG_FLAG_DIMLANE) return dl;
  return dimlane_is_default(p, dl, batch) ? dl : nullptr;
}

// The run-time-K dimension-in-lane body (mtg_dimlane_rt.h): same eligibility as the static dimension-in-lane variants
// (canonical SoA inputs, coefficient output only); taken where the plan has no static variant (K > 32, ...) -- or always /
// never with MTG_DL_RT=1 / 0.
static const MtgDimlaneRtEntry* pick_dimlane_rt(const mtg_plan* p, int64_t batch, const mtg_layout* L, const MtgParams& P,
                                                uint32_t flags, bool cost_only) {
  const MtgDimlaneRtEntry* rt = p->dimlane_rt;
  if (!rt || p->ctx->knob_dl_rt == 0 || p->ctx->knob_no_dimlane || cost_only || P.dfree || P.cost) return nullptr;
  if (p->dimlane && p->ctx->knob_dl_rt != 1) return nullptr;
  if (flags & (MTG_FLAG_GENERIC_KERNEL | MTG_FLAG_FUSED_DIMS | MTG_FLAG_SPLIT_DIMS)) return nullptr;
  { const int kind = dimlane_input_kind(p, L, batch); if (kind < 0 || kind > 1) return nullptr; }
  // the body keeps the batch size and its tile count in 32-bit integers (its input addresses are 64-bit, unlike the static
  // variants' 32-bit byte offsets)
  if (batch + rt->tpw >= (1ll << 31)) return nullptr;
  return rt;
}

// Variant choice of the fused / dimension-split forms: specialised kernels when the plan matches one; with few tiles (small
// batch) the dimension-split form puts Dtot/D times as many (lighter, 2-per-SIMD) waves on the machine.  nullptr: generic.
static const MtgSlabEntry* pick_slab(const mtg_plan* p, const MtgStaticEntry* var);
static const MtgStaticEntry* pick_static(const mtg_plan* p, int ntiles, uint32_t flags, bool coeffs_only) {
  const mtg_context* ctx = p->ctx;
  const MtgStaticEntry* var = nullptr;
  if (flags & MTG_FLAG_GENERIC_KERNEL) return nullptr;
  // Coefficient output only and a slab-output fused kernel for the shape: never the dimension-split form by default.  Its
  // 80-byte pieces complete sectors from different workgroups (1.21x write amplification, read-modify-write at the memory
  // side once the output is not cache-resident): with rotating buffers 15.4 / 25.7 us at B = 10k / 20k against 10.4 / 14.3 us
  // (profiles/r02_sweep_forms.txt).  Round 2 still sent SoA batches between 1.5 workgroups per CU of the dimension-in-lane
  // form (~16k) and 4 x CUs split-form workgroups (~21.8k) to the split form (found with mtg_plan_launch_form).
  if (coeffs_only && !(flags & MTG_FLAG_SPLIT_DIMS) && ctx->knob_force_dg <= 0 && !ctx->knob_prefer_rolled && pick_slab(p, p->fast))
    return p->fast;
  // Dimension-split form while ALL its workgroups (tiles x dimension groups) are resident at once at <= 2 waves per
  // SIMD (4 x CUs workgroups); beyond that it runs in rounds and the fused form -- no repeated factorisation, one
  // round up to 2 x CUs tiles -- wins (measured, N = 10 / K = 8 / D = 3: B = 20k 15.0 vs 15.4 us, B = 30k 25.5 vs
  // 18.3 us, B = 60k 44.1 vs 34.3 us).  Plans whose fused kernel spills ("heavy") keep the split form longer.
  bool auto_split = ntiles < 4 * ctx->n_cu;
  if (p->fast && p->fast_split && !p->fast->heavy)
    auto_split = (long long)ntiles * (p->D / p->fast_split->d) <= 4ll * ctx->n_cu;
  const bool want_split = (flags & MTG_FLAG_SPLIT_DIMS) || (!(flags & MTG_FLAG_FUSED_DIMS) && auto_split);
  var = (want_split && p->fast_split) ? p->fast_split : (p->fast ? p->fast : p->fast_split);
  if (var && var->heavy && !want_split) {   // large launch, spilling static kernel: the rolled form is faster
    const MtgStaticEntry* v = mtg_find_static(p->H, p->D, p->K, p->deriv, p->mask.data(), true);
    if (v) var = v;
  }
  if (ctx->knob_prefer_rolled) {
    const MtgStaticEntry* v = mtg_find_static(p->H, p->D, p->K, p->deriv, p->mask.data(), true);
    if (v) var = v;
  }
  if (ctx->knob_force_dg > 0) {
    const int dg = ctx->knob_force_dg;
    if (p->D % dg == 0) {
      const MtgStaticEntry* v = mtg_find_static(p->H, dg, p->K, p->deriv, p->mask.data());
      if (v) var = v;
    }
  }
  return var;
}

// fused static form, coefficient output only: the slab-output kernel (whole-sector stores, mtg_solve_slab_kernel)
static const MtgSlabEntry* pick_slab(const mtg_plan* p, const MtgStaticEntry* var) {
  if (!var || var->k <= 0 || var->d != p->D || p->ctx->knob_no_slab) return nullptr;
  return mtg_find_slab(p->H, p->D, p->K, p->deriv, p->mask.data());
}

struct PerturbedTimes { double h, lower_bound; };   // mtg_mellinger_cost_gradient: (K + 1) virtual problems per trajectory

// ---- one solve / update call: stage (host pointers) -> pick_form -> launch_<form> -> fetch (host pointers) ------------------
enum class SolveForm { kUpdate, kCoop, kDimlaneRt, kDimlane, kFused };   // kFused: slab-output / static / rolled / generic kernels

struct SolveCall {                  // everything a launcher needs, assembled once by solve_impl
  mtg_plan* p;
  int64_t batch;
  const mtg_layout* L;
  uint32_t flags;
  bool cost_only, wc;               // wc: extra outputs (cost and / or d_P) requested
  const PerturbedTimes* pert;
  hipStream_t st;
  MtgParams P;                      // device pointers, strides, tables
  int ntiles;                       // 64-trajectory tiles (x (K + 1) virtual problems for perturbed-time launches)
  int32_t* dts;                     // per-trajectory status on the device (or null)
  const MtgDimlaneRtEntry* rt = nullptr;    // set by pick_form for the form it chose
  const MtgDimlaneEntry* dl = nullptr;
};

static int workspace(mtg_plan* p, size_t need, double** out) {
  mtg_context* ctx = p->ctx;
  if (p->user_ws) {
    if (p->user_ws_bytes < need) return set_err(ctx, MTG_ERR_INVALID_ARGUMENT, "user workspace too small");
    *out = p->user_ws;
    return MTG_OK;
  }
  const int rc = ensure_buffer(ctx, &p->ws, &p->ws_bytes, need);
  if (rc == MTG_OK) *out = p->ws;
  return rc;
}

// The row-cooperative form (mtg_coop.hip): standard shapes (end vertices fully fixed, position-only interior vertices), D = 3,
// coefficient output only, non-negative strides, 32-bit input / output offsets, the step storage of the chain within one CU's LDS.
// Default range = where it was measured faster than the lane-per-half forms (profiles/r04d_coop_vs_default.jsonl: 0.54-0.87 of
// their time): LONG chains in launches of at most one 2-wave workgroup per CU -- a chain step costs ~2.4x the lane-instructions
// here, but its latency is ~1.2 us against 1.8-2.6 us, and four trajectory-halves share a wavefront instead of 21-64.
//   N = 12: K >= 16 (K >= 32: up to two workgroups per CU);  N = 10: K >= 64;  N = 8: K >= 80;
//   workgroups (four trajectories each) <= CUs x that factor, and all of them resident at once (LDS).
// MTG_FLAG_COOPERATIVE forces the form wherever it is eligible; option "coop" = 0 takes it out of the default choice, 1 makes
// it the choice wherever eligible.
static bool coop_eligible(const mtg_plan* p, int64_t batch, const mtg_layout* L, const MtgParams& P, bool cost_only) {
  if (p->D != 3 || p->H < 4 || p->H > 6 || p->K < 2 || cost_only || P.dfree || P.cost || P.pert_on) return false;
  const int full = (1 << p->H) - 1;
  if (p->mask[0] != full || p->mask[p->K] != full) return false;
  for (int v = 1; v < p->K; ++v) if (p->mask[v] != 1) return false;
  const size_t lds = mtg_coop_lds_bytes(p->H, p->D, p->K);
  if (lds == 0 || lds > 160 * 1024) return false;
  if (L->times_stride_b < 0 || L->times_stride_k < 0 || L->fixed_stride_b < 0 || L->fixed_stride_d < 0 || L->fixed_stride_c < 0) return false;
  const int64_t tmax = (batch - 1) * L->times_stride_b + (int64_t)(p->K - 1) * L->times_stride_k;
  const int64_t fmax = (batch - 1) * L->fixed_stride_b + (int64_t)(p->D - 1) * L->fixed_stride_d + (int64_t)(p->n_fixed - 1) * L->fixed_stride_c;
  return tmax * 8 < (1ll << 32) && fmax * 8 < (1ll << 32) && batch * p->K * p->D * p->N * 8 < (1ll << 32);
}
static bool pick_coop(const mtg_plan* p, int64_t batch, const mtg_layout* L, const MtgParams& P, uint32_t flags, bool cost_only) {
  if (!coop_eligible(p, batch, L, P, cost_only)) return false;
  if (flags & (MTG_FLAG_GENERIC_KERNEL | MTG_FLAG_FUSED_DIMS | MTG_FLAG_SPLIT_DIMS | MTG_FLAG_DIMLANE)) return false;
  if ((flags & MTG_FLAG_COOPERATIVE) || p->ctx->knob_coop == 1) return true;
  if (p->ctx->knob_coop == 0 || p->ctx->knob_dl_rt == 1) return false;   // (option "dl_rt" = 1 asks for the run-time-K body)
  const int kmin = p->H == 6 ? 16 : (p->H == 5 ? 64 : 80);
  if (p->K < kmin) return false;
  const int64_t wgs = (batch + 3) / 4;
  const int64_t resident = (int64_t)(160 * 1024 / mtg_coop_lds_bytes(p->H, p->D, p->K));   // workgroups per CU the LDS holds
  const int64_t per_cu = std::min<int64_t>((p->H == 6 && p->K >= 32) ? 2 : 1, resident);
  return wgs <= per_cu * p->ctx->n_cu;
}

// Which form a call takes: the run-time-K dimension-in-lane body where the plan has no static variant, the static
// dimension-in-lane variants inside their default range (or forced), else the fused family.  Same order as
// mtg_plan_launch_form reports.
static SolveForm pick_form(SolveCall& c, bool update_only) {
  if (update_only) return SolveForm::kUpdate;
  if (pick_coop(c.p, c.batch, c.L, c.P, c.flags, c.cost_only)) return SolveForm::kCoop;
  if ((c.rt = pick_dimlane_rt(c.p, c.batch, c.L, c.P, c.flags, c.cost_only))) return SolveForm::kDimlaneRt;
  if ((c.dl = pick_dimlane(c.p, c.batch, c.L, c.P, c.flags, c.cost_only))) return SolveForm::kDimlane;
  return SolveForm::kFused;
}

// setFreeConstraints path (LIN:500-508): compile-time-mask ("rolled") update kernel when the plan has one (all D dimensions in
// one launch), else generic
static int launch_update(SolveCall& c) {
  mtg_plan* p = c.p;
  mtg_context* ctx = p->ctx;
  const MtgStaticEntry* uv = (c.flags & MTG_FLAG_GENERIC_KERNEL) ? nullptr
                             : mtg_find_static(p->H, p->D, p->K, p->deriv, p->mask.data(), true);
  for (int dim0 = 0; dim0 < p->D; dim0 += 4) {
    const int dc = uv ? p->D : std::min(4, p->D - dim0);
    UpdateFn fn = uv ? uv->upd[c.wc ? 1 : 0] : mtg_pick_generic_update(p->H, dc, c.wc);
    if (!fn) return set_err(ctx, MTG_ERR_UNSUPPORTED, "no update kernel");
    MtgParams Q = c.P;
    Q.dim0 = dim0;
    const int grid = std::min(c.ntiles, ctx->n_cu * 16);
    size_t lds = (size_t)64 * ((size_t)(dc * p->N / 2) | 1) * 2 * sizeof(double);
    // whole-sector output (mtg_update_slab_kernel) for the rolled form; "no_slab" keeps the per-segment staging
    const int phase = ((size_t)p->K * p->D * p->N * 8) % 64 != 0 ? 1 : 0;
    if (uv && !ctx->knob_no_slab && uv->upd_slab[c.wc ? 1 : 0][phase] && uv->upd_slab_lds <= 64 * 1024) {
      fn = uv->upd_slab[c.wc ? 1 : 0][phase];
      lds = uv->upd_slab_lds;
    }
    hipLaunchKernelGGL(fn, dim3(grid), dim3(kWave), lds, c.st, Q, c.ntiles);
    if (uv) break;
  }
  return MTG_OK;
}

// row-cooperative form (mtg_coop.hip): one 2-wave workgroup per four trajectories, step storage in LDS
static int launch_coop(SolveCall& c) {
  mtg_plan* p = c.p;
  const MtgParams& P = c.P;
  const int rc = mtg_coop_launch((void*)c.st, p->H, p->D, p->K, p->deriv, c.batch, P.times, P.ts_b, P.ts_k, P.dfix, P.fs_b, P.fs_d, P.fs_c,
                                 P.coeffs, P.status, c.dts);
  if (rc != 0) return set_err(p->ctx, rc == 1 ? MTG_ERR_UNSUPPORTED : MTG_ERR_DEVICE, "row-cooperative launch failed");
  LaunchRecord r;
  r.valid = true; r.params = P; r.coop = true;
  p->last.push_back(r);
  return MTG_OK;
}

// run-time-K dimension-in-lane body: persistent 2-wave workgroups, two per CU; the head steps beyond the register tail and the
// LDS step area go through a lane-coalesced workspace
static int launch_dimlane_rt(SolveCall& c) {
  mtg_plan* p = c.p;
  mtg_context* ctx = p->ctx;
  const MtgDimlaneRtEntry* rt = c.rt;
  const int nt = (int)((c.batch + rt->tpw - 1) / rt->tpw);
  const int grid = std::min(nt, ctx->n_cu * 2);
  const int kc_max = (p->K + 1) / 2;
  double* rt_ws = nullptr;
  if (kc_max - 1 - rt->r_steps - rt->l_steps > 0) {      // head steps beyond the register tail and the LDS step area
    const size_t need = rt->step_bytes_per_lane * (size_t)(kc_max - 1 - rt->r_steps) * (size_t)grid * 2 * kWave;   // slots j - 1 of all head steps
    const int rc = workspace(p, need, &rt_ws);
    if (rc != MTG_OK) return rc;
  }
  const int aos = dimlane_input_kind(p, c.L, c.batch);
  if (rt->launch((void*)c.st, grid, c.P.times, c.P.dfix, c.P.coeffs, c.P.status, c.dts, (int)c.batch, p->K, nt, rt_ws, aos) != 0)
    return set_err(ctx, MTG_ERR_DEVICE, "run-time-K dimension-in-lane launch set-up failed");
  LaunchRecord r;
  r.valid = true; r.params = c.P; r.ntiles = nt; r.grid = grid; r.rt = rt; r.dl_ws = rt_ws; r.dl_aos = aos;
  p->last.push_back(r);
  return MTG_OK;
}

// dimension-in-lane form (mtg_dimlane.h): all dimensions of a trajectory in one wave, whole-sector coefficient stores
static int launch_dimlane(SolveCall& c) {
  mtg_plan* p = c.p;
  mtg_context* ctx = p->ctx;
  const MtgDimlaneEntry* dl = c.dl;
  const MtgParams& P = c.P;
  const int nt = (int)((c.batch + dl->tpw - 1) / dl->tpw);
  const int units = (nt + dl->np - 1) / dl->np;
  int grid = std::min(units, ctx->n_cu * ctx->knob_dl_grid_per_cu);
  double* dl_ws = nullptr;
  if (dl->ws_per_lane) {
    // long chains: part of the back-substitution data goes through the workspace; persistent workgroups only (two
    // 2-wave workgroups per CU, one wave per SIMD), so the workspace stays small enough to live in the Infinity Cache
    grid = std::min(units, ctx->n_cu * 4 / (2 * dl->np));
    const int rc = workspace(p, dl->ws_per_lane * (size_t)grid * dl->np * 2 * kWave, &dl_ws);
    if (rc != MTG_OK) return rc;
  }
  const int aos = dimlane_input_kind(p, c.L, c.batch);
  const int lrc = (P.dfree || P.cost)
                      ? dl->launch_extra((void*)c.st, grid, P.times, P.dfix, P.coeffs, P.status, c.dts, (int)c.batch, nt, dl_ws, aos,
                                         P.dfree, P.cost, P.ps_b, P.ps_d, P.ps_c)
                      : dl->launch((void*)c.st, grid, P.times, P.dfix, P.coeffs, P.status, c.dts, (int)c.batch, nt, dl_ws, aos);
  if (lrc != 0) return set_err(ctx, MTG_ERR_DEVICE, "dimension-in-lane launch set-up failed");
  LaunchRecord r;
  r.valid = true; r.params = P; r.ntiles = nt; r.grid = grid; r.dl = dl; r.dl_ws = dl_ws; r.dl_aos = aos;
  p->last.push_back(r);
  return MTG_OK;
}

// the fused family: slab-output kernel (whole-sector stores) where the shape has one, else static (fused / dimension-split),
// rolled (run-time K, workspace) or generic (run-time masks) kernels
static int launch_fused(SolveCall& c) {
  mtg_plan* p = c.p;
  mtg_context* ctx = p->ctx;
  const bool wc = c.wc, cost_only = c.cost_only;
  const int ntiles = c.ntiles;
  hipStream_t st = c.st;
  // variant choice: specialised kernels when the plan matches one; with few tiles (small batch) the
  // dimension-split form puts Dtot/D times as many (lighter, 2-per-SIMD) waves on the machine.
  const MtgStaticEntry* var = pick_static(p, ntiles, c.flags, !wc && !cost_only && !c.pert);
  const int vm = (p->K + 1) / 2;
  const int fm = p->H - __builtin_popcount((unsigned)p->mask[vm]);
  for (int dim0 = 0; dim0 < p->D; dim0 += 4) {
    const int dc = var ? var->d : std::min(4, p->D - dim0);
    const int ngroups = var ? p->D / var->d : 1;
    MtgParams Q = c.P;
    Q.dim0 = dim0;
    SolveFn fn;
    int grid;
    const bool needs_ws = !var || var->k < 0;   // generic and rolled kernels stream (G, g) through the workspace
    // fused static form, coefficient output only: the slab-output kernel (whole-sector stores, mtg_solve_slab_kernel)
    const MtgSlabEntry* slab = nullptr;
    if (!cost_only && !c.pert) slab = pick_slab(p, var);
    if (slab && wc && (!slab->extra || ctx->knob_no_slab_extra)) slab = nullptr;
    if (slab) {
      // (extra outputs -- cost / d_P -- through the slab-output kernel as well: the older fused kernel's 240-byte pieces
      // complete most sectors from two store instructions, 80-83 us at B = 125k with rotating buffers)
      const int pol = ctx->knob_slab_policy >= 0 ? ctx->knob_slab_policy : 1;
      SolveFn sfn = wc ? slab->extra : slab->fn[pol];
      bool& attr_set = wc ? p->slab_extra_attr_set : p->slab_attr_set[pol];
      const int sgrid = balanced_grid(ctx, ntiles, ctx->n_cu * 2);   // 63.5 KB of LDS per workgroup: two per CU, one wave per SIMD
      if (!attr_set) {
        MTG_HIP_TRY(ctx, hipFuncSetAttribute((const void*)sfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)slab->lds));
        attr_set = true;
      }
      hipLaunchKernelGGL(sfn, dim3(sgrid), dim3(kBlock), slab->lds, st, Q, ntiles);
      LaunchRecord r;
      r.valid = true; r.fn = sfn; r.params = Q; r.ntiles = ntiles; r.grid = sgrid; r.gridy = 1; r.lds = slab->lds;
      p->last.push_back(r);
      break;
    }
    if (var) {
      Q.ws = p->user_ws;   // unused by the static kernels (measurement builds park timestamps here)
      // few tiles => every workgroup finishes at about the same time: write-through stores avoid the serial
      // end-of-kernel L2 write-back; many tiles => plain write-back stores are faster
      const bool write_through = (long long)ntiles * ngroups <= 4ll * ctx->n_cu;
      fn = cost_only ? var->fn[4] : var->fn[(wc ? 1 : 0) + (write_through ? 2 : 0)];
      grid = std::min(ntiles, std::max(1, ctx->n_cu * 8 / ngroups));
    } else {
      fn = mtg_pick_generic_solve(p->H, dc, cost_only ? 2 : (wc ? 1 : 0));
      if (!fn) return set_err(ctx, MTG_ERR_UNSUPPORTED, "no generic kernel");
      grid = std::min(ntiles, ctx->n_cu * 4);
    }
    if (needs_ws) {
      if (var) grid = std::min(ntiles, std::max(1, ctx->n_cu * ctx->rolled_wg_per_cu / ngroups));
      const int kc = (p->K + 1) / 2;
      const size_t E = (size_t)p->H * p->H + (size_t)dc * p->H;
      const int rc = workspace(p, (size_t)kc * E * (size_t)grid * ngroups * kBlock * sizeof(double), &Q.ws);
      if (rc != MTG_OK) return rc;
      Q.ws_stride = (long long)grid * ngroups * kBlock;
    }
    // LDS: two coefficient staging buffers (64 rows x odd number of 16-byte chunks) + two exchange buffers
    const size_t stage = (size_t)64 * ((size_t)(dc * p->N / 2) | 1) * 2 * sizeof(double);
    const size_t lds = 2 * stage + (size_t)2 * (fm * (fm + 1) / 2 + dc * fm) * kWave * sizeof(double);
    hipLaunchKernelGGL(fn, dim3(grid, ngroups), dim3(kBlock), lds, st, Q, ntiles);
    LaunchRecord r;
    r.valid = true; r.fn = fn; r.params = Q; r.ntiles = ntiles; r.grid = grid; r.gridy = ngroups; r.lds = lds;
    p->last.push_back(r);
    if (var) break;
  }
  return MTG_OK;
}

// Host-pointer calls: inputs staged into the plan's device area ([times | d_fixed | d_free | coeffs | cost | per-trajectory
// status | status word]; small calls through ONE page-locked bounce buffer = one H2D DMA, the kernel, one D2H DMA), outputs and
// the call's OWN status word fetched back synchronously.
struct HostStaging {
  bool bounce = false;
  int64_t n_times = 0, n_fix = 0, n_fre = 0, n_coef = 0, n_ts = 0;
  int* status_dev = nullptr;        // this call's own status word (a host-pointer call reports its status itself; it must neither
                                    // collect nor clear the flags earlier asynchronous launches left in the context's word)
  const double *dt = nullptr, *dfx = nullptr;
  double *dco = nullptr, *dfr = nullptr, *dcs = nullptr;
  int32_t* dts = nullptr;
};

static int stage_host_inputs(mtg_plan* p, int64_t batch, const mtg_layout* L, const double* times, const double* d_fixed,
                             const double* d_free, bool want_cost, bool want_ts, bool update_only, hipStream_t st, HostStaging& h) {
  mtg_context* ctx = p->ctx;
  constexpr size_t kBounceLimit = 1u << 20;
  h.n_times = span(batch, L->times_stride_b, p->K, L->times_stride_k, 1, 0);
  h.n_fix = p->n_fixed ? span(batch, L->fixed_stride_b, p->D, L->fixed_stride_d, p->n_fixed, L->fixed_stride_c) : 0;
  h.n_fre = p->n_free ? span(batch, L->free_stride_b, p->D, L->free_stride_d, p->n_free, L->free_stride_c) : 0;
  h.n_coef = batch * p->K * p->D * p->N;
  h.n_ts = want_ts ? (batch + 1) / 2 : 0;   // doubles that hold `batch` int32
  const size_t need = (size_t)(h.n_times + h.n_fix + h.n_fre + h.n_coef + batch + h.n_ts + 1) * sizeof(double);
  int rc = ensure_buffer(ctx, &p->stage, &p->stage_bytes, need);
  if (rc != MTG_OK) return rc;
  double* s = p->stage;
  double* s_t = s; s += h.n_times;
  double* s_f = s; s += h.n_fix;
  double* s_p = s; s += h.n_fre;
  double* s_c = s; s += h.n_coef;
  double* s_j = s; s += batch;
  if (want_ts) h.dts = reinterpret_cast<int32_t*>(s);
  s += h.n_ts;
  h.status_dev = reinterpret_cast<int*>(s);
  const size_t n_in = (size_t)(h.n_times + h.n_fix + (update_only ? h.n_fre : 0));
  h.bounce = need <= kBounceLimit;
  if (h.bounce) {
    if (ctx->h_bounce_bytes < need) {
      if (ctx->h_bounce) hipHostFree(ctx->h_bounce);
      ctx->h_bounce = nullptr;
      ctx->h_bounce_bytes = 0;
      MTG_HIP_TRY(ctx, hipHostMalloc((void**)&ctx->h_bounce, kBounceLimit, hipHostMallocDefault));
      ctx->h_bounce_bytes = kBounceLimit;
    }
    std::memcpy(ctx->h_bounce, times, h.n_times * sizeof(double));
    if (h.n_fix) std::memcpy(ctx->h_bounce + h.n_times, d_fixed, h.n_fix * sizeof(double));
    if (update_only && h.n_fre) std::memcpy(ctx->h_bounce + h.n_times + h.n_fix, d_free, h.n_fre * sizeof(double));
    MTG_HIP_TRY(ctx, hipMemcpyAsync(s_t, ctx->h_bounce, n_in * sizeof(double), hipMemcpyHostToDevice, st));
  } else {
    MTG_HIP_TRY(ctx, hipMemcpyAsync(s_t, times, h.n_times * sizeof(double), hipMemcpyHostToDevice, st));
    if (h.n_fix) MTG_HIP_TRY(ctx, hipMemcpyAsync(s_f, d_fixed, h.n_fix * sizeof(double), hipMemcpyHostToDevice, st));
    if (update_only && h.n_fre) MTG_HIP_TRY(ctx, hipMemcpyAsync(s_p, d_free, h.n_fre * sizeof(double), hipMemcpyHostToDevice, st));
  }
  h.dt = s_t; h.dfx = s_f; h.dco = s_c;
  h.dfr = (d_free && h.n_fre) ? s_p : nullptr;
  h.dcs = want_cost ? s_j : nullptr;
  return MTG_OK;
}

static int fetch_host_outputs(mtg_plan* p, int64_t batch, double* coeffs, double* d_free, double* cost, int32_t* traj_status,
                              bool update_only, hipStream_t st, const HostStaging& h, int* host_status) {
  mtg_context* ctx = p->ctx;
  const int64_t n_times = h.n_times, n_fix = h.n_fix, n_fre = h.n_fre, n_coef = h.n_coef, n_ts = h.n_ts;
  if (h.bounce) {
    // [d_free | coeffs | cost | per-trajectory status | status word] sit back to back in the device staging area: one D2H DMA
    double* s_p = p->stage + n_times + n_fix;
    const size_t n_out = (size_t)(n_fre + n_coef + batch + n_ts + 1);
    MTG_HIP_TRY(ctx, hipMemcpyAsync(ctx->h_bounce, s_p, n_out * sizeof(double), hipMemcpyDeviceToHost, st));
    MTG_HIP_TRY(ctx, hipStreamSynchronize(st));
    std::memcpy(host_status, ctx->h_bounce + n_fre + n_coef + batch + n_ts, sizeof(int));
    if (!update_only && d_free && n_fre) std::memcpy(d_free, ctx->h_bounce, n_fre * sizeof(double));
    std::memcpy(coeffs, ctx->h_bounce + n_fre, n_coef * sizeof(double));
    if (cost) std::memcpy(cost, ctx->h_bounce + n_fre + n_coef, batch * sizeof(double));
    if (traj_status) std::memcpy(traj_status, ctx->h_bounce + n_fre + n_coef + batch, batch * sizeof(int32_t));
  } else {
    MTG_HIP_TRY(ctx, hipMemcpyAsync(coeffs, h.dco, n_coef * sizeof(double), hipMemcpyDeviceToHost, st));
    if (!update_only && d_free && n_fre) MTG_HIP_TRY(ctx, hipMemcpyAsync(d_free, h.dfr, n_fre * sizeof(double), hipMemcpyDeviceToHost, st));
    if (cost) MTG_HIP_TRY(ctx, hipMemcpyAsync(cost, h.dcs, batch * sizeof(double), hipMemcpyDeviceToHost, st));
    if (traj_status) MTG_HIP_TRY(ctx, hipMemcpyAsync(traj_status, h.dts, batch * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    // (h_status is the context's pinned word; the context lock is held, and mtg_context_sync overwrites it under the same lock)
    MTG_HIP_TRY(ctx, hipMemcpyAsync(ctx->h_status, h.status_dev, sizeof(int), hipMemcpyDeviceToHost, st));
    MTG_HIP_TRY(ctx, hipStreamSynchronize(st));
    *host_status = *ctx->h_status;
  }
  return MTG_OK;
}

// latency path of single-trajectory host callers: the lane code's host build on the calling thread (mtg_host.cpp); no device
// work, no lock
static int solve_on_host_backend(mtg_plan* p, int64_t batch, const mtg_layout* L, const double* times, const double* d_fixed,
                                 double* coeffs, double* d_free, double* cost, bool update_only, int32_t* traj_status) {
  MtgParams P;
  fill_common(p, P, batch, L);
  P.times = times; P.dfix = d_fixed; P.coeffs = coeffs; P.dfree = p->n_free ? d_free : nullptr; P.cost = cost;
  int st_word = 0;
  P.status = &st_word;
  P.tstatus = traj_status;
  P.vmask = p->mask.data(); P.offF = p->offF.data(); P.offP = p->offP.data();
  if (traj_status) std::memset(traj_status, 0, (size_t)batch * sizeof(int32_t));
  if (mtg_host_run(P, p->H, update_only) != 0) return MTG_ERR_UNSUPPORTED;
  if (!update_only && p->null_dim > 0 && p->n_free > 0) {      // structurally rank-deficient free system: every trajectory
    st_word |= MTG_FLAG_SINGULAR;
    if (traj_status) for (int64_t b = 0; b < batch; ++b) traj_status[b] |= MTG_FLAG_SINGULAR;
  }
  if (st_word & MTG_FLAG_BAD_TIME) return MTG_ERR_BAD_SEGMENT_TIME;
  if (st_word & MTG_FLAG_SINGULAR) return MTG_ERR_SINGULAR;
  return MTG_OK;
}

static int solve_impl(mtg_plan* p, int64_t batch, const mtg_layout* L, const double* times, const double* d_fixed,
                      double* coeffs, double* d_free, double* cost, uint32_t flags, bool update_only,
                      int32_t* traj_status = nullptr, const PerturbedTimes* pert = nullptr,
                      hipStream_t on_stream = nullptr, int* own_status_dev = nullptr, const double* explicit_rhs = nullptr) {
  // explicit_rhs (MTG_FLAG_REFINE's correction solve; with MTG_FLAG_GENERIC_KERNEL): [batch][D][n_free], added to the right-hand side
  // own_status_dev: a device status word of the CALL (zeroed here) instead of the context's -- flags of earlier asynchronous
  // launches stay where the next mtg_context_sync finds them
  const bool cost_only = !update_only && (flags & MTG_FLAG_COST_ONLY) != 0;
  if (!p || !L || !times || (!coeffs && !cost_only) || batch < 0) return MTG_ERR_INVALID_ARGUMENT;
  if (cost_only && (!cost || (flags & MTG_FLAG_HOST_POINTERS))) return MTG_ERR_INVALID_ARGUMENT;
  mtg_context* ctx = p->ctx;
  if (!cost_only && !(flags & MTG_FLAG_HOST_POINTERS) && (reinterpret_cast<uintptr_t>(coeffs) & 15))
    return set_err(ctx, MTG_ERR_INVALID_ARGUMENT, "coeffs must be 16-byte aligned");
  if (p->n_fixed > 0 && !d_fixed) return set_err(ctx, MTG_ERR_INVALID_ARGUMENT, "d_fixed is null");
  if (update_only && p->n_free > 0 && !d_free) return set_err(ctx, MTG_ERR_INVALID_ARGUMENT, "d_free is null");
  if (batch == 0) return MTG_OK;
  const bool host = (flags & MTG_FLAG_HOST_POINTERS) != 0;
  if (host && (flags & MTG_FLAG_HOST_BACKEND) && batch <= MTG_HOST_BACKEND_MAX_BATCH)
    return solve_on_host_backend(p, batch, L, times, d_fixed, coeffs, d_free, cost, update_only, traj_status);
  std::lock_guard<std::mutex> lock(ctx->mu);
  MTG_HIP_TRY(ctx, hipSetDevice(ctx->device));

  SolveCall c;
  c.p = p; c.batch = batch; c.L = L; c.flags = flags; c.cost_only = cost_only; c.pert = pert;
  c.st = on_stream ? on_stream : ctx->stream;   // on_stream: a side stream of a concurrent mixed request
  c.dts = traj_status;
  const double* dt = times; const double* dfx = d_fixed; double* dco = coeffs; double* dfr = d_free; double* dcs = cost;
  HostStaging h;
  if (host) {
    const int rc = stage_host_inputs(p, batch, L, times, d_fixed, d_free, cost != nullptr, traj_status != nullptr, update_only, c.st, h);
    if (rc != MTG_OK) return rc;
    dt = h.dt; dfx = h.dfx; dco = h.dco; dfr = h.dfr; dcs = h.dcs; c.dts = h.dts;
  }
  if (dcs) MTG_HIP_TRY(ctx, hipMemsetAsync(dcs, 0, (pert ? (size_t)(p->K + 1) : (size_t)1) * batch * sizeof(double), c.st));
  if (c.dts) MTG_HIP_TRY(ctx, hipMemsetAsync(c.dts, 0, batch * sizeof(int32_t), c.st));
  if (h.status_dev) MTG_HIP_TRY(ctx, hipMemsetAsync(h.status_dev, 0, sizeof(double), c.st));
  if (own_status_dev && !host) MTG_HIP_TRY(ctx, hipMemsetAsync(own_status_dev, 0, sizeof(double), c.st));

  fill_common(p, c.P, batch, L);
  if (h.status_dev) c.P.status = h.status_dev;
  else if (own_status_dev) c.P.status = own_status_dev;
  c.P.times = dt; c.P.dfix = dfx; c.P.coeffs = dco; c.P.dfree = (p->n_free ? dfr : nullptr); c.P.cost = dcs;
  c.P.tstatus = c.dts;
  if (explicit_rhs) { c.P.rhs = explicit_rhs; c.P.rh_b = (long long)p->D * p->n_free; c.P.rh_d = p->n_free; c.P.rh_c = 1; }
  c.wc = dcs != nullptr || (!update_only && c.P.dfree != nullptr);
  c.ntiles = (int)((batch + kWave - 1) / kWave);
  if (pert) {   // cost-only launch over (K + 1) x batch virtual problems; cost = [(K + 1)][batch]
    c.P.pert_on = 1; c.P.pert_seg = -1; c.P.pert_tpv = c.ntiles;
    c.P.pert_h = pert->h; c.P.pert_corr = pert->h / (p->K - 1.0); c.P.pert_lo = pert->lower_bound;
    c.ntiles *= p->K + 1;
  }
  p->last.clear();

  int rc = MTG_OK;
  switch (pick_form(c, update_only)) {
    case SolveForm::kUpdate: rc = launch_update(c); break;
    case SolveForm::kCoop: rc = launch_coop(c); break;
    case SolveForm::kDimlaneRt: rc = launch_dimlane_rt(c); break;
    case SolveForm::kDimlane: rc = launch_dimlane(c); break;
    case SolveForm::kFused: rc = launch_fused(c); break;
  }
  if (rc != MTG_OK) return rc;
  if (!update_only) flag_structurally_singular(p, c.st, c.P.status, c.P.tstatus, batch);
  MTG_HIP_TRY(ctx, hipGetLastError());

  if (host) {
    // A host-pointer call synchronises anyway: the status word (and the per-trajectory status) come back with the
    // results, and the call itself returns MTG_ERR_BAD_SEGMENT_TIME / MTG_ERR_SINGULAR -- no mtg_context_sync needed.
    int host_status = 0;
    rc = fetch_host_outputs(p, batch, coeffs, d_free, cost, traj_status, update_only, c.st, h, &host_status);
    if (rc != MTG_OK) return rc;
    return status_code(ctx, host_status);
  }
  return MTG_OK;
}

int mtg_plan_launch_form(const mtg_plan* p, int64_t batch, const mtg_layout* L, uint32_t flags) {
  if (!p || !L || batch <= 0 || (flags & (MTG_FLAG_HOST_POINTERS | MTG_FLAG_COST_ONLY))) return MTG_ERR_INVALID_ARGUMENT;
  MtgParams P;
  fill_common(p, P, batch, L);   // (no d_free / cost output: coefficient output only ...
  const bool extra = (flags & MTG_FLAG_QUERY_EXTRA_OUTPUTS) != 0;
  static double cost_stands_for_any_extra_output = 0.0;
  if (extra) P.cost = &cost_stands_for_any_extra_output;   // ... unless asked for the form of a call with extra outputs; never dereferenced)
  flags &= ~(uint32_t)MTG_FLAG_QUERY_EXTRA_OUTPUTS;
  if (pick_coop(p, batch, L, P, flags, false)) return 7;
  if (pick_dimlane_rt(p, batch, L, P, flags, false)) return 6;
  if (pick_dimlane(p, batch, L, P, flags, false)) return 5;
  const MtgStaticEntry* var = pick_static(p, (int)((batch + kWave - 1) / kWave), flags, !extra);
  if (!var) return 0;
  if (const MtgSlabEntry* slab = pick_slab(p, var)) {
    if (!extra || (slab->extra && !p->ctx->knob_no_slab_extra)) return 4;
  }
  if (var->k < 0) return 3;
  return var->d == p->D ? 1 : 2;
}

int mtg_plan_set_workspace(mtg_plan* p, void* device_ptr, size_t bytes) {
  if (!p) return MTG_ERR_INVALID_ARGUMENT;
  std::lock_guard<std::mutex> lock(p->ctx->mu);
  p->user_ws = bytes ? static_cast<double*>(device_ptr) : nullptr;
  p->user_ws_bytes = bytes;
  return MTG_OK;
}

int mtg_basic_solution_host(const mtg_plan* p, const double* times, const double* d_fixed, double* d_free, int32_t* rank) {
  if (!p || !times || (p->n_fixed > 0 && !d_fixed) || (p->n_free > 0 && !d_free)) return MTG_ERR_INVALID_ARGUMENT;
  for (int k = 0; k < p->K; ++k)
    if (!(times[k] > 0.0)) return MTG_ERR_BAD_SEGMENT_TIME;
  const int r = mtg_basic_solution_one(p->H, p->K, p->D, p->deriv, p->mask.data(), p->offF.data(), p->offP.data(), times, d_fixed, d_free);
  if (r < 0) return MTG_ERR_UNSUPPORTED;
  if (rank) *rank = r;
  return MTG_OK;
}

// MTG_FLAG_BASIC_SOLUTION: the ordinary solve, then -- synchronously -- the trajectories the LDL^T sweep flagged singular are
// solved again on the host (mtg_basic.cpp: column-pivoted QR of the dense R_PP, LIN:365-378), their coefficients recovered
// with the host build of the update path (LIN:263-283), and their rows of the outputs replaced.
static int solve_with_basic_solution(mtg_plan* p, int64_t batch, const mtg_layout* L, const double* times, const double* d_fixed,
                                     double* coeffs, double* d_free, double* cost, int32_t* traj_status, uint32_t flags) {
  if (!p || !L) return MTG_ERR_INVALID_ARGUMENT;
  if (flags & MTG_FLAG_COST_ONLY) return set_err(p->ctx, MTG_ERR_INVALID_ARGUMENT, "MTG_FLAG_BASIC_SOLUTION needs coefficient output");
  mtg_context* ctx = p->ctx;
  const bool host = (flags & MTG_FLAG_HOST_POINTERS) != 0;
  const uint32_t inner = flags & ~(uint32_t)MTG_FLAG_BASIC_SOLUTION;
  if (batch <= 0) return solve_impl(p, batch, L, times, d_fixed, coeffs, d_free, cost, inner, false, traj_status);
  std::vector<int32_t> ts((size_t)batch, 0);
  int32_t* dev_ts = nullptr;           // device-pointer calls: the per-trajectory status the kernels write
  int rc;
  // A structurally rank-deficient plan is solved through its SHADOW (the same problem with null_dim more slots fixed to zero:
  // a regular system, the LDL^T kernels at full speed and accuracy -- cost within 1e-11 of the reference's on 50-segment
  // chains of free vertices where the dense pivoted QR is at 1e-5): shadow d_fixed gathered from the caller's, coefficients and
  // cost written straight to the caller's buffers, d_free scattered back with zeros at the pinned slots.
  const bool use_shadow = p->shadow != nullptr;
  mtg_plan* q = use_shadow ? p->shadow : p;
  const int Dd = p->D, nfs = q->n_fixed, nps = q->n_free;
  mtg_layout SL = *L;
  if (use_shadow) {
    SL.fixed_stride_b = (int64_t)Dd * nfs; SL.fixed_stride_d = nfs; SL.fixed_stride_c = 1;
    SL.free_stride_b = (int64_t)Dd * nps; SL.free_stride_d = nps; SL.free_stride_c = 1;
  }
  const mtg_layout* QL = use_shadow ? &SL : L;
  if (host) {
    std::vector<double> sfx, sfr;
    const double* q_fixed = d_fixed;
    double* q_free = d_free;
    if (use_shadow) {
      sfx.assign((size_t)batch * Dd * std::max(nfs, 1), 0.0);
      sfr.assign((size_t)batch * Dd * std::max(nps, 1), 0.0);
      for (int64_t b = 0; b < batch; ++b)
        for (int dm = 0; dm < Dd; ++dm)
          for (int j = 0; j < nfs; ++j) {
            const int c = p->shadow_fixed_src[j];
            if (c >= 0) sfx[((size_t)b * Dd + dm) * nfs + j] = d_fixed[b * L->fixed_stride_b + dm * L->fixed_stride_d + c * L->fixed_stride_c];
          }
      q_fixed = sfx.data();
      q_free = d_free ? sfr.data() : nullptr;
    }
    rc = solve_impl(q, batch, QL, times, q_fixed, coeffs, q_free, cost, inner, false, ts.data());
    if (use_shadow && d_free && (rc == MTG_OK || rc == MTG_ERR_SINGULAR || rc == MTG_ERR_BAD_SEGMENT_TIME))
      for (int64_t b = 0; b < batch; ++b)
        for (int dm = 0; dm < Dd; ++dm)
          for (int j = 0; j < p->n_free; ++j) {
            const int c = p->free_in_shadow[j];
            d_free[b * L->free_stride_b + dm * L->free_stride_d + j * L->free_stride_c] = c >= 0 ? sfr[((size_t)b * Dd + dm) * nps + c] : 0.0;
          }
    // (bit 1 of the reported per-trajectory status: WHICH trajectories got a basic solution -- all of a deficient plan)
    if (traj_status) for (int64_t b = 0; b < batch; ++b) traj_status[b] = ts[b] | (use_shadow ? (int32_t)MTG_FLAG_SINGULAR : 0);
    if (rc != MTG_ERR_SINGULAR && rc != MTG_ERR_BAD_SEGMENT_TIME) return rc;
  } else {
    // The call has its OWN device status word and per-trajectory status, in a buffer of the plan: it neither reads nor clears
    // the context's word, so SINGULAR / BAD_TIME flags left by earlier asynchronous launches of this context are still there
    // for the caller's next mtg_context_sync (round 4 went through mtg_context_sync and lost them).
    int* own_word = nullptr;
    const double* q_fixed = d_fixed;
    double* q_free = d_free;
    {
      std::lock_guard<std::mutex> lock(ctx->mu);
      MTG_HIP_TRY(ctx, hipSetDevice(ctx->device));
      const int rb = ensure_buffer(ctx, &p->basic_status, &p->basic_status_bytes, sizeof(double) + (size_t)batch * sizeof(int32_t));
      if (rb != MTG_OK) return rb;
      own_word = reinterpret_cast<int*>(p->basic_status);
      dev_ts = (traj_status && !use_shadow) ? traj_status : reinterpret_cast<int32_t*>(p->basic_status + 1);
      if (use_shadow) {
        const size_t n_fx = (size_t)batch * Dd * std::max(nfs, 1), n_fr = (size_t)batch * Dd * std::max(nps, 1);
        const int rs = ensure_buffer(ctx, &p->shadow_buf, &p->shadow_buf_bytes, (n_fx + n_fr) * sizeof(double));
        if (rs != MTG_OK) return rs;
        double* sfx = p->shadow_buf;
        q_fixed = sfx;
        q_free = d_free ? p->shadow_buf + n_fx : nullptr;     // (d_P only when the caller asked for it)
        const long long n = (long long)batch * Dd * nfs;
        if (n > 0)
          hipLaunchKernelGGL(mtg_pin_gather_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_fixed, (long long)L->fixed_stride_b,
                             (long long)L->fixed_stride_d, (long long)L->fixed_stride_c, (const int*)p->d_shadow_maps, sfx, (long long)batch, Dd, nfs);
      }
    }
    rc = solve_impl(q, batch, QL, times, q_fixed, coeffs, q_free, cost, inner, false, dev_ts, nullptr, nullptr, own_word);
    if (rc != MTG_OK) return rc;
    {
      std::lock_guard<std::mutex> lock(ctx->mu);
      MTG_HIP_TRY(ctx, hipSetDevice(ctx->device));
      if (use_shadow && d_free && p->n_free > 0) {
        const long long n = (long long)batch * Dd * p->n_free;
        hipLaunchKernelGGL(mtg_pin_scatter_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const double*)q_free,
                           (const int*)(p->d_shadow_maps + nfs), d_free, (long long)L->free_stride_b, (long long)L->free_stride_d,
                           (long long)L->free_stride_c, (long long)batch, Dd, p->n_free, nps);
      }
      int word = 0;
      MTG_HIP_TRY(ctx, hipMemcpyAsync(ctx->h_status, own_word, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
      MTG_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
      word = *ctx->h_status;
      if (word != 0) MTG_HIP_TRY(ctx, hipMemcpy(ts.data(), dev_ts, (size_t)batch * sizeof(int32_t), hipMemcpyDeviceToHost));
      if (use_shadow && traj_status) {
        std::vector<int32_t> rep(ts);
        for (auto& x : rep) x |= (int32_t)MTG_FLAG_SINGULAR;
        MTG_HIP_TRY(ctx, hipMemcpy(traj_status, rep.data(), (size_t)batch * sizeof(int32_t), hipMemcpyHostToDevice));
      }
      if (word == 0) return MTG_OK;
    }
  }
  // one flagged trajectory after the other: gather its inputs (any strides), solve, recover, scatter
  const int K = p->K, D = p->D, nf = p->n_fixed, np = p->n_free, N = p->N;
  std::vector<double> t(K), fx((size_t)D * std::max(nf, 1)), fr((size_t)D * std::max(np, 1)), co((size_t)K * D * N);
  double cost1 = 0.0;
  bool any_bad_time = false;
  auto pull = [&](double* dst, const double* src, int64_t stride, int count) -> bool {   // dst[i] = src[i * stride]
    if (host) { for (int i = 0; i < count; ++i) dst[i] = src[(int64_t)i * stride]; return true; }
    return hipMemcpy2D(dst, sizeof(double), src, (size_t)stride * sizeof(double), sizeof(double), (size_t)count, hipMemcpyDeviceToHost) == hipSuccess;
  };
  auto push = [&](double* dst, int64_t stride, const double* src, int count) -> bool {   // dst[i * stride] = src[i]
    if (host) { for (int i = 0; i < count; ++i) dst[(int64_t)i * stride] = src[i]; return true; }
    return hipMemcpy2D(dst, (size_t)stride * sizeof(double), src, sizeof(double), sizeof(double), (size_t)count, hipMemcpyHostToDevice) == hipSuccess;
  };
  std::unique_lock<std::mutex> lock(ctx->mu, std::defer_lock);
  if (!host) { lock.lock(); MTG_HIP_TRY(ctx, hipSetDevice(ctx->device)); }
  for (int64_t b = 0; b < batch; ++b) {
    if (ts[b] & MTG_FLAG_BAD_TIME) { any_bad_time = true; continue; }
    if (!(ts[b] & MTG_FLAG_SINGULAR)) continue;
    bool ok = pull(t.data(), times + b * L->times_stride_b, std::max<int64_t>(L->times_stride_k, 1), K);
    for (int d = 0; d < D && ok && nf > 0; ++d)
      ok = pull(fx.data() + (size_t)d * nf, d_fixed + b * L->fixed_stride_b + d * L->fixed_stride_d, std::max<int64_t>(L->fixed_stride_c, 1), nf);
    if (!ok) return set_err(ctx, MTG_ERR_DEVICE, "basic solution: gathering a flagged trajectory failed");
    if (mtg_basic_solution_one(p->H, K, D, p->deriv, p->mask.data(), p->offF.data(), p->offP.data(), t.data(), fx.data(), fr.data()) < 0)
      return set_err(ctx, MTG_ERR_UNSUPPORTED, "basic solution: unsupported shape");
    // coefficients (and the cost) of this one trajectory: host build of the update path, contiguous AoS scratch
    MtgParams P;
    mtg_layout one;
    mtg_layout_aos(p, 1, &one);
    fill_common(p, P, 1, &one);
    int st_word = 0;
    P.times = t.data(); P.dfix = fx.data(); P.coeffs = co.data(); P.dfree = fr.data(); P.cost = cost ? &cost1 : nullptr;
    P.status = &st_word; P.tstatus = nullptr;
    P.vmask = p->mask.data(); P.offF = p->offF.data(); P.offP = p->offP.data();
    if (mtg_host_run(P, p->H, /*update=*/true) != 0) return set_err(ctx, MTG_ERR_UNSUPPORTED, "basic solution: no host update path");
    ok = push(coeffs + b * (int64_t)K * D * N, 1, co.data(), K * D * N);
    for (int d = 0; d < D && ok && d_free && np > 0; ++d)
      ok = push(d_free + b * L->free_stride_b + d * L->free_stride_d, std::max<int64_t>(L->free_stride_c, 1), fr.data() + (size_t)d * np, np);
    if (ok && cost) ok = push(cost + b, 1, &cost1, 1);
    if (!ok) return set_err(ctx, MTG_ERR_DEVICE, "basic solution: writing a trajectory back failed");
  }
  return any_bad_time ? set_err(ctx, MTG_ERR_BAD_SEGMENT_TIME, mtg_status_string(MTG_ERR_BAD_SEGMENT_TIME)) : MTG_OK;
}

// MTG_FLAG_REFINE (asynchronous, device pointers): x = the ordinary solve's d_P; r = -(R_PP x + R_PF d_F) in double-double
// (mtg_refine.hip); R_PP delta = r by the generic float64 kernel (zero fixed values, r as its explicit right-hand side);
// x += delta; coefficients (and the cost) recovered from x by the update path (LIN:263-283).
static int solve_refined(mtg_plan* p, int64_t batch, const mtg_layout* L, const double* times, const double* d_fixed,
                         double* coeffs, double* d_free, double* cost, int32_t* traj_status, uint32_t flags) {
  if (!p || !L) return MTG_ERR_INVALID_ARGUMENT;
  mtg_context* ctx = p->ctx;
  if (flags & (MTG_FLAG_HOST_POINTERS | MTG_FLAG_COST_ONLY | MTG_FLAG_BASIC_SOLUTION))
    return set_err(ctx, MTG_ERR_INVALID_ARGUMENT, "MTG_FLAG_REFINE: device pointers, coefficient output, not with MTG_FLAG_BASIC_SOLUTION");
  const uint32_t inner = flags & ~(uint32_t)MTG_FLAG_REFINE;
  if (batch <= 0 || p->n_free == 0) return solve_impl(p, batch, L, times, d_fixed, coeffs, d_free, cost, inner, false, traj_status);
  const size_t nfree = (size_t)batch * p->D * p->n_free, nfix = (size_t)batch * p->D * std::max(p->n_fixed, 1);
  double *xbuf, *rbuf, *dbuf, *zbuf;
  {
    std::lock_guard<std::mutex> lock(ctx->mu);
    MTG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int rb = ensure_buffer(ctx, &p->refine_buf, &p->refine_buf_bytes, (3 * nfree + nfix) * sizeof(double));
    if (rb != MTG_OK) return rb;
    xbuf = p->refine_buf; rbuf = xbuf + nfree; dbuf = rbuf + nfree; zbuf = dbuf + nfree;
    MTG_HIP_TRY(ctx, hipMemsetAsync(zbuf, 0, nfix * sizeof(double), ctx->stream));
  }
  // x lives in the caller's d_free when there is one, else in the plan's scratch (contiguous [B][D][n_free])
  mtg_layout XL = *L;
  double* x = d_free;
  if (!x) { x = xbuf; XL.free_stride_b = (int64_t)p->D * p->n_free; XL.free_stride_d = p->n_free; XL.free_stride_c = 1; }
  int rc = solve_impl(p, batch, &XL, times, d_fixed, coeffs, x, nullptr, inner, false, traj_status);
  if (rc != MTG_OK) return rc;
  {
    std::lock_guard<std::mutex> lock(ctx->mu);
    MTG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    MtgParams P;
    fill_common(p, P, batch, &XL);
    if (mtg_refine_residual_launch((void*)ctx->stream, p->H, p->K, p->D, p->deriv, P.h1off, P.vmask, P.offF, P.offP, (long long)batch, times,
                                   P.ts_b, P.ts_k, d_fixed, P.fs_b, P.fs_d, P.fs_c, x, P.ps_b, P.ps_d, P.ps_c, rbuf, p->n_free) != 0)
      return set_err(ctx, MTG_ERR_DEVICE, "MTG_FLAG_REFINE: residual launch failed");
  }
  // the correction solve: zero fixed values (contiguous), the residual as explicit right-hand side, delta contiguous
  mtg_layout CL = *L;
  CL.fixed_stride_b = (int64_t)p->D * p->n_fixed; CL.fixed_stride_d = p->n_fixed; CL.fixed_stride_c = 1;
  CL.free_stride_b = (int64_t)p->D * p->n_free; CL.free_stride_d = p->n_free; CL.free_stride_c = 1;
  const uint32_t generic = (inner & ~(uint32_t)(MTG_FLAG_FUSED_DIMS | MTG_FLAG_SPLIT_DIMS | MTG_FLAG_DIMLANE | MTG_FLAG_COOPERATIVE)) | MTG_FLAG_GENERIC_KERNEL;
  rc = solve_impl(p, batch, &CL, times, zbuf, coeffs, dbuf, nullptr, generic, false, nullptr, nullptr, nullptr, nullptr, rbuf);
  if (rc != MTG_OK) return rc;
  {
    std::lock_guard<std::mutex> lock(ctx->mu);
    MTG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (mtg_refine_axpy_launch((void*)ctx->stream, x, XL.free_stride_b, XL.free_stride_d, XL.free_stride_c, dbuf, (long long)batch, p->D, p->n_free) != 0)
      return set_err(ctx, MTG_ERR_DEVICE, "MTG_FLAG_REFINE: update launch failed");
  }
  return solve_impl(p, batch, &XL, times, d_fixed, coeffs, x, cost, inner, true);
}

// include/mtg_hip_lab.h: the double-double residual alone (what the tests compare with the residual formed at 50 digits)
extern "C" int mtg_lab_refine_residual(mtg_plan* p, int64_t batch, const mtg_layout* L, const double* times, const double* d_fixed,
                                       const double* d_free, double* rhs_out) {
  if (!p || !L || !times || !d_free || !rhs_out || batch < 0 || (p->n_fixed > 0 && !d_fixed)) return MTG_ERR_INVALID_ARGUMENT;
  if (batch == 0 || p->n_free == 0) return MTG_OK;
  mtg_context* ctx = p->ctx;
  std::lock_guard<std::mutex> lock(ctx->mu);
  MTG_HIP_TRY(ctx, hipSetDevice(ctx->device));
  MtgParams P;
  fill_common(p, P, batch, L);
  if (mtg_refine_residual_launch((void*)ctx->stream, p->H, p->K, p->D, p->deriv, P.h1off, P.vmask, P.offF, P.offP, (long long)batch, times,
                                 P.ts_b, P.ts_k, d_fixed, P.fs_b, P.fs_d, P.fs_c, d_free, P.ps_b, P.ps_d, P.ps_c, rhs_out, p->n_free) != 0)
    return set_err(ctx, MTG_ERR_DEVICE, "residual launch failed");
  return MTG_OK;
}

int mtg_solve_linear(mtg_plan* plan, int64_t batch, const mtg_layout* layout, const double* times,
                     const double* d_fixed, double* coeffs, double* d_free, double* cost, uint32_t flags) {
  if (flags & MTG_FLAG_REFINE) return solve_refined(plan, batch, layout, times, d_fixed, coeffs, d_free, cost, nullptr, flags);
  if (flags & MTG_FLAG_BASIC_SOLUTION) return solve_with_basic_solution(plan, batch, layout, times, d_fixed, coeffs, d_free, cost, nullptr, flags);
  return solve_impl(plan, batch, layout, times, d_fixed, coeffs, d_free, cost, flags, false);
}

int mtg_solve_linear_status(mtg_plan* plan, int64_t batch, const mtg_layout* layout, const double* times,
                            const double* d_fixed, double* coeffs, double* d_free, double* cost,
                            int32_t* trajectory_status, uint32_t flags) {
  if (flags & MTG_FLAG_REFINE) return solve_refined(plan, batch, layout, times, d_fixed, coeffs, d_free, cost, trajectory_status, flags);
  if (flags & MTG_FLAG_BASIC_SOLUTION)
    return solve_with_basic_solution(plan, batch, layout, times, d_fixed, coeffs, d_free, cost, trajectory_status, flags);
  return solve_impl(plan, batch, layout, times, d_fixed, coeffs, d_free, cost, flags, false, trajectory_status);
}

// The queue as ONE persistent launch (several when n > kSeqMax), coefficient output only: the slab-output fused kernel
// (mtg_solve_slab_queue_kernel) or the dimension-in-lane kernel (mtg_solve_dl_queue_kernel; canonical SoA inputs), chosen
// by the TOTAL number of trajectories the way single launches choose by their batch.  Returns 1 when the call does not
// qualify (the caller then enqueues one launch per batch), MTG_OK or an error otherwise.
static int sequence_as_queue(mtg_plan* p, int32_t n, int64_t batch, const mtg_layout* L, const double* const* times,
                             const double* const* d_fixed, double* const* coeffs, uint32_t flags) {
  mtg_context* ctx = p->ctx;
  if (n < 2 || batch <= 0 || ctx->knob_no_queue || p->null_dim > 0) return 1;
  if (flags & (MTG_FLAG_GENERIC_KERNEL | MTG_FLAG_SPLIT_DIMS | MTG_FLAG_SEQUENCE_ONE_LAUNCH_PER_BATCH)) return 1;
  const int32_t n_launch = std::min<int32_t>(n, kSeqMax);       // batches per launch
  const MtgSlabEntry* slab = nullptr;
  if (!(flags & MTG_FLAG_DIMLANE) && !ctx->knob_no_slab && p->fast && p->fast->k > 0 && p->fast->d == p->D)
    slab = mtg_find_slab(p->H, p->D, p->K, p->deriv, p->mask.data());
  if (slab && (!slab->queue || ((batch + kWave - 1) / kWave) * (int64_t)n_launch >= (1ll << 31))) slab = nullptr;
  const MtgDimlaneEntry* dl = p->dimlane;
  if (dl && (!dl->launch_queue || ctx->knob_no_dimlane || (flags & MTG_FLAG_FUSED_DIMS))) dl = nullptr;
  if (dl && (dimlane_input_kind(p, L, batch) < 0 ||
             padded16(batch) * 8 * (int64_t)std::max(p->K, p->n_fixed * p->D) >= (1ll << 32) ||
             ((batch + dl->tpw - 1) / dl->tpw) * (int64_t)n_launch >= (1ll << 31)))
    dl = nullptr;
  if (slab && dl && !(flags & MTG_FLAG_DIMLANE) && !dimlane_is_default(p, dl, batch * (int64_t)n_launch)) dl = nullptr;
  if (!slab && !dl) return 1;
  for (int32_t i = 0; i < n; ++i) {
    if (!times[i] || !coeffs[i] || (p->n_fixed > 0 && !d_fixed[i])) return MTG_ERR_INVALID_ARGUMENT;
    if (reinterpret_cast<uintptr_t>(coeffs[i]) & 15) return set_err(ctx, MTG_ERR_INVALID_ARGUMENT, "coeffs must be 16-byte aligned");
  }
  std::lock_guard<std::mutex> lock(ctx->mu);
  MTG_HIP_TRY(ctx, hipSetDevice(ctx->device));
  p->last.clear();
  const int64_t tpb = dl ? (batch + dl->tpw - 1) / dl->tpw : (batch + kWave - 1) / kWave;
  MtgParams P;
  fill_common(p, P, batch, L);
  P.times = times[0]; P.dfix = d_fixed ? d_fixed[0] : nullptr; P.coeffs = coeffs[0];
  if (!dl && !p->slab_queue_attr_set) {
    MTG_HIP_TRY(ctx, hipFuncSetAttribute((const void*)slab->queue, hipFuncAttributeMaxDynamicSharedMemorySize, (int)slab->lds));
    p->slab_queue_attr_set = true;
  }
  for (int32_t i0 = 0; i0 < n; i0 += kSeqMax) {
    MtgSeqQueue q;
    q.n = std::min<int32_t>(kSeqMax, n - i0);
    q.tiles_per_batch = (int)tpb;
    for (int i = 0; i < q.n; ++i) q.item[i] = MtgSeqItem{times[i0 + i], d_fixed ? d_fixed[i0 + i] : nullptr, coeffs[i0 + i]};
    for (int i = q.n; i < kSeqMax; ++i) q.item[i] = MtgSeqItem{nullptr, nullptr, nullptr};
    const int ntiles = q.n * (int)tpb;
    if (dl) {
      const int units = (ntiles + dl->np - 1) / dl->np;
      int grid = std::min(units, ctx->n_cu * 8);
      double* dl_ws = nullptr;
      if (dl->ws_per_lane) {   // long chains: persistent workgroups only, as in single launches
        grid = std::min(units, ctx->n_cu * 4 / (2 * dl->np));
        const size_t need = dl->ws_per_lane * (size_t)grid * dl->np * 2 * kWave;
        if (p->user_ws) {
          if (p->user_ws_bytes < need) return set_err(ctx, MTG_ERR_INVALID_ARGUMENT, "user workspace too small");
          dl_ws = p->user_ws;
        } else {
          int rc = ensure_buffer(ctx, &p->ws, &p->ws_bytes, need);
          if (rc != MTG_OK) return rc;
          dl_ws = p->ws;
        }
      }
      if (dl->launch_queue((void*)ctx->stream, grid, &q, ctx->d_status, (int)batch, ntiles, dl_ws, dimlane_input_kind(p, L, batch)) != 0)
        return set_err(ctx, MTG_ERR_DEVICE, "dimension-in-lane queue launch set-up failed");
    } else {
      const int grid = balanced_grid(ctx, ntiles, ctx->n_cu * 2);   // two workgroups per CU, one wave per SIMD (as the single-batch launch)
      hipLaunchKernelGGL(slab->queue, dim3(grid), dim3(kBlock), slab->lds, ctx->stream, P, ntiles, q);
    }
  }
  MTG_HIP_TRY(ctx, hipGetLastError());
  return MTG_OK;
}

int mtg_solve_linear_sequence_events(mtg_plan* plan, int32_t n, int64_t batch, const mtg_layout* layout,
                                     const double* const* times, const double* const* d_fixed, double* const* coeffs,
                                     uint32_t flags, void* start_event, void* stop_event) {
  if (!plan || !layout || n < 0 || !times || !coeffs || (plan->n_fixed > 0 && !d_fixed)) return MTG_ERR_INVALID_ARGUMENT;
  if (flags & (MTG_FLAG_HOST_POINTERS | MTG_FLAG_COST_ONLY)) return MTG_ERR_INVALID_ARGUMENT;
  mtg_context* ctx = plan->ctx;
  if (flags & MTG_FLAG_BASIC_SOLUTION) {
    // A queue stays asynchronous: a structurally rank-deficient plan runs the whole queue on its shadow (the pinned, regular
    // system -- the shadow's d_fixed of every batch gathered on the device first); on a regular plan the flag changes nothing.
    flags &= ~(uint32_t)MTG_FLAG_BASIC_SOLUTION;
    if (plan->null_dim > 0 && plan->n_free > 0 && n > 0 && batch > 0) {
      if (!plan->shadow) return set_err(ctx, MTG_ERR_UNSUPPORTED, "MTG_FLAG_BASIC_SOLUTION: this rank-deficient plan has no shadow plan");
      const size_t per = shadow_fixed_elems(plan, batch);
      mtg_layout SL;
      std::vector<const double*> sfx((size_t)n);
      {
        std::lock_guard<std::mutex> lock(ctx->mu);
        MTG_HIP_TRY(ctx, hipSetDevice(ctx->device));
        const int rs = ensure_buffer(ctx, &plan->shadow_buf, &plan->shadow_buf_bytes, per * (size_t)n * sizeof(double));
        if (rs != MTG_OK) return rs;
        for (int32_t i = 0; i < n; ++i) {
          if (!d_fixed[i]) return MTG_ERR_INVALID_ARGUMENT;
          shadow_gather_async(plan, batch, layout, d_fixed[i], plan->shadow_buf + per * (size_t)i, &SL, ctx->stream);
          sfx[(size_t)i] = plan->shadow_buf + per * (size_t)i;
        }
        MTG_HIP_TRY(ctx, hipGetLastError());
      }
      return mtg_solve_linear_sequence_events(plan->shadow, n, batch, &SL, times, sfx.data(), coeffs, flags, start_event, stop_event);
    }
  }
  if (start_event) {
    MTG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    MTG_HIP_TRY(ctx, hipEventRecord((hipEvent_t)start_event, ctx->stream));
  }
  int rc = sequence_as_queue(plan, n, batch, layout, times, d_fixed, coeffs, flags);
  if (rc == 1) {
    rc = MTG_OK;
    for (int32_t i = 0; i < n && rc == MTG_OK; ++i)
      rc = solve_impl(plan, batch, layout, times[i], d_fixed ? d_fixed[i] : nullptr, coeffs[i], nullptr, nullptr,
                      flags & ~(uint32_t)MTG_FLAG_SEQUENCE_ONE_LAUNCH_PER_BATCH, false);
  }
  if (stop_event) MTG_HIP_TRY(ctx, hipEventRecord((hipEvent_t)stop_event, ctx->stream));
  return rc;
}

int mtg_solve_linear_sequence(mtg_plan* plan, int32_t n, int64_t batch, const mtg_layout* layout,
                              const double* const* times, const double* const* d_fixed, double* const* coeffs,
                              uint32_t flags) {
  return mtg_solve_linear_sequence_events(plan, n, batch, layout, times, d_fixed, coeffs, flags, nullptr, nullptr);
}

namespace {
// J[b] = cost of the unperturbed problem; gradient[b][n] = (cost of variant n + 1 - J[b]) / h, written with the times' strides
__global__ void mtg_mellinger_grad_kernel(const double* __restrict__ cost_all, long long B, int K, double inv_h,
                                          double* __restrict__ J, double* __restrict__ grad, long long gs_b, long long gs_k) {
  const long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  if (cost_all == nullptr) {   // one segment: zero gradient
    for (int n = 0; n < K; ++n) grad[b * gs_b + n * gs_k] = 0.0;
    return;
  }
  const double j0 = cost_all[b];
  if (J) J[b] = j0;
  for (int n = 0; n < K; ++n) grad[b * gs_b + n * gs_k] = (cost_all[(long long)(n + 1) * B + b] - j0) * inv_h;
}
}  // namespace

int mtg_mellinger_cost_gradient(mtg_plan* plan, int64_t batch, const mtg_layout* layout, const double* times,
                                const double* d_fixed, double increment_time, double time_lower_bound, double* cost,
                                double* gradient) {
  if (!plan || !layout || !times || !gradient || batch < 0 || !(increment_time > 0.0)) return MTG_ERR_INVALID_ARGUMENT;
  if (batch == 0) return MTG_OK;
  mtg_context* ctx = plan->ctx;
  const int K = plan->K;
  if (K == 1) {   // polynomial_optimization_nonlinear_impl.h:295-302: one segment -> zero gradient
    int rc = MTG_OK;
    if (cost) rc = solve_impl(plan, batch, layout, times, d_fixed, nullptr, nullptr, cost, MTG_FLAG_COST_ONLY, false);
    if (rc != MTG_OK) return rc;
    std::lock_guard<std::mutex> lock(ctx->mu);
    MTG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(mtg_mellinger_grad_kernel, dim3((unsigned)((batch + 255) / 256)), dim3(256), 0, ctx->stream,
                       (const double*)nullptr, (long long)batch, 1, 0.0, (double*)nullptr, gradient,
                       (long long)layout->times_stride_b, (long long)layout->times_stride_k);
    MTG_HIP_TRY(ctx, hipGetLastError());
    return MTG_OK;
  }
  double* all = nullptr;
  {
    std::lock_guard<std::mutex> lock(ctx->mu);
    MTG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int rc = ensure_buffer(ctx, &plan->pert_cost, &plan->pert_cost_bytes, (size_t)(K + 1) * batch * sizeof(double));
    if (rc != MTG_OK) return rc;
    all = plan->pert_cost;
  }
  const PerturbedTimes pt{increment_time, time_lower_bound};
  const int rc = solve_impl(plan, batch, layout, times, d_fixed, nullptr, nullptr, all, MTG_FLAG_COST_ONLY, false, nullptr, &pt);
  if (rc != MTG_OK) return rc;
  std::lock_guard<std::mutex> lock(ctx->mu);
  hipLaunchKernelGGL(mtg_mellinger_grad_kernel, dim3((unsigned)((batch + 255) / 256)), dim3(256), 0, ctx->stream,
                     (const double*)all, (long long)batch, K, 1.0 / increment_time, cost, gradient,
                     (long long)layout->times_stride_b, (long long)layout->times_stride_k);
  MTG_HIP_TRY(ctx, hipGetLastError());
  return MTG_OK;
}

int mtg_update_segments_from_free(mtg_plan* plan, int64_t batch, const mtg_layout* layout, const double* times,
                                  const double* d_fixed, const double* d_free, double* coeffs, double* cost,
                                  uint32_t flags) {
  return solve_impl(plan, batch, layout, times, d_fixed, coeffs, const_cast<double*>(d_free), cost, flags, true);
}

// ---- mixed requests ---------------------------------------------------------------------------
struct MtgMultiGroup {
  const MtgStaticEntry* entry = nullptr;   // rolled configuration shared by the group's items
  std::vector<int> items;
  MtgParams* d_table = nullptr;
  MtgTileRef* d_tiles = nullptr;
  double* d_ws = nullptr;
  int ntiles = 0, grid = 0, ngroups = 1;    // ngroups: dimension groups (grid.y) of the launch
  size_t lds = 0;
  bool extra = false;                       // any item wants d_free / cost
  bool any = false;                         // cross-structure launch (mtg_solve_multi_any_kernel): items of several configurations
  int dg = 0;                               // dimensions per workgroup of the launch
  bool attr_set[4] = {false, false, false, false};
  int any_units = 1;                        // cross-structure launch: dimension groups per tile (units = ntiles * any_units)
};
struct MtgDlAnyGroup {                      // cross-structure dimension-in-lane launch (mtg_solve_dl_any_kernel)
  std::vector<int> items;
  MtgDlAnyItem* d_items = nullptr;
  MtgDlAnyUnit* d_units = nullptr;
  int* d_wg_begin = nullptr;                // [grid + 1]: workgroup w runs d_units[d_wg_begin[w] .. d_wg_begin[w + 1])
  double* d_ws = nullptr;
  int nunits = 0, grid = 0;
  bool shared_schedule = false;             // d_units / d_wg_begin belong to the context's schedule cache
  size_t d_items_bytes = 0;                 // d_items comes from (and returns to) the context's free list
  std::vector<MtgDlAnyItem> h_items;        // source of the asynchronous upload
};
struct mtg_multi {
  mtg_context* ctx = nullptr;
  MtgDlAnyGroup dl_any;
  std::vector<mtg_multi_item> items;
  std::vector<MtgMultiGroup> groups;
  std::vector<int> singles;                 // items launched through the ordinary path
  bool concurrent = false;                  // MTG_FLAG_CONCURRENT_ITEMS: singles spread over the context's side streams
  std::vector<int> lane_of;                 // [singles.size()] side stream of each single (longest-processing-time first)
  int n_lanes = 0;
  // MTG_FLAG_BASIC_SOLUTION: items of structurally rank-deficient plans run on the plan's shadow; their shadow d_fixed is gathered
  // from the caller's buffer in front of every solve, their d_free (when asked for) scattered back behind it
  struct ShadowFix { const mtg_plan* plan; int64_t batch; mtg_layout layout; const double* d_fixed; double* sfx; double* d_free; double* sfr; };
  std::vector<ShadowFix> shadow_fix;
  double* shadow_mem = nullptr;
};

static void multi_free(mtg_multi* m, bool context_locked) {
  hipSetDevice(m->ctx->device);
  hipStreamSynchronize(m->ctx->stream);
  if (m->dl_any.d_items) {
    std::unique_lock<std::mutex> lock(m->ctx->mu, std::defer_lock);
    if (!context_locked) lock.lock();
    m->ctx->dl_any_item_pool.push_back({(void*)m->dl_any.d_items, m->dl_any.d_items_bytes});
  }
  if (!m->dl_any.shared_schedule) {
    if (m->dl_any.d_units) hipFree(m->dl_any.d_units);
    if (m->dl_any.d_wg_begin) hipFree(m->dl_any.d_wg_begin);
  }
  // (dl_any.d_ws is the context's)
  for (MtgMultiGroup& g : m->groups) {
    if (g.d_table) hipFree(g.d_table);
    if (g.d_tiles) hipFree(g.d_tiles);
    if (g.d_ws) hipFree(g.d_ws);
  }
  if (m->shadow_mem) hipFree(m->shadow_mem);
  delete m;
}
int mtg_multi_destroy(mtg_multi* m) {
  if (!m) return MTG_OK;
  multi_free(m, false);
  return MTG_OK;
}

int mtg_multi_create(mtg_context* ctx, int32_t n_items, const mtg_multi_item* items, uint32_t flags, mtg_multi** out) {
  if (!ctx || !items || !out || n_items < 1) return MTG_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  std::vector<mtg_multi_item> patched;
  std::vector<mtg_multi::ShadowFix> fixes;
  double* shadow_mem = nullptr;
  if (flags & MTG_FLAG_BASIC_SOLUTION) {
    // the request stays asynchronous: an item of a structurally rank-deficient plan becomes an item of that plan's SHADOW (the
    // pinned, regular system: just another plan of the request); regular plans' items are unchanged
    flags &= ~(uint32_t)MTG_FLAG_BASIC_SOLUTION;
    patched.assign(items, items + n_items);
    size_t total = 0;
    for (int i = 0; i < n_items; ++i) {
      const mtg_multi_item& it = items[i];
      if (!it.plan || it.plan->ctx != ctx || it.batch < 0) return set_err(ctx, MTG_ERR_INVALID_ARGUMENT, "mtg_multi_create: bad item");
      if (it.plan->null_dim <= 0 || it.plan->n_free == 0 || it.batch == 0) continue;
      if (!it.plan->shadow) return set_err(ctx, MTG_ERR_UNSUPPORTED, "MTG_FLAG_BASIC_SOLUTION: a rank-deficient plan of the request has no shadow plan");
      if (!it.d_fixed) return set_err(ctx, MTG_ERR_INVALID_ARGUMENT, "mtg_multi_create: bad item");
      total += shadow_fixed_elems(it.plan, it.batch) + (it.d_free ? (size_t)it.batch * it.plan->D * std::max(it.plan->shadow->n_free, 1) : 0);
    }
    if (total > 0) {
      std::lock_guard<std::mutex> lock(ctx->mu);
      MTG_HIP_TRY(ctx, hipSetDevice(ctx->device));
      MTG_HIP_TRY(ctx, hipMalloc((void**)&shadow_mem, total * sizeof(double)));
      double* cur = shadow_mem;
      for (int i = 0; i < n_items; ++i) {
        mtg_multi_item& it = patched[(size_t)i];
        const mtg_plan* p = it.plan;
        if (p->null_dim <= 0 || p->n_free == 0 || it.batch == 0) continue;
        mtg_multi::ShadowFix fx{p, it.batch, it.layout, it.d_fixed, cur, it.d_free, nullptr};
        cur += shadow_fixed_elems(p, it.batch);
        // the item's layout with the shadow buffer's fixed-value strides (same rule as shadow_gather_async) ...
        const mtg_layout& L = items[i].layout;
        const int nfs = p->shadow->n_fixed, nps = p->shadow->n_free;
        const bool soa = L.fixed_stride_b == 1 && L.times_stride_b == 1 && L.times_stride_k >= it.batch && L.times_stride_k <= padded16(it.batch);
        if (soa) { it.layout.fixed_stride_b = 1; it.layout.fixed_stride_c = L.times_stride_k; it.layout.fixed_stride_d = (int64_t)nfs * L.times_stride_k; }
        else { it.layout.fixed_stride_b = (int64_t)p->D * nfs; it.layout.fixed_stride_d = nfs; it.layout.fixed_stride_c = 1; }
        if (it.d_free) {     // ... and d_P through a contiguous [B][D][n_free of the shadow] buffer, scattered back after the solve
          fx.sfr = cur;
          cur += (size_t)it.batch * p->D * std::max(nps, 1);
          it.layout.free_stride_b = (int64_t)p->D * nps; it.layout.free_stride_d = nps; it.layout.free_stride_c = 1;
          it.d_free = fx.sfr;
        }
        it.plan = p->shadow;
        it.d_fixed = fx.sfx;
        fixes.push_back(fx);
      }
      items = patched.data();
    }
  }
  struct ShadowMemGuard { double*& m; ~ShadowMemGuard() { if (m) hipFree(m); } } shadow_guard{shadow_mem};   // (released on every error return)
  for (int i = 0; i < n_items; ++i) {
    const mtg_multi_item& it = items[i];
    if (!it.plan || it.plan->ctx != ctx || it.batch < 0 || !it.times || !it.coeffs || (it.plan->n_fixed > 0 && !it.d_fixed))
      return set_err(ctx, MTG_ERR_INVALID_ARGUMENT, "mtg_multi_create: bad item");
    if (reinterpret_cast<uintptr_t>(it.coeffs) & 15) return set_err(ctx, MTG_ERR_INVALID_ARGUMENT, "coeffs must be 16-byte aligned");
  }
  std::lock_guard<std::mutex> lock(ctx->mu);
  MTG_HIP_TRY(ctx, hipSetDevice(ctx->device));
  mtg_multi* m = new (std::nothrow) mtg_multi();
  if (!m) return MTG_ERR_DEVICE;
  m->ctx = ctx;
  m->items.assign(items, items + n_items);
  if (flags & MTG_FLAG_CONCURRENT_ITEMS) {
    // One launch per item, each through the ordinary variant choice, on up to kSideStreams side streams (the HIP runtime
    // maps a process's streams onto 4 hardware queues: more streams add no overlap).  Longest-processing-time-first
    // assignment; the work estimate is chain length x N^2 x rounds of tiles.
    constexpr int kSideStreams = 4;
    m->concurrent = true;
    for (int i = 0; i < n_items; ++i)
      if (items[i].batch > 0) m->singles.push_back(i);
    auto est = [&](int i) {
      const mtg_plan* p = items[i].plan;
      const double rounds = std::max(1.0, (double)items[i].batch * p->D / (64.0 * 4.0 * ctx->n_cu));
      return (double)p->K * p->N * p->N * rounds;
    };
    std::stable_sort(m->singles.begin(), m->singles.end(), [&](int a, int b) { return est(a) > est(b); });
    m->n_lanes = std::min<int>(kSideStreams, (int)m->singles.size());
    std::vector<double> load(std::max(1, m->n_lanes), 0.0);
    for (size_t s = 0; s < m->singles.size(); ++s) {
      const int i = m->singles[s];
      int lane = (int)(std::min_element(load.begin(), load.end()) - load.begin());
      for (size_t r = 0; r < s; ++r)   // items of one plan share its workspace: same stream, in order
        if (items[m->singles[r]].plan == items[i].plan) lane = m->lane_of[r];
      load[lane] += est(i);
      m->lane_of.push_back(lane);
    }
    while ((int)ctx->side_streams.size() < m->n_lanes) {
      hipStream_t q = nullptr;
      hipEvent_t e = nullptr;
      if (hipStreamCreateWithFlags(&q, hipStreamNonBlocking) != hipSuccess ||
          hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
        if (q) hipStreamDestroy(q);
        delete m;
        return set_err(ctx, MTG_ERR_DEVICE, "mtg_multi_create: side stream creation failed");
      }
      ctx->side_streams.push_back(q);
      ctx->join_events.push_back(e);
    }
    if (!ctx->fork_event && hipEventCreateWithFlags(&ctx->fork_event, hipEventDisableTiming) != hipSuccess) {
      delete m;
      return set_err(ctx, MTG_ERR_DEVICE, "mtg_multi_create: event creation failed");
    }
    m->shadow_fix = std::move(fixes); m->shadow_mem = shadow_mem; shadow_mem = nullptr;
    *out = m;
    return MTG_OK;
  }
  // Items that can run their static dimension-in-lane configuration (canonical SoA inputs, coefficient output only) join
  // ONE cross-structure launch (mtg_solve_dl_any_kernel), whatever their N and K: back-substitution data in registers
  // instead of the rolled kernels' workspace traffic.
  std::vector<char> taken(n_items, 0);
  if (!(flags & (MTG_FLAG_FUSED_DIMS | MTG_FLAG_SPLIT_DIMS | MTG_FLAG_GENERIC_KERNEL)) && !ctx->knob_no_dimlane) {
    std::vector<int> cand;
    for (int i = 0; i < n_items; ++i) {
      const mtg_multi_item& it = items[i];
      const mtg_plan* p = it.plan;
      if (it.batch <= 0 || it.cost || (it.d_free && p->n_free > 0) || mtg_dl_any_index(p->dimlane) < 0) continue;
      const mtg_layout& L = it.layout;
      { const int kind = dimlane_input_kind(p, &L, it.batch); if (kind < 0 || kind > 1) continue; }   // (padded SoA: single / queue launches only)
      if (it.batch * 8 * (int64_t)std::max(p->K, p->n_fixed * p->D) >= (1ll << 32)) continue;
      cand.push_back(i);
    }
    if (cand.size() >= 2) {
      MtgDlAnyGroup& g = m->dl_any;
      auto work = [&](int a) { return (long long)items[a].plan->K * items[a].plan->N * items[a].plan->N; };
      std::stable_sort(cand.begin(), cand.end(), [&](int a, int b) { return work(a) > work(b); });
      g.items = cand;
      std::vector<MtgDlAnyItem> table(cand.size());
      std::vector<MtgDlAnyUnit> units;
      for (size_t bi = 0; bi < cand.size(); ++bi) {
        const mtg_multi_item& it = items[cand[bi]];
        taken[cand[bi]] = 1;
        const int tpw = it.plan->dimlane->tpw;
        table[bi] = MtgDlAnyItem{it.times, it.d_fixed, it.coeffs, (int)it.batch, mtg_dl_any_index(it.plan->dimlane),
                                 dimlane_input_kind(it.plan, &it.layout, it.batch), 0};
        g.nunits += (int)((it.batch + tpw - 1) / tpw);
      }
      g.grid = std::min(g.nunits, ctx->n_cu * 2);      // two 2-wave workgroups per CU: one wave per SIMD
      // the schedule of this STRUCTURE: cached per context (see mtg_context::DlAnySchedule)
      constexpr size_t kMaxDlAnySchedules = 64;
      std::vector<long long> key;
      key.reserve(2 + 2 * cand.size());
      key.push_back(g.grid);
      key.push_back(ctx->knob_dl_any_rr ? 1 : 0);
      for (size_t bi = 0; bi < cand.size(); ++bi) {
        key.push_back(table[bi].cfg);
        key.push_back((items[cand[bi]].batch + items[cand[bi]].plan->dimlane->tpw - 1) / items[cand[bi]].plan->dimlane->tpw);
      }
      const mtg_context::DlAnySchedule* hit = nullptr;
      for (const auto& sc : ctx->dl_any_schedules)
        if (sc.key == key) { hit = &sc; break; }
      bool ok = true;
      if (hit) {
        g.d_units = (MtgDlAnyUnit*)hit->d_units;
        g.d_wg_begin = hit->d_wg_begin;
        g.shared_schedule = true;
      } else {
        // Every workgroup gets its own unit list.  Default: greedy longest-processing-time assignment (units in order of
        // decreasing cost, each to the least-loaded workgroup; ties -> the lowest index, so the first `grid` units land on
        // workgroups 0, 1, 2, ... and neighbours start with the same configuration).  Cost model from the per-bucket kernel
        // times (profiles/r03b_configs.jsonl): ~0.03 us x K x (N/2)^2 + ~2.5 us per unit.  MTG_DL_ANY_SCHED=rr: round 2's
        // schedule (units w, w + grid, ... of the sorted list, every second round reversed).
        units.reserve((size_t)g.nunits);
        for (size_t bi = 0; bi < cand.size(); ++bi) {
          const int tpw = items[cand[bi]].plan->dimlane->tpw;
          const int nt = (int)((items[cand[bi]].batch + tpw - 1) / tpw);
          for (int t = 0; t < nt; ++t) units.push_back(MtgDlAnyUnit{(int)bi, t});
        }
        std::vector<std::vector<MtgDlAnyUnit>> lists(g.grid);
        if (ctx->knob_dl_any_rr) {
          for (size_t r = 0; r * (size_t)g.grid < units.size(); ++r) {
            const size_t lo = r * (size_t)g.grid, hi = std::min(units.size(), lo + (size_t)g.grid);
            const bool rev = (r & 1) && hi - lo == (size_t)g.grid;
            for (size_t u = lo; u < hi; ++u) lists[rev ? (hi - 1 - u) : (u - lo)].push_back(units[u]);
          }
        } else {
          auto cost = [&](const MtgDlAnyUnit& u) {
            const mtg_plan* pl = items[cand[u.item]].plan;
            return (long long)pl->K * pl->H * pl->H + 90;
          };
          typedef std::pair<long long, int> Load;      // (load, workgroup): min-heap
          std::priority_queue<Load, std::vector<Load>, std::greater<Load>> heap;
          for (int w = 0; w < g.grid; ++w) heap.push(Load(0, w));
          for (const MtgDlAnyUnit& u : units) {          // `units` is sorted by decreasing work already
            Load l = heap.top();
            heap.pop();
            lists[l.second].push_back(u);
            heap.push(Load(l.first + cost(u), l.second));
          }
        }
        std::vector<int> wg_begin(g.grid + 1, 0);
        units.clear();
        for (int w = 0; w < g.grid; ++w) {
          units.insert(units.end(), lists[w].begin(), lists[w].end());
          wg_begin[w + 1] = (int)units.size();
        }
        ok = hipMalloc((void**)&g.d_units, units.size() * sizeof(MtgDlAnyUnit)) == hipSuccess &&
             hipMalloc((void**)&g.d_wg_begin, wg_begin.size() * sizeof(int)) == hipSuccess &&
             hipMemcpy(g.d_wg_begin, wg_begin.data(), wg_begin.size() * sizeof(int), hipMemcpyHostToDevice) == hipSuccess &&
             hipMemcpy(g.d_units, units.data(), units.size() * sizeof(MtgDlAnyUnit), hipMemcpyHostToDevice) == hipSuccess;
        if (ok && ctx->dl_any_schedules.size() < kMaxDlAnySchedules) {
          mtg_context::DlAnySchedule sc;
          sc.key = std::move(key); sc.grid = g.grid; sc.nunits = g.nunits; sc.d_units = g.d_units; sc.d_wg_begin = g.d_wg_begin;
          ctx->dl_any_schedules.push_back(std::move(sc));
          g.shared_schedule = true;
        }
      }
      // workspace: one buffer per context, sized for the largest grid (requests of a context run in stream order)
      const size_t ws_bytes = std::max<size_t>(16, mtg_dl_any_ws_per_lane() * (size_t)(ctx->n_cu * 2) * 2 * kWave);
      if (ok && ctx->dl_any_ws_bytes < ws_bytes) {
        if (ctx->dl_any_ws) { hipStreamSynchronize(ctx->stream); hipFree(ctx->dl_any_ws); ctx->dl_any_ws = nullptr; ctx->dl_any_ws_bytes = 0; }
        ok = hipMalloc((void**)&ctx->dl_any_ws, ws_bytes) == hipSuccess;
        if (ok) ctx->dl_any_ws_bytes = ws_bytes;
      }
      g.d_ws = ctx->dl_any_ws;
      // item table: a buffer of the free list (or a new one), filled by an asynchronous copy on the context's stream -- the
      // launch that reads it is enqueued behind it
      const size_t ib = table.size() * sizeof(MtgDlAnyItem);
      if (ok) {
        for (size_t k2 = 0; k2 < ctx->dl_any_item_pool.size(); ++k2)
          if (ctx->dl_any_item_pool[k2].second >= ib) {
            g.d_items = (MtgDlAnyItem*)ctx->dl_any_item_pool[k2].first;
            g.d_items_bytes = ctx->dl_any_item_pool[k2].second;
            ctx->dl_any_item_pool.erase(ctx->dl_any_item_pool.begin() + (long)k2);
            break;
          }
        if (!g.d_items) {
          g.d_items_bytes = std::max<size_t>(ib, 4096);
          ok = hipMalloc((void**)&g.d_items, g.d_items_bytes) == hipSuccess;
        }
      }
      g.h_items = std::move(table);
      if (ok) ok = hipMemcpyAsync(g.d_items, g.h_items.data(), ib, hipMemcpyHostToDevice, ctx->stream) == hipSuccess;
      if (!ok) {
        multi_free(m, true);
        return set_err(ctx, MTG_ERR_DEVICE, "mtg_multi_create: device allocation failed");
      }
    }
  }
  // group the others by rolled configuration
  for (int i = 0; i < n_items; ++i) {
    if (taken[i]) continue;
    mtg_plan* p = items[i].plan;
    const MtgStaticEntry* e = (items[i].batch > 0 && p->K >= 2)
                                  ? mtg_find_static(p->H, p->D, p->K, p->deriv, p->mask.data(), true) : nullptr;
    if (!e || !e->multi[0]) {
      if (items[i].batch > 0) m->singles.push_back(i);
      continue;
    }
    MtgMultiGroup* g = nullptr;
    for (MtgMultiGroup& c : m->groups) if (c.entry == e) g = &c;
    if (!g) {
      m->groups.emplace_back();
      g = &m->groups.back();
      g->entry = e;
    }
    g->items.push_back(i);
  }
  // Cross-structure merge: groups whose (3-dimensional) rolled configurations are all covered by mtg_solve_multi_any_kernel
  // become ONE launch (config 4: N = 8, 10 and 12 buckets together) -- streams would not overlap them (see the kernel).
  {
    std::vector<size_t> anyable;
    for (size_t gi = 0; gi < m->groups.size(); ++gi)
      if (m->groups[gi].entry->d == 3 && mtg_any_cfg_index(m->groups[gi].entry) >= 0) anyable.push_back(gi);
    if (anyable.size() >= 2) {
      MtgMultiGroup merged;
      merged.any = true;
      merged.entry = m->groups[anyable[0]].entry;
      for (size_t gi : anyable) merged.items.insert(merged.items.end(), m->groups[gi].items.begin(), m->groups[gi].items.end());
      for (size_t r = anyable.size(); r-- > 0;) m->groups.erase(m->groups.begin() + anyable[r]);
      m->groups.push_back(merged);
    }
  }
  // a group of one gains nothing from the merged form: leave it to the ordinary path (static variants, heuristics)
  for (size_t gi = 0; gi < m->groups.size();) {
    if (m->groups[gi].items.size() < 2) {
      m->singles.push_back(m->groups[gi].items[0]);
      m->groups.erase(m->groups.begin() + gi);
    } else {
      ++gi;
    }
  }
  long long total_tiles = 0;
  for (const MtgMultiGroup& g : m->groups)
    for (int i : g.items) total_tiles += (items[i].batch + kWave - 1) / kWave;
  for (MtgMultiGroup& g : m->groups) {
    const int D = g.entry->d;
    // tiles, longest chain first (work per tile ~ K N^2)
    std::vector<int> order(g.items.begin(), g.items.end());
    auto work = [&](int a) { return (long long)items[a].plan->K * items[a].plan->N * items[a].plan->N; };
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return work(a) > work(b); });
    // few tiles: the one-dimension-per-workgroup form of the same configurations (D x the workgroups, lighter waves)
    // while all of them are resident at once -- the same rule as for single-plan launches (flags override)
    const bool want_split = D > 1 && ((flags & MTG_FLAG_SPLIT_DIMS) ||
                                      (!(flags & MTG_FLAG_FUSED_DIMS) && total_tiles * D <= 4ll * ctx->n_cu));
    std::vector<const MtgStaticEntry*> ent(order.size());
    bool split_ok = want_split;
    for (size_t bi = 0; bi < order.size(); ++bi) {
      const mtg_plan* p = items[order[bi]].plan;
      ent[bi] = mtg_find_static(p->H, p->D, p->K, p->deriv, p->mask.data(), true);
      const MtgStaticEntry* es = want_split ? mtg_find_static(p->H, 1, p->K, p->deriv, p->mask.data(), true) : nullptr;
      if (!es || !es->multi[0] || (g.any && mtg_any_cfg_index(es) < 0)) split_ok = false;
    }
    int Dw = D;
    if (split_ok) {
      for (size_t bi = 0; bi < order.size(); ++bi) {
        const mtg_plan* p = items[order[bi]].plan;
        ent[bi] = mtg_find_static(p->H, 1, p->K, p->deriv, p->mask.data(), true);
      }
      g.entry = ent[0];
      g.ngroups = D;
      Dw = 1;
    }
    g.dg = Dw;
    std::vector<MtgTileRef> tiles;
    std::vector<MtgParams> table(order.size());
    int kc_max = 1;
    size_t E = 0;
    g.lds = 0;
    for (size_t bi = 0; bi < order.size(); ++bi) {
      const mtg_multi_item& it = items[order[bi]];
      const int H = it.plan->H;
      const int nt = (int)((it.batch + kWave - 1) / kWave);
      const int cfg = g.any ? mtg_any_cfg_index(ent[bi]) : 0;
      for (int t = 0; t < nt; ++t) tiles.push_back(MtgTileRef{(int)bi, t, cfg});
      kc_max = std::max(kc_max, (it.plan->K + 1) / 2);
      g.extra = g.extra || it.cost != nullptr || (it.d_free != nullptr && it.plan->n_free > 0);
      E = std::max(E, (size_t)H * H + (size_t)Dw * H);
      const int fm = H - __builtin_popcount((unsigned)ent[bi]->mi);
      const size_t stage = (size_t)64 * ((size_t)(Dw * 2 * H / 2) | 1) * 2 * sizeof(double);
      g.lds = std::max(g.lds, 2 * stage + (size_t)2 * (fm * (fm + 1) / 2 + Dw * fm) * kWave * sizeof(double));
    }
    g.ntiles = (int)tiles.size();
    g.grid = std::min(g.ntiles, std::max(1, ctx->n_cu * 4 / g.ngroups));
    if (g.any) {   // one-dimensional grid over (tile, dimension group) units: as many workgroups as are resident at once
      g.grid = std::min(g.ntiles * g.ngroups, ctx->n_cu * 2);
      g.any_units = g.ngroups;
      g.ngroups = 1;
    }
    const size_t ws_bytes = (size_t)kc_max * E * (size_t)g.grid * g.ngroups * kBlock * sizeof(double);
    if (hipMalloc((void**)&g.d_ws, ws_bytes) != hipSuccess ||
        hipMalloc((void**)&g.d_table, table.size() * sizeof(MtgParams)) != hipSuccess ||
        hipMalloc((void**)&g.d_tiles, tiles.size() * sizeof(MtgTileRef)) != hipSuccess) {
      multi_free(m, true);
      return set_err(ctx, MTG_ERR_DEVICE, "mtg_multi_create: device allocation failed");
    }
    for (size_t bi = 0; bi < order.size(); ++bi) {
      const mtg_multi_item& it = items[order[bi]];
      MtgParams& P = table[bi];
      fill_common(it.plan, P, it.batch, &it.layout);
      P.times = it.times; P.dfix = it.d_fixed; P.coeffs = it.coeffs;
      P.dfree = it.plan->n_free ? it.d_free : nullptr;
      P.cost = it.cost;
      P.ws = g.d_ws;
      P.ws_stride = (long long)g.grid * g.ngroups * kBlock;
    }
    if (hipMemcpy(g.d_table, table.data(), table.size() * sizeof(MtgParams), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(g.d_tiles, tiles.data(), tiles.size() * sizeof(MtgTileRef), hipMemcpyHostToDevice) != hipSuccess) {
      multi_free(m, true);
      return set_err(ctx, MTG_ERR_DEVICE, "mtg_multi_create: table upload failed");
    }
  }
  m->shadow_fix = std::move(fixes); m->shadow_mem = shadow_mem; shadow_mem = nullptr;
  *out = m;
  return MTG_OK;
}

int mtg_multi_launch_count(const mtg_multi* m) {
  return m ? (int)(m->groups.size() + m->singles.size() + (m->dl_any.nunits > 0 ? 1 : 0)) : 0;
}

static int multi_solve_body(mtg_multi* m);
int mtg_multi_solve(mtg_multi* m) {
  if (!m) return MTG_ERR_INVALID_ARGUMENT;
  mtg_context* ctx = m->ctx;
  if (!m->shadow_fix.empty()) {      // MTG_FLAG_BASIC_SOLUTION items: the shadows' d_fixed from the callers' current values
    std::lock_guard<std::mutex> lock(ctx->mu);
    MTG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    mtg_layout unused;
    for (const mtg_multi::ShadowFix& fx : m->shadow_fix) shadow_gather_async(fx.plan, fx.batch, &fx.layout, fx.d_fixed, fx.sfx, &unused, ctx->stream);
    MTG_HIP_TRY(ctx, hipGetLastError());
  }
  const int rc = multi_solve_body(m);
  if (rc != MTG_OK) return rc;
  if (!m->shadow_fix.empty()) {      // ... and their d_P back into the callers' layout, exact zeros at the pinned slots
    std::lock_guard<std::mutex> lock(ctx->mu);
    MTG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    for (const mtg_multi::ShadowFix& fx : m->shadow_fix) {
      if (!fx.d_free || fx.plan->n_free == 0) continue;
      const long long n = (long long)fx.batch * fx.plan->D * fx.plan->n_free;
      hipLaunchKernelGGL(mtg_pin_scatter_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const double*)fx.sfr,
                         (const int*)(fx.plan->d_shadow_maps + fx.plan->shadow->n_fixed), fx.d_free, (long long)fx.layout.free_stride_b,
                         (long long)fx.layout.free_stride_d, (long long)fx.layout.free_stride_c, (long long)fx.batch, fx.plan->D, fx.plan->n_free,
                         fx.plan->shadow->n_free);
    }
    MTG_HIP_TRY(ctx, hipGetLastError());
  }
  return MTG_OK;
}
static int multi_solve_body(mtg_multi* m) {
  mtg_context* ctx = m->ctx;
  {
    std::lock_guard<std::mutex> lock(ctx->mu);
    MTG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (m->dl_any.nunits > 0) {
      const MtgDlAnyGroup& g = m->dl_any;
      if (mtg_dl_any_launch((void*)ctx->stream, g.grid, g.d_items, g.d_units, g.d_wg_begin, ctx->d_status, g.d_ws) != 0)
        return set_err(ctx, MTG_ERR_DEVICE, "cross-structure dimension-in-lane launch set-up failed");
    }
    for (MtgMultiGroup& g : m->groups) {
      for (int i : g.items) {
        const mtg_multi_item& it = m->items[i];
        if (it.cost) MTG_HIP_TRY(ctx, hipMemsetAsync(it.cost, 0, it.batch * sizeof(double), ctx->stream));
      }
      // few tiles: write-through stores (no serial end-of-kernel L2 write-back), as for single-plan launches
      const bool write_through = (long long)g.ntiles * g.ngroups * g.any_units <= 4ll * ctx->n_cu;
      const int variant = (g.extra ? 1 : 0) + (write_through ? 2 : 0);
      SolveMultiFn fn = g.any ? mtg_multi_any_fn(g.dg, variant) : g.entry->multi[variant];
      if (g.any && g.lds > 64 * 1024 && !g.attr_set[variant]) {
        MTG_HIP_TRY(ctx, hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds));
        g.attr_set[variant] = true;
      }
      hipLaunchKernelGGL(fn, dim3(g.grid, g.ngroups), dim3(kBlock), g.lds, ctx->stream, (const MtgParams*)g.d_table,
                         (const MtgTileRef*)g.d_tiles, g.ntiles);
      for (int i : g.items)   // (merged groups are compile-time-mask shapes with fully fixed ends: never rank-deficient; kept for symmetry)
        flag_structurally_singular(m->items[i].plan, ctx->stream, ctx->d_status, nullptr, m->items[i].batch);
    }
    MTG_HIP_TRY(ctx, hipGetLastError());
  }
  if (m->concurrent) {
    // fork: the side streams start behind the work already queued on the context's stream
    MTG_HIP_TRY(ctx, hipSetDevice(ctx->device));
    MTG_HIP_TRY(ctx, hipEventRecord(ctx->fork_event, ctx->stream));
    for (int l = 0; l < m->n_lanes; ++l) MTG_HIP_TRY(ctx, hipStreamWaitEvent(ctx->side_streams[l], ctx->fork_event, 0));
    int rc_all = MTG_OK;
    for (size_t s = 0; s < m->singles.size() && rc_all == MTG_OK; ++s) {
      const mtg_multi_item& it = m->items[m->singles[s]];
      rc_all = solve_impl(it.plan, it.batch, &it.layout, it.times, it.d_fixed, it.coeffs, it.d_free, it.cost, 0, false,
                          nullptr, nullptr, ctx->side_streams[m->lane_of[s]]);
    }
    // join (also after a failed enqueue: the context's stream must not run ahead of what was launched)
    for (int l = 0; l < m->n_lanes; ++l) {
      MTG_HIP_TRY(ctx, hipEventRecord(ctx->join_events[l], ctx->side_streams[l]));
      MTG_HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->join_events[l], 0));
    }
    return rc_all;
  }
  for (int i : m->singles) {
    const mtg_multi_item& it = m->items[i];
    const int rc = solve_impl(it.plan, it.batch, &it.layout, it.times, it.d_fixed, it.coeffs, it.d_free, it.cost, 0, false);
    if (rc != MTG_OK) return rc;
  }
  return MTG_OK;
}

int mtg_time_last_solve(mtg_plan* p, int iters, double* mean_us) {
  if (!p || !mean_us || iters < 1) return MTG_ERR_INVALID_ARGUMENT;
  mtg_context* ctx = p->ctx;
  if (p->last.empty()) return set_err(ctx, MTG_ERR_INVALID_ARGUMENT, "no recorded solve launch");
  std::lock_guard<std::mutex> lock(ctx->mu);
  MTG_HIP_TRY(ctx, hipSetDevice(ctx->device));
  hipEvent_t e0, e1;
  MTG_HIP_TRY(ctx, hipEventCreate(&e0));
  if (hipEventCreate(&e1) != hipSuccess) { hipEventDestroy(e0); return set_err(ctx, MTG_ERR_DEVICE, "hipEventCreate"); }
  struct EventGuard { hipEvent_t a, b; ~EventGuard() { hipEventDestroy(a); hipEventDestroy(b); } } guard{e0, e1};
  MTG_HIP_TRY(ctx, hipEventRecord(e0, ctx->stream));
  for (int i = 0; i < iters; ++i) {
    for (const LaunchRecord& r : p->last) {
      if (r.coop) {
        mtg_coop_launch((void*)ctx->stream, p->H, p->D, p->K, p->deriv, r.params.B, r.params.times, r.params.ts_b, r.params.ts_k,
                        r.params.dfix, r.params.fs_b, r.params.fs_d, r.params.fs_c, r.params.coeffs, r.params.status, r.params.tstatus);
        continue;
      }
      if (r.rt) {
        r.rt->launch((void*)ctx->stream, r.grid, r.params.times, r.params.dfix, r.params.coeffs, r.params.status, r.params.tstatus,
                     (int)r.params.B, r.params.K, r.ntiles, r.dl_ws, r.dl_aos);
        continue;
      }
      if (r.dl) {
        if (r.params.dfree || r.params.cost)
          r.dl->launch_extra((void*)ctx->stream, r.grid, r.params.times, r.params.dfix, r.params.coeffs, r.params.status,
                             r.params.tstatus, (int)r.params.B, r.ntiles, r.dl_ws, r.dl_aos, r.params.dfree, r.params.cost,
                             r.params.ps_b, r.params.ps_d, r.params.ps_c);
        else
          r.dl->launch((void*)ctx->stream, r.grid, r.params.times, r.params.dfix, r.params.coeffs, r.params.status,
                       r.params.tstatus, (int)r.params.B, r.ntiles, r.dl_ws, r.dl_aos);
        continue;
      }
      // (the cost accumulators are not re-zeroed between the timed launches: values are irrelevant here, and a memset
      // node per iteration would be timed as part of the kernel)
      hipLaunchKernelGGL(r.fn, dim3(r.grid, r.gridy), dim3(kBlock), r.lds, ctx->stream, r.params, r.ntiles);
    }
  }
  MTG_HIP_TRY(ctx, hipEventRecord(e1, ctx->stream));
  MTG_HIP_TRY(ctx, hipEventSynchronize(e1));
  float ms = 0.f;
  MTG_HIP_TRY(ctx, hipEventElapsedTime(&ms, e0, e1));
  *mean_us = (double)ms * 1000.0 / iters;
  return MTG_OK;
}

int mtg_selftest_rcp(mtg_context* ctx, int n, double* max_rel_err) {
  if (!ctx || !max_rel_err || n == 0) return MTG_ERR_INVALID_ARGUMENT;
  std::lock_guard<std::mutex> lock(ctx->mu);
  MTG_HIP_TRY(ctx, hipSetDevice(ctx->device));
  double* d = nullptr;
  MTG_HIP_TRY(ctx, hipMalloc((void**)&d, sizeof(double)));
  MTG_HIP_TRY(ctx, hipMemsetAsync(d, 0, sizeof(double), ctx->stream));
  hipLaunchKernelGGL(mtg_rcp_selftest_kernel, dim3((std::abs(n) + 255) / 256), dim3(256), 0, ctx->stream, std::abs(n), d, n < 0 ? ((-n) & 3) : 2);
  MTG_HIP_TRY(ctx, hipMemcpyAsync(max_rel_err, d, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  MTG_HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  MTG_HIP_TRY(ctx, hipFree(d));
  return MTG_OK;
}

}  // extern "C"
