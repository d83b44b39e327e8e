// mtg_comm.hip -- one PROCESS per GPU: the final gather of the coefficient shards over RCCL / xGMI, behind the C ABI.
//
// north_star: "batches of independent trajectories shard embarrassingly across the 8 GPUs of one node with RCCL over xGMI only
// for the final gather".  The solve needs no communication: rank r owns the contiguous slice
// mtg_shard_range(B, world, r) and solves it with its own context / plan.  What a C++ consumer that runs one process per GPU
// (the reference's ROS nodes are single processes: mav_trajectory_generation_ros/src/time_evaluation_node.cpp:348-357) lacked in
// rounds 1-4 was the gather itself -- the only RCCL path went through torch.distributed (mav_trajectory_generation_amd/dist.py).
// Here: a communicator bound to a context (ncclCommInitRank on the context's device), an all-gather of equal-sized shards, and
// the chunked solve + gather in which chunk i's ncclAllGather runs on the communicator's own stream while chunk i + 1 is solved
// on the context's stream (at 240 MB per rank -- config 3 -- the gather takes ~10x the solve: 7 xGMI links x ~153 GB/s per GPU,
// point-to-point, so the chunks are sized for the links, not for a switch).
//
// librccl.so is opened on first use (dlopen): a single-GPU consumer of libmtg_hip.so neither links nor loads it.  The entry
// points take no RCCL types: the unique id is an opaque 128-byte blob the caller ships from rank 0 to the other ranks over
// whatever channel it has (a file, a socket, MPI).
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/mtg_hip.h"

extern "C" int mtg_context_stream_device(mtg_context* ctx, void** stream, int* device);

namespace {

// the handful of RCCL entry points used, with their C signatures (rccl/rccl.h: ncclResult_t is an int enum, ncclSuccess = 0;
// ncclUniqueId is a 128-byte struct passed BY VALUE to ncclCommInitRank; ncclFloat64 = 8)
struct UniqueId { char bytes[MTG_COMM_UNIQUE_ID_BYTES]; };
typedef int (*GetUniqueIdFn)(UniqueId*);
typedef int (*CommInitRankFn)(void** comm, int nranks, UniqueId id, int rank);
typedef int (*CommDestroyFn)(void* comm);
typedef int (*AllGatherFn)(const void* send, void* recv, size_t count, int dtype, void* comm, hipStream_t stream);
typedef const char* (*GetErrorStringFn)(int);
constexpr int kNcclFloat64 = 8;

struct Rccl {
  void* lib = nullptr;
  GetUniqueIdFn get_unique_id = nullptr;
  CommInitRankFn comm_init_rank = nullptr;
  CommDestroyFn comm_destroy = nullptr;
  AllGatherFn all_gather = nullptr;
  GetErrorStringFn error_string = nullptr;
  std::string why;
};

Rccl* rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, []() {
    for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
      r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (r.lib) break;
    }
    if (!r.lib) { r.why = std::string("librccl.so not found: ") + dlerror(); return; }
    r.get_unique_id = (GetUniqueIdFn)dlsym(r.lib, "ncclGetUniqueId");
    r.comm_init_rank = (CommInitRankFn)dlsym(r.lib, "ncclCommInitRank");
    r.comm_destroy = (CommDestroyFn)dlsym(r.lib, "ncclCommDestroy");
    r.all_gather = (AllGatherFn)dlsym(r.lib, "ncclAllGather");
    r.error_string = (GetErrorStringFn)dlsym(r.lib, "ncclGetErrorString");
    if (!r.get_unique_id || !r.comm_init_rank || !r.comm_destroy || !r.all_gather) r.why = "librccl.so lacks an expected entry point";
  });
  return r.why.empty() ? &r : nullptr;
}

}  // namespace

struct mtg_comm {
  mtg_context* ctx = nullptr;
  void* comm = nullptr;           // ncclComm_t
  int rank = 0, world = 1, device = 0;
  hipStream_t solve_stream = nullptr;   // the context's stream
  hipStream_t comm_stream = nullptr;    // the gathers' own stream
  std::vector<hipEvent_t> solved;       // chunk c has been solved (recorded on the context's stream)
  hipEvent_t gathered = nullptr;        // the last gather has been enqueued (joined back onto the context's stream)
  std::string last_error;
};

static int comm_err(mtg_comm* c, int code, const std::string& msg) {
  if (c) c->last_error = msg;
  return code;
}

extern "C" {

int mtg_comm_unique_id(void* out_id) {
  if (!out_id) return MTG_ERR_INVALID_ARGUMENT;
  Rccl* r = rccl();
  if (!r) return MTG_ERR_UNSUPPORTED;
  UniqueId id;
  std::memset(&id, 0, sizeof(id));
  if (r->get_unique_id(&id) != 0) return MTG_ERR_DEVICE;
  std::memcpy(out_id, &id, sizeof(id));
  return MTG_OK;
}

int mtg_comm_create(mtg_context* ctx, int32_t rank, int32_t world, const void* unique_id, mtg_comm** out) {
  if (!ctx || !out || !unique_id || world < 1 || rank < 0 || rank >= world) return MTG_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  Rccl* r = rccl();
  if (!r) return MTG_ERR_UNSUPPORTED;
  mtg_comm* c = new (std::nothrow) mtg_comm();
  if (!c) return MTG_ERR_DEVICE;
  c->ctx = ctx; c->rank = rank; c->world = world;
  void* st = nullptr;
  if (mtg_context_stream_device(ctx, &st, &c->device) != MTG_OK) { delete c; return MTG_ERR_INVALID_ARGUMENT; }
  c->solve_stream = (hipStream_t)st;
  UniqueId id;
  std::memcpy(&id, unique_id, sizeof(id));
  if (hipSetDevice(c->device) != hipSuccess || hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&c->gathered, hipEventDisableTiming) != hipSuccess) {
    mtg_comm_destroy(c);
    return MTG_ERR_DEVICE;
  }
  const int rc = r->comm_init_rank(&c->comm, world, id, rank);      // collective: every rank of the job calls it
  if (rc != 0) {
    c->comm = nullptr;
    mtg_comm_destroy(c);
    return MTG_ERR_DEVICE;
  }
  *out = c;
  return MTG_OK;
}

int mtg_comm_destroy(mtg_comm* c) {
  if (!c) return MTG_OK;
  hipSetDevice(c->device);
  if (c->comm_stream) hipStreamSynchronize(c->comm_stream);
  if (c->comm) { Rccl* r = rccl(); if (r) r->comm_destroy(c->comm); }
  for (hipEvent_t e : c->solved) hipEventDestroy(e);
  if (c->gathered) hipEventDestroy(c->gathered);
  if (c->comm_stream) hipStreamDestroy(c->comm_stream);
  delete c;
  return MTG_OK;
}

int mtg_comm_rank(const mtg_comm* c) { return c ? c->rank : MTG_ERR_INVALID_ARGUMENT; }
int mtg_comm_world(const mtg_comm* c) { return c ? c->world : MTG_ERR_INVALID_ARGUMENT; }
const char* mtg_comm_last_error(const mtg_comm* c) { return c ? c->last_error.c_str() : ""; }

// gathered[r][n_doubles] <- rank r's local[n_doubles], every rank the same count; ordered after the work already queued on the
// context's stream, runs on the communicator's stream, and the context's stream waits for it (asynchronous: mtg_context_sync or
// mtg_comm_sync to wait on the host).
static int all_gather_on_comm_stream(mtg_comm* c, const double* local, int64_t n_doubles, double* gathered, hipEvent_t after) {
  Rccl* r = rccl();
  if (hipStreamWaitEvent(c->comm_stream, after, 0) != hipSuccess) return comm_err(c, MTG_ERR_DEVICE, "hipStreamWaitEvent");
  const int rc = r->all_gather(local, gathered, (size_t)n_doubles, kNcclFloat64, c->comm, c->comm_stream);
  if (rc != 0) return comm_err(c, MTG_ERR_DEVICE, std::string("ncclAllGather: ") + (r->error_string ? r->error_string(rc) : "error"));
  return MTG_OK;
}

static int ensure_events(mtg_comm* c, size_t n) {
  while (c->solved.size() < n) {
    hipEvent_t e;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return comm_err(c, MTG_ERR_DEVICE, "hipEventCreate");
    c->solved.push_back(e);
  }
  return MTG_OK;
}

int mtg_comm_all_gather(mtg_comm* c, const double* local, int64_t n_doubles, double* gathered) {
  if (!c || !local || !gathered || n_doubles < 0) return MTG_ERR_INVALID_ARGUMENT;
  if (n_doubles == 0) return MTG_OK;
  if (hipSetDevice(c->device) != hipSuccess) return comm_err(c, MTG_ERR_DEVICE, "hipSetDevice");
  int rc = ensure_events(c, 1);
  if (rc != MTG_OK) return rc;
  if (hipEventRecord(c->solved[0], c->solve_stream) != hipSuccess) return comm_err(c, MTG_ERR_DEVICE, "hipEventRecord");
  rc = all_gather_on_comm_stream(c, local, n_doubles, gathered, c->solved[0]);
  if (rc != MTG_OK) return rc;
  if (hipEventRecord(c->gathered, c->comm_stream) != hipSuccess || hipStreamWaitEvent(c->solve_stream, c->gathered, 0) != hipSuccess)
    return comm_err(c, MTG_ERR_DEVICE, "joining the communicator's stream failed");
  return MTG_OK;
}

// This rank's `batch` trajectories solved in n_chunks pieces, each piece all-gathered as soon as it is solved: chunk i's gather
// (communicator's stream) overlaps chunk i + 1's solve (context's stream).  batch must be the same on every rank and a
// multiple of n_chunks.  Inputs in ANY layout: chunk c reads trajectories [c Bc, (c + 1) Bc) through the layout's batch
// strides.  local_coeffs [batch][K][D][N]; gathered is chunk-major like dist.ChunkedSolveGather:
// gathered[c][r][i] = rank r's trajectory c Bc + i, i.e. [n_chunks][world][Bc][K][D][N].
int mtg_comm_solve_all_gather(mtg_comm* c, mtg_plan* plan, int64_t batch, const mtg_layout* layout, const double* times,
                              const double* d_fixed, double* local_coeffs, double* gathered, int32_t n_chunks, uint32_t flags) {
  if (!c || !plan || !layout || !times || !local_coeffs || !gathered || batch < 0 || n_chunks < 1) return MTG_ERR_INVALID_ARGUMENT;
  if (mtg_plan_context(plan) != c->ctx) return comm_err(c, MTG_ERR_INVALID_ARGUMENT, "the plan lives on another context than the communicator");
  if (flags & (MTG_FLAG_HOST_POINTERS | MTG_FLAG_COST_ONLY | MTG_FLAG_BASIC_SOLUTION)) return MTG_ERR_INVALID_ARGUMENT;
  if (batch == 0) return MTG_OK;
  if (batch % n_chunks) return comm_err(c, MTG_ERR_INVALID_ARGUMENT, "batch must be a multiple of n_chunks");
  int32_t N = 0, D = 0, K = 0;
  int rc = mtg_plan_get_shape(plan, &N, &D, &K, nullptr);
  if (rc != MTG_OK) return rc;
  if (hipSetDevice(c->device) != hipSuccess) return comm_err(c, MTG_ERR_DEVICE, "hipSetDevice");
  rc = ensure_events(c, (size_t)n_chunks);
  if (rc != MTG_OK) return rc;
  const int64_t bc = batch / n_chunks;
  const int64_t per_chunk = bc * (int64_t)K * D * N;       // doubles per chunk: coeffs [Bc][K][D][N]
  // (a failure in chunk i leaves the gathers of chunks < i in flight on the communicator's stream: the solve stream is joined to it on
  // EVERY way out, so that whatever the caller enqueues next on the context runs behind them)
  auto join = [&]() {
    return hipEventRecord(c->gathered, c->comm_stream) == hipSuccess && hipStreamWaitEvent(c->solve_stream, c->gathered, 0) == hipSuccess;
  };
  for (int32_t ch = 0; ch < n_chunks; ++ch) {
    // Chunk = the trajectories [ch bc, (ch + 1) bc) of the caller's buffers through the layout's batch strides.  For AoS inputs a chunk
    // is itself a canonical AoS batch and takes the same kernels as the whole batch; for canonical SoA inputs the rows keep the
    // FULL batch's stride, which the dimension-in-lane / queue kernels do not take (their row stride is the batch size or its
    // multiple of 16): such chunks run the strided fused kernels -- correct, slower.  Callers that want the fast path with SoA data
    // hand over pre-chunked buffers (dist.ChunkedSolveGather does) or AoS.
    const double* t = times + ch * bc * layout->times_stride_b;
    const double* f = d_fixed ? d_fixed + ch * bc * layout->fixed_stride_b : nullptr;
    double* co = local_coeffs + ch * per_chunk;
    rc = mtg_solve_linear(plan, bc, layout, t, f, co, nullptr, nullptr, flags);
    if (rc != MTG_OK) { join(); return rc; }
    if (hipEventRecord(c->solved[ch], c->solve_stream) != hipSuccess) { join(); return comm_err(c, MTG_ERR_DEVICE, "hipEventRecord"); }
    rc = all_gather_on_comm_stream(c, co, per_chunk, gathered + (int64_t)ch * c->world * per_chunk, c->solved[ch]);
    if (rc != MTG_OK) { join(); return rc; }
  }
  if (!join()) return comm_err(c, MTG_ERR_DEVICE, "joining the communicator's stream failed");
  return MTG_OK;
}

int mtg_comm_sync(mtg_comm* c) {
  if (!c) return MTG_ERR_INVALID_ARGUMENT;
  if (hipSetDevice(c->device) != hipSuccess || hipStreamSynchronize(c->comm_stream) != hipSuccess) return comm_err(c, MTG_ERR_DEVICE, "hipStreamSynchronize");
  return mtg_context_sync(c->ctx);
}

}  // extern "C"
