// mtg_extrema.hip -- batched magnitude extrema + feasibility time scaling.
//
// Replaces, for a batch of solved trajectories (coeffs [B][K][D][N] as written by mtg_solve_linear):
//   * Trajectory::computeMinMaxMagnitude (src/trajectory.cpp:190-227) over Segment::computeMinMaxMagnitude-
//     Candidates / selectMinMaxMagnitudeFromCandidates (src/segment.cpp:83-184)            -> mtg_minmax_magnitude
//   * Trajectory::computeMaxVelocityAndAcceleration (:343-361) + scaleSegmentTimesToMeetConstraints (:385-429)
//                                                                          -> mtg_scale_segment_times_to_meet_constraints
// The per-lane algorithm (real-root isolation by a compile-time derivative chain instead of the reference's
// Jenkins-Traub) lives in mtg_extrema_lane.h.  Mapping: one lane per (trajectory, segment); blockIdx.y selects the
// derivative (velocity / acceleration are searched in one launch for the scaling path).  The kernels are
// FP64-issue bound, not HBM bound: ~10^4 FMAs per lane against 240 B of coefficients read (round 4: two independent chains
// at a time and lane-aligned refinement rounds, mtg_extrema_lane.h).  The two root buffers of each lane live in LDS ([slot][lane]
// layout: lanes in lock-step hit distinct banks); everything else is in registers with compile-time indices.
#include <hip/hip_runtime.h>

#include "../../include/mtg_hip.h"
#include "mtg_extrema_lane.h"

extern "C" int mtg_context_extrema_split(const mtg_context* ctx);   // measurement knob "extrema_split" (mtg_hip_lab.h): -1 auto

namespace {

constexpr int kThreads = 128;   // two root buffers of L - 1 doubles per lane in LDS: 128 x 2 x 21 x 8 B = 43 KB at most

struct ExtremaParams {
  const double* coeffs;   // [B][K][D][N]
  const double* times;    // times[b*ts_b + k*ts_k]
  long long ts_b, ts_k;
  double* seg_out;        // [n_slots][B][K][4]  (t_min, v_min, t_max, v_max), segment-local times
  double* traj_out;       // [n_slots][B][4]
  int* traj_seg;          // [n_slots][B][2] (segment of the minimum, of the maximum) or null
  long long B;
  int N, K, D;
  unsigned mask;
  int der[2];
  int split;              // lanes that share one root search (launch_seg): 1, 2 or 4
  int rolled;             // one code body for all levels of the derivative chain (mtg_extrema_lane.h, Level) or one per level
};

template <int COLS>
struct LdsRoots {
  double* p;   // the search's column: element i at p[i * COLS]
  __device__ double& operator[](int i) { return p[i * COLS]; }
};

// SPLIT lanes share one (trajectory, segment) root search (mtg_extrema_lane.h, Share): a launch with about one wavefront per SIMD
// or fewer is bound by the LATENCY of a lane's chain of refinements (a lone wavefront of one-lane searches takes ~150 us whatever
// the batch), and two / four lanes shorten that chain.  The lanes of a search are neighbours in one wavefront and address the same
// LDS column; lane `part` 0 writes the result.  Results do not depend on SPLIT (bit-identical by construction: a bracket's
// refinement does not depend on which lane refines it; tests/test_extrema.py).
template <int NMAX, int SPLIT, bool ROLLED>
__device__ __forceinline__ void mtg_minmax_seg_body(const ExtremaParams& P, double* lds) {
  constexpr int COLS = kThreads / SPLIT;
  const long long total = P.B * P.K;
  const long long idx = ((long long)blockIdx.x * kThreads + threadIdx.x) / SPLIT;
  const int part = threadIdx.x % SPLIT;
  if (idx >= total) return;
  const int slot = blockIdx.y;
  const long long b = idx / P.K;
  const int seg = (int)(idx - b * P.K);
  const double T = P.times[b * P.ts_b + (long long)seg * P.ts_k];
  LdsRoots<COLS> roots{lds + threadIdx.x / SPLIT};
  const mtgx::MinMax mm = mtgx::segment_minmax<NMAX, LdsRoots<COLS>, ROLLED>(P.coeffs + idx * (long long)(P.D * P.N), P.N, P.D, P.mask,
                                                                             P.der[slot], T, roots, mtgx::Share{part, part + 1, SPLIT});
  if (part != 0) return;
  double* o = P.seg_out + ((long long)slot * total + idx) * 4;
  reinterpret_cast<double2*>(o)[0] = make_double2(mm.t_min, mm.v_min);
  reinterpret_cast<double2*>(o)[1] = make_double2(mm.t_max, mm.v_max);
}

// NMAX0 / NMAX1: compile-time bound on the derivative polynomial's coefficient count of derivative slot 0 / slot 1 (blockIdx.y).
// The time-scaling path searches velocity and acceleration in one launch: with one bound for both, the acceleration slot walked
// the velocity's two extra (identically zero) levels of the derivative chain.
template <int NMAX0, int NMAX1, int SPLIT, bool ROLLED>
__global__ __launch_bounds__(kThreads) void mtg_minmax_seg_kernel(ExtremaParams P) {
  extern __shared__ double lds[];
  if constexpr (NMAX1 == NMAX0) {
    mtg_minmax_seg_body<NMAX0, SPLIT, ROLLED>(P, lds);
  } else {
    if (blockIdx.y == 0) mtg_minmax_seg_body<NMAX0, SPLIT, ROLLED>(P, lds);
    else mtg_minmax_seg_body<NMAX1, SPLIT, ROLLED>(P, lds);
  }
}

// Trajectory::computeMinMaxMagnitude's outer loop (trajectory.cpp:199-225): first segment with a strictly
// smaller / larger value wins.
__global__ void mtg_minmax_traj_kernel(ExtremaParams P) {
  const long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= P.B) return;
  const int slot = blockIdx.y;
  const double* s = P.seg_out + ((long long)slot * P.B + b) * P.K * 4;
  double t_min = 0.0, v_min = DBL_MAX, t_max = 0.0, v_max = -DBL_MAX;
  int k_min = 0, k_max = 0;
  for (int k = 0; k < P.K; ++k) {
    const double2 lo = reinterpret_cast<const double2*>(s)[2 * k];
    const double2 hi = reinterpret_cast<const double2*>(s)[2 * k + 1];
    if (lo.y < v_min) { v_min = lo.y; t_min = lo.x; k_min = k; }
    if (hi.y > v_max) { v_max = hi.y; t_max = hi.x; k_max = k; }
  }
  double* o = P.traj_out + ((long long)slot * P.B + b) * 4;
  reinterpret_cast<double2*>(o)[0] = make_double2(t_min, v_min);
  reinterpret_cast<double2*>(o)[1] = make_double2(t_max, v_max);
  if (P.traj_seg) {
    P.traj_seg[((long long)slot * P.B + b) * 2 + 0] = k_min;
    P.traj_seg[((long long)slot * P.B + b) * 2 + 1] = k_max;
  }
}

struct ScaleParams {
  double* coeffs;
  double* times;
  long long ts_b, ts_k;
  double* seg_out;          // [2][B][K][4]
  double* traj_out;         // [2][B][4]: slot 0 velocity, slot 1 acceleration
  double* scaling;          // [B] or null: product of the applied factors
  int* within;              // [B] or null
  long long B;
  int N, K, D;
  double v_max, a_max;
  int max_iterations;
};

// The WHOLE loop of scaleSegmentTimesToMeetConstraints (trajectory.cpp:392-426) for one trajectory from ONE extrema search:
// stretching every segment time by s is a re-parametrisation p(t / s), so the next round's maxima are known exactly --
// velocity / s, acceleration / s^2, at s times the old instant -- and the reference's re-computation (:397) only reproduces
// them up to round-off.  total: product of the factors applied; total_chk: the factor in force at the LAST check (what the
// tables show afterwards: "the maxima seen by the last round's check").  Round 3 ran the root search again in every round.
struct ScaleLoop { double total, total_chk; bool within; };
__device__ __forceinline__ ScaleLoop mtg_scale_loop(double v_act, double a_act, double v_max, double a_max, int max_iterations) {
  ScaleLoop r{1.0, 1.0, false};
  for (int it = 0; it < max_iterations; ++it) {
    r.total_chk = r.total;
    const double s = mtgx::violation_scaling(v_act, a_act, v_max, a_max, r.within);
    if (r.within) break;   // the reference breaks out before scaling (:405-407)
    r.total *= s;
    v_act /= s;
    a_act /= s * s;
  }
  return r;
}

// lane per (b, segment, dim): Polynomial::scalePolynomialInTime (polynomial.cpp:199-205) + Segment::setTime with the factor
// of the whole loop.  Reads the UNSCALED trajectory maxima; mtg_scale_finish_kernel (launched after it) updates the tables.
__global__ void mtg_scale_kernel(ScaleParams P) {
  const long long total = P.B * P.K * P.D;
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const long long bk = idx / P.D;
  const int d = (int)(idx - bk * P.D);
  const long long b = bk / P.K;
  const int seg = (int)(bk - b * P.K);
  const ScaleLoop r = mtg_scale_loop(P.traj_out[b * 4 + 3], P.traj_out[(P.B + b) * 4 + 3], P.v_max, P.a_max, P.max_iterations);
  if (r.total == 1.0) return;
  const double inv = 1.0 / r.total;
  double* c = P.coeffs + idx * (long long)P.N;
  double scale = 1.0;
  for (int n = 0; n < P.N; ++n) {
    c[n] *= scale;
    scale *= inv;
  }
  if (d == 0) P.times[b * P.ts_b + (long long)seg * P.ts_k] *= r.total;
}

// lane per trajectory, after mtg_scale_kernel: per-trajectory outputs, and the extrema tables as the last round's check saw
// them (instants x factor, velocity / factor, acceleration / factor^2)
__global__ void mtg_scale_finish_kernel(ScaleParams P) {
  const long long b = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= P.B) return;
  const ScaleLoop r = mtg_scale_loop(P.traj_out[b * 4 + 3], P.traj_out[(P.B + b) * 4 + 3], P.v_max, P.a_max, P.max_iterations);
  if (P.within) P.within[b] = r.within ? 1 : 0;
  if (P.scaling) P.scaling[b] = r.total;
  const double f = r.total_chk;
  if (f == 1.0) return;
  const double inv = 1.0 / f;
  for (int slot = 0; slot < 2; ++slot) {
    const double vs = slot == 0 ? inv : inv * inv;
    double* t = P.traj_out + ((long long)slot * P.B + b) * 4;
    t[0] *= f; t[1] *= vs; t[2] *= f; t[3] *= vs;
    double* s = P.seg_out + ((long long)slot * P.B + b) * P.K * 4;
    for (int k = 0; k < P.K; ++k) { s[4 * k] *= f; s[4 * k + 1] *= vs; s[4 * k + 2] *= f; s[4 * k + 3] *= vs; }
  }
}

template <int NMAX, int NMAX1 = NMAX>
void launch_seg(const ExtremaParams& P, int n_slots, hipStream_t stream) {
  constexpr int L = 2 * NMAX - 2;
  const long long total = P.B * P.K;
  const dim3 grid((unsigned)((P.split * total + kThreads - 1) / kThreads), n_slots);
  const size_t lds = (size_t)(kThreads / P.split) * 2 * (L - 1) * sizeof(double);   // two root buffers per search (mtg_extrema_lane.h)
#define MTG_XL(S, R) hipLaunchKernelGGL((mtg_minmax_seg_kernel<NMAX, R ? NMAX : NMAX1, S, R>), grid, dim3(kThreads), lds, stream, P)
  if (P.rolled) {   // (measurement knob: one or two lanes, one bound for all slots)
    if (P.split == 2) MTG_XL(2, true); else MTG_XL(1, true);
  } else if (P.split == 4) MTG_XL(4, false);
  else if (P.split == 2) MTG_XL(2, false);
  else MTG_XL(1, false);
#undef MTG_XL
}

// Lanes per search by launch size (round 5; profiles/r05_extrema_lanes_per_search.jsonl, N = 10 / K = 8 / D = 3, velocity search, us):
//   wavefronts at one lane per search      313     625    1024    1250    1500    2048
//   one lane                               149     185     188     193     197     215
//   two lanes                              136     138     140     183     222     244
//   four lanes                             112     145     178     209     237     316
// A wavefront costs the same whatever its number of live lanes, so more lanes per search cost instructions (two: +39 %) and pay
// only while the chip has SIMDs to spare -- up to one one-lane wavefront per SIMD (1024).  At 1250 (10k x 8 segments) two lanes
// measured 179 vs 194 us in one process and 224 vs 207 us inside bench.py: not a default there.  (Also measured and dropped: the
// whole rounds of wavefronts at one lane per search and only the SURPLUS searches on four lanes, in one launch -- 186 us at 1250.)
constexpr long long kFourLanesMaxWaves = 400, kTwoLanesMaxWaves = 1100;
int launch_minmax(ExtremaParams& P, int n_slots, hipStream_t stream, int option) {
  // option (measurement knob "extrema_split"): -1 default; bits 0-1: lanes per search (1, 2, 3 = four; 0 = by launch size); bit 2:
  // ONE code body for all levels of the derivative chain.  Measured (profiles/r04f_extrema_variants.jsonl, 10k x 8 segments):
  // per-level bodies 181 / 177 us (one / two lanes per search), one body 221 / 211 us -- the 60 KB of straight-line code are NOT
  // what bounds the kernel (its zero-padded chains cost more than the instruction fetches they save): per-level bodies stay.
  const int split_option = option < 0 ? 0 : (option & 3);
  P.rolled = option < 0 ? 0 : ((option & 4) ? 1 : 0);
  const long long waves = (P.B * P.K + 63) / 64 * n_slots;   // at one lane per search, all derivative slots of the launch
  P.split = split_option == 3 ? 4 : (split_option != 0 ? split_option : (waves <= kFourLanesMaxWaves ? 4 : (waves <= kTwoLanesMaxWaves ? 2 : 1)));
  if (P.rolled && P.split == 4) P.split = 2;
  int n_d = 0;
  for (int s = 0; s < n_slots; ++s) {
    const int nd = P.N - P.der[s];
    if (nd > n_d) n_d = nd;
  }
  // (velocity + acceleration of the time-scaling path: slot 1's polynomial is one coefficient shorter)
  const bool shorter1 = n_slots == 2 && P.N - P.der[0] == n_d && P.N - P.der[1] == n_d - 1;
  if (n_d <= 7) launch_seg<7>(P, n_slots, stream);
  else if (n_d <= 8) { if (shorter1) launch_seg<8, 7>(P, n_slots, stream); else launch_seg<8>(P, n_slots, stream); }
  else if (n_d <= 9) { if (shorter1) launch_seg<9, 8>(P, n_slots, stream); else launch_seg<9>(P, n_slots, stream); }
  else if (n_d <= 10) { if (shorter1) launch_seg<10, 9>(P, n_slots, stream); else launch_seg<10>(P, n_slots, stream); }
  else if (n_d <= 11) { if (shorter1) launch_seg<11, 10>(P, n_slots, stream); else launch_seg<11>(P, n_slots, stream); }
  else launch_seg<12>(P, n_slots, stream);
  if (P.traj_out)
    hipLaunchKernelGGL(mtg_minmax_traj_kernel, dim3((unsigned)((P.B + 255) / 256), n_slots), dim3(256), 0, stream, P);
  return hipGetLastError() == hipSuccess ? MTG_OK : MTG_ERR_DEVICE;
}

int check_common(mtg_context* ctx, int n_coeffs, int n_segments, int dimension, int64_t batch, const void* coeffs,
                 const void* times) {
  if (!ctx || !coeffs || !times || batch < 0 || n_coeffs < 2 || n_coeffs > MTG_MAX_N || n_segments < 1 ||
      dimension < 1 || dimension > 32)
    return MTG_ERR_INVALID_ARGUMENT;
  return MTG_OK;
}

}  // namespace

extern "C" int mtg_context_stream_device(mtg_context* ctx, void** stream, int* device);

extern "C" int mtg_minmax_magnitude(mtg_context* ctx, int32_t n_coeffs, int32_t n_segments, int32_t dimension,
                                    int64_t batch, const double* coeffs, const double* times, int64_t times_stride_b,
                                    int64_t times_stride_k, int32_t derivative, uint32_t dimension_mask,
                                    double* segment_minmax, double* trajectory_minmax, int32_t* trajectory_segment_idx) {
  int rc = check_common(ctx, n_coeffs, n_segments, dimension, batch, coeffs, times);
  if (rc != MTG_OK) return rc;
  // N - derivative - 1 >= 0 (polynomial.cpp:70-73); a trajectory-level result needs the per-segment table
  if (!segment_minmax || derivative < 0 || n_coeffs - derivative - 1 < 0) return MTG_ERR_INVALID_ARGUMENT;
  const uint32_t all = dimension >= 32 ? 0xffffffffu : ((1u << dimension) - 1u);
  if (dimension_mask == 0) dimension_mask = all;
  if (dimension_mask & ~all) return MTG_ERR_INVALID_ARGUMENT;   // "dimensions out of bounds" (segment.cpp:102-107)
  if (batch == 0) return MTG_OK;
  void* stream = nullptr;
  int device = 0;
  rc = mtg_context_stream_device(ctx, &stream, &device);
  if (rc != MTG_OK) return rc;
  if (hipSetDevice(device) != hipSuccess) return MTG_ERR_DEVICE;
  ExtremaParams P;
  P.coeffs = coeffs; P.times = times; P.ts_b = times_stride_b; P.ts_k = times_stride_k;
  P.seg_out = segment_minmax; P.traj_out = trajectory_minmax; P.traj_seg = trajectory_minmax ? trajectory_segment_idx : nullptr;
  P.B = batch; P.N = n_coeffs; P.K = n_segments; P.D = dimension; P.mask = dimension_mask;
  P.der[0] = derivative; P.der[1] = derivative;
  return launch_minmax(P, 1, (hipStream_t)stream, mtg_context_extrema_split(ctx));
}

extern "C" int mtg_scale_segment_times_to_meet_constraints(mtg_context* ctx, int32_t n_coeffs, int32_t n_segments,
                                                           int32_t dimension, int64_t batch, double* coeffs,
                                                           double* times, int64_t times_stride_b, int64_t times_stride_k,
                                                           double v_max, double a_max, int32_t max_iterations,
                                                           double* workspace, double* scaling, int32_t* within_range) {
  int rc = check_common(ctx, n_coeffs, n_segments, dimension, batch, coeffs, times);
  if (rc != MTG_OK) return rc;
  if (!workspace || max_iterations < 1 || !(v_max > 0.0) || !(a_max > 0.0) || n_coeffs < 3) return MTG_ERR_INVALID_ARGUMENT;
  if (batch == 0) return MTG_OK;
  void* stream = nullptr;
  int device = 0;
  rc = mtg_context_stream_device(ctx, &stream, &device);
  if (rc != MTG_OK) return rc;
  if (hipSetDevice(device) != hipSuccess) return MTG_ERR_DEVICE;
  ExtremaParams P;
  P.coeffs = coeffs; P.times = times; P.ts_b = times_stride_b; P.ts_k = times_stride_k;
  P.seg_out = workspace;
  P.traj_out = workspace + (size_t)2 * batch * n_segments * 4;
  P.traj_seg = nullptr;
  P.B = batch; P.N = n_coeffs; P.K = n_segments; P.D = dimension;
  P.mask = dimension >= 32 ? 0xffffffffu : ((1u << dimension) - 1u);   // "whatever dimensions we have" (trajectory.cpp:346-347)
  P.der[0] = 1;   // derivative_order::VELOCITY
  P.der[1] = 2;   // derivative_order::ACCELERATION
  ScaleParams S;
  S.coeffs = coeffs; S.times = times; S.ts_b = times_stride_b; S.ts_k = times_stride_k; S.seg_out = P.seg_out; S.traj_out = P.traj_out;
  S.scaling = scaling; S.within = within_range; S.B = batch; S.N = n_coeffs; S.K = n_segments; S.D = dimension;
  S.v_max = v_max; S.a_max = a_max; S.max_iterations = max_iterations;
  // ONE root search (velocity and acceleration in one launch), then the whole check / stretch loop analytically (mtg_scale_loop)
  rc = launch_minmax(P, 2, (hipStream_t)stream, mtg_context_extrema_split(ctx));
  if (rc != MTG_OK) return rc;
  const long long total = (long long)batch * n_segments * dimension;
  hipLaunchKernelGGL(mtg_scale_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, S);
  hipLaunchKernelGGL(mtg_scale_finish_kernel, dim3((unsigned)((batch + 255) / 256)), dim3(256), 0, (hipStream_t)stream, S);
  return hipGetLastError() == hipSuccess ? MTG_OK : MTG_ERR_DEVICE;
}
