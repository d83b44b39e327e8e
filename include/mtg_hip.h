/*
 * mtg_hip.h -- C ABI of libmtg_hip.so: batched PolynomialOptimization<N>::solveLinear()
 * for AMD MI355X (gfx950).  Eigen-free, torch-free: plain pointers and sizes.
 *
 * This is the drop-in boundary for the reference's linear-optimiser hot path.  The
 * reference exposes that path only as a header-only C++ class template, so "what the
 * reference's FFI would bind" is the set of member functions below; each entry point
 * cites the reference interface it replaces, with
 *   LINH = mav_trajectory_generation/include/mav_trajectory_generation/polynomial_optimization_linear.h
 *   LIN  = mav_trajectory_generation/include/mav_trajectory_generation/impl/polynomial_optimization_linear_impl.h
 *
 * Data conventions (all IEEE double):
 *   N  = coefficients per polynomial (even, 2..12; polynomial.h:44 kMaxN), h = N/2
 *   K  = segments, D = dimensions, B = independent trajectories in the batch
 *   fixed_mask[v] bit p set  <=>  derivative p is a fixed constraint at vertex v
 *       (a vertex keeps exactly h slots; LIN:206-221.  Constraints of order >= h are
 *        dropped by the reference, LIN:84-105 -- callers drop them before building masks)
 *   d_fixed : values of the fixed slots, ordered lexicographically by (vertex, derivative)
 *             = the reference's fixed_constraints_compact_ (LINH:288-295, LIN:238-246)
 *   d_free  : the free slots in the same order = free_constraints_compact_ (LINH:194-206)
 *   coeffs  : [B][K][D][N], increasing powers c0..c(N-1) (LINH:43-44, polynomial.h:34-35),
 *             i.e. segments_[k][dim].getCoefficients() of trajectory b (LIN:276-280)
 * Input layouts are described by element strides so that both
 *   AoS  times[B][K], d_fixed[B][D][n_fixed]   (what a host caller naturally holds) and
 *   SoA  times[K][B], d_fixed[D][n_fixed][B]   (fully coalesced on the device)
 * are accepted; mtg_layout_aos()/mtg_layout_soa() fill the stride block.
 *
 * Error convention: every function returns MTG_OK (0) or a negative mtg_status; nothing
 * aborts or throws (the reference CHECK-aborts; the C++ veneer maps codes back to that).
 */
#ifndef MTG_HIP_H_
#define MTG_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MTG_MAX_N 12

typedef enum mtg_status {
  MTG_OK = 0,
  MTG_ERR_INVALID_ARGUMENT = -1, /* LIN:60-65, :76, :289, :502-504 CHECKs                  */
  MTG_ERR_BAD_SEGMENT_TIME = -2, /* LIN:297 CHECK_GT(segment_time, 0)                      */
  MTG_ERR_SINGULAR = -3,         /* non-positive pivot in R_PP (rank-deficient problem)    */
  MTG_ERR_DEVICE = -4,           /* HIP runtime error; see mtg_last_error_string           */
  MTG_ERR_NO_DEVICE = -5,        /* no gfx950 device / HIP runtime unusable                */
  MTG_ERR_UNSUPPORTED = -6
} mtg_status;

typedef struct mtg_context mtg_context; /* (device, stream, scratch) -- one per host thread */
typedef struct mtg_plan mtg_plan;       /* a7 gather tables for one (N, K, D, d, masks)     */

/* Replaces the constructor + setupFromVertices() structure part: LINH:57, LIN:57-109 and
 * setupConstraintReorderingMatrix LIN:182-260 (pure indexing, done once per plan). */
typedef struct mtg_plan_desc {
  int32_t n_coeffs;               /* N                                                      */
  int32_t dimension;              /* D (LINH:57)                                            */
  int32_t n_segments;             /* K = vertices - 1 (LIN:72)                              */
  int32_t derivative_to_optimize; /* d in [0, N/2-1] (LIN:60-65)                            */
  const uint32_t* fixed_mask;     /* [K+1], see above                                       */
} mtg_plan_desc;

typedef struct mtg_plan_info {
  int32_t n_all;          /* getNumberAllConstraints()   LINH:218 = N*K                     */
  int32_t n_fixed;        /* getNumberFixedConstraints() LINH:219                           */
  int32_t n_free;         /* getNumberFreeConstraints()  LINH:220                           */
  int32_t kernel_variant; /* 0 generic (run-time K/masks), 1 static, 2 static dim-split only, 3 rolled */
  int64_t algorithmic_bytes_per_trajectory; /* 8*(K + D*n_fixed + K*D*N), SURVEY 8(d)       */
} mtg_plan_info;

typedef struct mtg_layout {
  int64_t times_stride_b, times_stride_k;               /* times[b*sb + k*sk]               */
  int64_t fixed_stride_b, fixed_stride_d, fixed_stride_c; /* d_fixed[b*sb + dim*sd + col*sc] */
  int64_t free_stride_b, free_stride_d, free_stride_c;    /* d_free likewise                 */
} mtg_layout;

enum {
  MTG_FLAG_HOST_POINTERS = 1u << 0, /* all buffers are host memory: the library stages them  */
  MTG_FLAG_GENERIC_KERNEL = 1u << 1, /* force the generic kernel (tests / A-B measurements)  */
  MTG_FLAG_FUSED_DIMS = 1u << 2,     /* all D dimensions in one workgroup (large batches)     */
  MTG_FLAG_SPLIT_DIMS = 1u << 3,     /* one dimension group per workgroup (small batches);    */
                                     /* default: chosen from the batch size                   */
  MTG_FLAG_COST_ONLY = 1u << 4,      /* only cost[] is produced (coeffs may be NULL): the     */
                                     /* objective evaluations of the time optimisers          */
                                     /* (polynomial_optimization_nonlinear_impl.h:313-359,    */
                                     /* :569-571) need J = computeCost(), not the segments    */
  MTG_FLAG_DIMLANE = 1u << 5,        /* force the dimension-in-lane launch form where the     */
                                     /* plan and the call are eligible (canonical SoA or AoS  */
                                     /* inputs -- mtg_layout_soa / mtg_layout_aos --, coeffs  */
                                     /* only); default: chosen from the batch size            */
  MTG_FLAG_HOST_BACKEND = 1u << 6,   /* with MTG_FLAG_HOST_POINTERS and batch <=               */
                                     /* MTG_HOST_BACKEND_MAX_BATCH: solve on the calling      */
                                     /* thread with the host build of the kernels' lane code  */
                                     /* (no launch, no PCIe): the single-trajectory calls of  */
                                     /* the reference's nlopt loops (polynomial_optimization_ */
                                     /* nonlinear_impl.h:569-571).  Ignored otherwise.        */
  MTG_FLAG_CONCURRENT_ITEMS = 1u << 7, /* mtg_multi_create: no merging -- every item runs as its */
                                     /* own best launch, spread over the context's side       */
                                     /* streams (longest chains first), forked from and       */
                                     /* joined back onto the context's stream                 */
  MTG_FLAG_QUERY_EXTRA_OUTPUTS = 1u << 9, /* mtg_plan_launch_form only: the form of a call that */
                                     /* also asks for the cost and / or d_free                */
  MTG_FLAG_SEQUENCE_ONE_LAUNCH_PER_BATCH = 1u << 8, /* mtg_solve_linear_sequence: never merge  */
                                     /* the queue into one persistent launch (latency of the  */
                                     /* single launches; A/B measurements)                    */
  MTG_FLAG_COOPERATIVE = 1u << 11,   /* force the row-cooperative launch form (16 lanes per   */
                                     /* trajectory-half: the LATENCY form for small launches  */
                                     /* of long chains) where the plan and the call are       */
                                     /* eligible (standard shapes, D = 3, coefficients only); */
                                     /* default: chosen from the batch size and chain length  */
  MTG_FLAG_REFINE = 1u << 12,        /* mtg_solve_linear / _status, device pointers: ONE step */
                                     /* of iterative refinement on the free derivatives --    */
                                     /* the residual -(R_PP d_P + R_PF d_F) formed in double- */
                                     /* double from the exact unit-time table (csrc/          */
                                     /* mtg_refine.hip), the correction solved and the        */
                                     /* coefficients recovered by the ordinary float64        */
                                     /* kernels.  For problems whose float64 solution is      */
                                     /* conditioning-limited (N = 12, d < N/2 - 1: every      */
                                     /* float64 evaluation, the reference's own included, is  */
                                     /* 1e-8 .. 2e-6 from the 50-digit solution) the result   */
                                     /* is good to ~1e-13.  An ACCURACY mode: five launches,  */
                                     /* the correction through the generic kernel -- measured */
                                     /* 10-25x a plain solve (N = 12 / K = 32 at 100k: 5.2 ms */
                                     /* against 0.5 ms); asynchronous.  Not with COST_ONLY /  */
                                     /* HOST_POINTERS / BASIC_SOLUTION.                       */
  MTG_FLAG_BASIC_SOLUTION = 1u << 10 /* mtg_solve_linear / _status: reference behaviour on     */
                                     /* RANK-DEFICIENT free systems (LIN:365-378: the rank-   */
                                     /* revealing SparseQR returns a basic solution and       */
                                     /* solveLinear() returns true).  Trajectories the sweep  */
                                     /* flags singular are solved on the host (column-pivoted */
                                     /* QR of the dense R_PP, Eigen's rank threshold, free    */
                                     /* variables beyond the rank zero) and their outputs     */
                                     /* replaced; the call then reports MTG_OK for them (bit 2 */
                                     /* stays set in trajectory_status: WHICH ones were basic).*/
                                     /* The call is SYNCHRONOUS and carries its OWN status     */
                                     /* word (host and device pointers alike): flags raised   */
                                     /* by earlier asynchronous launches stay in the context  */
                                     /* for the next mtg_context_sync.  Not with              */
                                     /* MTG_FLAG_COST_ONLY.  WHICH minimiser: a structurally  */
                                     /* deficient plan is solved through its shadow -- the    */
                                     /* LOWEST free slots (vertex, derivative < d) that       */
                                     /* complete the fixed ones pinned to zero -- i.e. ONE of */
                                     /* the minimisers: same cost and constraints as the      */
                                     /* reference's basic solution, differing from it by an   */
                                     /* element of the cost's null space (one polynomial of   */
                                     /* degree < d over the whole trajectory); which columns  */
                                     /* SparseQR drops depends on COLAMD's order.             */
                                     /* mtg_solve_linear_sequence* and mtg_multi_create accept */
                                     /* the flag and stay ASYNCHRONOUS: structurally deficient */
                                     /* plans run on their shadow (a regular plan of the      */
                                     /* queue / request like any other); there is no per-     */
                                     /* trajectory host fall-back there -- a trajectory whose */
                                     /* factorisation breaks down stays flagged in the        */
                                     /* context's status word.                                */
};
#define MTG_HOST_BACKEND_MAX_BATCH 64

/* ---- context ------------------------------------------------------------------------- */
/* stream: a hipStream_t (as void*) to enqueue on, or NULL for the library's own stream.   */
int mtg_context_create(int device, void* stream, mtg_context** out);
int mtg_context_destroy(mtg_context* ctx);
/* Waits for everything enqueued on the context; returns MTG_ERR_BAD_SEGMENT_TIME /
 * MTG_ERR_SINGULAR if any trajectory of a device-pointer solve since the last sync raised it
 * (also solves replayed from a captured hipGraph: the status word is read from the device on
 * every call).  Host-pointer calls report their status themselves.                        */
int mtg_context_sync(mtg_context* ctx);
const char* mtg_last_error_string(const mtg_context* ctx);
const char* mtg_status_string(int status);

/* ---- plan ---------------------------------------------------------------------------- */
int mtg_plan_create(mtg_context* ctx, const mtg_plan_desc* desc, mtg_plan** out);
int mtg_plan_destroy(mtg_plan* plan);
int mtg_plan_get_info(const mtg_plan* plan, mtg_plan_info* out);
/* The context a plan was created on (its stream, status word and error text): host objects that hold a plan and may be
 * used from another thread than their creator synchronise THIS context, not their own thread's.                     */
mtg_context* mtg_plan_context(const mtg_plan* plan);
/* (N, D, K, d) of the description the plan was created from (any out-pointer may be NULL).                                  */
int mtg_plan_get_shape(const mtg_plan* plan, int32_t* n_coeffs, int32_t* dimension, int32_t* n_segments, int32_t* derivative_to_optimize);
/* STRUCTURAL rank deficiency of the plan's free system R_PP (0: regular).  The cost's null space is the polynomials of degree
 * < derivative_to_optimize over the whole trajectory; what the fixed slots leave of it is a property of the constraint pattern,
 * not of a batch's values (under-constrained problems: fewer than d independent position / velocity / ... constraints in the
 * whole trajectory).  Every trajectory of such a plan is flagged MTG_ERR_SINGULAR (bit 1 of the per-trajectory status) by every
 * solve -- the reference's rank-revealing SparseQR returns a basic solution there (LIN:365-378): ask for it with
 * MTG_FLAG_BASIC_SOLUTION -- such a plan is then solved through its "shadow" (the same pattern with that many more slots
 * fixed to zero: a regular system, on the device like any other; free variables beyond the rank come back as exact zeros).
 * On regular plans the kernels flag a trajectory only when the factorisation breaks down (a non-positive or NaN pivot).
 * LIMITATION: the rank is computed at GENERIC vertex instants.  A Birkhoff-type pattern (a derivative fixed without the lower
 * ones) can be regular generically and singular at a batch's actual times -- e.g. d = 3, ends position-only, the middle vertex
 * velocity-only, EQUAL segment times: p(0), p'(T), p(2T) are dependent on the quadratics.  Such a plan reports 0 here, and a
 * trajectory with exactly those times is caught only by the breakdown guard (a pivot of round-off size and either sign), where
 * the reference's QR still returns a basic solution.  Hermite-type patterns (every fixed derivative with all lower ones: every
 * pattern of the reference's tests and generators) are not affected.                                                        */
int mtg_plan_rank_deficiency(const mtg_plan* plan);
/* The same number from the description alone (host arithmetic only: no context, no device): fixed_mask[n_segments + 1] as in
 * mtg_plan_desc.  Negative: mtg_status.                                                                                      */
int mtg_structural_rank_deficiency(int32_t n_coeffs, int32_t n_segments, int32_t derivative_to_optimize, const uint32_t* fixed_mask);
/* Which kernel form a device-pointer mtg_solve_linear(plan, batch, layout, ..., flags) call with coefficient output only
 * (with MTG_FLAG_QUERY_EXTRA_OUTPUTS: a call that also returns the cost / d_free) takes on this device: 0 generic (run-time K / masks), 1 fused static, 2 dimension-split static, 3 rolled (run-time K),
 * 4 fused with slab output (whole-sector stores), 5 dimension-in-lane (one unrolled body per chain length), 6 dimension-in-lane
 * with a run-time chain length (one body per polynomial order), 7 row-cooperative (16 lanes per trajectory-half; small
 * launches of long chains).  Negative: mtg_status.                                                                     */
int mtg_plan_launch_form(const mtg_plan* plan, int64_t batch, const mtg_layout* layout, uint32_t flags);
void mtg_layout_aos(const mtg_plan* plan, int64_t batch, mtg_layout* out);
void mtg_layout_soa(const mtg_plan* plan, int64_t batch, mtg_layout* out);
/* SoA with the row stride padded to the next multiple of 16 trajectories (times[K][Bs], d_fixed[D][n_fixed][Bs], d_free likewise;
 * Bs = (batch + 15) & ~15): every 16-trajectory row piece starts on a 128-byte boundary whatever the batch size -- with the plain
 * SoA stride a batch of 12 500 (100 000-byte rows) reads 1.75x its input bytes.  Read by the dimension-in-lane kernels like
 * the two canonical layouts (any other kernel takes it as the general strided layout it is).                              */
void mtg_layout_soa_padded(const mtg_plan* plan, int64_t batch, mtg_layout* out);

/* Optional: caller-owned scratch for the generic kernels' back-substitution workspace (no allocation inside
 * mtg_solve_linear then, e.g. for graph capture).  bytes == 0 restores library-managed scratch.               */
int mtg_plan_set_workspace(mtg_plan* plan, void* device_ptr, size_t bytes);

/* ---- device memory for host code that must not see HIP headers --------------------------
 * (the reference is host-only: its containers live in host memory, LINH:249-283.)  Allocation on
 * the context's device; the copies are ordered on the context's stream and return when done.  */
int mtg_device_malloc(mtg_context* ctx, size_t bytes, void** device_ptr);
int mtg_device_free(mtg_context* ctx, void* device_ptr);
int mtg_copy_to_device(mtg_context* ctx, void* dst_device, const void* src_host, size_t bytes);
int mtg_copy_to_host(mtg_context* ctx, void* dst_host, const void* src_device, size_t bytes);

/* ---- the hot path -------------------------------------------------------------------- */
/* Replaces updateSegmentTimes() + solveLinear() (LINH:101,108; LIN:286-305, :339-379,
 * incl. constructR :308-336 and updateSegmentsFromCompactConstraints :263-283) for `batch`
 * trajectories sharing one plan.  Asynchronous on the context's stream.
 *   times   in   segment times (> 0)
 *   d_fixed in   fixed constraint values
 *   coeffs  out  [batch][K][D][N]
 *   d_free  out  optional (NULL): optimised free constraints  (getFreeConstraints LINH:194)
 *   cost    out  optional (NULL): [batch], computeCost() LIN:124-140 = 0.5 * sum c^T Q c  */
int mtg_solve_linear(mtg_plan* plan, int64_t batch, const mtg_layout* layout,
                     const double* times, const double* d_fixed, double* coeffs,
                     double* d_free, double* cost, uint32_t flags);

/* mtg_solve_linear with a per-trajectory status output: trajectory_status[b] (int32, optional, NULL = none) receives
 * 0, or the OR of 1 (a segment time <= 0: LIN:297 CHECK_GT) and 2 (non-positive pivot: rank-deficient free system,
 * where the reference's rank-revealing SparseQR LIN:365-367 would still return a basic solution) for trajectory b --
 * the batch-wide codes MTG_ERR_BAD_SEGMENT_TIME / MTG_ERR_SINGULAR say THAT a trajectory failed, this says WHICH.
 * The library zero-fills it.  Device pointer, or host pointer with MTG_FLAG_HOST_POINTERS.
 * With MTG_FLAG_HOST_POINTERS every solve / update entry is synchronous and returns the batch status itself.     */
int mtg_solve_linear_status(mtg_plan* plan, int64_t batch, const mtg_layout* layout,
                            const double* times, const double* d_fixed, double* coeffs,
                            double* d_free, double* cost, int32_t* trajectory_status, uint32_t flags);

/* The basic solution of ONE trajectory's free system on the host (what MTG_FLAG_BASIC_SOLUTION applies to the flagged
 * trajectories of a batch; LIN:360-375 with a rank-revealing factorisation): times [K], d_fixed [D][n_fixed] ->
 * d_free [D][n_free] (host pointers); *rank (optional) receives the numerical rank of R_PP (n_free: regular system).
 * Coefficients follow from mtg_update_segments_from_free (LIN:263-283).                                          */
int mtg_basic_solution_host(const mtg_plan* plan, const double* times, const double* d_fixed, double* d_free, int32_t* rank);

/* A queue of n INDEPENDENT batches of the same plan (same batch size, layout and flags): solve i reads times[i] /
 * d_fixed[i] and writes coeffs[i] (host arrays of n DEVICE pointers; no batch may read what another one writes), enqueued
 * on the context's stream by ONE host call -- what a pipeline that streams batch after batch through the solver does (the
 * reference's counterpart is a loop over setupFromVertices + solveLinear, polynomial_timing_evaluation.cpp:104-110).
 * Because the batches are independent the library overlaps them: plans with a slab-output kernel (the standard shapes)
 * run the whole queue as ONE persistent launch whose workgroups walk the tiles of all batches, batch-major (the pointer
 * triples travel in the kernel arguments, up to 96 batches per launch): no drain / launch gap between batches, the store
 * tail of batch i under the elimination of batch i + 1.  Results are bit-identical to n mtg_solve_linear calls -- EXCEPT where a
 * single call of that batch size would take the row-cooperative form by default (mtg_plan_launch_form = 7: N = 12 / K >= 16,
 * N = 10 / K >= 64, N = 8 / K >= 80 in launches of at most one or two 4-trajectory workgroups per CU): that form eliminates in
 * another order, and the queue (and mtg_multi_*) agrees with it to round-off x conditioning, not bit for bit.  Other
 * plans, and calls with MTG_FLAG_SEQUENCE_ONE_LAUNCH_PER_BATCH, enqueue one launch per batch back to back.           */
int mtg_solve_linear_sequence(mtg_plan* plan, int32_t n, int64_t batch, const mtg_layout* layout,
                              const double* const* times, const double* const* d_fixed, double* const* coeffs,
                              uint32_t flags);
/* The same with two caller-owned hipEvent_t (either may be NULL) recorded on the context's stream immediately before the
 * first and after the last launch, inside the call: a caller that times the queue with events does not measure its own
 * latency between recording the start event and enqueueing the first launch (several microseconds from Python -- as
 * much as a kernel of this library).                                                                                  */
int mtg_solve_linear_sequence_events(mtg_plan* plan, int32_t n, int64_t batch, const mtg_layout* layout,
                                     const double* const* times, const double* const* d_fixed, double* const* coeffs,
                                     uint32_t flags, void* start_event, void* stop_event);

/* Replaces updateSegmentTimes() + setFreeConstraints() (LIN:500-508): coefficients from
 * caller-provided free constraints, no solve (the nonlinear optimiser's path,
 * polynomial_optimization_nonlinear_impl.h:695-696).  d_free is an INPUT here.            */
int mtg_update_segments_from_free(mtg_plan* plan, int64_t batch, const mtg_layout* layout,
                                  const double* times, const double* d_fixed,
                                  const double* d_free, double* coeffs, double* cost,
                                  uint32_t flags);

/* ---- the time optimisers' cost / gradient step ------------------------------------------------
 * Replaces, for a batch, PolynomialOptimizationNonLinear<N>::getCostAndGradientMellinger
 * (impl/polynomial_optimization_nonlinear_impl.h:287-364): per trajectory the reference runs K + 1 updateSegmentTimes +
 * solveLinear + computeCost -- the current times, and for each segment n the times with T_n += h and every other
 * T_i -= h / (K - 1) (h = increment_time, 0.1 in the reference), clamped from below to time_lower_bound
 * (kOptimizationTimeLowerBound = 0.1, polynomial_optimization_nonlinear.h:31) -- and returns J_d and the forward
 * differences (J_n - J_d) / h.  Here all (K + 1) * batch solves are ONE cost-only launch; the perturbed times are formed
 * in the kernel from `times` (nothing is materialised), a second small kernel forms the differences.
 *   cost     out optional [batch]: J_d
 *   gradient out [batch][K] with the strides of `times` in `layout` (gradient[b*times_stride_b + n*times_stride_k])
 * K == 1: zero gradient (impl:295-302).  Device pointers; asynchronous on the context's stream; flags raised by any of
 * the virtual problems surface at mtg_context_sync.                                                            */
int mtg_mellinger_cost_gradient(mtg_plan* plan, int64_t batch, const mtg_layout* layout, const double* times,
                                const double* d_fixed, double increment_time, double time_lower_bound, double* cost,
                                double* gradient);

/* ---- next step after the path: batched sampling ---------------------------------------- */
/* Replaces, for a batch, Trajectory::evaluateRange (src/trajectory.cpp:81-141) / Polynomial::evaluate
 * (polynomial.h:137-149) as used by sampleTrajectoryInRange (src/trajectory_sampling.cpp:45-110):
 *   out[b][i][der][dim] = der-th derivative of dimension dim at t_i = t_start + i*dt,  i < n_samples,
 * der < n_derivatives (5 = position .. snap), from coeffs[batch][K][D][N] and times[b*stride_b + k*stride_k].
 * Samples beyond a trajectory's end are evaluated at its end; n_valid[b] (optional) = number of samples with
 * t_i <= total time.  Device pointers (out 16-byte aligned); asynchronous on the context's stream.          */
int mtg_sample_range(mtg_context* ctx, int32_t n_coeffs, int32_t n_segments, int32_t dimension, int64_t batch,
                     const double* coeffs, const double* times, int64_t times_stride_b, int64_t times_stride_k,
                     double t_start, double dt, int32_t n_samples, int32_t n_derivatives, double* out,
                     int32_t* n_valid);

/* ---- next step after the path: batched magnitude extrema and feasibility time scaling ---- */
/* Replaces, for a batch, Trajectory::computeMinMaxMagnitude (src/trajectory.cpp:190-227) over
 * Segment::computeMinMaxMagnitudeCandidates / selectMinMaxMagnitudeFromCandidates (src/segment.cpp:83-184):
 * extrema over [0, T_k] of  sqrt(sum_{dim in mask} (p_dim^(derivative)(t))^2)  for every segment.
 *   dimension_mask       bit d set = dimension d enters the magnitude (0 = all `dimension` of them); with exactly
 *                        one bit the candidates are the critical points of that polynomial (segment.cpp:124-131)
 *   segment_minmax   out [batch][K][4] = (t_min, v_min, t_max, v_max), times relative to the segment start
 *                        (Extremum::time, extremum.h:40-41); required, 16-byte aligned
 *   trajectory_minmax out optional [batch][4]: first segment with the strictly smallest / largest value
 *   trajectory_segment_idx out optional [batch][2] = (Extremum::segment_idx of the minimum, of the maximum)
 * The real roots of the magnitude derivative are isolated directly instead of running Jenkins-Traub
 * (src/rpoly/rpoly_ak1.cpp) -- same extrema values; extremum TIMES agree only where the root is well conditioned.
 * Needs n_coeffs - derivative - 1 >= 0 (polynomial.cpp:70-73).  Device pointers; asynchronous.               */
int mtg_minmax_magnitude(mtg_context* ctx, int32_t n_coeffs, int32_t n_segments, int32_t dimension, int64_t batch,
                         const double* coeffs, const double* times, int64_t times_stride_b, int64_t times_stride_k,
                         int32_t derivative, uint32_t dimension_mask, double* segment_minmax,
                         double* trajectory_minmax, int32_t* trajectory_segment_idx);

/* Replaces, for a batch, Trajectory::scaleSegmentTimesToMeetConstraints(v_max, a_max) (src/trajectory.cpp:385-429)
 * on top of computeMaxVelocityAndAcceleration (:343-361): `max_iterations` rounds of { maximum |velocity| and
 * |acceleration| over all dimensions; if either exceeds its bound by more than 1e-3 relative, stretch every
 * segment time by s = max(1, v_act/v_max, sqrt(a_act/a_max)) and rescale the coefficients by (1/s)^n }.
 * The reference allows up to 20 rounds but stretching by s meets both bounds exactly, so round 2 only verifies;
 * pass 2 for the reference's result, 1 to skip the verification.  In place on coeffs [batch][K][D][N] and times.
 *   workspace     [8 * batch * (K + 1)] doubles, 16-byte aligned; afterwards holds the last round's
 *                 per-segment tables [2][batch][K][4] and per-trajectory extrema [2][batch][4] (0 = velocity,
 *                 1 = acceleration), evaluated BEFORE that round's scaling
 *   scaling   out optional [batch]: total factor applied to the segment times
 *   within_range out optional [batch]: the reference's return value (1 = bounds met at the last check)       */
int mtg_scale_segment_times_to_meet_constraints(mtg_context* ctx, int32_t n_coeffs, int32_t n_segments,
                                                int32_t dimension, int64_t batch, double* coeffs, double* times,
                                                int64_t times_stride_b, int64_t times_stride_k, double v_max,
                                                double a_max, int32_t max_iterations, double* workspace,
                                                double* scaling, int32_t* within_range);

/* ---- mixed requests: several plans in one launch ------------------------------------------
 * What a caller of the reference does with a list of independent PolynomialOptimization<N>
 * problems of different structure (BASELINE config 4: N in {8, 10, 12}, 4..32 segments): each
 * structure is a plan + a batch.  A mixed request is created once -- items with canonical SoA
 * (times[K][B], d_fixed[D][n_fixed][B]) or AoS inputs and coefficient output only whose plan has a
 * static dimension-in-lane configuration (the config-4 shapes: N = 8 / 10 / 12, K = 4 / 8 / 16 /
 * 32, D = 3) join ONE cross-structure launch whatever their N and K; of the others, items that
 * share D, the start / interior / end constraint pattern and the derivative (any K >= 2; N = 8 /
 * 10 / 12 standard shapes also across N) are merged into ONE launch of the run-time-K kernels
 * whose tiles are ordered longest-chain-first -- and solved as often as wanted
 * with new values in the same device buffers (same structure, new segment times: every
 * iteration of a time optimiser).  mtg_multi_solve only enqueues kernels (and memsets of the
 * cost outputs) on the context's stream: it can be stream-captured into a hipGraph.
 * Items that cannot be merged run as ordinary mtg_solve_linear launches.
 * Device pointers only; all plans must belong to `ctx`.                                       */
typedef struct mtg_multi mtg_multi;
typedef struct mtg_multi_item {
  mtg_plan* plan;
  int64_t batch;
  mtg_layout layout;
  const double* times;     /* device */
  const double* d_fixed;   /* device */
  double* coeffs;          /* device, 16-byte aligned, [batch][K][D][N] */
  double* d_free;          /* device, optional */
  double* cost;            /* device, optional */
} mtg_multi_item;
/* flags: 0, or MTG_FLAG_FUSED_DIMS / MTG_FLAG_SPLIT_DIMS to fix the launch geometry of the merged groups (a caller
 * that runs several mixed requests concurrently knows the total load; the default looks at this request only);
 * or MTG_FLAG_CONCURRENT_ITEMS: one launch per item through the ordinary variant choice (static / dimension-in-lane
 * kernels where the plan has them), the items overlapping on the device on up to 4 internal streams.               */
int mtg_multi_create(mtg_context* ctx, int32_t n_items, const mtg_multi_item* items, uint32_t flags, mtg_multi** out);
int mtg_multi_solve(mtg_multi* multi);              /* asynchronous on the context's stream */
int mtg_multi_launch_count(const mtg_multi* multi); /* kernel launches one mtg_multi_solve enqueues */
int mtg_multi_destroy(mtg_multi* multi);

/* ---- synthetic inputs and result check on the device (no tensor library needed) -------------
 * mtg_generate_waypoints: fills `times` / `d_fixed` (device pointers, any layout) of `batch` trajectories of the plan's
 * structure with the reference's random-waypoint recipe (src/vertex.cpp:27-82 createRandomVertices: positions uniform in
 * [-box, box]^D, consecutive vertices more than 0.2 apart, start / goal at rest; :255-272 estimateSegmentTimesNfabian with
 * the default magic constant 6.5); other fixed derivatives of interior vertices: random direction, magnitude <= v_max /
 * a_max / 1.  yaw_dimension != 0 with D = 4: the last dimension is an angle in [-3 pi, 3 pi].  Counter-based random
 * numbers: reproducible per (seed, trajectory), distribution-equivalent (not bit-equal) to the mt19937 original.
 * Asynchronous on the context's stream.
 * mtg_compare_coefficients: max over polynomials of ||a - b||_inf / ||b||_inf (the parity tests' norm-wise measure) and the
 * max absolute difference of two device buffers of n_polynomials x n_coeffs doubles; synchronous.                      */
int mtg_generate_waypoints(mtg_plan* plan, int64_t batch, const mtg_layout* layout, uint64_t seed, double box, double v_max,
                           double a_max, int32_t yaw_dimension, double* times, double* d_fixed);
int mtg_compare_coefficients(mtg_context* ctx, const double* a, const double* b, int64_t n_polynomials, int32_t n_coeffs,
                             double* max_normwise_rel, double* max_abs);

/* ---- several GPUs from one host process -----------------------------------------------------
 * The path shards by independent trajectories (LIN:339-379 touches only one optimiser's state): shard s of G is the
 * contiguous range mtg_shard_range(batch, G, s), solved on device s by that device's own context / plan / stream, with
 * no communication during the solve.  A device group = one context + one plan of the same description per device.
 * mtg_device_group_solve_linear fans one call out (asynchronous: the shards run concurrently); the caller holds the
 * per-device buffers (times[s], d_fixed[s], coeffs[s]: device pointers on device s, shard-local SoA (soa != 0) or AoS).
 * mtg_device_group_gather_coeffs is the optional final gather into one [batch][K][D][N] buffer: host memory (root < 0)
 * or memory of device `root` (peer copies over xGMI).  `devices` = NULL: devices 0 .. n-1.  The multi-process form
 * (one rank per GPU, RCCL all_gather over xGMI) is mav_trajectory_generation_amd/dist.py; both split the batch alike. */
typedef struct mtg_device_group mtg_device_group;
void mtg_shard_range(int64_t batch, int32_t n_shards, int32_t shard, int64_t* lo, int64_t* hi);
int mtg_device_group_create(int32_t n_devices, const int32_t* devices, const mtg_plan_desc* desc, mtg_device_group** out);
int mtg_device_group_destroy(mtg_device_group* group);
int mtg_device_group_size(const mtg_device_group* group);
mtg_context* mtg_device_group_context(mtg_device_group* group, int32_t shard);
mtg_plan* mtg_device_group_plan(mtg_device_group* group, int32_t shard);
int mtg_device_group_solve_linear(mtg_device_group* group, int64_t batch, int32_t soa, const double* const* times,
                                  const double* const* d_fixed, double* const* coeffs, uint32_t flags);
int mtg_device_group_sync(mtg_device_group* group);
int mtg_device_group_gather_coeffs(mtg_device_group* group, int64_t batch, const double* const* coeffs, int32_t root,
                                   double* dst);

/* ---- one PROCESS per GPU: the final gather over RCCL / xGMI (csrc/mtg_comm.hip) ----------------------------------------
 * north_star: batches shard across the GPUs of a node "with RCCL over xGMI only for the final gather".  Rank r of `world`
 * processes owns the slice mtg_shard_range(B, world, r), solves it with its own context / plan (no communication), and -- when
 * one consumer needs every shard on every device -- gathers the coefficient shards through a communicator:
 *   rank 0: mtg_comm_unique_id(id); ships the MTG_COMM_UNIQUE_ID_BYTES bytes to the other ranks (file, socket, MPI ...);
 *   every rank: mtg_comm_create(ctx, rank, world, id, &comm)   -- collective (ncclCommInitRank on the context's device);
 *   mtg_comm_all_gather(comm, local, n, gathered)              -- gathered[r][n] <- rank r's local[n] (equal n on every rank);
 *   mtg_comm_solve_all_gather(...)                             -- the batch in n_chunks pieces, chunk i's gather (on the
 *       communicator's own stream) under chunk i + 1's solve (on the context's stream): at 240 MB per rank the gather takes
 *       ~10x the solve (7 xGMI links x ~153 GB/s per GPU), so the solve hides behind it, not the other way round.
 *       gathered is chunk-major: [n_chunks][world][batch / n_chunks][K][D][N]; batch equal on every rank, a multiple of n_chunks.
 * Everything is asynchronous (ordered after the context's stream, which in turn waits for the last gather); mtg_comm_sync waits
 * for both streams and reports the context's status.  librccl.so is opened on first use (dlopen) -- a single-GPU consumer neither
 * links nor loads it; MTG_ERR_UNSUPPORTED where it is absent.  No RCCL types cross the boundary.                              */
typedef struct mtg_comm mtg_comm;
#define MTG_COMM_UNIQUE_ID_BYTES 128
int mtg_comm_unique_id(void* out_id /* MTG_COMM_UNIQUE_ID_BYTES bytes */);
int mtg_comm_create(mtg_context* ctx, int32_t rank, int32_t world, const void* unique_id, mtg_comm** out);
int mtg_comm_destroy(mtg_comm* comm);
int mtg_comm_rank(const mtg_comm* comm);
int mtg_comm_world(const mtg_comm* comm);
const char* mtg_comm_last_error(const mtg_comm* comm);
int mtg_comm_all_gather(mtg_comm* comm, const double* local, int64_t n_doubles, double* gathered);
int mtg_comm_solve_all_gather(mtg_comm* comm, mtg_plan* plan, int64_t batch, const mtg_layout* layout, const double* times,
                              const double* d_fixed, double* local_coeffs, double* gathered, int32_t n_chunks, uint32_t flags);
int mtg_comm_sync(mtg_comm* comm);

/* ---- measurement hooks (bench.py / tests) --------------------------------------------- */
/* Re-runs the last mtg_solve_linear launch of this plan `iters` times back-to-back on the
 * context's stream, bracketed by hipEvents recorded on that same stream; returns the mean
 * kernel-side duration per launch in microseconds.  (Queue launches of mtg_solve_linear_sequence
 * are not recorded: MTG_ERR_INVALID_ARGUMENT after one -- time those with the _events form.)   */
int mtg_time_last_solve(mtg_plan* plan, int iters, double* mean_us);
/* Accuracy self-test of the device reciprocal used by the LDL^T pivots: max relative error
 * of rcp(x) vs 1/x over n pseudo-random positive doubles.  (n < 0: diagnostic -- |n| samples with
 * (|n| & 3) Newton steps instead of the shipped two.)                                      */
int mtg_selftest_rcp(mtg_context* ctx, int n, double* max_rel_err);

#ifdef __cplusplus
}
#endif
#endif /* MTG_HIP_H_ */
