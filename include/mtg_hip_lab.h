/* mtg_hip_lab.h -- measurement knobs of libmtg_hip.so.  NOT part of the drop-in boundary (include/mtg_hip.h): nothing a
 * consumer of the solver needs lives here.  The library never reads the environment; A/B measurements (tools/) and the
 * tests that force one kernel form against another set these per context, by name.  The Python layer
 * (mav_trajectory_generation_amd.Context) forwards the environment variables MTG_<NAME> of the process it runs in.
 *
 *   name               value                       effect (default = shipped behaviour)
 *   "force_dg"         1..4, 0 = off               dimension-group size of the specialised fused / split kernels
 *   "prefer_rolled"    0 / 1                       rolled (run-time K) kernel even where a static one exists
 *   "no_dimlane"       0 / 1                       never the dimension-in-lane forms
 *   "no_slab"          0 / 1                       fused form without the slab-output kernel
 *   "no_slab_extra"    0 / 1                       extra outputs (cost / d_P) through the older fused kernel
 *   "no_dl_extra"      0 / 1                       extra outputs never through the dimension-in-lane kernels
 *   "no_queue"         0 / 1                       mtg_solve_linear_sequence as one launch per batch
 *   "no_balance"       0 / 1                       persistent grids not evened out over their rounds
 *   "dl_rt"            -1 default, 0 never, 1 always   run-time-K dimension-in-lane body
 *   "dl_grid_per_cu"   >= 1 (default 8)            workgroups per CU of a non-workspace dimension-in-lane launch
 *   "dl_any_sched_rr"  0 / 1                       round 2's unit schedule of the cross-structure launch
 *   "slab_policy"      -1 default, 0 / 1           store policy of the slab-output kernels (write-back / nt sc1)
 *   "rolled_wg_per_cu" >= 1 (default 4)            persistent workgroups per CU of the rolled kernels
 *   "dl_max_units"     -1 default, >= 0            upper limit of the dimension-in-lane default range (x CUs)
 *   "coop"             -1 default, 0 never, 1 always   row-cooperative form where eligible
 *   "extrema_split"    -1 default; bits 0-1: lanes that share one root search of the extrema kernels (1, 2, 3 = four; 0 = by
 *                      launch size, the default); bit 2: one code body for all levels of the derivative chain
 *   "sample_generic"   0 / 1                       mtg_sample_range never through its compile-time-shape kernels
 * Returns MTG_OK, or MTG_ERR_INVALID_ARGUMENT for an unknown name.                                                  */
#ifndef MTG_HIP_LAB_H_
#define MTG_HIP_LAB_H_
#include "mtg_hip.h"
#ifdef __cplusplus
extern "C" {
#endif
int mtg_context_set_option(mtg_context* ctx, const char* name, int value);

/* EVIDENCE variant of one step of the path (SURVEY.md section 7 "K1-alt"; the MFMA clause of BASELINE.json's north_star) -- no
 * solve entry point goes through it.  The per-segment cost matrices  H_k = A_k^-T Q_k A_k^-1  of
 * impl/polynomial_optimization_linear_impl.h:318 (A: LIN:112-121, its inverse LIN:143-179, Q: LIN:568-583) for `n_segments`
 * segment times, h_out [n_segments][n_coeffs][n_coeffs] row-major:
 *   variant 1   LITERALLY: both N x N x N products on the FP64 matrix cores (v_mfma_f64_16x16x4_f64, N <= 12 padded to one
 *               16 x 16 tile), one wavefront per segment, A^-1 and Q assembled per segment;
 *   variant 0   by the unit-time scaling identity the solve kernels use (H(T) = T^(1-2d) S H(1) S, constant table).
 * n_coeffs even in 2..12, 0 <= derivative < n_coeffs / 2; device pointers; asynchronous on the context's stream.
 * Measured: the literal form is 4.9x slower and reaches 5.5 % of the FP64 MFMA peak (profiles/r01_literal_h_mfma.txt);
 * checked against the 50-digit oracle in tests/test_gpu_literal_mfma.py.                                              */
int mtg_lab_segment_cost_matrices(mtg_context* ctx, int32_t n_coeffs, int32_t derivative, int64_t n_segments,
                                  const double* times, double* h_out, int32_t variant);

/* Shader clock while the caller's work runs: one wavefront on a stream of its own compares the shader-clock counter (s_memtime)
 * with the constant 100 MHz counter (s_memrealtime) over `duration_us`.  start returns immediately (the probe is enqueued);
 * finish waits for the probe, returns MHz and the probed interval, and releases it.  bench.py brackets its `sustained` run
 * with it (the FP64-heavy solve kernels run power-limited: 1.75-1.96 GHz instead of the 2.4 GHz the peaks are quoted at).   */
/* The residual of MTG_FLAG_REFINE alone: rhs_out [batch][D][n_free] (contiguous, device) = -(R_PP d_free + R_PF d_fixed), formed in
 * double-double (csrc/mtg_refine.hip) from device buffers in `layout`; asynchronous on the context's stream.  The tests compare it
 * with the residual formed at 50 digits.                                                                                       */
int mtg_lab_refine_residual(mtg_plan* plan, int64_t batch, const mtg_layout* layout, const double* times, const double* d_fixed,
                            const double* d_free, double* rhs_out);
typedef struct mtg_lab_clock_probe mtg_lab_clock_probe;
int mtg_lab_clock_probe_start(mtg_context* ctx, double duration_us, mtg_lab_clock_probe** out);
int mtg_lab_clock_probe_finish(mtg_lab_clock_probe* probe, double* shader_mhz, double* measured_us);
#ifdef __cplusplus
}
#endif
#endif
