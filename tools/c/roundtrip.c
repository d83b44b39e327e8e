/* Plain C consumer of libmtg_hip.so (no HIP headers, no tensor library): generate a random-waypoint batch on the device,
 * solve it with the default kernel choice and with the generic kernel, compare the two results on the device, time the
 * default launch; then a QUEUE of four independent batches in one call (mtg_solve_linear_sequence: one persistent launch) against
 * the same batches solved one by one.  What a cgo / JNI / FFI binding of include/mtg_hip.h does, as one file.
 * build: gcc -std=c11 -O2 -Iinclude tools/c/roundtrip.c -Lmav_trajectory_generation_amd/csrc -lmtg_hip \
 *            -Wl,-rpath,'$ORIGIN/../../mav_trajectory_generation_amd/csrc' -o tools/c/roundtrip
 * usage: roundtrip [batch = 100000] [segments = 8]                                                                   */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "mtg_hip.h"

#define CHECK(call)                                                                                          \
  do {                                                                                                       \
    int rc_ = (call);                                                                                        \
    if (rc_ != MTG_OK) {                                                                                     \
      fprintf(stderr, "%s -> %d (%s): %s\n", #call, rc_, mtg_status_string(rc_), mtg_last_error_string(ctx)); \
      return 1;                                                                                              \
    }                                                                                                        \
  } while (0)

int main(int argc, char** argv) {
  const int64_t batch = argc > 1 ? atoll(argv[1]) : 100000;
  const int K = argc > 2 ? atoi(argv[2]) : 8;
  enum { N = 10, D = 3, DERIV = 4 };
  mtg_context* ctx = NULL;
  if (mtg_context_create(0, NULL, &ctx) != MTG_OK) { fprintf(stderr, "no HIP device\n"); return 2; }
  uint32_t* mask = (uint32_t*)malloc((size_t)(K + 1) * sizeof(uint32_t));
  for (int v = 0; v <= K; ++v) mask[v] = (v == 0 || v == K) ? 31u : 1u;   /* ends fully fixed, interior: position only */
  mtg_plan_desc desc = {N, D, K, DERIV, mask};
  mtg_plan* plan = NULL;
  CHECK(mtg_plan_create(ctx, &desc, &plan));
  mtg_plan_info info;
  CHECK(mtg_plan_get_info(plan, &info));
  mtg_layout lay;
  mtg_layout_soa(plan, batch, &lay);
  void *times = NULL, *dfix = NULL, *ca = NULL, *cb = NULL;
  const size_t ncoef = (size_t)batch * K * D * N;
  CHECK(mtg_device_malloc(ctx, (size_t)batch * K * sizeof(double), &times));
  CHECK(mtg_device_malloc(ctx, (size_t)batch * D * info.n_fixed * sizeof(double), &dfix));
  CHECK(mtg_device_malloc(ctx, ncoef * sizeof(double), &ca));
  CHECK(mtg_device_malloc(ctx, ncoef * sizeof(double), &cb));
  CHECK(mtg_generate_waypoints(plan, batch, &lay, 2024, 10.0, 3.0, 5.0, 0, (double*)times, (double*)dfix));
  CHECK(mtg_solve_linear(plan, batch, &lay, (const double*)times, (const double*)dfix, (double*)cb, NULL, NULL, MTG_FLAG_GENERIC_KERNEL));
  CHECK(mtg_solve_linear(plan, batch, &lay, (const double*)times, (const double*)dfix, (double*)ca, NULL, NULL, 0));
  CHECK(mtg_context_sync(ctx));                     /* reports a bad segment time / singular system of any trajectory */
  double rel = 0.0, abs_ = 0.0, us = 0.0;
  CHECK(mtg_compare_coefficients(ctx, (const double*)ca, (const double*)cb, (int64_t)batch * K * D, N, &rel, &abs_));
  CHECK(mtg_time_last_solve(plan, 50, &us));
  double first[N];
  CHECK(mtg_copy_to_host(ctx, first, ca, sizeof(first)));
  printf("batch %lld, %d segments: default vs generic kernel max norm-wise rel diff %.3e (abs %.3e); %.2f us per launch = %.3g "
         "trajectories/s, %.1f%% of 8 TB/s; c[0][0][x] = %.6f %.6f %.6f ...\n",
         (long long)batch, K, rel, abs_, us, (double)batch / us * 1e6,
         100.0 * (double)batch * (double)info.algorithmic_bytes_per_trajectory / us * 1e-3 / 8000.0, first[0], first[1], first[2]);
  /* a queue of independent batches: four quarter-size batches, each with its own inputs and outputs */
  double rel_q = 0.0;
  {
    enum { NQ = 4 };
    const int64_t bq = batch / NQ > 0 ? batch / NQ : 1;
    mtg_layout lq;
    mtg_layout_soa(plan, bq, &lq);
    void *tq[NQ], *fq[NQ], *cq[NQ], *cs[NQ];
    const double* tp[NQ]; const double* fp[NQ]; double* cp[NQ];
    for (int i = 0; i < NQ; ++i) {
      CHECK(mtg_device_malloc(ctx, (size_t)bq * K * sizeof(double), &tq[i]));
      CHECK(mtg_device_malloc(ctx, (size_t)bq * D * info.n_fixed * sizeof(double), &fq[i]));
      CHECK(mtg_device_malloc(ctx, (size_t)bq * K * D * N * sizeof(double), &cq[i]));
      CHECK(mtg_device_malloc(ctx, (size_t)bq * K * D * N * sizeof(double), &cs[i]));
      CHECK(mtg_generate_waypoints(plan, bq, &lq, 7000 + (uint64_t)i, 10.0, 3.0, 5.0, 0, (double*)tq[i], (double*)fq[i]));
      tp[i] = (const double*)tq[i]; fp[i] = (const double*)fq[i]; cp[i] = (double*)cq[i];
    }
    CHECK(mtg_solve_linear_sequence(plan, NQ, bq, &lq, tp, fp, cp, 0));                 /* one call, one launch */
    for (int i = 0; i < NQ; ++i)
      CHECK(mtg_solve_linear(plan, bq, &lq, tp[i], fp[i], (double*)cs[i], NULL, NULL, 0));   /* one by one */
    CHECK(mtg_context_sync(ctx));
    for (int i = 0; i < NQ; ++i) {
      double r = 0.0, a = 0.0;
      CHECK(mtg_compare_coefficients(ctx, cp[i], (const double*)cs[i], bq * K * D, N, &r, &a));
      if (r > rel_q) rel_q = r;
      mtg_device_free(ctx, tq[i]); mtg_device_free(ctx, fq[i]); mtg_device_free(ctx, cq[i]); mtg_device_free(ctx, cs[i]);
    }
    printf("queue of %d x %lld trajectories in one call vs one call each: max norm-wise rel diff %.3e\n", NQ, (long long)bq, rel_q);
  }
  mtg_device_free(ctx, times); mtg_device_free(ctx, dfix); mtg_device_free(ctx, ca); mtg_device_free(ctx, cb);
  mtg_plan_destroy(plan);
  mtg_context_destroy(ctx);
  free(mask);
  return (rel < 1e-10 && rel_q < 1e-10) ? 0 : 3;
}
