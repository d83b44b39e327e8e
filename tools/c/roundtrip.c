/* Plain C consumer of libmtg_hip.so (no HIP headers, no tensor library): generate a random-waypoint batch on the device,
 * solve it with the default kernel choice and with the generic kernel, compare the two results on the device, time the
 * default launch; then a QUEUE of four independent batches in one call (mtg_solve_linear_sequence: one persistent launch) against
 * the same batches solved one by one; an under-constrained batch with MTG_FLAG_BASIC_SOLUTION; the chunked solve + RCCL all-gather
 * of a one-process-per-GPU job (mtg_comm_*).  What a cgo / JNI / FFI binding of include/mtg_hip.h does, as one file.
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
  /* an UNDER-CONSTRAINED problem (one segment, only the two end positions fixed): the reference's rank-revealing SparseQR
   * returns a basic solution and solveLinear() returns true (LIN:365-378).  Without the flag the call reports
   * MTG_ERR_SINGULAR; with MTG_FLAG_BASIC_SOLUTION it returns MTG_OK, the straight line between the points costs nothing */
  int basic_ok = 0;
  {
    uint32_t m2[2] = {1u, 1u};
    mtg_plan_desc d2 = {N, D, 1, DERIV, m2};
    mtg_plan* p2 = NULL;
    CHECK(mtg_plan_create(ctx, &d2, &p2));
    mtg_layout l2;
    mtg_layout_aos(p2, 3, &l2);
    const double t2[3] = {1.0, 1.5, 2.5};
    const double f2[3 * D * 2] = {0, 1, 0, 2, 0, 3, 1, -1, 2, -2, 3, -3, 0.5, 0.25, 0.5, 0.75, -1, 4};
    double c2[3 * D * N], j2[3];
    const int rc_plain = mtg_solve_linear(p2, 3, &l2, t2, f2, c2, NULL, j2, MTG_FLAG_HOST_POINTERS);
    CHECK(mtg_solve_linear(p2, 3, &l2, t2, f2, c2, NULL, j2, MTG_FLAG_HOST_POINTERS | MTG_FLAG_BASIC_SOLUTION));
    double worst = 0.0;
    for (int b = 0; b < 3; ++b)
      for (int d = 0; d < D; ++d) {
        const double* c = c2 + (b * D + d) * N;
        double end = 0.0, tp_ = 1.0;
        for (int j = 0; j < N; ++j) { end += c[j] * tp_; tp_ *= t2[b]; }
        const double e0 = c[0] - f2[(b * D + d) * 2], e1 = end - f2[(b * D + d) * 2 + 1];
        if ((e0 < 0 ? -e0 : e0) > worst) worst = e0 < 0 ? -e0 : e0;
        if ((e1 < 0 ? -e1 : e1) > worst) worst = e1 < 0 ? -e1 : e1;
      }
    basic_ok = rc_plain == MTG_ERR_SINGULAR && worst < 1e-9 && j2[0] < 1e-9 && j2[1] < 1e-9 && j2[2] < 1e-9;
    printf("under-constrained batch: plain call -> %d (%s); with MTG_FLAG_BASIC_SOLUTION -> ok, end-point error %.2e, cost %.2e\n",
           rc_plain, mtg_status_string(rc_plain), worst, j2[0] + j2[1] + j2[2]);
    mtg_plan_destroy(p2);
  }
  /* the final gather of a one-process-per-GPU job through the C ABI's own RCCL communicator (here: the one-rank job this
   * process is -- rank 0 makes the id, every rank would create its communicator from it): the batch solved in four chunks,
   * each chunk all-gathered on the communicator's stream under the next chunk's solve; gathered = [chunks][world][B / chunks]... */
  int gather_ok = 0;
  {
    char id[MTG_COMM_UNIQUE_ID_BYTES];
    mtg_comm* comm = NULL;
    const int rc_id = mtg_comm_unique_id(id);
    if (rc_id == MTG_ERR_UNSUPPORTED) {
      printf("RCCL gather: librccl.so not present, skipped\n");
      gather_ok = 1;
    } else {
      CHECK(rc_id);
      CHECK(mtg_comm_create(ctx, 0, 1, id, &comm));
      const int chunks = batch % 4 == 0 ? 4 : 1;
      void* gathered = NULL;
      CHECK(mtg_device_malloc(ctx, ncoef * sizeof(double), &gathered));
      CHECK(mtg_comm_solve_all_gather(comm, plan, batch, &lay, (const double*)times, (const double*)dfix, (double*)cb, (double*)gathered, chunks, 0));
      CHECK(mtg_comm_sync(comm));
      double r = 1.0, a = 1.0;
      CHECK(mtg_compare_coefficients(ctx, (const double*)gathered, (const double*)ca, (int64_t)batch * K * D, N, &r, &a));
      gather_ok = r == 0.0 && mtg_comm_world(comm) == 1;
      printf("RCCL gather through mtg_comm_solve_all_gather (%d chunks, world %d): gathered vs single solve rel diff %.3e\n", chunks,
             mtg_comm_world(comm), r);
      mtg_device_free(ctx, gathered);
      mtg_comm_destroy(comm);
    }
  }
  mtg_device_free(ctx, times); mtg_device_free(ctx, dfix); mtg_device_free(ctx, ca); mtg_device_free(ctx, cb);
  mtg_plan_destroy(plan);
  mtg_context_destroy(ctx);
  free(mask);
  return (rel < 1e-10 && rel_q < 1e-10 && basic_ok && gather_ok) ? 0 : 3;
}
