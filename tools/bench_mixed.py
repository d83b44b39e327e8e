"""BASELINE config 4 as ONE request: 30k trajectories = 12 buckets (N in {8, 10, 12} x K in {4, 8, 16, 32}, D = 3) x 2500,
device-resident; one stream (buckets queue behind each other) vs the buckets spread over several HIP streams
(MixedBatchSolver.solve_device).  Wall time per mixed batch from events on the caller's stream."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mav_trajectory_generation_amd as m

per_bucket = int(sys.argv[1]) if len(sys.argv) > 1 else 2500
# modes run in separate processes (python tools/bench_mixed.py 2500 merged): streams created by one section stay alive
# in torch's pool and share the hardware queues with the next section's, which hides the overlap being measured
modes = sys.argv[2].split(",") if len(sys.argv) > 2 else ["streams", "graph", "merged", "concurrent"]
ctx = m.Context(0)
buckets, algo_bytes = [], 0
NS = [int(x) for x in os.environ.get("MIXED_NS", "8,10,12").split(",")]
for (N, d) in ((8, 3), (10, 4), (12, 5)):
    if N not in NS:
        continue
    for K in (4, 8, 16, 32):
        masks = m.ends_full_masks(N, K, 1)
        t, f = m.random_waypoint_batch(per_bucket, K, 3, N, masks, seed=11 + K, device="cuda", layout="soa")
        buckets.append(dict(n_coeffs=N, derivative=d, masks=masks, times=t, d_fixed=f, layout="soa"))
        algo_bytes += per_bucket * 8 * (K + 3 * (N + K - 1) + K * 3 * N)
total = per_bucket * len(buckets)
for n_streams in ((1, 2, 4, 8) if "streams" in modes else ()):
    solver = m.MixedBatchSolver(ctx, n_streams=n_streams)
    for _ in range(5):
        solver.solve_device(buckets)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 30
    e0.record()
    for _ in range(reps):
        out = solver.solve_device(buckets)
    e1.record()
    torch.cuda.synchronize()
    solver.sync()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print(json.dumps(dict(config="config4-mixed", per_bucket=per_bucket, n_streams=n_streams, us_per_mixed_batch=round(us, 1),
                          traj_per_s=total / us * 1e6, GBps=algo_bytes / us * 1e-3, frac_8TBps=algo_bytes / us * 1e-3 / 8000.0)))
    solver.close()

# the same request captured once into a hipGraph (fork over 4 streams + join) and replayed: one host call per mixed batch
for n_streams in ((1, 4, 8, 12) if "graph" in modes else ()):
    solver = m.MixedBatchSolver(ctx, n_streams=n_streams)
    graph, out = solver.capture(buckets)
    for _ in range(5):
        graph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 50
    e0.record()
    for _ in range(reps):
        graph.replay()
    e1.record()
    torch.cuda.synchronize()
    solver.sync()
    us = e0.elapsed_time(e1) * 1e3 / reps
    # replayed results == eager results
    eager = solver.solve_device(buckets)
    torch.cuda.synchronize()
    same = all(torch.equal(a[0], b[0]) for a, b in zip(out, eager))
    print(json.dumps(dict(config="config4-mixed-graph", per_bucket=per_bucket, n_streams=n_streams, us_per_mixed_batch=round(us, 1),
                          traj_per_s=total / us * 1e6, GBps=algo_bytes / us * 1e-3, frac_8TBps=algo_bytes / us * 1e-3 / 8000.0,
                          replay_equals_eager=bool(same))))
    del graph
    solver.close()

# one C call per mixed batch: every bucket its own best launch, spread over the library context's side streams
if "concurrent" in modes:
    solver = m.MixedBatchSolver(ctx, n_streams=1)
    req = solver.concurrent(buckets)
    for _ in range(5):
        req.solve()
    torch.cuda.synchronize()
    import time
    for reps in (1, 50):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0.record()
        for _ in range(reps):
            req.solve()
        e1.record()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        solver.sync()
        us = e0.elapsed_time(e1) * 1e3 / reps
        one = m.MixedBatchSolver(ctx, n_streams=1)
        eager = one.solve_device(buckets)
        torch.cuda.synchronize()
        same = all(torch.equal(a[0], b[0]) for a, b in zip(req.out, eager))
        one.close()
        print(json.dumps(dict(config="config4-mixed-concurrent", per_bucket=per_bucket, launches=req.launch_count, back_to_back=reps,
                              us_per_mixed_batch=round(us, 1), host_enqueue_us=round((t1 - t0) * 1e6 / reps, 1),
                              traj_per_s=total / us * 1e6, GBps=algo_bytes / us * 1e-3,
                              frac_8TBps=algo_bytes / us * 1e-3 / 8000.0, equals_per_bucket_launches=bool(same))))
    req.close()
    solver.close()

# buckets of equal structure-up-to-K merged into one launch each (mtg_multi_*): 3 launches on 3 streams, eager and as a graph
if "merged" not in modes:
    sys.exit(0)
solver = m.MixedBatchSolver(ctx, n_streams=3)
req = solver.merged(buckets, dims=os.environ.get("MIXED_DIMS", "auto"))
for mode in ("eager", "eager-library-stream", "graph"):
    graph = req.capture() if mode == "graph" else None
    run = (lambda: graph.replay()) if graph is not None else (lambda: req.solve())
    # "eager": called from torch's default stream -- every call forks onto and joins back from the library's stream (two
    # cross-stream event waits, ~10 us of device time each on this runtime); "eager-library-stream": the caller works on
    # the library's stream (what a C caller of mtg_multi_solve does), no cross-stream waits
    import contextlib
    scope = torch.cuda.stream(ctx.stream) if mode == "eager-library-stream" else contextlib.nullcontext()
    with scope:
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 50
        e0.record()
        for _ in range(reps):
            run()
        e1.record()
        torch.cuda.synchronize()
    solver.sync()
    us = e0.elapsed_time(e1) * 1e3 / reps
    one = m.MixedBatchSolver(ctx, n_streams=1)
    eager = one.solve_device(buckets)
    torch.cuda.synchronize()
    same = all(torch.equal(a[0], b[0]) for a, b in zip(req.out, eager))
    one.close()
    print(json.dumps(dict(config="config4-mixed-merged-" + mode, per_bucket=per_bucket, launches=req.launch_count,
                          us_per_mixed_batch=round(us, 1), traj_per_s=total / us * 1e6, GBps=algo_bytes / us * 1e-3,
                          frac_8TBps=algo_bytes / us * 1e-3 / 8000.0, equals_per_bucket_launches=bool(same))))
req.close()
solver.close()
