#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric: trajectories/s for 8-segment, N=10, 3-D minimum-snap solveLinear().

One "step" = one pass of the hot path (fused updateSegmentTimes + constructR + solve + coefficient recovery: ONE kernel
launch through the C ABI, mtg_solve_linear) over one batch of synthetic random-waypoint trajectories that is already
resident in HBM.  The timed loop ROTATES over `--buffer-sets` (default 16) independent input/output buffer sets, so that
with the default workload (24 MB per set, 382 MB in total > the 256 MiB Infinity Cache) every step's reads and writes
really go to HBM; the same loop over ONE resident set is reported beside it (`extra.resident_buffers`).

N GPUs = N processes, one per GPU (`--gpus N` spawns them itself via torch.distributed.run when not already launched
that way).  The path shards embarrassingly: every rank solves its own batch, RCCL is used only for the
barrier / max-reduce of the timing and -- reported separately, never part of `value` -- the optional final all_gather of
the coefficients, chunked so that chunk i's gather overlaps chunk i+1's solve  => "scaling": "weak".

`--config` picks the BASELINE.json configuration (per-GPU share): 2 = 10k x (K=8, N=10, D=3) [default, the configuration
the metric is quoted on], 3 = 125k of the same (1M over 8 GPUs), 5 = 12.5k x (K=16, N=10, D=4, velocity + acceleration
fixed at interior vertices; 100k over 8 GPUs).

Prints ONE JSON line (rank 0).  `roofline.achieved` = algorithmic bytes per launch (SURVEY.md 8(d):
8*(K + D*n_fixed + K*D*N) per trajectory) / mean launch duration, measured with HIP events recorded on the launch stream
around the SAME timed region that `value` comes from.  `cpu_baseline` = the reference algorithm's CPU restatement
(oracle/, the checker -- never the product path) timed on this box's host cores.
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)

CONFIGS = {
    2: dict(name="BASELINE configs[1]", batch=10_000, K=8, D=3, interior=1, yaw=False),
    3: dict(name="BASELINE configs[2] per-GPU share (1M / 8)", batch=125_000, K=8, D=3, interior=1, yaw=False),
    5: dict(name="BASELINE configs[4] per-GPU share (100k / 8)", batch=12_500, K=16, D=4, interior=7, yaw=True),
}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=None, help="trajectories per step per GPU (default: the config's)")
    ap.add_argument("--buffer-sets", type=int, default=16, help="independent input/output buffer sets the timed loop "
                    "rotates over (1 = everything stays resident in the Infinity Cache)")
    ap.add_argument("--layout", default="soa", choices=["aos", "soa"])
    ap.add_argument("--dims", default="auto", choices=["auto", "fused", "split", "dimlane"], help="kernel launch form")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true", help="N > 1: skip the solve + all_gather measurement")
    ap.add_argument("--gather-chunks", type=int, default=4)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to "
                    "exercise the multi-process path on a box with fewer GPUs than ranks)")
    ap.add_argument("--same-device", action="store_true", help="all ranks on HIP device 0 (multi-process tests on a 1-GPU box)")
    ap.add_argument("--plumbing-only", action="store_true", help="spawn / rendezvous / reduce only, no GPU work (CPU test)")
    ap.add_argument("--extra", action="store_true", help="also time the 1M per-launch batch and the host-pointer paths")
    ap.add_argument("--no-extras", action="store_true",
                    help="only the timed steps (profiling runs: the kernel statistics then cover the same launches as the metric)")
    return ap.parse_args(argv)


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def respawn_under_launcher(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start N ranks ourselves (one process per GPU,
    torch.distributed.run, rendezvous on 127.0.0.1) and relay rank 0's line."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env)


def cpu_baseline(n_sample, seed):
    """Reference-algorithm CPU restatement (oracle/cpu_ref.cpp, the checker -- never the product path) timed on
    this box's host cores over a bounded sample of the same workload (~10-15 s of CPU work)."""
    import mav_trajectory_generation_amd as m
    from oracle import cpu_ref
    masks = m.ends_full_masks(10, 8)
    t, f = m.random_waypoint_batch(n_sample, 8, 3, 10, masks, seed=seed, device="cpu")
    out = cpu_ref.timed_baseline(10, 4, masks, t.numpy(), f.numpy(), target_seconds=12.0)
    # beside it: the reference's own code (oracle/_ref/libmtg_ref.so, compiled from /root/reference against the
    # Eigen/glog container stand-ins -- slower than real Eigen, hence reported as context, not as the baseline value)
    from oracle import ref_linear
    if ref_linear.available():
        out["reference_build"] = ref_linear.timed_baseline(10, 4, masks, t.numpy()[:20_000], f.numpy()[:20_000],
                                                           target_seconds=2.0)
    return out


def profile_traffic(batch):
    """HBM bytes per launch as measured by the committed rocprofv3 PMC passes (profiles/*_pmc_traffic.json): evidence
    from a separate profiling run of the same command, NOT measured in this run -- hence its own key and the file name."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")), reverse=True):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        if d.get("batch") == batch:
            return {"file": os.path.relpath(path, ROOT), "hbm_bytes_per_launch": d.get("hbm_bytes_per_launch"),
                    "note": d.get("note", "separate rocprofv3 --pmc run of this command")}
    return None


class SolveLoop:
    """The timed loop: `steps` solves (one kernel launch each) rotating over the buffer sets, enqueued by ONE call of
    mtg_solve_linear_sequence (C ABI) so that the host enqueues faster than the GPU executes (a Python loop over
    mtg_solve_linear costs ~8 us per call -- more than the kernel)."""

    def __init__(self, plan, sets, layout, dims):
        from mav_trajectory_generation_amd import _lib as L
        self.plan = plan
        self.flags = {"auto": 0, "fused": L.FLAG_FUSED_DIMS, "split": L.FLAG_SPLIT_DIMS, "dimlane": L.FLAG_DIMLANE}[dims]
        self.batch = sets[0][2].shape[0]
        self.lay = plan.layout(self.batch, layout)
        self.ptrs = [(t.data_ptr(), f.data_ptr(), co.data_ptr()) for (t, f, co) in sets]
        self._cache = {}

    def _arrays(self, steps, first):
        key = (steps, first % len(self.ptrs))
        if key not in self._cache:
            n = len(self.ptrs)
            arr = [(ctypes.c_void_p * steps)(*[self.ptrs[(first + i) % n][j] for i in range(steps)]) for j in range(3)]
            self._cache[key] = arr
        return self._cache[key]

    def prepare(self, steps, first=0):
        self._arrays(steps, first)

    def run(self, steps, first=0, start_event=None, stop_event=None):
        """start_event / stop_event: torch.cuda.Event objects (already recorded once, so that their hipEvent_t exists)
        recorded by the library right before the first / after the last launch -- the measured interval then does not
        contain Python's latency between an `event.record()` and the first launch.  (Measured: it makes no difference to
        the 20-step figure, 8.4-9.0 us against 8.1-8.2 us for 1000+ steps -- the excess of short runs is the GPU coming
        out of the idle period behind the contract's barrier, not host latency.)"""
        if steps <= 0:
            return
        t, f, c = self._arrays(steps, first)
        ev = [ctypes.c_void_p(e.cuda_event) if e is not None else None for e in (start_event, stop_event)]
        rc = self.plan.lib.mtg_solve_linear_sequence_events(self.plan.handle, steps, self.batch, ctypes.byref(self.lay),
                                                            t, f, c, self.flags, ev[0], ev[1])
        if rc != 0:
            raise RuntimeError(f"mtg_solve_linear_sequence failed: {rc}")


def main():
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    launched = "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not launched:
        raise SystemExit(respawn_under_launcher(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")

    import torch
    import torch.distributed as dist

    if args.plumbing_only:
        if world > 1:
            dist.init_process_group(args.backend if args.backend != "nccl" else "gloo")
            assert dist.get_world_size() == args.gpus
            tt = torch.tensor([float(rank + 1)], dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            assert float(tt.item()) == world
            dist.barrier()
        if rank == 0:
            print(json.dumps({"plumbing_only": True, "n_gpus": world, "steps": args.steps, "warmup": args.warmup}))
        if world > 1:
            dist.destroy_process_group()
        return

    import mav_trajectory_generation_amd as m

    device_index = 0 if args.same_device else local
    torch.cuda.set_device(device_index)
    dev = torch.device("cuda", device_index)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)
        assert dist.get_world_size() == args.gpus

    cfg = CONFIGS[args.config]
    N, K, D, d = 10, cfg["K"], cfg["D"], 4
    B = args.batch if args.batch is not None else cfg["batch"]
    masks = m.ends_full_masks(N, K, cfg["interior"])
    ctx = m.Context(device_index)
    plan = m.Plan(ctx, N, D, K, d, masks)
    nsets = max(1, args.buffer_sets)
    # keep the rotating footprint bounded (config 3: 300 MB per set)
    set_bytes = B * (plan.bytes_per_trajectory)
    while nsets > 2 and nsets * set_bytes > 24 * 2**30:
        nsets //= 2

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def make_set(seed):
        t, f = m.random_waypoint_batch(B, K, D, N, masks, seed=seed, device=dev, layout=args.layout, yaw_dim=cfg["yaw"])
        return t, f, torch.zeros((B, K, D, N), dtype=torch.float64, device=dev)

    with torch.cuda.stream(ctx.stream):
        sets = [make_set(1234 + rank + 1000 * s) for s in range(nsets)]
        loop = SolveLoop(plan, sets, args.layout, args.dims)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(ctx.stream)   # (creates the underlying hipEvent_t handles: torch makes them on first record)
        e1.record(ctx.stream)

        def finish():
            while not e1.query():   # spin on the end event (a blocking synchronize alone wakes up ~30 us late) ...
                pass
            barrier()               # ... then the contract's barrier + torch.cuda.synchronize()

        def timed(loop_, steps, warmup):
            loop_.prepare(warmup, 0)
            loop_.prepare(steps, warmup)
            loop_.run(warmup)
            barrier()
            t0 = time.perf_counter()
            loop_.run(steps, first=warmup, start_event=e0, stop_event=e1)   # events recorded inside the C call
            finish()
            return time.perf_counter() - t0, e0.elapsed_time(e1) * 1e3 / steps   # wall seconds, event us per launch

        # set-up, not a step: every buffer set is touched once (first use of fresh allocations: page-table / TLB fills),
        # as in any pipeline that has been running for longer than one rotation
        loop.run(nsets)
        torch.cuda.synchronize()
        dt, kern_us = timed(loop, args.steps, args.warmup)
        ctx.sync()  # raises if any trajectory flagged bad time / singular
        for (_, _, co) in sets:
            assert torch.isfinite(co).all()

        extra = {}
        if rank == 0 and not args.no_extras:
            # the same loop over ONE resident buffer set (inputs and outputs stay in the 256 MiB Infinity Cache)
            res = SolveLoop(plan, sets[:1], args.layout, args.dims)
            steps_r = max(args.steps, 200)
            res.run(20)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            e0.record(ctx.stream)
            res.run(steps_r)
            e1.record(ctx.stream)
            torch.cuda.synchronize()
            dt_r = time.perf_counter() - t0
            us_r = e0.elapsed_time(e1) * 1e3 / steps_r
            extra["resident_buffers"] = {"buffer_sets": 1, "steps": steps_r, "launch_us": us_r,
                                         "traj_per_s": B * steps_r / dt_r,
                                         "frac_of_8TBps": B * plan.bytes_per_trajectory / us_r * 1e-3 / HBM_PEAK_GBS}
            # a longer rotating run of the same loop (the driver's --steps can be as small as 20: 150 us of GPU work)
            if args.steps < 1000:
                loop.run(20)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                e0.record(ctx.stream)
                loop.run(2000)
                e1.record(ctx.stream)
                torch.cuda.synchronize()
                dt_l = time.perf_counter() - t0
                us_l = e0.elapsed_time(e1) * 1e3 / 2000
                extra["rotating_buffers_2000_steps"] = {"buffer_sets": nsets, "steps": 2000, "launch_us": us_l,
                                                        "traj_per_s": B * 2000 / dt_l,
                                                        "frac_of_8TBps": B * plan.bytes_per_trajectory / us_l * 1e-3 / HBM_PEAK_GBS}
            # steady state of a pipeline that feeds TWO streams (independent batches: nothing orders them): the launches of
            # one stream fill the gap between dependent launches of the other (~1.0-1.3 us) and overlap its store tail.
            # Reported beside the one-stream numbers, never as `value`.
            if args.config == 2 and nsets >= 4:
                ctx2 = m.Context(device_index)
                plan2 = m.Plan(ctx2, N, D, K, d, masks)
                la = SolveLoop(plan, sets[0::2], args.layout, args.dims)
                lb = SolveLoop(plan2, sets[1::2], args.layout, args.dims)
                chunk, rounds = 8, 125
                for _ in range(3):
                    la.run(chunk)
                    lb.run(chunk)
                torch.cuda.synchronize()
                ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                t0 = time.perf_counter()
                e0.record(ctx.stream)
                for _ in range(rounds):
                    la.run(chunk)
                    lb.run(chunk)
                ea.record(ctx.stream)
                eb.record(ctx2.stream)
                torch.cuda.synchronize()
                dt_2 = time.perf_counter() - t0
                n2 = 2 * chunk * rounds
                us_2 = max(e0.elapsed_time(ea), e0.elapsed_time(eb)) * 1e3 / n2
                ctx2.sync()
                extra["two_streams_steady_state"] = {"buffer_sets": nsets, "launches": n2, "us_per_launch": us_2,
                                                     "traj_per_s": B * n2 / dt_2,
                                                     "frac_of_8TBps": B * plan.bytes_per_trajectory / us_2 * 1e-3 / HBM_PEAK_GBS}
                plan2.close()
                ctx2.close()
            # context for small launches: a write-only fill of the same coefficient buffer (zero compute, zero reads)
            co0 = sets[0][2]
            co0_copy = co0.clone()
            co0.fill_(0.0)
            e0.record(ctx.stream)
            for _ in range(50):
                co0.fill_(0.0)
            e1.record(ctx.stream)
            torch.cuda.synchronize()
            fill_us = e0.elapsed_time(e1) * 1e3 / 50
            co0.copy_(co0_copy)
            del co0_copy
            if args.config == 2:
                # the per-GPU share of BASELINE configs[2] (1M over 8 GPUs) always; --extra adds 1M on this GPU
                for big in ((125_000, 1_000_000) if args.extra else (125_000,)):
                    tb, fb = m.random_waypoint_batch(big, K, D, N, masks, seed=99, device=dev, layout=args.layout)
                    cb = torch.empty((big, K, D, N), dtype=torch.float64, device=dev)
                    plan.solve(tb, fb, layout=args.layout, coeffs=cb, dims=args.dims)
                    torch.cuda.synchronize()
                    us = plan.time_last_solve(20)
                    extra[f"batch_{big}"] = {"kernel_us": us, "traj_per_s": big / us * 1e6,
                                             "GBps": big * plan.bytes_per_trajectory / us * 1e-3,
                                             "frac_of_8TBps": big * plan.bytes_per_trajectory / us * 1e-3 / HBM_PEAK_GBS}
                    del tb, fb, cb
        else:
            fill_us = None
        if args.extra and rank == 0 and args.config == 2:
            # host buffers in / out (MTG_FLAG_HOST_POINTERS): PCIe-inclusive rate, never the reported value.
            t, f, _ = sets[0]
            th = (t.t().contiguous() if args.layout == "soa" else t).cpu()
            fh = (f.permute(2, 0, 1).contiguous() if args.layout == "soa" else f).cpu()
            for tag, tt_, ff_ in (("pageable", th.numpy(), fh.numpy()),
                                  ("pinned", th.pin_memory().numpy(), fh.pin_memory().numpy())):
                co_h = torch.empty((B, K, D, N), dtype=torch.float64, pin_memory=(tag == "pinned")).numpy()
                plan.solve_host(tt_, ff_, want_free=False, want_cost=False, coeffs=co_h)
                t1 = time.perf_counter()
                for _ in range(5):
                    plan.solve_host(tt_, ff_, want_free=False, want_cost=False, coeffs=co_h)
                extra[f"host_pointers_{tag}_pcie_inclusive_traj_per_s"] = 5 * B / (time.perf_counter() - t1)

        gather = None
        if world > 1 and not args.no_gather:
            # solve + final all_gather of the coefficients: chunked, chunk i's gather overlaps chunk i+1's
            # solve; reported beside the solve-only number, never part of `value`
            from mav_trajectory_generation_amd import dist as mdist
            t, f, _ = sets[0]
            runner = mdist.ChunkedSolveGather(plan, t, f, layout=args.layout, n_chunks=args.gather_chunks)
            for _ in range(3):
                runner.run()
            barrier()
            reps = 10
            t0 = time.perf_counter()
            for _ in range(reps):
                runner.run()
            barrier()
            both = (time.perf_counter() - t0) / reps
            barrier()
            t0 = time.perf_counter()
            for _ in range(reps):
                runner.run(gather=False)
            barrier()
            solve_only = (time.perf_counter() - t0) / reps
            barrier()
            t0 = time.perf_counter()
            for _ in range(reps):
                runner.run(solve=False)
            barrier()
            gather_only = (time.perf_counter() - t0) / reps
            gather = {"chunks": runner.n_chunks, "solve_plus_gather_ms": both * 1e3, "solve_only_ms": solve_only * 1e3,
                      "gather_only_ms": gather_only * 1e3,
                      "gathered_bytes_per_rank": world * B * K * D * N * 8, "backend": args.backend}

    if world > 1:
        vals = torch.tensor([dt, kern_us], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(vals, op=dist.ReduceOp.MAX)
        dt, kern_us = float(vals[0].item()), float(vals[1].item())
        if gather is not None:
            g = torch.tensor([gather["solve_plus_gather_ms"], gather["solve_only_ms"], gather["gather_only_ms"]],
                             dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
            dist.all_reduce(g, op=dist.ReduceOp.MAX)
            gather["solve_plus_gather_ms"], gather["solve_only_ms"], gather["gather_only_ms"] = [float(x) for x in g]
            gather["traj_per_s_with_gather"] = world * B / (gather["solve_plus_gather_ms"] * 1e-3)

    if rank == 0:
        bytes_per_launch = B * plan.bytes_per_trajectory
        achieved = bytes_per_launch / (kern_us * 1e-6) / 1e9
        what = (f"batch of {B} random-waypoint trajectories per GPU per step ({cfg['name']}): {K} segments, N=10, "
                f"dim={D}, snap" + (", velocity + acceleration fixed at interior vertices" if cfg["interior"] == 7 else "")
                + f"; one kernel launch per step on ONE stream (latency figure, no overlap between steps), rotating over "
                  f"{nsets} independent input/output buffer sets ({nsets * set_bytes / 2**20:.0f} MiB) resident in HBM; "
                  f"inputs {args.layout.upper()}, coeffs [B][K][D][N]")
        out = {
            "metric": "trajectories/sec (8-seg, N=10, 3D min-snap solveLinear)" if args.config != 5 else
                      "trajectories/sec (16-seg, N=10, 4D min-snap solveLinear, config 5)",
            "value": world * B * args.steps / dt,
            "unit": "trajectories/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": what, "baseline_config": args.config, "buffer_sets": nsets,
                       "kernel_variant": plan.kernel_variant, "launch_form": args.dims,
                       "bytes_per_trajectory": plan.bytes_per_trajectory},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "traffic_from_profile": profile_traffic(B),
                         "kernel_us": kern_us, "bytes_per_launch": bytes_per_launch,
                         "kernel_us_is": "HIP events on the launch stream around the timed steps / steps "
                                         "(launch-to-launch period of the back-to-back stream)",
                         "output_fill_only_us": fill_us},
        }
        if extra:
            out["extra"] = extra
        if gather is not None:
            out["gather"] = gather
        if not args.no_cpu_baseline and world == 1 and args.config == 2:
            out["cpu_baseline"] = cpu_baseline(200_000, 4321)
            out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
