#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric: trajectories/s for 8-segment, N=10, 3-D minimum-snap solveLinear().

One "step" = one pass of the hot path (fused updateSegmentTimes + constructR + solve + coefficient recovery,
one kernel launch) over one batch of `--batch` synthetic random-waypoint trajectories that is already resident
in HBM.  N GPUs = N processes (torch.distributed / RCCL only for the barrier + max-reduce of the timing; the
path shards embarrassingly, no data-path collective) each solving its own batch => weak scaling.

Prints ONE JSON line (rank 0).  `roofline` = algorithmic bytes per launch (8*(K + D*n_fixed +
K*D*N) = 2392 B/trajectory) / mean kernel duration measured with hipEvents on the launch stream;
`cpu_baseline` = the reference algorithm's CPU restatement (oracle/) timed on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


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


def measured_traffic(batch):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/*_pmc_traffic.json, produced by
    tools/gpu_profile.sh on the same command); None if no measurement for this batch size is on file."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")), reverse=True):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        if d.get("batch") == batch:
            return d.get("hbm_bytes_per_launch")
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=10_000, help="trajectories per step per GPU (BASELINE configs[1])")
    ap.add_argument("--layout", default="soa", choices=["aos", "soa"])
    ap.add_argument("--dims", default="auto", choices=["auto", "fused", "split"], help="kernel launch geometry")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to "
                    "exercise the multi-process path on a box with fewer GPUs than ranks)")
    ap.add_argument("--device", type=int, default=None, help="force the HIP device index (multi-process tests)")
    ap.add_argument("--extra", action="store_true", help="also time the 125k / 1M per-launch batches")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import mav_trajectory_generation_amd as m

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.device is not None:
        local = args.device
    if world > 1:
        torch.cuda.set_device(local)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(args.backend)
    else:
        torch.cuda.set_device(local if args.device is not None else 0)
        local = local if args.device is not None else 0
    dev = torch.device("cuda", local)

    N, K, D, d = 10, 8, 3, 4
    masks = m.ends_full_masks(N, K)
    ctx = m.Context(local)
    plan = m.Plan(ctx, N, D, K, d, masks)
    B = args.batch

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.cuda.stream(ctx.stream):
        t, f = m.random_waypoint_batch(B, K, D, N, masks, seed=1234 + rank, device=dev, layout=args.layout)
        coeffs = torch.empty((B, K, D, N), dtype=torch.float64, device=dev)
        for _ in range(args.warmup):
            plan.solve(t, f, layout=args.layout, coeffs=coeffs, dims=args.dims)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            plan.solve(t, f, layout=args.layout, coeffs=coeffs, dims=args.dims)
        barrier()
        dt = time.perf_counter() - t0
        ctx.sync()  # raises if any trajectory flagged bad time / singular
        assert torch.isfinite(coeffs).all()

        # kernel-only duration: hipEvents on the launch stream (inside the library)
        kern_us = plan.time_last_solve(max(50, args.steps))
        # context for small launches: a write-only fill of the same coefficient buffer (zero compute, zero reads)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        coeffs.fill_(0.0)
        e0.record(ctx.stream)
        for _ in range(50):
            coeffs.fill_(0.0)
        e1.record(ctx.stream)
        torch.cuda.synchronize()
        fill_us = e0.elapsed_time(e1) * 1e3 / 50
        extra = {}
        if rank == 0:
            # the per-GPU share of BASELINE configs[2] (1M over 8 GPUs) always; --extra adds 1M on this GPU + host paths
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
        if args.extra and rank == 0:
            # host buffers in / out (MTG_FLAG_HOST_POINTERS): PCIe-inclusive rate, never the reported value.  Pageable
            # numpy arrays go through the runtime's staged copies; page-locked ones (here: torch pinned tensors viewed as
            # numpy) are DMA'd directly.
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

    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    if rank == 0:
        bytes_per_launch = B * plan.bytes_per_trajectory
        achieved = bytes_per_launch / (kern_us * 1e-6) / 1e9
        out = {
            "metric": "trajectories/sec (8-seg, N=10, 3D min-snap solveLinear)",
            "value": world * B * args.steps / dt,
            "unit": "trajectories/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"batch of {B} random-waypoint trajectories per GPU per step, 8 segments, N=10, "
                                   f"dim=3, snap (BASELINE configs[1]); inputs {args.layout.upper()} resident in HBM, "
                                   f"coeffs [B][K][D][N]",
                       "kernel_variant": plan.kernel_variant, "bytes_per_trajectory": plan.bytes_per_trajectory},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic(B),
                         "kernel_us": kern_us, "bytes_per_launch": bytes_per_launch,
                         "output_fill_only_us": fill_us},
        }
        if extra:
            out["extra"] = extra
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(200_000, 4321)
            out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
