#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric: trajectories/s for 8-segment, N=10, 3-D minimum-snap solveLinear().

One "step" = one pass of the hot path (fused updateSegmentTimes + constructR + solve + coefficient recovery) over one batch
of synthetic random-waypoint trajectories that is already resident in HBM.  The K timed steps are K INDEPENDENT batches
handed to the library by ONE call of the C ABI's mtg_solve_linear_sequence_events, rotating over `--buffer-sets` (default
16) independent input/output buffer sets (24 MB per set, 382 MB in total > the 256 MiB Infinity Cache: every step's reads
and writes really go to HBM).  `--sequence queue` (default): the library runs the queue as ONE persistent launch whose
workgroups walk the tiles of all batches (<= 96 batches per launch) -- the throughput form, `value`;
`--sequence launches`: one kernel launch per batch back to back on one stream -- the latency form of round 2, reported
beside it as `extra.one_launch_per_batch`.  The same loop over ONE resident buffer set is `extra.resident_buffers`.

N GPUs = N processes, one per GPU (`--gpus N` spawns them itself via torch.distributed.run when not already launched
that way).  The path shards embarrassingly: every rank solves its own batches, RCCL is used only for the
barrier / max-reduce of the timing and -- reported separately, never part of `value` -- the optional final all_gather of
the coefficients, chunked so that chunk i's gather overlaps chunk i+1's solve  => "scaling": "weak".

`--config` picks the BASELINE.json configuration (per-GPU share): 2 = 10k x (K=8, N=10, D=3) [default, the configuration
the metric is quoted on], 3 = 125k of the same (1M over 8 GPUs), 4 = the mixed request (N in {8, 10, 12} x K in {4, 8, 16,
32}, D = 3: 12 buckets x 2500 = 30k trajectories per step, ONE cross-structure launch per step through mtg_multi_solve),
5 = 12.5k x (K=16, N=10, D=4, velocity + acceleration fixed at interior vertices; 100k over 8 GPUs).
The default run (config 2) also times the SURVEY 8(f) rows -- sampling, extrema / time scaling, Mellinger cost + gradient --
as `extra.next` (`--no-next` skips them).

Prints ONE JSON line (rank 0).  `roofline.achieved` = algorithmic bytes (8*(K + D*n_fixed + K*D*N) per
trajectory) of the timed steps / their duration on the device, measured with HIP events recorded by the library on the
launch stream immediately around the SAME launches that `value` comes from.  `cpu_baseline` = the reference algorithm's
CPU restatement (oracle/, the checker -- never the product path) timed on this box's host cores.
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)

CONFIGS = {
    2: dict(name="BASELINE configs[1]", batch=10_000, K=8, D=3, interior=1, yaw=False),
    3: dict(name="BASELINE configs[2] per-GPU share (1M / 8)", batch=125_000, K=8, D=3, interior=1, yaw=False),
    4: dict(name="BASELINE configs[3]: mixed N / K request", batch=2_500, K=None, D=3, interior=1, yaw=False),
    5: dict(name="BASELINE configs[4] per-GPU share (100k / 8)", batch=12_500, K=16, D=4, interior=7, yaw=True),
}


WATCHDOG = {"hard_exit": False}   # set when a watchdog thread is stuck inside a collective init: leave through os._exit


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS))
    ap.add_argument("--batch", type=int, default=None, help="trajectories per step per GPU (default: the config's)")
    ap.add_argument("--buffer-sets", type=int, default=None, help="independent input/output buffer sets the timed loop "
                    "rotates over (1 = everything stays resident in the Infinity Cache).  Default (round 5): enough sets that "
                    "the timed steps touch NO set the warm-up / settle phases used (steps + warmup) and that the rotation is "
                    ">= 1.25 GiB, 5x the 256 MiB Infinity Cache -- at least 16 (rounds 2-4 rotated over 16 sets = 382 MiB "
                    "for config 2, 1.4x the cache; measured in round 5: 0.665 with 16, 0.644 with 48, 0.639 with 128 sets)")
    ap.add_argument("--layout", default=None, choices=["aos", "soa", "soa16"],
                    help="input layout (default: soa; config 5: soa16 = SoA with the row stride padded to a multiple of 16 "
                         "trajectories, so that the row pieces of a 12 500-trajectory batch start on 128-byte boundaries)")
    ap.add_argument("--dims", default="auto", choices=["auto", "fused", "split", "dimlane"], help="kernel launch form")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true", help="N > 1: skip the solve + all_gather measurement")
    ap.add_argument("--gather-chunks", type=int, default=4)
    ap.add_argument("--gather-via-mtg-comm", action="store_true", help="(default since round 6; kept for old command lines)")
    ap.add_argument("--no-gather-via-mtg-comm", action="store_true", help="nccl: do NOT measure the chunked solve + gather through the "
                    "C ABI's own RCCL communicator (mtg_comm_*, north_star's gather).  Default: measured at every N, its "
                    "ncclCommInitRank on a watchdog thread (--mtg-comm-init-timeout); a rank that cannot join in time makes ALL "
                    "ranks fall back to the torch.distributed figure, and the line says so")
    ap.add_argument("--mtg-comm-init-timeout", type=float, default=90.0, help="seconds the watchdog waits for ncclCommInitRank")
    ap.add_argument("--no-live-traffic", action="store_true", help="do not measure roofline.traffic in this invocation (default at N = 1: "
                    "after its own measurements rank 0 re-runs this command's timed region as child processes under rocprofv3 --pmc "
                    "FETCH_SIZE / WRITE_SIZE -- separate passes, calibrated in the same visit, tools/gpu_profile3.sh -- which adds about "
                    "a minute; without it, or when rocprofv3 is not there, the committed profile of the same arguments is quoted)")
    ap.add_argument("--sustained-seconds", type=float, default=1.0, help="length of the `sustained` run: continuous queue launches "
                    "over the HBM-sized rotation with the shader clock probed next to them (0: skip)")
    ap.add_argument("--exercise-collectives", action="store_true",
                    help="run every collective branch of the N > 1 path on a ONE-rank group (RCCL on a 1-GPU box: init, "
                         "all_reduce, all_gather, barrier, the chunked solve + all_gather_into_tensor); extra.collectives_exercised")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to "
                    "exercise the multi-process path on a box with fewer GPUs than ranks)")
    ap.add_argument("--same-device", action="store_true", help="all ranks on HIP device 0 (multi-process tests on a 1-GPU box)")
    ap.add_argument("--plumbing-only", action="store_true", help="spawn / rendezvous / reduce only, no GPU work (CPU test)")
    ap.add_argument("--extra", action="store_true", help="also time the 1M per-launch batch and the host-pointer paths")
    ap.add_argument("--sequence", default="queue", choices=["queue", "launches"],
                    help="how the library runs the K independent batches of the timed region: 'queue' = one persistent launch "
                         "over all of them (throughput form), 'launches' = one kernel launch per batch (latency form)")
    ap.add_argument("--no-next", action="store_true", help="skip the SURVEY 8(f) rows (sampling, extrema / time scaling, "
                                                           "Mellinger cost + gradient: extra.next, config 2 only)")
    ap.add_argument("--settle-ms", type=float, default=50.0, help="untimed set-up: milliseconds of the same work before the "
                    "contract's warm-up + timed steps (a fresh process starts on an idle GPU; 0: none). The line keeps the "
                    "region measured before it as `cold_start`")
    ap.add_argument("--no-parity", action="store_true", help="skip the parity sample of the timed launch's outputs (the metric's "
                    "second half: max coefficient rel-err vs the reference build / the port)")
    ap.add_argument("--parity-samples", type=int, default=768, help="trajectories of the timed launch's outputs that are "
                    "compared with the oracles (spread over >= 3 of the rotating buffer sets the TIMED launch wrote)")
    ap.add_argument("--timed-repeats", type=int, default=0,
                    help="diagnostic: repeat the warm-up + timed region this many more times AFTER the contract's one and "
                         "report each repeat's host-clock breakdown (extra.timed_region_repeats); never part of `value`")
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


def profile_traffic(batch, config, steps, warmup):
    """HBM bytes per STEP (one batch / one mixed request) as measured by the committed rocprofv3 PMC passes of THIS command
    (profiles/r04_config<N>_pmc_traffic.json, tools/gpu_profile3.sh: FETCH_SIZE / WRITE_SIZE in their own passes, calibrated in
    the same visit): evidence from a separate profiling run, NOT measured in this run -- the file name travels with the number.
    Only a profile of the same arguments (batch, --steps, --warmup) counts."""
    for name in (f"r06_config{config}_pmc_traffic.json", f"r05_config{config}_pmc_traffic.json", f"r04_config{config}_pmc_traffic.json",
                 {2: "r03z_driver_args_pmc_traffic.json", 4: "r03z_config4_pmc_traffic.json"}.get(config)):
        path = os.path.join(ROOT, "profiles", name) if name else None
        if not path or not os.path.exists(path):
            continue
        try:
            d = json.load(open(path))
        except Exception:
            continue
        pa = (d.get("bench_args") or "").split()
        def arg(flag, default):
            return int(pa[pa.index(flag) + 1]) if flag in pa else default
        if d.get("batch") != batch or arg("--steps", 200) != steps or arg("--warmup", 20) != warmup:
            continue
        return {"file": os.path.relpath(path, ROOT), "hbm_bytes_per_step": d.get("hbm_bytes_per_step"),
                "hbm_read_bytes_per_step": d.get("hbm_read_bytes_per_step"), "hbm_write_bytes_per_step": d.get("hbm_write_bytes_per_step"),
                "algorithmic_bytes_per_step": d.get("algorithmic_bytes_per_step"), "bench_args_of_the_profile": d.get("bench_args"),
                "note": d.get("note", "separate rocprofv3 --pmc run of this command")}
    return None


def live_traffic(args, batch):
    """roofline.traffic measured BY THIS INVOCATION (a stored profile cannot notice a regression between
    the profile visit and the run): the same command's timed region re-run as child processes under rocprofv3 --pmc (FETCH_SIZE and
    WRITE_SIZE in separate passes with --kernel-trace only, the counters calibrated on 2 GiB streams in the same visit -- exactly
    tools/gpu_profile3.sh, as /opt/skills/guides/MI355X_MICROARCH.md prescribes).  None when that is not possible here."""
    import shutil
    import subprocess
    if os.environ.get("MTG_BENCH_PROFILED_CHILD") or not shutil.which("rocprofv3"):
        return None
    calib = os.path.join(ROOT, "tools", "micro", "fetch_calib")
    if not os.path.exists(calib):
        return None
    tag = f"live{os.getpid()}"
    argv = ["--config", str(args.config), "--steps", str(args.steps), "--warmup", str(args.warmup), "--batch", str(batch),
            "--layout", args.layout, "--dims", args.dims, "--sequence", args.sequence, "--settle-ms", str(args.settle_ms),
            "--sustained-seconds", "0", "--no-live-traffic"]
    if args.buffer_sets is not None:
        argv += ["--buffer-sets", str(args.buffer_sets)]
    env = dict(os.environ, MTG_BENCH_PROFILED_CHILD="1", GRAFT_REPO_ROOT=ROOT)
    t0 = time.perf_counter()
    try:
        subprocess.run(["bash", os.path.join(ROOT, "tools", "gpu_profile3.sh"), tag] + argv, env=env, cwd=ROOT, timeout=420,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=False)
        path = os.path.join(ROOT, "gpurun_out", f"{tag}_pmc_traffic.json")
        d = json.load(open(path))
        import glob
        for f in glob.glob(os.path.join(ROOT, "gpurun_out", tag + "_*")):     # (the script's per-pass logs and copies)
            try:
                os.remove(f)
            except OSError:
                pass
        if not d.get("hbm_bytes_per_step"):
            return None
        return {"file": None, "measured": "by this invocation: child runs of the same command under rocprofv3 --pmc (tools/gpu_profile3.sh)",
                "seconds": time.perf_counter() - t0, "hbm_bytes_per_step": d.get("hbm_bytes_per_step"),
                "hbm_read_bytes_per_step": d.get("hbm_read_bytes_per_step"), "hbm_write_bytes_per_step": d.get("hbm_write_bytes_per_step"),
                "algorithmic_bytes_per_step": d.get("algorithmic_bytes_per_step"), "kernel": d.get("kernel"),
                "rocprof_us_per_step": d.get("rocprof_us_per_step"), "calibration_true_bytes_per_counted_byte": d.get("calibration_true_bytes_per_counted_byte"),
                "note": d.get("note")}
    except Exception:   # noqa: BLE001  (evidence, never a reason to fail the bench)
        return None


PARITY_TOL = 1e-9   # BASELINE.json north_star: coefficients within 1e-9 (relative, per polynomial) of the Eigen reference


class ParitySample:
    """The metric's second half ("max coeff rel-err vs Eigen"): rows of the coefficient buffers that the TIMED launch is
    going to write are overwritten with NaN beforehand (set-up, a few hundred 2-KB rows), pulled to the host right after the
    timed region together with their inputs, and compared -- in the cpu_baseline leg, the only place of this file that may
    call oracle/ -- with the C++ restatement (`port`) and, where the prebuilt library travelled, the reference's own code
    (oracle/_ref/libmtg_ref.so).  A row the timed launch did not write stays NaN and fails the check."""

    def __init__(self, nsets, steps, warmup, n_rows, seed=20260925, settle_sets=()):
        import numpy as np
        timed_sets = [(warmup + i) % nsets for i in range(steps)]
        warm_sets = {i % nsets for i in range(warmup)} | set(settle_sets)   # (+ what the settle phase rewrites after the prefill)
        uniq = list(dict.fromkeys(timed_sets))
        only_timed = [s for s in uniq if s not in warm_sets]
        # sets nobody but the timed launch writes between the prefill and the read-back; else every set it writes
        self.after_warmup = len(only_timed) < min(3, len(uniq))
        pool = uniq if self.after_warmup else only_timed
        k = min(len(pool), 4)
        self.sets = [pool[(j * len(pool)) // k] for j in range(k)]
        self.n_rows, self.rng, self.items, self.fill = n_rows, np.random.default_rng(seed), [], {}

    def add(self, set_index, label, n, deriv, masks, t, f, co, layout, flat=None, flat_offset=0):
        """One (buffer set, problem shape) source; t / f / co are the device tensors of that set.  flat / flat_offset: the
        1-D tensor `co` is a view of and its first element's index there (config 4: the twelve buckets of a buffer set share
        one allocation, so that the NaN prefill is ONE small launch per buffer set)."""
        import torch
        B = co.shape[0]
        cnt = min(B, self._per_item)
        idx = torch.from_numpy(self.rng.choice(B, size=cnt, replace=False)).to(co.device).sort().values
        row = co[0].numel()
        base = co.view(-1) if flat is None else flat
        elems = (flat_offset + idx[:, None] * row + torch.arange(row, device=co.device)[None, :]).reshape(-1)
        self.items.append(dict(set=set_index, label=label, n=n, deriv=deriv, masks=list(masks), t=t, f=f, co=co, idx=idx,
                               layout=layout))
        key = base.data_ptr()
        if key not in self.fill:
            self.fill[key] = [base, []]
        self.fill[key][1].append(elems)

    def plan(self, n_sources_per_set):
        self._per_item = max(1, -(-self.n_rows // max(1, len(self.sets) * n_sources_per_set)))

    def seal(self):
        import torch
        self.fill = {k: [b, torch.cat(e)] for k, (b, e) in self.fill.items()}

    def prefill(self):
        for base, elems in self.fill.values():     # one small launch per buffer set
            base[elems] = float("nan")

    def collect(self):
        """Right after the timed region: the sampled rows and their inputs to the host (AoS, the oracles' layout)."""
        for it in self.items:
            idx = it["idx"]
            t, f = it["t"], it["f"]
            soa = it["layout"] in ("soa", "soa16")    # (padded SoA: the sampled columns are below the batch size)
            it["times_h"] = (t[:, idx].t() if soa else t[idx]).contiguous().cpu().numpy()
            it["fixed_h"] = (f[:, :, idx].permute(2, 0, 1) if soa else f[idx]).contiguous().cpu().numpy()
            it["coeffs_h"] = it["co"][idx].cpu().numpy()
            for key in ("t", "f", "co"):
                it[key] = None
        self.fill = {}

    def check(self):
        """(cpu_baseline leg) compare with the oracles; returns the `parity` object of the line."""
        import numpy as np
        from oracle import cpu_ref, ref_linear
        have_ref = ref_linear.available()

        def relerr(c, cref):   # per trajectory: max over (segment, dimension) of ||c - cref||_inf / ||cref||_inf
            num = np.abs(c - cref).max(axis=-1)
            den = np.abs(cref).max(axis=-1)
            e = num / np.where(den == 0, 1.0, den)
            e = np.where(np.isfinite(e), e, np.inf)
            return e.reshape(e.shape[0], -1).max(axis=1)

        def arbitrate(it, rows, c_port, c_ref):
            """Rows above the tolerance: the 50-digit solution (oracle/oracle_mp.py) says which side is off.  All float64
            evaluations of the reference's formulas sit 1e-11 .. 1e-9 (long, ill-conditioned N = 10 chains) from it."""
            from oracle import oracle_mp
            out = []
            for r in rows[:24]:     # EVERY row above the tolerance (round 4 arbitrated three per source; a batch with more than 24 fails)
                c_mp = oracle_mp.solve_batch(it["n"], it["deriv"], it["masks"], it["times_h"][r:r + 1], it["fixed_h"][r:r + 1])[0]
                c_mp = np.asarray(c_mp, dtype=np.float64)
                row = {"label": it["label"], "gpu_vs_50_digit_solution": float(relerr(it["coeffs_h"][r:r + 1], c_mp)[0]),
                       "port_vs_50_digit_solution": float(relerr(c_port[r:r + 1], c_mp)[0])}
                if c_ref is not None:
                    row["reference_build_vs_50_digit_solution"] = float(relerr(c_ref[r:r + 1], c_mp)[0])
                out.append(row)
            return out

        per_n, n_tot, unwritten, arbitrated, not_arbitrated = {}, 0, 0, [], 0
        for it in self.items:
            c = it["coeffs_h"]
            unwritten += int(np.isnan(c).any(axis=(1, 2, 3)).sum())
            c_port = cpu_ref.solve_batch(it["n"], it["deriv"], it["masks"], it["times_h"], it["fixed_h"], want_free=False,
                                         want_cost=False)[0]
            e_port = relerr(c, c_port)
            e_ref = None
            if have_ref:
                c_ref = ref_linear.solve_batch(it["n"], it["deriv"], it["masks"], it["times_h"], it["fixed_h"], want_free=False,
                                               want_cost=False)[0]
                e_ref = relerr(c, c_ref)
            if it["n"] <= 10:
                worst = np.maximum(e_port, e_ref) if e_ref is not None else e_port
                over = [int(r) for r in np.argsort(-worst) if worst[r] > PARITY_TOL and np.isfinite(worst[r])]
                if over:
                    arbitrated += arbitrate(it, over, c_port, c_ref if have_ref else None)
                    not_arbitrated += max(0, len(over) - 24)
            d = per_n.setdefault(it["n"], {"port": [], "ref": []})
            d["port"].append(e_port)
            if e_ref is not None:
                d["ref"].append(e_ref)
            n_tot += c.shape[0]
        out_n, ok = {}, unwritten == 0
        for n, d in sorted(per_n.items()):
            ep = np.concatenate(d["port"])
            er = np.concatenate(d["ref"]) if d["ref"] else None
            row = {"n": int(ep.size), "max_rel_err_vs_port": float(ep.max()), "median_rel_err_vs_port": float(np.median(ep)),
                   "max_rel_err_vs_reference_build": None if er is None else float(er.max()),
                   "median_rel_err_vs_reference_build": None if er is None else float(np.median(er))}
            if n <= 10:
                row["tol"] = PARITY_TOL
                row["ok"] = bool(ep.max() <= PARITY_TOL and (er is None or er.max() <= PARITY_TOL))
                if not row["ok"] and arbitrated and np.isfinite(ep.max()):
                    # EVERY sample above the tolerance was arbitrated: accepted only if the GPU result is within the tolerance of
                    # the 50-digit solution on each of them
                    row["ok"] = row["ok_by_arbitration"] = bool(not_arbitrated == 0 and
                                                                all(a["gpu_vs_50_digit_solution"] <= PARITY_TOL for a in arbitrated))
            else:
                # N = 12: cond(A) reaches 1e17; every float64 evaluation of the reference's formulas (the compiled reference
                # included) sits ~1e-8 (worst trajectories 1e-6) from the 50-digit solution -- the tolerance is the reference's
                # own distance to the truth (tests/test_gpu_vs_reference.py: median < 1e-8, max < 1e-5)
                row["tol"] = {"median": 1e-8, "max": 1e-5}
                worst = er if er is not None else ep
                row["ok"] = bool(np.median(worst) <= 1e-8 and worst.max() <= 1e-5)
            ok = ok and row["ok"]
            out_n[str(n)] = row
        allp = np.concatenate([np.concatenate(d["port"]) for d in per_n.values()])
        allr = [np.concatenate(d["ref"]) for d in per_n.values() if d["ref"]]
        le10 = [k for k in per_n if k <= 10]
        out = {"metric": "max over sampled trajectories, segments and dimensions of ||c_gpu - c_ref||_inf / ||c_ref||_inf "
                         "(SURVEY.md 8(d) parity metric), outputs of the TIMED launch",
               "n": n_tot, "buffer_sets_sampled": self.sets, "rows_prefilled_with_nan": True,
               "prefilled": "after the warm-up steps" if self.after_warmup else "before the warm-up steps (sets the warm-up does not write)",
               "rows_not_written_by_the_timed_launch": unwritten, "tol": PARITY_TOL,
               "max_rel_err_vs_port": float(max(np.concatenate(per_n[k]["port"]).max() for k in le10)) if le10 else float(allp.max()),
               "max_rel_err_vs_reference_build": (float(max(np.concatenate(per_n[k]["ref"]).max() for k in (le10 or per_n)))
                                                  if allr else None),
               "reference_build": "oracle/_ref/libmtg_ref.so (the reference's own code against the Eigen/glog stand-ins)" if have_ref
                                  else "not shipped to this box",
               "port": "oracle/cpu_ref.cpp (C++ restatement of the reference algorithm)",
               "ok": bool(ok)}
        if arbitrated:
            out["above_tol_arbitrated_by_the_50_digit_solution"] = arbitrated
        if len(out_n) > 1 or any(k > 10 for k in per_n):
            out["per_n"] = out_n
            out["max_rel_err_is"] = "over the N <= 10 samples (the north-star tolerance); N = 12: per_n, with the tolerance the tests use"
        return out


class SolveLoop:
    """The timed loop: `steps` independent batches rotating over the buffer sets, handed to the library by ONE call of
    mtg_solve_linear_sequence_events (C ABI).  per_batch=False: the library runs the queue as one persistent launch
    (<= 96 batches per launch); per_batch=True: one kernel launch per batch, enqueued back to back (a Python loop over
    mtg_solve_linear would cost ~8 us per call -- more than the kernel)."""

    def __init__(self, plan, sets, layout, dims, per_batch=False):
        from mav_trajectory_generation_amd import _lib as L
        self.plan = plan
        self.flags = {"auto": 0, "fused": L.FLAG_FUSED_DIMS, "split": L.FLAG_SPLIT_DIMS, "dimlane": L.FLAG_DIMLANE}[dims]
        if per_batch:
            self.flags |= L.FLAG_SEQUENCE_ONE_LAUNCH_PER_BATCH
        self.batch = sets[0][2].shape[0]
        self.lay = plan.layout(self.batch, layout)
        self.ptrs = [(t.data_ptr(), f.data_ptr(), co.data_ptr()) for (t, f, co) in sets]
        self._cache = {}

    def _arrays(self, steps, first, within=None):
        n = within or len(self.ptrs)        # within: rotate over the first `within` buffer sets only (the settle phase)
        key = (steps, first % n, n)
        if key not in self._cache:
            arr = [(ctypes.c_void_p * steps)(*[self.ptrs[(first + i) % n][j] for i in range(steps)]) for j in range(3)]
            self._cache[key] = arr
        return self._cache[key]

    def prepare(self, steps, first=0, within=None):
        self._arrays(steps, first, within)

    def run(self, steps, first=0, start_event=None, stop_event=None, within=None):
        """start_event / stop_event: torch.cuda.Event objects (already recorded once, so that their hipEvent_t exists)
        recorded by the library right before the first / after the last launch -- the measured interval then does not
        contain Python's latency between an `event.record()` and the first launch."""
        if steps <= 0:
            return
        t, f, c = self._arrays(steps, first, within)
        ev = [ctypes.c_void_p(e.cuda_event) if e is not None else None for e in (start_event, stop_event)]
        rc = self.plan.lib.mtg_solve_linear_sequence_events(self.plan.handle, steps, self.batch, ctypes.byref(self.lay),
                                                            t, f, c, self.flags, ev[0], ev[1])
        if rc != 0:
            raise RuntimeError(f"mtg_solve_linear_sequence failed: {rc}")


class MixedLoop:
    """BASELINE config 4: one step = ONE mixed request of 12 buckets (N in {8, 10, 12} x K in {4, 8, 16, 32}, D = 3) x
    `per_bucket` trajectories, rotating over the buffer sets (each set = its own inputs and outputs for all twelve buckets).
    per_step_call=True: one mtg_multi_solve call (one cross-structure kernel launch) per step -- the latency form.
    per_step_call=False (default, the analogue of the queue of config 2): the K independent requests of a timed region are
    handed to the library as ONE request of 12 K items (mtg_multi_create takes any number of items), i.e. one cross-structure
    launch whose persistent workgroups are assigned the units of all K requests by the longest-processing-time schedule.
    Round 5: that request is BUILT INSIDE run(), i.e. inside the timed region (rounds 3-4 built it once outside and reported an
    amortised figure): the items of a buffer set are kept as a packed mtg_multi_item array, a region's request is the
    concatenation of K such arrays and one mtg_multi_create call -- whose unit schedule the library caches per structure."""

    SHAPES = [(n, d, k) for (n, d) in ((8, 3), (10, 4), (12, 5)) for k in (4, 8, 16, 32)]

    def __init__(self, m, ctx, per_bucket, nsets, dev, seed, per_step_call=False):
        self.solver = m.MixedBatchSolver(ctx, n_streams=1)
        self.per_step_call = per_step_call
        self.sets, self.reqs, self.packed, self.live, self.build_us, self.create_us = [], [], [], [], {}, {}
        self.m, self.ctx = m, ctx
        import torch
        for s in range(nsets):
            buckets = []
            # the twelve coefficient buffers of a set are views of ONE allocation (16-byte aligned pieces: N is even)
            flat = torch.zeros((sum(per_bucket * k * 3 * n for (n, _, k) in self.SHAPES),), dtype=torch.float64, device=dev)
            off = 0
            for (n, d, k) in self.SHAPES:
                masks = m.ends_full_masks(n, k, 1)
                t, f = m.random_waypoint_batch(per_bucket, k, 3, n, masks, seed=seed + 1000 * s + k + n, device=dev, layout="soa")
                co = flat[off:off + per_bucket * k * 3 * n].view(per_bucket, k, 3, n)
                buckets.append(dict(n_coeffs=n, derivative=d, masks=masks, times=t, d_fixed=f, layout="soa", coeffs=co,
                                    flat=flat, flat_offset=off))
                off += per_bucket * k * 3 * n
            self.sets.append(buckets)
            self.reqs.append(self.solver.merged(buckets))
            self.packed.append(m.pack_multi_items([dict(plan=self.solver.plan_for(b["n_coeffs"], 3, len(b["masks"]) - 1, b["derivative"], b["masks"], 0),
                                                        times=b["times"], d_fixed=b["d_fixed"], coeffs=b["coeffs"], layout="soa") for b in buckets]))
        # position-only interior vertices => n_fixed = N + K - 1
        self.bytes_per_step = sum(per_bucket * 8 * (k + 3 * (n + k - 1) + k * 3 * n) for (n, _, k) in self.SHAPES)
        self.per_step = per_bucket * len(self.SHAPES)
        self.launches_per_step = self.reqs[0].launch_count

    def prepare(self, steps, first=0):
        """Outside any timed region: requests of earlier runs are destroyed here (mtg_multi_destroy waits for the stream)."""
        while len(self.live) > 1:
            self.live.pop(0).close()

    def launches(self, steps):
        return steps * self.launches_per_step if self.per_step_call else self.launches_per_step

    def run(self, steps, first=0, start_event=None, stop_event=None):
        if steps <= 0:
            return
        stream = self.solver.ctx.stream
        req = None
        if not self.per_step_call:
            t0 = time.perf_counter()
            n = len(self.sets)
            packed = np.concatenate([self.packed[(first + i) % n] for i in range(steps)]) if steps > 1 else self.packed[first % n]
            req = self.m.PackedMultiSolve(self.ctx, packed)
            self.live.append(req)
            self.build_us[(steps, first % n)] = (time.perf_counter() - t0) * 1e6   # concatenate + mtg_multi_create, every run
            self.create_us[(steps, first % n)] = req.create_us
        if start_event is not None:
            start_event.record(stream)
        if self.per_step_call:
            n = len(self.reqs)
            for i in range(steps):
                self.reqs[(first + i) % n].solve()
        else:
            req.solve(ordered=False)   # (events are recorded on ctx.stream around the call; the bench syncs explicitly)
        if stop_event is not None:
            stop_event.record(stream)

    def outputs(self):
        return [b["coeffs"] for s in self.sets for b in s]


FP64_VALU_PEAK = {"simds": 1024, "clock_hz": 2.4e9, "cycles_per_wave_instruction": 4.0, "tflops": 78.6}


def fp64_issue_roofline(row, us, B):
    """The compute-bound (f) rows against the FP64 VECTOR peak: one VALU instruction per 4 cycles and SIMD (for FMAs: 78.6 TFLOP/s,
    /opt/skills/guides/MI355X_MICROARCH.md).  VALU instructions per call from the committed rocprofv3 PMC pass of the same row at
    this size (profiles/r04_next_rows_pmc.json, tools/gpu_profile_rows.sh: its own profiling run); the duration is this run's."""
    path = next((q for q in (os.path.join(ROOT, "profiles", f"r0{r}_next_rows_pmc.json") for r in (6, 5, 4)) if os.path.exists(q)), "")
    if B != 10_000 or not os.path.exists(path):
        return None
    try:
        d = json.load(open(path)).get(row)
    except Exception:
        return None
    if not d or not d.get("valu_insts_per_call"):
        return None
    pk = FP64_VALU_PEAK
    issue_s = d["valu_insts_per_call"] * pk["cycles_per_wave_instruction"] / (pk["simds"] * pk["clock_hz"])
    frac = issue_s / (us * 1e-6)
    f64 = d.get("f64_insts_per_call")
    if f64 and sum(f64.values()) > 0:
        # round 6: flops from the FP64 instruction counters (wave instructions x 64 lanes; FMA = 2 flops, MUL / ADD / TRANS = 1)
        flops = 64.0 * (2.0 * f64.get("FMA_F64", 0.0) + f64.get("MUL_F64", 0.0) + f64.get("ADD_F64", 0.0) + f64.get("TRANS_F64", 0.0))
        achieved = flops / (us * 1e-6) * 1e-12
        return {"bound": "fp64", "peak": pk["tflops"], "unit": "TFLOP/s", "frac": achieved / pk["tflops"], "achieved": achieved,
                "achieved_is": "FP64 flops from SQ_INSTS_VALU_{FMA,MUL,ADD,TRANS}_F64 of the committed PMC pass (FMA = 2 flops per lane) / this run's duration",
                "fp64_wave_instructions_per_call": f64, "valu_instructions_per_call": d["valu_insts_per_call"],
                "fp64_share_of_valu_instructions": sum(f64.values()) / d["valu_insts_per_call"],
                "valu_issue_us_at_peak": issue_s * 1e6, "valu_issue_frac": frac,
                "valu_issue_frac_is": "all VALU instructions x 4 cycles / (1024 SIMDs x 2.4 GHz) / duration: how busy the issue slots are, whatever they issue",
                "counters_from": os.path.relpath(path, ROOT)}
    return {"bound": "fp64", "peak": pk["tflops"], "unit": "TFLOP/s", "valu_instructions_per_call": d["valu_insts_per_call"],
            "valu_issue_us_at_peak": issue_s * 1e6, "frac": frac, "achieved": frac * pk["tflops"],
            "achieved_is": "upper bound: as if every VALU instruction were an FP64 FMA (64 lanes x 2 flops)",
            "counters_from": os.path.relpath(path, ROOT)}


def next_rows(m, ctx, plan, sets, B, K, D, N):
    """SURVEY 8(f) rows on the config's batch (coefficients of buffer set 0): device durations from events on the library's
    stream, each with the roofline that bounds it."""
    import torch
    t, f, co = sets[0]
    tt = t.t().contiguous() if t.shape[0] == K else t          # [B][K] for the post-solve kernels
    out = {}
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def timed(fn, reps):
        fn()
        torch.cuda.synchronize()
        e0.record(ctx.stream)
        for _ in range(reps):
            fn()
        e1.record(ctx.stream)
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / reps

    # N3: batched sampling, 100 samples per trajectory, position .. snap (HBM-bound: the write stream)
    ns, nd = 100, 5
    us = timed(lambda: m.sample_range(ctx, co, tt, 0.0, 0.05, ns, nd), 20)
    by = B * ns * nd * D * 8 + B * (K * D * N + K) * 8
    out["sample"] = {"what": f"{B} trajectories x {ns} samples x derivatives 0..{nd - 1} (Trajectory::evaluateRange)",
                     "us": us, "samples_per_s": B * ns / us * 1e6,
                     "roofline": {"bound": "hbm", "bytes": by, "achieved_GBps": by / us * 1e-3, "frac": by / us * 1e-3 / HBM_PEAK_GBS}}
    # N4: velocity + acceleration maxima per segment, then the feasibility time scaling (FP64-issue-bound root isolation)
    us_v = timed(lambda: m.minmax_magnitude(ctx, co, tt, 1), 5)
    out["extrema"] = {"what": f"{B} x {K} segments, velocity magnitude extrema (Trajectory::computeMinMaxMagnitude)", "us": us_v,
                      "segment_derivatives_per_s": B * K / us_v * 1e6,
                      "roofline": fp64_issue_roofline("extrema", us_v, B) or {"bound": "fp64", "frac": None},
                      "hbm_frac": B * K * (D * N * 8 + 32) / us_v * 1e-3 / HBM_PEAK_GBS}
    us_s = timed(lambda: m.scale_segment_times_to_meet_constraints(ctx, co.clone(), tt.clone(), 2.0, 3.0), 3)
    out["time_scaling"] = {"what": f"{B} trajectories, scaleSegmentTimesToMeetConstraints(v_max 2, a_max 3), incl. the copies "
                                   f"of its inputs", "us": us_s, "traj_per_s": B / us_s * 1e6,
                           "roofline": fp64_issue_roofline("time_scaling", us_s, B) or {"bound": "fp64", "frac": None}}
    # N2: Mellinger cost + gradient = (K + 1) x B perturbed-time cost-only solves in one launch
    lay = "soa" if t.shape[0] == K else "aos"
    us_m = timed(lambda: m.mellinger_cost_and_gradient(plan, t, f, layout=lay), 10)
    out["mellinger"] = {"what": f"{B} trajectories x {K + 1} perturbed-time cost-only solves (getCostAndGradientMellinger)",
                        "us": us_m, "solves_per_s": B * (K + 1) / us_m * 1e6,
                        "roofline": fp64_issue_roofline("mellinger", us_m, B) or {"bound": "fp64", "frac": None},
                        "hbm_frac": B * (K + 1) * (K + D * plan.n_fixed + 1) * 8 / us_m * 1e-3 / HBM_PEAK_GBS}
    return out


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
        ranks_seen, devices = 1, [0]
        if world > 1:
            dist.init_process_group(args.backend if args.backend != "nccl" else "gloo")
            assert dist.get_world_size() == args.gpus
            tt = torch.tensor([float(rank + 1)], dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            assert float(tt.item()) == world
            ones = torch.ones(1, dtype=torch.float64)
            dist.all_reduce(ones)
            ranks_seen = int(ones.item())
            dv = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
            dist.all_gather(dv, torch.tensor([0 if args.same_device else local], dtype=torch.int64))
            devices = [int(x.item()) for x in dv]
            dist.barrier()
        if rank == 0:
            print(json.dumps({"plumbing_only": True, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                              "ranks_seen": ranks_seen, "rank_devices": devices}))
        if world > 1:
            dist.destroy_process_group()
        return

    import mav_trajectory_generation_amd as m

    device_index = 0 if args.same_device else local
    torch.cuda.set_device(device_index)
    dev = torch.device("cuda", device_index)
    grouped = world > 1 or args.exercise_collectives
    if grouped:
        if not launched:     # one rank, no launcher: rendezvous with ourselves
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)
        assert dist.get_world_size() == args.gpus
    red_dev = dev if (grouped and args.backend == "nccl") else "cpu"
    ranks_seen, rank_devices = 1, [device_index]
    if grouped:   # evidence that the collective backend really spans `world` ranks, and where each rank runs
        ones = torch.ones(1, dtype=torch.float64, device=red_dev)
        dist.all_reduce(ones)
        ranks_seen = int(ones.item())
        dv = [torch.zeros(1, dtype=torch.int64, device=red_dev) for _ in range(world)]
        dist.all_gather(dv, torch.tensor([device_index], dtype=torch.int64, device=red_dev))
        rank_devices = [int(x.item()) for x in dv]
        assert ranks_seen == world

    cfg = CONFIGS[args.config]
    if args.layout is None:
        args.layout = "soa16" if args.config == 5 else "soa"
    mixed = args.config == 4
    N, K, D, d = 10, (cfg["K"] or 8), cfg["D"], 4
    B = args.batch if args.batch is not None else cfg["batch"]
    masks = m.ends_full_masks(N, K, cfg["interior"])
    ctx = m.Context(device_index)
    plan = m.Plan(ctx, N, D, K, d, masks)
    per_batch = args.sequence == "launches"
    if args.buffer_sets is None:
        if mixed:
            nsets = max(16, min(args.steps + args.warmup, 32))
        else:
            per_set = B * plan.bytes_per_trajectory
            nsets = max(16, args.steps + args.warmup, -(-(5 * 2**28) // per_set))
            nsets = min(nsets, 256, max(2, (24 * 2**30) // per_set))
    else:
        nsets = max(1, args.buffer_sets)

    def barrier():
        if grouped:
            dist.barrier()
        torch.cuda.synchronize()

    def make_set(seed):
        t, f = m.random_waypoint_batch(B, K, D, N, masks, seed=seed, device=dev, layout=args.layout, yaw_dim=cfg["yaw"])
        return t, f, torch.zeros((B, K, D, N), dtype=torch.float64, device=dev)

    with torch.cuda.stream(ctx.stream):
        if mixed:
            loop = MixedLoop(m, ctx, B, nsets, dev, 1234 + rank, per_step_call=per_batch)
            sets = None
            bytes_per_step, traj_per_step = loop.bytes_per_step, loop.per_step
            set_bytes = bytes_per_step
        else:
            # keep the rotating footprint bounded (config 3: 300 MB per set)
            set_bytes = B * (plan.bytes_per_trajectory)
            while nsets > 2 and nsets * set_bytes > 24 * 2**30:
                nsets //= 2
            sets = [make_set(1234 + rank + 1000 * s) for s in range(nsets)]
            loop = SolveLoop(plan, sets, args.layout, args.dims, per_batch)
            bytes_per_step, traj_per_step = B * plan.bytes_per_trajectory, B
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(ctx.stream)   # (creates the underlying hipEvent_t handles: torch makes them on first record)
        e1.record(ctx.stream)

        stamps = {}

        def finish():
            while not e1.query():   # spin on the end event (a blocking synchronize alone wakes up ~30 us late) ...
                pass
            stamps["done"] = time.perf_counter()
            barrier()               # ... then the contract's barrier + torch.cuda.synchronize()

        def timed(loop_, steps, warmup, parity_=None, prefilled=False):
            loop_.prepare(warmup, 0)
            loop_.prepare(steps, warmup)
            if parity_ is not None and not parity_.after_warmup and not prefilled:
                parity_.prefill()       # NaN into the sampled output rows (sets the warm-up steps do not write)
            loop_.run(warmup)
            if parity_ is not None and parity_.after_warmup:
                parity_.prefill()
            barrier()
            t0 = time.perf_counter()
            loop_.run(steps, first=warmup, start_event=e0, stop_event=e1)   # events recorded around the launches
            stamps["enqueued"] = time.perf_counter()
            finish()
            t1 = time.perf_counter()
            # where the wall clock of the timed region goes (host side): the library call that enqueues the launch(es), the wait
            # for the stop event, the contract's barrier + synchronize
            stamps["breakdown_us"] = {"enqueue_call": (stamps["enqueued"] - t0) * 1e6, "until_stop_event": (stamps["done"] - stamps["enqueued"]) * 1e6,
                                      "barrier_and_synchronize": (t1 - stamps["done"]) * 1e6, "wall": (t1 - t0) * 1e6}
            return t1 - t0, e0.elapsed_time(e1) * 1e3 / steps   # wall seconds, device us per step

        def side_run(loop_, steps, warm=20):
            """(device us per step, wall s) of `steps` steps outside the contract's timed region (extras)."""
            loop_.prepare(warm, 0)
            loop_.prepare(steps, 0)
            loop_.run(warm)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            loop_.run(steps, start_event=e0, stop_event=e1)
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) * 1e3 / steps, time.perf_counter() - t0

        def side_entry(us, wall, steps, sets_):
            return {"buffer_sets": sets_, "steps": steps, "us_per_step": us, "traj_per_s": traj_per_step * steps / wall,
                    "frac_of_8TBps": bytes_per_step / us * 1e-3 / HBM_PEAK_GBS}

        # The settle phase of the single-structure configs rotates over the warm-up's buffer sets only, so that the parity
        # sample's NaN prefill can precede it: NO host work but the library calls themselves then sits between the settle
        # phase and the timed call (visit K of round 4, profiles/r04k_*: the same call costs 5 us of host time in a loop,
        # 7-8 us right after such calls, and 12-21 us after the prefill's few torch launches -- 5-10 % of a 20-step region).
        settle_within = max(1, min(args.warmup, nsets)) if (args.settle_ms > 0 and not mixed) else None
        # (built BEFORE the settle phase: no host work may sit between that phase and the contract's warm-up + timed steps)
        parity = None
        if rank == 0 and not args.no_parity:
            # (config 4: >= 512 sampled trajectories per polynomial order)
            parity = ParitySample(nsets, args.steps, args.warmup, args.parity_samples * (3 if mixed else 1),
                                  settle_sets=range(settle_within or 0))
            if mixed:
                parity.plan(len(loop.SHAPES))
                for si in parity.sets:
                    for b in loop.sets[si]:
                        parity.add(si, f"N={b['n_coeffs']} K={len(b['masks']) - 1}", b["n_coeffs"], b["derivative"], b["masks"],
                                   b["times"], b["d_fixed"], b["coeffs"], "soa", b["flat"], b["flat_offset"])
            else:
                parity.plan(1)
                for si in parity.sets:
                    parity.add(si, f"N={N} K={K} D={D}", N, d, masks, sets[si][0], sets[si][1], sets[si][2], args.layout)
            parity.seal()
        # set-up, not a step: every buffer set is touched once (first use of fresh allocations: page-table / TLB fills),
        # as in any pipeline that has been running for longer than one rotation
        loop.run(nsets)
        torch.cuda.synchronize()
        cold, prefilled = None, False
        if args.settle_ms > 0:
            # A fresh process starts on a GPU that has just come out of idle (rocm-smi: sclk level 543 MHz); the contract's W
            # warm-up steps are ~30 us of work, and the first timed region then runs 10-15 % slower than every later one
            # (tools/gpu_r3_t.sh: 98-102 us vs 87-91 us per 20-batch launch).  Reported, not hidden: the same warm-up + timed
            # region is run ONCE before the settle phase and kept as `cold_start` in the line; then the device is kept busy
            # with the same work for settle_ms (set-up, not a step), then the contract's warm-up + timed steps follow.
            dt_c, us_c = timed(loop, args.steps, args.warmup)
            cold = {"kernel_us_per_step": us_c, "frac": bytes_per_step / us_c * 1e-3 / HBM_PEAK_GBS, "wall_us": dt_c * 1e6,
                    "units_per_s_this_rank": traj_per_step * args.steps / dt_c, "wall_breakdown_us": dict(stamps["breakdown_us"]),
                    "is": "the same warm-up + timed region, run once BEFORE the settle phase: what a process that starts on an idle GPU sees first"}
            if settle_within and parity is not None and not parity.after_warmup:
                parity.prefill()
                prefilled = True
            kw = {"within": settle_within} if settle_within else {}
            loop.prepare(nsets, 0, **kw)
            t_end = time.perf_counter() + args.settle_ms * 1e-3
            while time.perf_counter() < t_end:
                loop.run(nsets, **kw)
                torch.cuda.synchronize()
        dt, step_us = timed(loop, args.steps, args.warmup, parity, prefilled)
        wall_breakdown = dict(stamps["breakdown_us"])
        ctx.sync()  # raises if any trajectory flagged bad time / singular
        if parity is not None:
            parity.collect()    # the sampled rows of the TIMED launch's outputs, before anything else rewrites the buffers
        for co in (loop.outputs() if mixed else [x[2] for x in sets]):
            assert torch.isfinite(co).all()

        extra, fill_us, peer = {}, None, None
        # ---- `sustained` (round 6, first-class field): the same queue launches, continuously for >= --sustained-seconds over the
        # same HBM-sized rotation (calls of up to 96 batches, enqueued back to back: no host gap), every rank at once; the shader
        # clock is probed next to them by a one-wave kernel on its own stream (s_memtime against the 100 MHz s_memrealtime:
        # include/mtg_hip_lab.h) -- the FP64-heavy kernels run power-limited, which a 90-us burst after a 50-ms settle does not show
        sustained = None
        if args.sustained_seconds > 0 and not args.no_extras and not per_batch:
            chunk = min(96, nsets) if not mixed else min(16, nsets)
            est_us = max(step_us, 0.5) * chunk
            calls = max(8, int(args.sustained_seconds * 1e6 / est_us) + 1)
            loop.prepare(chunk, 0)
            loop.run(chunk)
            barrier()
            probe = None
            try:
                probe = ctx.clock_probe_start(min(0.6 * args.sustained_seconds * 1e6, 0.6 * est_us * calls))
            except Exception:   # noqa: BLE001  (the probe is evidence, not part of the measurement)
                probe = None
            t0 = time.perf_counter()
            e0.record(ctx.stream)
            for _ in range(calls):
                loop.run(chunk)
            e1.record(ctx.stream)
            finish()
            wall_s = time.perf_counter() - t0
            dev_us = e0.elapsed_time(e1) * 1e3 / (calls * chunk)
            mhz = None
            if probe is not None:
                try:
                    mhz, _ = ctx.clock_probe_finish(probe)
                except Exception:   # noqa: BLE001
                    mhz = None
            if grouped:
                w = torch.tensor([wall_s, dev_us], dtype=torch.float64, device=red_dev)
                dist.all_reduce(w, op=dist.ReduceOp.MAX)
                wall_s, dev_us = float(w[0].item()), float(w[1].item())
            sustained = {"value": world * traj_per_step * calls * chunk / wall_s, "unit": "trajectories/s", "seconds": wall_s,
                         "steps": calls * chunk, "calls": calls, "batches_per_call": chunk, "buffer_sets": nsets,
                         "device_us_per_step": dev_us, "roofline_frac": bytes_per_step / dev_us * 1e-3 / HBM_PEAK_GBS,
                         "shader_clock_mhz": mhz,
                         "is": "the queue form run continuously (calls enqueued back to back, max over ranks) over the same rotation as "
                               "`value`; shader clock = s_memtime / s_memrealtime of a one-wave probe kernel running next to it (this rank)"}
            ctx.sync()
        # ---- the queue form on AoS inputs (times[B][K], d_fixed[B][D][n_fixed]: the order a reference-side caller's Vertex::Vector
        # packs into), same protocol as value_other_form: W warm-up + K timed steps, wall clock
        aos_peer = None
        if rank == 0 and not args.no_extras and not mixed and args.layout != "aos":
            try:
                aos_sets = []
                for (t_, f_, co_) in sets:
                    ta = t_[:, :B].t().contiguous()
                    fa = f_[:, :, :B].permute(2, 0, 1).contiguous()
                    aos_sets.append((ta, fa, co_))
                aos_loop = SolveLoop(plan, aos_sets, "aos", args.dims, per_batch)
                aos_loop.run(len(aos_sets))
                torch.cuda.synchronize()
                us, wall = side_run(aos_loop, args.steps, warm=args.warmup)
                aos_peer = {"value": world * traj_per_step * args.steps / wall, "unit": "trajectories/s", "device_us_per_step": us,
                            "roofline_frac": bytes_per_step / us * 1e-3 / HBM_PEAK_GBS, "steps": args.steps, "warmup": args.warmup,
                            "input_layout": "aos", "launch_form": plan.launch_form(B, "aos"),
                            "is": "the same steps with AoS inputs (times [B][K], d_fixed [B][D][n_fixed] -- the reference's natural order), "
                                  "same output buffers, this rank's clock"}
                ctx.sync()
                del aos_loop, aos_sets
            except Exception as e:   # noqa: BLE001
                aos_peer = {"error": repr(e)[:300]}
        if args.timed_repeats > 0:
            reps_ = []
            for _ in range(args.timed_repeats):
                dt_r, us_r = timed(loop, args.steps, args.warmup)
                reps_.append(dict(stamps["breakdown_us"], device_us_per_step=us_r))
            extra["timed_region_repeats"] = reps_
        if rank == 0 and not args.no_extras:
            # The OTHER hand-over form of the same K steps under the same protocol (W warm-up steps, K timed steps, wall clock
            # around them), reported as a PEER of `value`: one kernel launch per step (what a caller whose batches arrive one
            # at a time gets) when `value` is the queue form, and vice versa.
            other = (MixedLoop(m, ctx, B, nsets, dev, 1234 + rank, per_step_call=not per_batch) if mixed
                     else SolveLoop(plan, sets, args.layout, args.dims, not per_batch))
            if mixed:
                other.run(nsets)
            us, wall = side_run(other, args.steps, warm=args.warmup)
            peer = {"sequence": "queue" if per_batch else "launches", "value": world * traj_per_step * args.steps / wall,
                    "unit": "trajectories/s", "ms_per_step": wall / args.steps * 1e3, "device_us_per_step": us,
                    "roofline_frac": bytes_per_step / us * 1e-3 / HBM_PEAK_GBS, "steps": args.steps, "warmup": args.warmup,
                    "is": ("one library call and one kernel launch PER STEP, back to back on one stream -- the form a caller whose "
                           "batches arrive one at a time can reproduce (this rank's clock)") if not per_batch else
                          "the K steps handed to the library together (one persistent launch per <= 96 steps)"}
            del other
        if rank == 0 and not args.no_extras and mixed:
            us, wall = side_run(loop, 96, warm=16)
            extra["rotating_buffers_96_steps"] = side_entry(us, wall, 96, nsets)
            other = MixedLoop(m, ctx, B, nsets, dev, 1234 + rank, per_step_call=not per_batch)
            other.run(nsets)
            us, wall = side_run(other, 96, warm=16)
            extra["one_launch_per_request" if not per_batch else "requests_merged_into_one_launch"] = dict(
                side_entry(us, wall, 96, nsets), note="one mtg_multi_solve call (one cross-structure launch) per mixed request"
                if not per_batch else "the 96 requests as one 12 x 96-item request")
        if rank == 0 and not args.no_extras and not mixed:
            big = set_bytes * nsets > 8 * 2**30          # config 3: keep the extras short
            long_steps = 200 if big else 2000
            # the other form of the same loop: one launch per batch (latency form) when `value` is the queue, and vice versa
            other = SolveLoop(plan, sets, args.layout, args.dims, not per_batch)
            us, wall = side_run(other, 200)
            extra["one_launch_per_batch" if not per_batch else "queue_one_persistent_launch"] = dict(
                side_entry(us, wall, 200, nsets),
                note=("one kernel launch per batch, back to back on ONE stream: the launch-to-launch period (latency form)"
                      if not per_batch else "the queue as one persistent launch per <= 96 batches (throughput form)"))
            # the same loop over ONE resident buffer set (inputs and outputs stay in the 256 MiB Infinity Cache)
            res = SolveLoop(plan, sets[:1], args.layout, args.dims, per_batch)
            us, wall = side_run(res, max(args.steps, 200))
            extra["resident_buffers"] = side_entry(us, wall, max(args.steps, 200), 1)
            # a longer rotating run of the same loop (the driver's --steps can be as small as 20)
            if args.steps < long_steps:
                us, wall = side_run(loop, long_steps)
                extra[f"rotating_buffers_{long_steps}_steps"] = side_entry(us, wall, long_steps, nsets)
            if args.steps != 96 and not per_batch and not big:
                us, wall = side_run(loop, 96, warm=96)
                extra["rotating_buffers_96_steps_one_full_launch"] = side_entry(us, wall, 96, nsets)
            if nsets > 16:
                # the rotation rounds 2-4 reported `value` on (16 sets: 382 MiB for config 2, 1.4x the Infinity Cache), same steps
                small = SolveLoop(plan, sets[:16], args.layout, args.dims, per_batch)
                us, wall = side_run(small, args.steps, warm=args.warmup)
                extra["rotation_over_16_sets_as_in_rounds_2_to_4"] = side_entry(us, wall, args.steps, 16)
            if args.layout == "soa16":
                # the same steps with the PLAIN SoA stride (the layout of rounds 1-3 and of every other config): the caller-side
                # layout is part of the number -- both are in the line
                def make_plain(seed):
                    t_, f_ = m.random_waypoint_batch(B, K, D, N, masks, seed=seed, device=dev, layout="soa", yaw_dim=cfg["yaw"])
                    return t_, f_, torch.zeros((B, K, D, N), dtype=torch.float64, device=dev)
                plain = SolveLoop(plan, [make_plain(777 + 1000 * s_) for s_ in range(min(nsets, 32))], "soa", args.dims, per_batch)
                us, wall = side_run(plain, args.steps, warm=args.warmup)
                extra["plain_soa_input_layout"] = dict(side_entry(us, wall, args.steps, min(nsets, 32)),
                                                       note="inputs [K][B] / [D][n_fixed][B] with row stride B (not padded to 16)")
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
                # ONE launch over the per-GPU share of BASELINE configs[2] (1M over 8 GPUs) always; --extra adds 1M on this GPU
                for bigb in ((125_000, 1_000_000) if args.extra else (125_000,)):
                    tb, fb = m.random_waypoint_batch(bigb, K, D, N, masks, seed=99, device=dev, layout=args.layout)
                    cb = torch.empty((bigb, K, D, N), dtype=torch.float64, device=dev)
                    plan.solve(tb, fb, layout=args.layout, coeffs=cb, dims=args.dims, batch=bigb)
                    torch.cuda.synchronize()
                    us = plan.time_last_solve(20)
                    extra[f"batch_{bigb}"] = {"kernel_us": us, "traj_per_s": bigb / us * 1e6,
                                              "GBps": bigb * plan.bytes_per_trajectory / us * 1e-3,
                                              "frac_of_8TBps": bigb * plan.bytes_per_trajectory / us * 1e-3 / HBM_PEAK_GBS}
                    del tb, fb, cb
            if args.config == 2 and not args.no_next and args.layout != "soa16":
                extra["next"] = next_rows(m, ctx, plan, sets, B, K, D, N)
        if args.extra and rank == 0 and args.config == 2:
            # host buffers in / out (MTG_FLAG_HOST_POINTERS): PCIe-inclusive rate, never the reported value.
            t, f, _ = sets[0]
            th = (t[:, :B].t().contiguous() if args.layout != "aos" else t).cpu()
            fh = (f[:, :, :B].permute(2, 0, 1).contiguous() if args.layout != "aos" else f).cpu()
            for tag, tt_, ff_ in (("pageable", th.numpy(), fh.numpy()),
                                  ("pinned", th.pin_memory().numpy(), fh.pin_memory().numpy())):
                co_h = torch.empty((B, K, D, N), dtype=torch.float64, pin_memory=(tag == "pinned")).numpy()
                plan.solve_host(tt_, ff_, want_free=False, want_cost=False, coeffs=co_h)
                t1 = time.perf_counter()
                for _ in range(5):
                    plan.solve_host(tt_, ff_, want_free=False, want_cost=False, coeffs=co_h)
                extra[f"host_pointers_{tag}_pcie_inclusive_traj_per_s"] = 5 * B / (time.perf_counter() - t1)

        gather = None
        if grouped and not args.no_gather and not mixed:
            # solve + final all_gather of the coefficients: chunked, chunk i's gather overlaps chunk i+1's
            # solve; reported beside the solve-only number, never part of `value`
            from mav_trajectory_generation_amd import dist as mdist
            t, f, _ = sets[0]
            g_layout = args.layout
            if g_layout == "soa16":     # (the chunked runner slices the batch: plain SoA copies of the same inputs)
                t, f, g_layout = t[:, :B].contiguous(), f[:, :, :B].contiguous(), "soa"
            runner = mdist.ChunkedSolveGather(plan, t, f, layout=g_layout, n_chunks=args.gather_chunks)
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
            runner.run()
            barrier()
            own_slice_ok = bool(torch.equal(runner.gathered[:, rank], runner.local))   # this rank's slice of the gathered buffer
            gather = {"chunks": runner.n_chunks, "own_slice_matches_local_solve": own_slice_ok, "solve_plus_gather_ms": both * 1e3, "solve_only_ms": solve_only * 1e3,
                      "gather_only_ms": gather_only * 1e3,
                      "gathered_bytes_per_rank": world * B * K * D * N * 8, "backend": args.backend}
            if args.backend == "nccl" and not args.no_gather_via_mtg_comm:
                # North_star's gather, at every N (round 6; rounds 4-5: opt-in for N > 1): the same chunked solve + gather through
                # the C ABI's OWN RCCL communicator (mtg_comm_*, csrc/mtg_comm.hip: what a C++ consumer running one process per GPU
                # calls -- no torch.distributed in the data path; the unique id travels over the process group that exists anyway).
                # A second communicator's ncclCommInitRank is a collective: a rank that cannot join would leave the others waiting,
                # and the driver's scaling run must not hang on an extra.  So the init runs on a WATCHDOG thread; every rank then
                # reports over torch.distributed whether it joined in time, and only if ALL did is the communicator used.
                # Otherwise the line carries the torch.distributed figure and says so (`gather.via_mtg_comm.fell_back`), and the
                # process leaves through os._exit after printing (a thread stuck inside RCCL cannot be joined).
                try:
                    import threading
                    ids = [mdist.Communicator.unique_id() if rank == 0 else None]
                    dist.broadcast_object_list(ids, src=0)
                    box = {}

                    def init_comm():
                        # (the thread: the communicator's init AND its first collectives -- three warm-up solve + gather calls and a
                        # sync -- so that a collective that cannot complete is caught by the watchdog too, not only the init)
                        try:
                            if os.environ.get("MTG_BENCH_TEST_COMM_HANG"):   # (tests/test_bench.py: the watchdog's way out, end to end)
                                time.sleep(1e6)
                            torch.cuda.set_device(device_index)          # (the current device is per thread)
                            c_ = mdist.Communicator(ctx, rank, world, ids[0])
                            box["init_s"] = time.perf_counter() - t_init
                            with torch.cuda.stream(ctx.stream):
                                for _ in range(3):
                                    box["loc"], box["gat"] = c_.solve_all_gather(plan, t, f, layout=g_layout, n_chunks=args.gather_chunks)
                                c_.sync()
                            box["comm"] = c_
                        except Exception as e_:   # noqa: BLE001
                            box["error"] = repr(e_)[:300]
                    th = threading.Thread(target=init_comm, daemon=True)
                    t_init = time.perf_counter()
                    th.start()
                    th.join(args.mtg_comm_init_timeout)
                    joined = torch.tensor([1.0 if "comm" in box else 0.0], dtype=torch.float64, device=red_dev)
                    dist.all_reduce(joined, op=dist.ReduceOp.MIN)
                    if float(joined.item()) < 1.0:
                        stuck = th.is_alive()
                        if stuck:
                            WATCHDOG["hard_exit"] = True
                        raise RuntimeError("mtg_comm init + first collectives did not complete on every rank within %.0f s (this rank: %s); "
                                           "gather figures are torch.distributed's" % (args.mtg_comm_init_timeout,
                                           "timed out" if stuck else box.get("error", "joined")))
                    comm, loc, gat = box["comm"], box["loc"], box["gat"]
                    init_s = box["init_s"]
                    barrier()
                    t0 = time.perf_counter()
                    for _ in range(reps):
                        comm.solve_all_gather(plan, t, f, layout=g_layout, n_chunks=args.gather_chunks, local=loc, gathered=gat)
                    comm.sync()
                    barrier()
                    via = (time.perf_counter() - t0) / reps
                    gather["via_mtg_comm"] = {"solve_plus_gather_ms": via * 1e3, "own_slice_matches_local_solve": bool(torch.equal(gat[:, rank].reshape(loc.shape), loc)),
                                              "matches_torch_distributed_gather": bool(torch.equal(gat, runner.gathered)),
                                              "init_s": init_s, "default": True,
                                              "is": "mtg_comm_solve_all_gather: ncclAllGather per chunk on the communicator's stream under the next chunk's solve"}
                    comm.close()
                except Exception as e:   # noqa: BLE001
                    gather["via_mtg_comm"] = {"error": repr(e)[:300], "fell_back": "torch.distributed all_gather_into_tensor (the figures above)"}

    per_rank = None
    if grouped:
        mine = torch.tensor([dt, step_us], dtype=torch.float64, device=red_dev)
        allv = [torch.zeros(2, dtype=torch.float64, device=red_dev) for _ in range(world)]
        dist.all_gather(allv, mine)
        per_rank = [{"rank": r, "device": rank_devices[r], "wall_s": float(v[0]), "device_us_per_step": float(v[1]),
                     "units_per_s": traj_per_step * args.steps / float(v[0]),
                     "roofline_frac": bytes_per_step / float(v[1]) * 1e-3 / HBM_PEAK_GBS}   # each rank's own launch vs ITS HBM
                    for r, v in enumerate(allv)]
        vals = mine.clone()
        dist.all_reduce(vals, op=dist.ReduceOp.MAX)
        dt, step_us = float(vals[0].item()), float(vals[1].item())
        if gather is not None:
            g = torch.tensor([gather["solve_plus_gather_ms"], gather["solve_only_ms"], gather["gather_only_ms"]],
                             dtype=torch.float64, device=red_dev)
            dist.all_reduce(g, op=dist.ReduceOp.MAX)
            gather["solve_plus_gather_ms"], gather["solve_only_ms"], gather["gather_only_ms"] = [float(x) for x in g]
            gather["traj_per_s_with_gather"] = world * B / (gather["solve_plus_gather_ms"] * 1e-3)

    if rank == 0:
        achieved = bytes_per_step / (step_us * 1e-6) / 1e9
        if mixed:
            form = (f"ONE library call (mtg_multi_solve) = {loop.launches_per_step} cross-structure kernel launch(es) per step"
                    if per_batch else
                    f"the {args.steps} timed steps are {args.steps} independent requests handed to the library as ONE request of "
                    f"{12 * args.steps} items (mtg_multi_create / mtg_multi_solve) = {loop.launches_per_step} cross-structure kernel "
                    f"launch(es); one launch per request: extra.one_launch_per_request")
            what = (f"mixed request of {traj_per_step} random-waypoint trajectories per GPU per step ({cfg['name']}): 12 buckets = "
                    f"N in {{8 jerk, 10 snap, 12}} x K in {{4, 8, 16, 32}} segments x {B} trajectories, dim=3; {form}; rotating "
                    f"over {nsets} independent input/output buffer sets ({nsets * set_bytes / 2**20:.0f} MiB) resident in HBM; "
                    f"inputs SOA, coeffs [B][K][D][N] per bucket")
            launches, batches_per_launch = loop.launches(args.steps), (None if per_batch else args.steps)
        else:
            queue = not per_batch
            launches = (args.steps + 95) // 96 if queue else args.steps
            batches_per_launch = min(args.steps, 96) if queue else 1
            form = (f"the {args.steps} timed steps are {args.steps} independent batches handed to the library in ONE "
                    f"mtg_solve_linear_sequence call and run as {launches} persistent kernel launch(es) whose workgroups walk the "
                    f"tiles of all batches (throughput form; the one-launch-per-batch latency figure is extra.one_launch_per_batch)"
                    if queue else "one kernel launch per step on ONE stream (latency figure, no overlap between steps)")
            what = (f"batch of {B} random-waypoint trajectories per GPU per step ({cfg['name']}): {K} segments, N=10, "
                    f"dim={D}, snap" + (", velocity + acceleration fixed at interior vertices" if cfg["interior"] == 7 else "")
                    + f"; {form}; rotating over {nsets} independent input/output buffer sets "
                      f"({nsets * set_bytes / 2**20:.0f} MiB) resident in HBM; inputs {args.layout.upper()}"
                      + (" (SoA, row stride padded to a multiple of 16 trajectories: mtg_layout_soa_padded)" if args.layout == "soa16" else "")
                      + ", coeffs [B][K][D][N]")
        traffic_prof = None
        if world == 1 and not args.no_live_traffic and not args.no_extras:
            traffic_prof = live_traffic(args, B)
        if traffic_prof is None:
            traffic_prof = profile_traffic(traj_per_step, args.config, args.steps, args.warmup)
        if args.settle_ms > 0:
            what += (f"; set-up before the contract's warm-up + timed steps: {args.settle_ms:g} ms of the same work (a fresh process "
                     f"starts on an idle GPU; the timed region measured before that phase is reported as cold_start)"
                     + (f", as {nsets}-batch calls rotating over the warm-up's {settle_within} buffer set(s)" if settle_within else ""))
        out = {
            "metric": {2: "trajectories/sec (8-seg, N=10, 3D min-snap solveLinear)",
                       3: "trajectories/sec (8-seg, N=10, 3D min-snap solveLinear)",
                       4: "trajectories/sec (mixed N = 8/10/12, 4-32 segments, 3D solveLinear, config 4)",
                       5: "trajectories/sec (16-seg, N=10, 4D min-snap solveLinear, config 5)"}[args.config],
            "value": world * traj_per_step * args.steps / dt,
            "unit": "trajectories/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": what, "baseline_config": args.config, "buffer_sets": nsets, "input_layout": args.layout,
                       "kernel_variant": plan.kernel_variant, "launch_form": args.dims, "sequence": args.sequence,
                       "bytes_per_trajectory": None if mixed else plan.bytes_per_trajectory,
                       "trajectories_per_step": traj_per_step},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         # HBM bytes per LAUNCH from the PMC counters (the committed, calibrated profile of this command with
                         # these arguments; null when there is none) and their ratio to the algorithmic bytes
                         "traffic": (traffic_prof["hbm_bytes_per_step"] * args.steps / launches
                                     if traffic_prof and traffic_prof.get("hbm_bytes_per_step") else None),
                         "traffic_over_algorithmic": (traffic_prof["hbm_bytes_per_step"] / bytes_per_step
                                                      if traffic_prof and traffic_prof.get("hbm_bytes_per_step") else None),
                         "traffic_from_profile": traffic_prof,
                         "kernel_us": step_us * args.steps / launches, "launches": launches,
                         "batches_per_launch": batches_per_launch,
                         "bytes_per_launch": bytes_per_step * args.steps / launches,
                         "device_us_per_step": step_us, "bytes_per_step": bytes_per_step,
                         "kernel_us_is": "HIP events recorded on the launch stream immediately before the first and after the "
                                         "last launch of the timed steps / launches (max over ranks)",
                         "output_fill_only_us": fill_us},
            "ranks_seen": ranks_seen, "rank_devices": rank_devices,
            "timed_region_wall_us": wall_breakdown,      # (this rank's host clock; `value` = units / max-over-ranks wall)
            "settle_ms": args.settle_ms, "cold_start": cold,
            # the same warm-up + timed region as the first thing a fresh process does (no settle phase in front of it)
            "value_cold": None if cold is None else world * cold["units_per_s_this_rank"],
            "value_other_form": peer,
            "sustained": sustained,
            "value_aos_inputs": aos_peer,
        }
        if mixed and not per_batch:
            # the merged request of the timed region is built INSIDE it (MixedLoop.run): `value` pays for it
            key = (args.steps, args.warmup % nsets)
            b_us = loop.build_us.get(key)
            out["request_build"] = {"inside_timed_region": True, "host_us": b_us, "mtg_multi_create_us": loop.create_us.get(key),
                                    "share_of_timed_region_wall": (b_us * 1e-6 / dt) if b_us is not None else None,
                                    "is": "every run of the timed region builds its 12 x K-item request: K packed item arrays "
                                          "concatenated + ONE mtg_multi_create (plan handles are looked up once per buffer set; the "
                                          "library caches the longest-processing-time unit schedule per structure and uploads only "
                                          "the item table, asynchronously) -- rounds 3-4 built the request once outside the region "
                                          "(2.45 ms) and reported an amortised `value`; value_other_form is one PRE-BUILT 12-item "
                                          "request per step"}
        if per_rank is not None:
            out["per_rank"] = per_rank
        if args.exercise_collectives:
            out["collectives_exercised"] = {"backend": args.backend, "world": world, "ranks_seen": ranks_seen,
                                            "calls": ["init_process_group", "all_reduce", "all_gather", "barrier"]
                                                     + (["all_gather_into_tensor (chunked solve + gather)"] if gather else [])}
        if extra:
            out["extra"] = extra
        if gather is not None:
            out["gather"] = gather
    if not args.no_cpu_baseline and args.config == 2:
        # rank 0's host cores, after every GPU measurement (the other ranks wait in the barrier below)
        if rank == 0:
            out["cpu_baseline"] = cpu_baseline(200_000, 4321)
            out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
    if rank == 0 and parity is not None:
        # the metric's second half (cpu_baseline leg: the only place that may call oracle/ -- as the checker)
        out["parity"] = parity.check()
        if not out["parity"]["ok"]:
            print("bench.py: PARITY FAILED: " + json.dumps(out["parity"]), file=sys.stderr)
    if rank == 0:
        print(json.dumps(out))
        sys.stdout.flush()
    if grouped:
        dist.barrier()
        if WATCHDOG["hard_exit"]:      # a thread of this process is stuck inside ncclCommInitRank: it cannot be joined
            os._exit(0)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
