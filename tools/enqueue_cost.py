"""Host cost of ONE mtg_solve_linear_sequence_events call (the bench's timed region is one such call for 20 batches of 10k):
median / quartiles of the call's own duration and of call + wait-for-stop-event over REPS repetitions, each after a device
synchronize.  usage: enqueue_cost.py [PACKAGE_ROOT]   (PACKAGE_ROOT: a tree holding mav_trajectory_generation_amd/, default: this one)"""
import ctypes, json, os, sys, time
root = os.path.abspath(sys.argv[1]) if len(sys.argv) > 1 else os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import torch
import mav_trajectory_generation_amd as m
assert os.path.abspath(m.__file__).startswith(root), m.__file__
N, K, D, d, B, STEPS, SETS, REPS = 10, 8, 3, 4, 10000, 20, 16, 400
ctx = m.Context(0)
masks = m.ends_full_masks(N, K)
plan = m.Plan(ctx, N, D, K, d, masks)
dev = torch.device("cuda", 0)
with torch.cuda.stream(ctx.stream):
    sets = []
    for s in range(SETS):
        t, f = m.random_waypoint_batch(B, K, D, N, masks, seed=s, device=dev, layout="soa")
        sets.append((t, f, torch.zeros((B, K, D, N), dtype=torch.float64, device=dev)))
    lay = plan.layout(B, "soa")
    arr = [(ctypes.c_void_p * STEPS)(*[sets[i % SETS][j].data_ptr() for i in range(STEPS)]) for j in range(3)]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(ctx.stream); e1.record(ctx.stream)
    torch.cuda.synchronize()
    fn = plan.lib.mtg_solve_linear_sequence_events
    res = {}
    for tag, ev in (("with_events", (ctypes.c_void_p(e0.cuda_event), ctypes.c_void_p(e1.cuda_event))), ("no_events", (None, None))):
        call, total, kern = [], [], []
        for r in range(REPS + 50):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            rc = fn(plan.handle, STEPS, B, ctypes.byref(lay), arr[0], arr[1], arr[2], 0, ev[0], ev[1])
            t1 = time.perf_counter()
            if ev[1] is not None:
                while not e1.query():
                    pass
            else:
                torch.cuda.synchronize()
            t2 = time.perf_counter()
            assert rc == 0
            if r >= 50:
                call.append((t1 - t0) * 1e6); total.append((t2 - t0) * 1e6)
                if ev[1] is not None:
                    kern.append(e0.elapsed_time(e1) * 1e3)
        q = lambda v: [round(sorted(v)[int(len(v) * p)], 2) for p in (0.25, 0.5, 0.75)]
        res[tag] = {"call_us_q25_50_75": q(call), "call_plus_wait_us": q(total), "kernel_us": q(kern) if kern else None}
    # the bench's sequence: a 5-batch warm-up call, synchronize, then the timed 20-batch call with events
    arr5 = [(ctypes.c_void_p * 5)(*[sets[i % SETS][j].data_ptr() for i in range(5)]) for j in range(3)]
    ev = (ctypes.c_void_p(e0.cuda_event), ctypes.c_void_p(e1.cuda_event))
    call, total = [], []
    for r in range(60):
        fn(plan.handle, 5, B, ctypes.byref(lay), arr5[0], arr5[1], arr5[2], 0, None, None)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn(plan.handle, STEPS, B, ctypes.byref(lay), arr[0], arr[1], arr[2], 0, ev[0], ev[1])
        t1 = time.perf_counter()
        while not e1.query():
            pass
        t2 = time.perf_counter()
        torch.cuda.synchronize()
        if r >= 10:
            call.append((t1 - t0) * 1e6); total.append((t2 - t0) * 1e6)
    res["after_5_batch_warmup_call"] = {"call_us_q25_50_75": q(call), "call_plus_wait_us": q(total)}
print(json.dumps({"package_root": root, "steps": STEPS, "batch": B, **res}))
