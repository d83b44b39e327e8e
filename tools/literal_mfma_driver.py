"""The MFMA evidence variant through the library (include/mtg_hip_lab.h: mtg_lab_segment_cost_matrices): literal contraction on the
FP64 matrix cores against the scaling identity, N = 8 / 10 / 12, 2^20 segments; time per launch and the implied MFMA rate.
python tools/literal_mfma_driver.py [n_segments] [reps]   (rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES ... wraps it)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mav_trajectory_generation_amd as m

nseg = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ctx = m.Context(0)
with torch.cuda.stream(ctx.stream):
    t = 1.0 + 19.0 * torch.rand(nseg, dtype=torch.float64, device="cuda")
    for n in (8, 10, 12):
        d = n // 2 - 1
        row = {"N": n, "derivative": d, "segments": nseg}
        outs = {}
        for variant, name in ((1, "literal_mfma"), (0, "scaling_identity")):
            outs[variant] = ctx.lab_segment_cost_matrices(n, d, t, variant)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(ctx.stream)
            for _ in range(reps):
                ctx.lab_segment_cost_matrices(n, d, t, variant)
            e1.record(ctx.stream)
            torch.cuda.synchronize()
            row[name + "_us"] = round(e0.elapsed_time(e1) * 1e3 / reps, 1)
        kp = (n + 3) // 4 * 4
        mfma_flops = 2 * (kp // 4) * 2 * 16 * 16 * 4          # two products of kp / 4 tiles of 16 x 16 x 4
        row["mfma_issue_tflops"] = round(mfma_flops * nseg / (row["literal_mfma_us"] * 1e-6) * 1e-12, 2)
        row["share_of_78p6_tf_fp64_mfma_peak"] = round(row["mfma_issue_tflops"] / 78.6, 4)
        row["useful_tflops_2x2xN3"] = round(4.0 * n ** 3 * nseg / (row["literal_mfma_us"] * 1e-6) * 1e-12, 2)
        row["literal_over_identity"] = round(row["literal_mfma_us"] / row["scaling_identity_us"], 2)
        scale = outs[0].abs().amax(dim=(1, 2)).clamp_min(1e-300)
        row["max_rel_diff_literal_vs_identity"] = float(((outs[1] - outs[0]).abs().amax(dim=(1, 2)) / scale).max())
        print(json.dumps(row), flush=True)
ctx.sync()
