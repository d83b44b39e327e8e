#!/bin/bash
# Round-2 evidence run: bench line (driver's arguments), rocprofv3 kernel stats + calibrated PMC traffic of the bench
# command (rotating and resident buffers), per-config kernel times, microbenchmarks.  Outputs under gpurun_out/r02/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r02; mkdir -p $OUT
cd $R
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_args.json 2> $OUT/bench_driver_args.err
python bench.py --steps 2000 --warmup 50 --no-cpu-baseline --extra > $OUT/bench_2000_steps.json 2> $OUT/bench_2000.err
python bench.py --config 3 --steps 100 --warmup 10 --no-cpu-baseline > $OUT/bench_config3.json 2>> $OUT/bench_2000.err
python bench.py --config 5 --steps 200 --warmup 10 --no-cpu-baseline > $OUT/bench_config5.json 2>> $OUT/bench_2000.err
python bench.py --gpus 2 --backend gloo --same-device --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | grep "^{" > $OUT/bench_two_ranks_one_gpu.json
bash tools/gpu_profile2.sh r02_rotating
bash tools/gpu_profile2.sh r02_resident --buffer-sets 1
cp $R/gpurun_out/r02_rotating_b10000_kernel_stats.csv $R/gpurun_out/r02_rotating_b10000_pmc_traffic.json $R/gpurun_out/r02_resident_b10000_kernel_stats.csv $R/gpurun_out/r02_resident_b10000_pmc_traffic.json $OUT/ 2>/dev/null
python tools/bench_configs.py 2>&1 | grep "^{" > $OUT/configs.jsonl
python tools/bench_configs.py long 2>&1 | grep "^{" >> $OUT/configs.jsonl
(for n in 10 8 12; do python tools/bench_other_k.py $n 2>&1 | grep "^{"; done) > $OUT/other_chain_lengths.jsonl
python tools/bench_mixed.py 2500 merged 2>&1 | grep "^{" > $OUT/mixed_config4.jsonl
python tools/bench_mixed.py 10000 merged 2>&1 | grep "^{" >> $OUT/mixed_config4.jsonl
tools/micro/stream_overlap.bin > $OUT/stream_overlap_microbench.txt 2>&1
tools/micro/launch_geometry > $OUT/launch_geometry_microbench.txt 2>&1
tools/lab/b10k_lab 10000 1 > $OUT/lab_resident.txt 2>&1
tools/lab/b10k_lab 10000 16 > $OUT/lab_rotating.txt 2>&1
tools/lab/b10k_lab_t 10000 1 | tail -12 > $OUT/lab_phase_timing.txt 2>&1
(for v in "" "LAB_FIX_IN=1" "LAB_FIX_OUT=1" "LAB_FIX_IN=1 LAB_FIX_OUT=1"; do echo "== 16 buffer sets $v"; env $v tools/lab/b10k_lab 10000 16 | grep "dim-in-lane, nt sc1 stores  \|dim-in-lane, sc1 stores  " | grep -v "max poly"; done) > $OUT/lab_rotation_sides.txt 2>&1
(for a in "125000 1" "125000 4" "1000000 1"; do echo "== B, buffer sets: $a"; tools/lab/b10k_lab $a | grep "fused\|slab"; done) > $OUT/lab_large_batch.txt 2>&1
python tools/enqueue_probe.py 16 > $OUT/enqueue_probe.txt 2>&1
tools/cpp/polynomial_timing_evaluation > $OUT/veneer_timing_evaluation.txt 2>&1
bash tools/sweep_forms.sh > $OUT/sweep_forms.txt 2>&1
ls $OUT
