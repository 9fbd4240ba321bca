#!/bin/bash
# End-of-round evidence on one MI355X -> gpurun_out/final_*  (python tools/collect_final_r04.py copies it into profiles/r04_*).
#   tools/final_profiles_r04.sh            (~3 min of GPU time)
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$root"
python bench.py > gpurun_out/final_bench_default.json 2> gpurun_out/final_bench_default.err
# kernel tables: every launch alone on the device (the parallel branches of the step switched off -- rocprofv3 serialises
# overlapping launches anyway, and a launch that shares the chip is not a measurement of the kernel)
WM_VIEW_BRANCHES=0 WM_DINO_TEACHER_STREAM=0 WM_PROFILE_MARKER=sgd_step bash tools/profile_bench.sh final_trace_simclr_r18 10 14 --no-kernel-timer
WM_VIEW_BRANCHES=0 WM_DINO_TEACHER_STREAM=0 WM_PROFILE_MARKER=adamw_kernel bash tools/profile_bench.sh final_trace_dino_vit_tiny 10 14 --workload dino_vit_tiny --no-kernel-timer
WM_PROFILE_MARKER=adamw_kernel bash tools/profile_bench.sh final_trace_mae_vit_small_16 10 14 --workload mae_vit_small_16 --no-kernel-timer
# the same two steps as the timed region runs them (branches on), for the record of what the profiler does to them
WM_PROFILE_MARKER=sgd_step bash tools/profile_bench.sh final_trace_simclr_r18_branches 10 14 --no-kernel-timer
WM_PROFILE_MARKER=adamw_kernel bash tools/profile_bench.sh final_trace_dino_vit_tiny_branches 10 14 --workload dino_vit_tiny --no-kernel-timer
for w in dino_vit_tiny dino_vit_small mae_vit_small_16 mae_vit_b_32; do
  python bench.py --workload $w --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/final_bench_$w.json 2>/dev/null
done
python bench.py --workload knn_allpairs --batch 64 --steps 200 --warmup 20 > gpurun_out/final_bench_knn_allpairs_b64.json 2>/dev/null
python bench.py --workload knn_allpairs --steps 50 --warmup 5 > gpurun_out/final_bench_knn_allpairs.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp && cd "$root"
rm -rf gpurun_out/final_trace_knn gpurun_out/final_trace_knn_pipe
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final_trace_knn -o p -- python3 tools/knn_one.py 64 bf16 > /dev/null 2>&1
python3 tools/summarize_profile.py gpurun_out/final_trace_knn gpurun_out/final_trace_knn_b64.md 5 "python tools/knn_one.py 64 bf16 (5 calls of wm_knn_topk: 64 bf16 queries x 811 457 x 128 bank, k = 8) under rocprofv3 --kernel-trace --stats; per 'step' = per call"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final_trace_knn_pipe -o p -- python3 tools/knn_pipelined_trace.py > /dev/null 2>&1
python3 tools/summarize_knn_trace.py gpurun_out/final_trace_knn_pipe gpurun_out/final_trace_knn_pipelined.md > /dev/null
rm -rf gpurun_out/final_trace_knn gpurun_out/final_trace_knn_pipe
# HBM traffic (rocprofv3 --pmc, separate FETCH_SIZE / WRITE_SIZE passes, eager steps; the launches of the roofline brackets:
# both views in one stream)
WM_VIEW_BRANCHES=0 bash tools/pmc_bench.sh final_hbm_simclr_r18 sgd_step "python bench.py --no-graph --steps 3 --warmup 1 (SimCLR ResNet-18, bs 256), eager steps" --steps 3 --warmup 1
WM_DINO_TEACHER_STREAM=0 bash tools/pmc_bench.sh final_hbm_dino_vit_tiny adamw_kernel "python bench.py --workload dino_vit_tiny --no-graph --steps 3 --warmup 1, eager steps" --workload dino_vit_tiny --steps 3 --warmup 1
bash tools/pmc_bench.sh final_hbm_mae_vit_small_16 adamw_kernel "python bench.py --workload mae_vit_small_16 --no-graph --steps 3 --warmup 1, eager steps" --workload mae_vit_small_16 --steps 3 --warmup 1
bash tools/pmc_knn.sh final_hbm_knn_b64 "python tools/knn_one.py 64 bf16 (5 calls of wm_knn_topk, 64 bf16 queries x 811 457 x 128)" 64 bf16
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final_smoke.txt 2>&1
ls -la gpurun_out/final_*
