#!/bin/bash
# Round evidence on the MI355X box: GPU test log, smoke, bench lines, rocprofv3 kernel stats of the
# bench command (pipelined default and sequential), PMC HBM traffic.  Writes gpurun_out/<tag>_*.
tag=${1:-r03}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -rA 2>&1 | tail -170 > gpurun_out/${tag}_pytest_gpu.log)
(timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${tag}_smoke.log 2>&1)
(timeout 400 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err)
(timeout 250 python bench.py --gpus 2 --steps 3 --warmup 3 --no-cpu-baseline 2>/dev/null | grep -a -o '{"metric.*' > gpurun_out/${tag}_bench_2ranks_shared_gpu.json)
(timeout 250 python bench.py --gpus 2 --config xarm --scaling strong --steps 3 --warmup 3 --no-cpu-baseline 2>/dev/null | grep -a -o '{"metric.*' > gpurun_out/${tag}_bench_dp2_xarm_strong_shared_gpu.json)
(timeout 500 python bench.py --gpus 8 --config a1_scaled --scaling strong --steps 2 --warmup 3 --no-cpu-baseline 2>/dev/null | grep -a -o '{"metric.*' > gpurun_out/${tag}_bench_dp8_a1_scaled_strong_shared_gpu.json)
cd /tmp && export TMPDIR=/tmp
for mode in 1 0; do
  rm -rf /tmp/prof$mode
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof$mode -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --pipeline $mode > /tmp/prof$mode.log 2>&1
  DB=$(ls /tmp/prof$mode/*/*.db /tmp/prof$mode/*.db 2>/dev/null | head -1)
  python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $DB 1 > $GRAFT_REPO_ROOT/gpurun_out/${tag}_rocprof_kernel_stats_pipeline$mode.csv 2>> /tmp/prof$mode.log
  grep -a -o '{"metric.*' /tmp/prof$mode.log | head -c 300 >> $GRAFT_REPO_ROOT/gpurun_out/${tag}_rocprof_kernel_stats_pipeline$mode.csv
done
bash $GRAFT_REPO_ROOT/tools/pmc_bench.sh > $GRAFT_REPO_ROOT/gpurun_out/${tag}_pmc_hbm_traffic.csv 2>&1
cd $GRAFT_REPO_ROOT
(timeout 200 python tools/phase_times.py > gpurun_out/${tag}_phase_times.txt 2>&1)
(timeout 200 python tools/scan_time.py > gpurun_out/${tag}_fused_scan_times.txt 2>&1)
(timeout 200 python tools/trace_shapes.py > gpurun_out/${tag}_contraction_call_sites.txt 2>&1)
(timeout 200 python tools/imag_time.py > gpurun_out/${tag}_fused_imagination_times.txt 2>&1)
(for d in 0 2 6 14; do DD_IMG_DBG=$d timeout 100 python tools/conv_image_probe.py 2>&1 | tail -1; done; DD_UP_IMAGE=0 timeout 100 python tools/conv_image_probe.py 2>&1 | tail -1) > gpurun_out/${tag}_image_layer_probe.txt
(timeout 100 python tools/conv_same_time.py > gpurun_out/${tag}_conv_same_times.txt 2>&1)
(timeout 200 python bench.py --cnn resnet --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/${tag}_bench_resnet.json 2>/dev/null)
(timeout 120 python tools/graph_stress.py --iters 45 > gpurun_out/${tag}_graph_stress.log 2>&1)
# the other BASELINE configs at their per-GPU shard (configs[2] 50 / 2 GPUs, [3] 64 / 4, [4] 256 / 8)
(for c in "a1 --batch 16 --length 16" "xarm --batch 25 --length 50" "ur5_multicam --batch 16 --length 64" "a1_scaled --batch 32 --length 64"; do timeout 400 python bench.py --config $c --no-cpu-baseline 2>/dev/null | grep -a -o '{"metric.*'; done > gpurun_out/${tag}_bench_other_configs.jsonl)
