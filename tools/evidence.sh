#!/bin/bash
# Round evidence on the MI355X box (run through gpurun): GPU test log, smoke, the bench line, rocprofv3
# kernel stats of the train loop (default schedule and pipeline), PMC HBM traffic, phase / scan /
# imagination / call-site timings, the other BASELINE config shards.  Writes gpurun_out/<tag>_*;
# copy what is to be judged into profiles/.
#   gpurun --timeout 2400 -- 'bash tools/evidence.sh r06'
tag=${1:-r06}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
(timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -rA 2>&1 | tail -260 > gpurun_out/${tag}_pytest_gpu.log)
(timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${tag}_smoke.log 2>&1)
(timeout 600 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err)
# multi-rank plumbing on the one GPU (ranks share the device, gloo): not a measurement
(timeout 250 python bench.py --gpus 2 --steps 3 --warmup 3 --no-cpu-baseline --pmc off 2>/dev/null | grep -a -o '{"metric.*' > gpurun_out/${tag}_bench_2ranks_shared_gpu.json)
(timeout 250 python bench.py --gpus 2 --config xarm --batch 50 --length 50 --scaling strong --steps 3 --warmup 3 --no-cpu-baseline --pmc off 2>/dev/null | grep -a -o '{"metric.*' > gpurun_out/${tag}_bench_dp2_xarm_strong_shared_gpu.json)
(timeout 500 python bench.py --gpus 8 --config a1_scaled --batch 256 --length 64 --scaling strong --steps 2 --warmup 3 --no-cpu-baseline --pmc off 2>/dev/null | grep -a -o '{"metric.*' > gpurun_out/${tag}_bench_dp8_a1_scaled_strong_shared_gpu.json)
cd /tmp && export TMPDIR=/tmp
K=10; W=3
for mode in 0 1; do
  rm -rf /tmp/prof$mode
  DD_PIPE_TUNE=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof$mode -o b -- python $GRAFT_REPO_ROOT/bench.py --child --steps $K --warmup $W --no-cpu-baseline --pmc off --pipeline $mode > /tmp/prof$mode.log 2>&1
  DB=$(ls /tmp/prof$mode/*/*.db /tmp/prof$mode/*.db 2>/dev/null | head -1)
  # (the child run makes max(W, 3) + K train calls and nothing else)
  python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $DB $((K + W)) $W > $GRAFT_REPO_ROOT/gpurun_out/${tag}_rocprof_kernel_stats_pipeline$mode.csv 2>> /tmp/prof$mode.log
  python $GRAFT_REPO_ROOT/tools/timeline.py $DB $((K + W)) $W 2>/dev/null | grep -v columns | head -40 > $GRAFT_REPO_ROOT/gpurun_out/${tag}_timeline_pipeline$mode.txt
done
cd $GRAFT_REPO_ROOT
bash tools/pmc_bench.sh > gpurun_out/${tag}_pmc_hbm_traffic.txt 2>&1
(timeout 200 python tools/phase_times.py > gpurun_out/${tag}_phase_times.txt 2>&1)
(timeout 200 python tools/scan_time.py > gpurun_out/${tag}_fused_scan_times.txt 2>&1)
(timeout 200 python tools/trace_shapes.py > gpurun_out/${tag}_contraction_call_sites.txt 2>&1)
(timeout 200 python tools/ln_bench.py > gpurun_out/${tag}_ln_bandwidth.txt 2>&1)
(timeout 200 python tools/conv_wgrad_probe.py 2>&1 | grep -v amdgpu > gpurun_out/${tag}_image_layer_probe.txt)
(timeout 200 python tools/phase_timeline.py 2>&1 | grep -v amdgpu > gpurun_out/${tag}_phase_timeline_pipelined.txt)
(for r in 16 32; do echo "rows per workgroup: $r"; DD_IMAG_ROWS=$r timeout 200 python tools/imag_time.py 2>&1 | grep -v amdgpu; done > gpurun_out/${tag}_fused_imagination_times.txt)
(timeout 200 python bench.py --cnn resnet --steps 4 --warmup 2 --no-cpu-baseline --pmc off > gpurun_out/${tag}_bench_resnet.json 2>/dev/null)
(timeout 120 python tools/graph_stress.py --iters 45 > gpurun_out/${tag}_graph_stress.log 2>&1)
(timeout 300 python tools/dp_preflight.py --gpus 1 2>&1 | grep -a preflight > gpurun_out/${tag}_dp_preflight_1rank_rccl.log)
# the other BASELINE configs at their per-GPU shard (configs[0] 16, [2] 50 / 2 GPUs, [3] 64 / 4, [4] 256 / 8)
(for c in "a1 --batch 16 --length 16 --horizon 5" "xarm --batch 25 --length 50" "ur5_multicam --batch 16 --length 64" "a1_scaled --batch 32 --length 64"; do timeout 400 python bench.py --config $c --no-cpu-baseline --pmc off 2>/dev/null | grep -a -o '{"metric.*'; done > gpurun_out/${tag}_bench_other_configs.jsonl)
tail -3 gpurun_out/${tag}_pytest_gpu.log; cat gpurun_out/${tag}_smoke.log | tail -2; head -c 600 gpurun_out/${tag}_bench.json
