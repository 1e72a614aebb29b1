#!/bin/bash
# Round-4 GPU session 1: the role-separated loop in the product library.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
(DD_WS_KMIN=256 DD_WS_KMIN_TC=256 timeout 400 python -m pytest tests/test_hip_ops.py -q -x -k "gemm or conv" 2>&1 | tail -5) > gpurun_out/r04_s1_pytest_ws_k256.log
(timeout 300 python -m pytest tests/test_graph_lifecycle_gpu.py -q -x 2>&1 | tail -8) > gpurun_out/r04_s1_pytest_lifecycle.log
(timeout 500 python tools/r04/ws_ab.py) > gpurun_out/r04_ws_selector.txt 2>&1
(timeout 400 python bench.py --no-cpu-baseline > gpurun_out/r04_s1_bench.json 2> gpurun_out/r04_s1_bench.err)
(bash tools/pmc_gemm2.sh 4096 4096 4096) > gpurun_out/r04_s1_pmc_gemm_4096.txt 2>&1
tail -3 gpurun_out/r04_s1_pytest_ws_k256.log gpurun_out/r04_s1_pytest_lifecycle.log
cat gpurun_out/r04_ws_selector.txt | head -70
cat gpurun_out/r04_s1_bench.json | head -c 3000
tail -5 gpurun_out/r04_s1_bench.err
cat gpurun_out/r04_s1_pmc_gemm_4096.txt | tail -22
