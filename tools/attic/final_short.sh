#!/bin/bash
# The part of tools/final_evidence.sh that depends on the whole tree: PMC HBM traffic record (stamped
# with the kernel-source hash), bench line, smoke, GPU test log.  Writes gpurun_out/<tag>_*.
# usage: final_short.sh <tag> [pytest selection...]   (default selection: the whole GPU suite)
tag=${1:-r03}; shift
sel=${@:-tests}
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
bash $GRAFT_REPO_ROOT/tools/pmc_bench.sh > $GRAFT_REPO_ROOT/gpurun_out/${tag}_pmc_hbm_traffic.csv 2>&1
cd $GRAFT_REPO_ROOT
mkdir -p profiles && cp gpurun_out/pmc_hbm_traffic.json profiles/pmc_hbm_traffic.json
(timeout 400 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err)
(timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${tag}_smoke.log 2>&1)
(timeout 1500 python -m pytest $sel -m gpu -q --timeout 900 -rA 2>&1 | tail -220 > gpurun_out/${tag}_pytest_gpu.log)
tail -3 gpurun_out/${tag}_pytest_gpu.log; tail -2 gpurun_out/${tag}_smoke.log; head -c 300 gpurun_out/${tag}_bench.json
