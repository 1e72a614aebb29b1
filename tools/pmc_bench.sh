#!/bin/bash
# HBM traffic of the contraction kernels during the train step (separate --pmc passes, as
# MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE do not fit one pass), written as the
# record bench.py falls back to when rocprofv3 is not on the box (gpurun_out/pmc_hbm_traffic.json ->
# commit it as profiles/pmc_hbm_traffic.json).  bench.py measures the same live by default.
cd $GRAFT_REPO_ROOT && python - <<'PY'
import json, os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import bench
rec = bench.pmc_live([])
if rec is None:
    sys.exit("PMC passes failed")
rec["source"] = rec["source"].replace("measured in this invocation", "tools/pmc_bench.sh")
out = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "pmc_hbm_traffic.json")
json.dump(rec, open(out, "w"), indent=1)
print(json.dumps(rec, indent=1))
print("wrote", out)
PY
