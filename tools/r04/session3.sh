#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
(timeout 300 python -m pytest tests/test_hip_ops.py -q -x -k "ln_act or role_separated" 2>&1 | tail -4) > gpurun_out/r04_s3_pytest.log
(timeout 200 python tools/ln_bench.py) > gpurun_out/r04_ln_bandwidth.txt 2>&1
(timeout 300 python bench.py --no-cpu-baseline --pmc off > gpurun_out/r04_s3_bench.json 2> gpurun_out/r04_s3_bench.err)
cat gpurun_out/r04_s3_pytest.log gpurun_out/r04_ln_bandwidth.txt
python - <<'PY'
import json
d = json.load(open('gpurun_out/r04_s3_bench.json'))
print({k: d[k] for k in ('ms_per_step', 'resident', 'pipelined')}, d['roofline']['achieved'], d['roofline']['kernel_time_ms'])
PY
tail -3 gpurun_out/r04_s3_bench.err
