#!/bin/bash
# HBM traffic of the contraction kernels during bench.py (separate --pmc passes, as
# MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE do not fit one pass).
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  DD_PIPE_TUNE=0 PYTHONPATH=$GRAFT_REPO_ROOT timeout 300 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_$c -o b --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 3 --no-cpu-baseline > /tmp/pmc_$c.log 2>&1
done
python - <<PY
import csv, glob, collections, re
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"/tmp/pmc_{c}/**/*counter_collection*.csv", recursive=True)
    if not f: print("no output for", c); continue
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] != c: continue
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        key = "k_mfma_gemm" if "k_mfma_gemm" in name else ("k_splitk_reduce" if "splitk" in name else ("k_col2im" if "col2im" in name else "other"))
        agg[key][0] += 1; agg[key][1] += float(r["Counter_Value"])
    res[c] = agg
print("counter,kernel_class,dispatches,sum_KiB,avg_KiB_per_launch")
for c, agg in res.items():
    for k, (n, v) in agg.items():
        print(f"{c},{k},{n},{v:.1f},{v/n:.2f}")
# the record bench.py reads (profiles/pmc_hbm_traffic.json): per launch of the contraction kernels,
# FETCH_SIZE doubled (gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md section HBM), stamped
# with the hash of the kernel sources it was measured on (bench.py ignores a stale record)
import json, os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import bench
if "FETCH_SIZE" in res and "WRITE_SIZE" in res and res["FETCH_SIZE"]["k_mfma_gemm"][0]:
    nf, vf = res["FETCH_SIZE"]["k_mfma_gemm"]; nw, vw = res["WRITE_SIZE"]["k_mfma_gemm"]
    rec = dict(source="rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/pmc_bench.sh) around "
               "python bench.py --steps 3 --warmup 3 --no-cpu-baseline; FETCH_SIZE doubled (gfx950 counts 128-B "
               "requests at 64 B); counters are L2-miss side (Infinity-Cache hits included)",
               kernel="k_mfma_gemm_s3<*>", dispatches=nf, fetch_kib_per_launch_reported=round(vf / nf, 2),
               write_kib_per_launch=round(vw / nw, 2),
               bytes_per_launch=int(1024 * (2 * vf / nf + vw / nw)), kernel_sources_sha=bench.kernel_sources_sha())
    out = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "pmc_hbm_traffic.json")
    json.dump(rec, open(out, "w"), indent=1)
    print("wrote", out)
PY
