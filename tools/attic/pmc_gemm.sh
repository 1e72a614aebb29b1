#!/bin/bash
# PMC counters for one GEMM shape: tools/pmc_gemm.sh M N K
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc1
PYTHONPATH=$GRAFT_REPO_ROOT timeout 120 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --kernel-trace -d /tmp/pmc1 -o g --output-format csv -- python $GRAFT_REPO_ROOT/tools/gemm_one.py $1 $2 $3 > /tmp/pmc.log 2>&1
ls -R /tmp/pmc1 | head
python - <<PY
import csv, glob, collections
for f in glob.glob("/tmp/pmc1/**/*counter_collection*.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    print(rows[0].keys())
    agg = collections.defaultdict(list)
    for r in rows:
        if "mfma_gemm" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items(): print(k, sum(v)/len(v), len(v))
for f in glob.glob("/tmp/pmc1/**/*kernel_trace*.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    d = [ (int(r["End_Timestamp"])-int(r["Start_Timestamp"])) for r in rows if "mfma_gemm" in r["Kernel_Name"]]
    print("durations ns", d)
PY
grep -v "^W2026\|^I2026" /tmp/pmc.log | tail -5 | cut -c1-300
