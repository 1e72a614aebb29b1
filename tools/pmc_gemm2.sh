#!/bin/bash
# PMC counters of the contraction loop for one GEMM shape, two passes (8 SQ counters each):
#   tools/pmc_gemm2.sh M N K [ta tb]
cd /tmp && export TMPDIR=/tmp
P1="SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS"
P2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_MISC"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1)); rm -rf /tmp/pmc$i
  PYTHONPATH=$GRAFT_REPO_ROOT timeout 120 rocprofv3 --pmc $P --kernel-trace -d /tmp/pmc$i -o g --output-format csv -- python $GRAFT_REPO_ROOT/tools/gemm_one.py "$@" > /tmp/pmc$i.log 2>&1
done
python - <<PY
import csv, glob, collections
for i in (1, 2):
  for f in glob.glob(f"/tmp/pmc{i}/**/*counter_collection*.csv", recursive=True):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "mfma_gemm" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items(): print(f"{k:28s} {sum(v)/len(v):16.0f}  ({len(v)} launches)")
for f in glob.glob("/tmp/pmc1/**/*kernel_trace*.csv", recursive=True):
    d = [(int(r["End_Timestamp"])-int(r["Start_Timestamp"])) for r in csv.DictReader(open(f)) if "mfma_gemm" in r["Kernel_Name"]]
    print("durations ns", d[:6])
PY
grep -v "^W2026\|^I2026" /tmp/pmc2.log | tail -3 | cut -c1-200
