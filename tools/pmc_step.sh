#!/bin/bash
# Matrix-pipe utilisation and shader clock of every kernel class of the real train step:
# one rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES, GRBM_GUI_ACTIVE) + kernel trace
# over a short child run of bench.py (default schedule).  Per class: launches, time, the effective
# shader clock and MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x active cycles)
# (MI355X_MICROARCH.md: the counter counts cycles, 32 per v_mfma_f32_32x32x16_bf16).  Active cycles =
# SQ_BUSY_CYCLES / 32 shader engines (the normalisation of profiles/r01_pmc_mfma_lds.txt; exact for
# kernels that keep every shader engine busy for their whole duration, i.e. the GPU-filling ones;
# GRBM_GUI_ACTIVE / 8 XCDs is printed beside it - its window is wider than a short kernel).
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_step
DD_PIPE_TUNE=0 PYTHONPATH=$GRAFT_REPO_ROOT timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d /tmp/pmc_step -o s --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --child --steps 3 --warmup 3 --no-cpu-baseline --pmc off > /tmp/pmc_step.log 2>&1
python - <<'PY'
import csv, glob, collections, re
cc = glob.glob("/tmp/pmc_step/**/*counter_collection*.csv", recursive=True)
kt = glob.glob("/tmp/pmc_step/**/*kernel_trace*.csv", recursive=True)
if not cc or not kt:
    raise SystemExit("no rocprofv3 output: " + open("/tmp/pmc_step.log").read()[-600:])
dur = {}
for r in csv.DictReader(open(kt[0])):
    dur[r.get("Dispatch_Id") or r.get("Dispatch_ID")] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
def cls(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    for key in ("ConvWgrad", "ConvUp", "ConvDown", "ConvSame"):
        if key in name: return "k_mfma_gemm_s3<" + key + ">"
    m = re.match(r"(?:void )?(k_\w+)", name)
    return m.group(1) if m else name[:40]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
seen = collections.defaultdict(set)
for r in csv.DictReader(open(cc[0])):
    c = cls(r["Kernel_Name"]); d = r.get("Dispatch_Id") or r.get("Dispatch_ID")
    agg[c][r["Counter_Name"]] += float(r["Counter_Value"])
    if d not in seen[c]:
        seen[c].add(d); agg[c]["ns"] += dur.get(d, 0); agg[c]["n"] += 1
print("kernel_class,launches,total_ms,avg_us,shader_clock_GHz(SQ_BUSY/32),mfma_busy(SQ_BUSY/32),clock_GHz(GRBM/8)")
for c, v in sorted(agg.items(), key=lambda kv: -kv[1]["ns"])[:24]:
    ns = max(v["ns"], 1.0)
    cyc = v["SQ_BUSY_CYCLES"] / 32.0
    clock = cyc / ns                                        # cycles per ns = GHz
    busy = v["SQ_VALU_MFMA_BUSY_CYCLES"] / max(1024.0 * cyc, 1.0)
    print(f"{c},{int(v['n'])},{ns / 1e6:.3f},{ns / 1e3 / max(v['n'], 1):.1f},{clock:.2f},{busy:.3f},{v['GRBM_GUI_ACTIVE'] / 8.0 / ns:.2f}")
PY
