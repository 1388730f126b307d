#!/bin/bash
# PMC comparison of conv variants: tools/pmc_probe.sh   (prints per-kernel counters + durations)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmc_probe; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT/p1 -- python3 $R/tools/probe_conv.py > $OUT/p1.log 2>&1 || { tail -5 $OUT/p1.log; exit 1; }
python3 - <<PY
import csv, glob, collections
disp = {}
for f in glob.glob("$OUT/p1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        disp[r["Dispatch_Id"]] = (r["Kernel_Name"][:55], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
rows = collections.defaultdict(dict)
for f in glob.glob("$OUT/p1/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
# probe order: for each shape: (pro,res) = (1,1),(1,0),(0,1),(0,0), each 1 warm + 5 reps
ids = sorted(rows, key=int)
for k in range(0, len(ids), 6):
    grp = ids[k+1:k+6]
    if not grp: break
    name = disp[grp[0]][0]
    ns = sum(disp[i][1] for i in grp) / len(grp)
    c = {n: sum(rows[i][n] for i in grp) / len(grp) for n in rows[grp[0]]}
    clk = c["GRBM_GUI_ACTIVE"] / 8 / ns
    print(f"{name[14:50]} {ns/1e3:8.1f} us clk {clk:.2f} GHz mfma_busy {c['SQ_VALU_MFMA_BUSY_CYCLES']/(c['GRBM_GUI_ACTIVE']/8*1024):.3f} wait_any {c['SQ_WAIT_ANY']/c['SQ_WAVE_CYCLES']:.3f} wait_inst {c['SQ_WAIT_INST_ANY']/c['SQ_WAVE_CYCLES']:.3f} active {c['SQ_ACTIVE_INST_ANY']/c['SQ_WAVE_CYCLES']:.3f} valu_insts {c['SQ_INSTS_VALU']/1e6:.0f}M wavecyc {c['SQ_WAVE_CYCLES']*4/1e9:.2f}G")
PY
