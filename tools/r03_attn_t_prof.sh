#!/bin/bash
# kernel durations + counters of the temporal attention kernels at the headline shapes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_attn_t; mkdir -p $OUT
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/attn_bench.py --reps 20 > $OUT/trace.log 2>&1 || { tail -5 $OUT/trace.log; exit 1; }
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc1 -- python3 $R/tools/attn_bench.py --reps 5 > $OUT/pmc1.log 2>&1 || { tail -5 $OUT/pmc1.log; exit 1; }
timeout -k 10 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $OUT/pmc2 -- python3 $R/tools/attn_bench.py --reps 5 > $OUT/pmc2.log 2>&1 || { tail -5 $OUT/pmc2.log; exit 1; }
timeout -k 10 200 rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $OUT/pmc3 -- python3 $R/tools/attn_bench.py --reps 5 > $OUT/pmc3.log 2>&1 || { tail -5 $OUT/pmc3.log; }
python3 - <<PY
import csv, glob, collections
for f in glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "attn" in r["Name"]: print(r["Name"][:80], r["Calls"], "avg us", float(r["AverageNs"]) / 1e3, "min", float(r["MinNs"]) / 1e3, "max", float(r["MaxNs"]) / 1e3)
for f in sorted(glob.glob("$OUT/pmc*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        if "attn_temporal" not in r["Kernel_Name"]: continue
        k = r["Kernel_Name"].split("(")[0][-60:] + " grid " + r["Grid_Size"]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
    for k, v in agg.items():
        print(k, "dispatches", len(cnt[k]), {c: round(x / len(cnt[k])) for c, x in v.items()})
PY
