#!/bin/bash
# rocprofv3 evidence for bench.py (run on the GPU box via gpurun):  tools/profile_bench.sh <tag>
#   1) --kernel-trace --stats of the bench command          -> gpurun_out/prof_<tag>/ (kernel_stats.csv)
#   2) PMC passes (separate runs, counters only + kernel trace): FETCH_SIZE, WRITE_SIZE, MFMA busy / cycles
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=$1; OUT=$R/gpurun_out/prof_$TAG; mkdir -p $OUT
CMD="python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-roofline --no-fp32-ref --no-dropin --no-full-window"
# the same command un-profiled, with the live roofline: algorithmic FLOPs per launch for tools/make_pmc_dominant.py's MFMA check
timeout -k 10 300 python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-fp32-ref --no-dropin --no-full-window > $OUT/bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/trace.log 2>&1 || { tail -5 $OUT/trace.log; exit 1; }
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc$i -- $CMD > $OUT/pmc$i.log 2>&1 || { echo "pmc pass $i failed"; tail -5 $OUT/pmc$i.log; exit 1; }
done
python3 - <<PY
import csv, glob, collections, json
out = {}
for f in glob.glob("$OUT/pmc*/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:70]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k].add(r["Dispatch_Id"])
    for k, v in agg.items():
        out.setdefault(k, {"dispatches": len(cnt[k])}).update({c: x for c, x in v.items()})
stats = {}
for f in glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        stats[r["Name"].split("(")[0][:70]] = dict(calls=int(r["Calls"]), total_ns=int(r["TotalDurationNs"]), avg_ns=float(r["AverageNs"]), pct=float(r["Percentage"]))
json.dump(dict(pmc=out, stats=stats), open("$OUT/summary.json", "w"), indent=1)
top = sorted(stats.items(), key=lambda kv: -kv[1]["total_ns"])[:6]
for k, v in top: print(k, v, {c: round(x) for c, x in out.get(k, {}).items()})
PY
