#!/bin/bash
# average duration of kernels matching a pattern over a short bench run:  tools/kernel_avg.sh <pattern> [bench flags]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; PAT=$1; shift; OUT=$R/gpurun_out/kavg; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-roofline --no-fp32-ref --no-dropin "$@" > $OUT/trace.log 2>&1 || { tail -5 $OUT/trace.log; exit 1; }
python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    for r in rows:
        if "$PAT" in r["Name"]: print(r["Name"][:70], r["Calls"], "avg us %.2f" % (float(r["AverageNs"]) / 1e3), "total ms %.3f" % (float(r["TotalDurationNs"]) / 1e6))
    print("all kernels, ms per step: %.3f" % (tot / 5 / 1e6))
PY
