#!/bin/bash
# PMC passes for the conv kernel (separate passes; never combined with sys/hip tracing).
# usage: tools/pmc_conv.sh "<bench_conv --only filter>" <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; F="$1"; TAG="$2"; OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
i=0
for C in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
         "FETCH_SIZE" "WRITE_SIZE" \
         "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/p$i -- python3 $R/tools/bench_conv.py --reps 2 --only "$F" > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/p$i.log; exit 1; }
done
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$OUT/p*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        for k, v in agg.items():
            if "conv3x3" in k or "igemm" in k:
                print(k, {c: round(x) for c, x in v.items()})
PY
