#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03m; mkdir -p $O; cd $R
VD_LIB=tools/_timing/rs.so timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "wino or conv" 2>&1 | tail -2
for v in "" rs "" rs "" rs; do
  echo "== variant ${v:-product}" >> $O/ab.log
  if [ -z "$v" ]; then timeout -k 10 200 python tools/s64_bench.py --kernel r64 --reps 8 >> $O/ab.log 2>&1
  else VD_LIB=tools/_timing/$v.so timeout -k 10 200 python tools/s64_bench.py --kernel r64 --reps 8 >> $O/ab.log 2>&1; fi
done
grep "variant\|class total" $O/ab.log
