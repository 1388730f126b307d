#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03k; mkdir -p $O; cd $R
for v in "" spread spread6w spread2 spread2w wonly "" spread spread6w spread2 spread2w wonly; do
  echo "== variant ${v:-product}" >> $O/ab.log
  if [ -z "$v" ]; then timeout -k 10 200 python tools/s64_bench.py --kernel r64 --quick --reps 10 >> $O/ab.log 2>&1
  else VD_LIB=tools/_timing/$v.so timeout -k 10 200 python tools/s64_bench.py --kernel r64 --quick --reps 10 >> $O/ab.log 2>&1; fi
done
grep -v amdgpu.ids $O/ab.log
VD_LIB=tools/_timing/spread2w.so timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "wino or conv" 2>&1 | tail -3
