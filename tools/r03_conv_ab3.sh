#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03l; mkdir -p $O; cd $R
VD_LIB=tools/_timing/sw.so timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_engine.py -m gpu -x -q -k "wino or conv or golden or oracle or batch or windows" 2>&1 | tail -3
for v in "" s1 sw "" s1 sw "" s1 sw; do
  echo "== variant ${v:-product}" >> $O/ab.log
  if [ -z "$v" ]; then timeout -k 10 200 python tools/s64_bench.py --kernel r64 --reps 8 >> $O/ab.log 2>&1
  else VD_LIB=tools/_timing/$v.so timeout -k 10 200 python tools/s64_bench.py --kernel r64 --reps 8 >> $O/ab.log 2>&1; fi
done
grep "variant\|class total" $O/ab.log
for v in "" s1 sw "" s1 sw; do
  if [ -z "$v" ]; then timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fp32-ref --no-dropin --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('product', d['value'], d['ms_per_step'])"
  else VD_LIB=tools/_timing/$v.so timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fp32-ref --no-dropin --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'])"; fi
done | tee -a $O/ab.log
