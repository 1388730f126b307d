#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03r; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_engine.py -m gpu -x -q -k "wino or conv or golden or oracle or batch or windows or full_size" 2>&1 | tail -3
for v in 0 1 0 1 0 1; do
  echo "== VD_R64_2PASS=$v" >> $O/ab.log
  VD_R64_2PASS=$v timeout -k 10 200 python tools/s64_bench.py --kernel r64 --reps 8 >> $O/ab.log 2>&1
done
grep -v amdgpu $O/ab.log | grep "2PASS\|class total\| 64 \| 32 "
for v in 0 1 0 1; do
  VD_R64_2PASS=$v timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fp32-ref --no-dropin --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('2PASS=$v', d['value'], d['ms_per_step'])"
done | tee -a $O/ab.log
