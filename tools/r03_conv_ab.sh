#!/bin/bash
# same-box A/B of the conv kernel: tools/_timing/base.so (previous commit) vs the tree's library
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-r03e}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_engine.py -m gpu -x -q -k "conv or wino or golden or oracle or batch" > $O/tests_conv.log 2>&1; rc=$?; tail -4 $O/tests_conv.log; [ $rc -ne 0 ] && exit $rc
for v in base "" base ""; do
  echo "== variant ${v:-product}" >> $O/ab.log
  if [ -z "$v" ]; then timeout -k 10 200 python tools/s64_bench.py --kernel r64 --reps 8 >> $O/ab.log 2>&1
  else VD_LIB=tools/_timing/$v.so timeout -k 10 200 python tools/s64_bench.py --kernel r64 --reps 8 >> $O/ab.log 2>&1; fi
done
grep -v amdgpu.ids $O/ab.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fp32-ref --no-dropin > $O/bench.json 2> $O/bench.err; tail -c 700 $O/bench.json
VD_LIB=tools/_timing/base.so timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fp32-ref --no-dropin > $O/bench_base.json 2> $O/bench_base.err; tail -c 700 $O/bench_base.json
echo ALL_OK
