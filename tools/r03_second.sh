#!/bin/bash
# round-3 box 2: parity tests with split-K / zero-early / PF3, B=1 and headline benches, conv ablations (VD_R64_SKIP builds)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-r03b}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; rc=$?; tail -5 $O/tests.log; [ $rc -ne 0 ] && exit $rc
VD_PROF_DUMP=1 timeout -k 10 300 python bench.py --batch 1 --frames 16 --steps 40 --warmup 5 --no-cpu-baseline --no-fp32-ref > $O/bench_b1.json 2> $O/bench_b1.err; rc=$?; tail -c 900 $O/bench_b1.json; [ $rc -ne 0 ] && { tail -20 $O/bench_b1.err; exit $rc; }
VD_R64_NO_KSPLIT=1 VD_LIB=tools/_timing/pf1.so timeout -k 10 300 python bench.py --batch 1 --frames 16 --steps 40 --warmup 5 --no-cpu-baseline --no-fp32-ref --no-dropin > $O/bench_b1_old.json 2> $O/bench_b1_old.err; tail -c 300 $O/bench_b1_old.json
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fp32-ref > $O/bench.json 2> $O/bench.err; rc=$?; tail -c 900 $O/bench.json; [ $rc -ne 0 ] && { tail -20 $O/bench.err; exit $rc; }
VD_LIB=tools/_timing/pf1.so timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fp32-ref --no-dropin > $O/bench_pf1.json 2> $O/bench_pf1.err; tail -c 300 $O/bench_pf1.json
for v in "" z0 s1 s2 s4 s16 s64 s128 s134 s150; do
  echo "== variant ${v:-product}" >> $O/ablate.log
  if [ -z "$v" ]; then timeout -k 10 120 python tools/s64_bench.py --kernel r64 --quick --reps 8 >> $O/ablate.log 2>&1
  else VD_LIB=tools/_timing/$v.so timeout -k 10 120 python tools/s64_bench.py --kernel r64 --quick --reps 8 >> $O/ablate.log 2>&1; fi
done
cat $O/ablate.log
echo ALL_OK
