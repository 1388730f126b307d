#!/bin/bash
# r04 measurement pass 1 (one box): executor test, conv ablations (MFMA + VALU only etc.), split accuracy of both split modes, full GPU test suite
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04f; mkdir -p $O; cd $R
timeout -k 10 300 python -m pytest tests/test_gpu_engine.py -q -k "window_executor_equals" > $O/t_exec.log 2>&1; rc=$?; tail -3 $O/t_exec.log; [ $rc -ge 124 ] && exit $rc
bash tools/ab.sh r04f "tools/conv_bench.py --quick --reps 6" tools/_timing/old_f16.so product > /dev/null 2>&1 || exit 124
grep -E "==|class total" $O/ab.log | tail -12
for m in f16x3 bf16x6; do VD_MATH=$m timeout -k 10 300 python tools/split_accuracy.py > $O/split_accuracy_$m.json 2> $O/split_accuracy_$m.err || { tail -5 $O/split_accuracy_$m.err; }; python -c "
import json; d=json.load(open('$O/split_accuracy_$m.json')); print('$m', 'worst ratio max', d['worst_ratio_max'], 'mean', d['worst_ratio_mean'], 'bias/mean', d['worst_bias_over_mean_err'])"; done
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/tests.log 2>&1; rc=$?; tail -6 $O/tests.log; [ $rc -ge 124 ] && exit $rc
echo PASS1_DONE
