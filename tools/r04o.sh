#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04p; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -q -x -k "winograd or conv" > $O/tests.log 2>&1; rc=$?; tail -3 $O/tests.log
[ $rc -ne 0 ] && exit $rc
bash tools/ab.sh r04p "tools/conv_bench.py" product tools/_timing/r64_b2load.so > /dev/null
grep -E "^==|class total" $O/ab.log | tail -12
