#!/bin/bash
# z128: accuracy test, stamps of the timing build, conv census A/B against r64
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r04k; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -q -x -k "winograd_split or conv_is_deterministic" > $O/tests.log 2>&1; rc=$?; tail -5 $O/tests.log
[ $rc -ne 0 ] && exit $rc
[ -f tools/_timing/z128t.so ] && { VD_LIB=tools/_timing/z128t.so timeout -k 10 300 python tools/conv_bench.py --quick 2>&1 | grep -v amdgpu.ids | tee $O/stamps.log; }
bash tools/ab.sh r04k "tools/conv_bench.py --quick" product VD_CONV_Z128=0:product > /dev/null
grep -E "^==|class total" $O/ab.log | tail -12
