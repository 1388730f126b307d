#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-r03h}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q > $O/tests_ops.log 2>&1; rc=$?; tail -4 $O/tests_ops.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 900 python -m pytest tests/test_gpu_engine.py -m gpu -x -q > $O/tests_eng.log 2>&1; rc=$?; tail -4 $O/tests_eng.log; [ $rc -ne 0 ] && exit $rc
for env in "VD_GS_RS=0" "VD_GS_RS=1" "VD_GS_RS=1 VD_GS_NO192=1" "VD_GS_RS=0 VD_GS_NO192=1"; do
  echo "== $env" | tee -a $O/gs_ab.log
  env $env timeout -k 10 200 python tools/gs_bench.py 2>&1 | grep -v amdgpu.ids | tee -a $O/gs_ab.log
done
for env in "VD_GS_RS=0" "VD_GS_RS=1" "VD_GS_RS=1 VD_GS_NO192=1"; do
  echo "== bench $env" | tee -a $O/gs_ab.log
  env $env timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fp32-ref --no-dropin 2> $O/bench.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], {k: v['ms'] for k, v in d['kernel_classes'].items() if 'gemm' in k})" | tee -a $O/gs_ab.log
done
echo ALL_OK
