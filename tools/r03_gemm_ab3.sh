#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03q; mkdir -p $O; cd $R
VD_LIB=tools/_timing/pf128.so timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "linear or gemm or split" 2>&1 | tail -2
for v in "" pf128 "" pf128; do
  echo "== variant ${v:-product}" | tee -a $O/ab.log
  if [ -z "$v" ]; then VD_GS_NO192=1 timeout -k 10 200 python tools/gs_bench.py 2>&1 | grep -v amdgpu | tee -a $O/ab.log
  else VD_GS_NO192=1 VD_LIB=tools/_timing/$v.so timeout -k 10 200 python tools/gs_bench.py 2>&1 | grep -v amdgpu | tee -a $O/ab.log; fi
done
