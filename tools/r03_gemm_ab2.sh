#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03n; mkdir -p $O; cd $R
VD_LIB=tools/_timing/slots.so timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q 2>&1 | tail -2
VD_LIB=tools/_timing/slots.so timeout -k 10 900 python -m pytest tests/test_gpu_engine.py -m gpu -x -q 2>&1 | tail -2
for v in "" slots "" slots; do
  echo "== variant ${v:-product}" | tee -a $O/ab.log
  if [ -z "$v" ]; then timeout -k 10 200 python tools/gs_bench.py 2>&1 | grep -v amdgpu | tee -a $O/ab.log
  else VD_LIB=tools/_timing/$v.so timeout -k 10 200 python tools/gs_bench.py 2>&1 | grep -v amdgpu | tee -a $O/ab.log; fi
done
for v in "" slots "" slots; do
  if [ -z "$v" ]; then timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fp32-ref --no-dropin 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('product', d['value'], d['ms_per_step'], {k: v['ms'] for k, v in d['kernel_classes'].items() if 'gemm' in k})"
  else VD_LIB=tools/_timing/$v.so timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fp32-ref --no-dropin 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['value'], d['ms_per_step'], {k: v['ms'] for k, v in d['kernel_classes'].items() if 'gemm' in k})"; fi
done | tee -a $O/ab.log
