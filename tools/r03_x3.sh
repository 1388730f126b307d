#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03s; mkdir -p $O; cd $R
VD_MATH=bf16x3 timeout -k 10 300 python tools/x3_check.py 2> $O/x3.err | tee $O/x3_check.json; tail -3 $O/x3.err
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q 2>&1 | tail -3
timeout -k 10 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d.get('fp32_mfma_only'), d.get('bf16x3_declared_reduced_mode'))"
VD_MATH=bf16x3 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dropin 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('x3', d['value'], d['ms_per_step'], {k: v['ms'] for k, v in d['kernel_classes'].items()})"
