#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-r03c}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_engine.py -m gpu -x -q -k "use_gradient_method" > $O/tests_guided.log 2>&1; rc=$?; tail -40 $O/tests_guided.log
VD_PROF_DUMP=1 timeout -k 10 300 python bench.py --batch 1 --frames 16 --steps 40 --warmup 5 --no-cpu-baseline --no-fp32-ref --no-dropin > $O/bench_b1.json 2> $O/bench_b1.err; tail -c 600 $O/bench_b1.json
exit $rc
