#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-r03g}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; rc=$?; tail -8 $O/tests.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fp32-ref > $O/bench.json 2> $O/bench.err; tail -c 500 $O/bench.json
echo ALL_OK
