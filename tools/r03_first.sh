#!/bin/bash
# round-3 first box: parity tests, headline bench, the B=1 x T=16 shard (eager / graph) with a per-launch dump
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-r03a}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; rc=$?; tail -5 $O/tests.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; rc=$?; tail -c 1500 $O/bench.json; [ $rc -ne 0 ] && { tail -20 $O/bench.err; exit $rc; }
VD_PROF_DUMP=1 timeout -k 10 300 python bench.py --batch 1 --frames 16 --steps 40 --warmup 5 --no-cpu-baseline --no-fp32-ref > $O/bench_b1.json 2> $O/bench_b1.err; rc=$?; tail -c 1500 $O/bench_b1.json; [ $rc -ne 0 ] && { tail -20 $O/bench_b1.err; exit $rc; }
timeout -k 10 300 python bench.py --batch 1 --frames 16 --steps 40 --warmup 5 --executor graph --no-cpu-baseline --no-fp32-ref --no-roofline --no-dropin > $O/bench_b1_graph.json 2> $O/bench_b1_graph.err; rc=$?; tail -c 600 $O/bench_b1_graph.json; [ $rc -ne 0 ] && { tail -20 $O/bench_b1_graph.err; exit $rc; }
VD_PROF_DUMP=1 timeout -k 10 300 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-fp32-ref --no-dropin > $O/bench_dump.json 2> $O/bench_dump.err
echo ALL_OK
