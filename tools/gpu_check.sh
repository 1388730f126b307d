#!/bin/bash
# One GPU-box pass: parity tests, the headline bench, the configs[4]-shaped bench, a 2-rank rehearsal of the self-launching
# bench on one card (gloo, both ranks on device 0).  Usage (via gpurun): bash tools/gpu_check.sh <tag>
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-now}; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; rc=$?; tail -5 $O/tests.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; rc=$?; tail -c 1500 $O/bench.json; [ $rc -ne 0 ] && { tail -20 $O/bench.err; exit $rc; }
timeout -k 10 300 python bench.py --executor graph --steps 20 --warmup 5 --no-cpu-baseline --no-fp32-ref --no-roofline --no-dropin > $O/bench_graph.json 2> $O/bench_graph.err; rc=$?; tail -c 400 $O/bench_graph.json; [ $rc -ne 0 ] && { tail -20 $O/bench_graph.err; exit $rc; }
timeout -k 10 300 python bench.py --image-size 128 --batch 8 --frames 20 --obs 10 --respacing ddim50 --steps 5 --warmup 2 --no-cpu-baseline --no-fp32-ref > $O/bench_c4.json 2> $O/bench_c4.err; rc=$?; tail -c 1200 $O/bench_c4.json; [ $rc -ne 0 ] && { tail -20 $O/bench_c4.err; exit $rc; }
timeout -k 10 300 python bench.py --image-size 128 --batch 4 --frames 16 --obs 4 --respacing "" --steps 10 --warmup 2 --no-cpu-baseline --no-fp32-ref > $O/bench_c3.json 2> $O/bench_c3.err; rc=$?; tail -c 600 $O/bench_c3.json; [ $rc -ne 0 ] && { tail -20 $O/bench_c3.err; exit $rc; }
VD_BENCH_BACKEND=gloo VD_BENCH_ALL_ON_DEVICE0=1 timeout -k 10 300 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --no-fp32-ref --no-roofline > $O/bench_2rank.json 2> $O/bench_2rank.err; rc=$?; tail -c 600 $O/bench_2rank.json; [ $rc -ne 0 ] && { tail -20 $O/bench_2rank.err; exit $rc; }
VD_BENCH_BACKEND=gloo VD_BENCH_ALL_ON_DEVICE0=1 timeout -k 10 300 python bench.py --gpus 2 --scaling strong --steps 5 --warmup 2 --no-cpu-baseline --no-fp32-ref --no-roofline --no-dropin > $O/bench_2rank_strong.json 2> $O/bench_2rank_strong.err; rc=$?; tail -c 600 $O/bench_2rank_strong.json; [ $rc -ne 0 ] && { tail -20 $O/bench_2rank_strong.err; exit $rc; }
timeout -k 10 300 python bench.py --batch 1 --frames 16 --steps 40 --warmup 5 --no-cpu-baseline --no-fp32-ref > $O/bench_B1_T16.json 2> $O/bench_B1_T16.err; rc=$?; tail -c 400 $O/bench_B1_T16.json; [ $rc -ne 0 ] && { tail -20 $O/bench_B1_T16.err; exit $rc; }
timeout -k 10 300 python bench.py --batch 1 --frames 16 --steps 40 --warmup 5 --executor graph --no-cpu-baseline --no-fp32-ref --no-roofline --no-dropin > $O/bench_B1_T16_graph.json 2> $O/bench_B1_T16_graph.err; rc=$?; tail -c 300 $O/bench_B1_T16_graph.json; [ $rc -ne 0 ] && { tail -20 $O/bench_B1_T16_graph.err; exit $rc; }
timeout -k 10 300 python tools/full_sampler_bench.py > $O/full_sampler.json 2> $O/full_sampler.err; rc=$?; tail -c 500 $O/full_sampler.json; [ $rc -ne 0 ] && { tail -20 $O/full_sampler.err; exit $rc; }
echo ALL_OK
