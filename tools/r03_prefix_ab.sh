#!/bin/bash
# same-box A/B: window executor without / with the prefix cache, headline window and the configs[4]-shaped one
cd $GRAFT_REPO_ROOT
F="--executor graph --no-cpu-baseline --no-fp32-ref --no-roofline --no-dropin"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["config"].get("cached_frames"))'
for rep in 1 2; do
  for pc in "" "--prefix-cache"; do
    echo "== headline $pc"; timeout -k 10 200 python bench.py --steps 20 --warmup 5 $F $pc | python -c "$P" || exit 1
    echo "== configs[4] window (obs 10 of 20) $pc"; timeout -k 10 200 python bench.py --image-size 128 --batch 8 --frames 20 --obs 10 --respacing ddim50 --steps 5 --warmup 2 $F $pc | python -c "$P" || exit 1
    echo "== B=1 T=16 $pc"; timeout -k 10 200 python bench.py --batch 1 --frames 16 --steps 40 --warmup 5 $F $pc | python -c "$P" || exit 1
  done
done
