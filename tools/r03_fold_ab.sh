#!/bin/bash
# same-box A/B: GroupNorm statistics folded by their own launch (VD_NO_GN_FOLD_FUSE=1) against inside the activation pass
cd $GRAFT_REPO_ROOT
F="--no-cpu-baseline --no-fp32-ref --no-roofline --no-dropin"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"])'
for rep in 1 2; do
  for mode in off on; do
    if [ $mode = off ]; then export VD_NO_GN_FOLD_FUSE=1; else unset VD_NO_GN_FOLD_FUSE; fi
    echo "== fold in the pass: $mode, headline"; timeout -k 10 200 python bench.py --steps 20 --warmup 5 $F | python -c "$P" || exit 1
    echo "== fold in the pass: $mode, B=1 T=16"; timeout -k 10 200 python bench.py --batch 1 --frames 16 --steps 40 --warmup 5 $F | python -c "$P" || exit 1
    echo "== fold in the pass: $mode, configs[4] window"; timeout -k 10 200 python bench.py --image-size 128 --batch 8 --frames 20 --obs 10 --respacing ddim50 --steps 5 --warmup 2 $F | python -c "$P" || exit 1
  done
done
