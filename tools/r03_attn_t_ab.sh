#!/bin/bash
# same-box A/B of the whole step: temporal attention on the VALU kernel (VD_ATTN_T=valu) against the matrix-pipe kernel
cd $GRAFT_REPO_ROOT
F="--no-cpu-baseline --no-fp32-ref --no-roofline --no-dropin"
for rep in 1 2; do
  for mode in valu mfma; do
    if [ $mode = valu ]; then export VD_ATTN_T=valu; else unset VD_ATTN_T; fi
    echo "== $mode headline";  timeout -k 10 200 python bench.py --steps 20 --warmup 5 $F | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" || exit 1
    echo "== $mode B=1 T=16";  timeout -k 10 200 python bench.py --batch 1 --frames 16 --steps 40 --warmup 5 $F | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" || exit 1
    echo "== $mode configs[4] window"; timeout -k 10 200 python bench.py --image-size 128 --batch 8 --frames 20 --obs 10 --respacing ddim50 --steps 5 --warmup 2 $F | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])" || exit 1
  done
done
