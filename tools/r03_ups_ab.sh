#!/bin/bash
# same-box A/B: Upsample convs as F(2x2,3x3) on the upsampled map (VD_UPS_PHASE=0) against the sub-pixel form
cd $GRAFT_REPO_ROOT
F="--no-cpu-baseline --no-fp32-ref --no-roofline --no-dropin"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], (d.get("kernel_classes") or {}).get("conv3x3_wino_r64_kernel"))'
for rep in 1 2; do
  for mode in 0 1; do
    export VD_UPS_PHASE=$mode
    echo "== VD_UPS_PHASE=$mode headline"; timeout -k 10 200 python bench.py --steps 20 --warmup 5 $F | python -c "$P" || exit 1
    echo "== VD_UPS_PHASE=$mode configs[4] window"; timeout -k 10 200 python bench.py --image-size 128 --batch 8 --frames 20 --obs 10 --respacing ddim50 --steps 5 --warmup 2 $F | python -c "$P" || exit 1
    echo "== VD_UPS_PHASE=$mode B=1 T=16"; timeout -k 10 200 python bench.py --batch 1 --frames 16 --steps 40 --warmup 5 $F | python -c "$P" || exit 1
  done
done
