#!/bin/bash
# same-box A/B of two library builds:  tools/r03_variant_ab.sh <a.so> <b.so>
cd $GRAFT_REPO_ROOT
F="--no-cpu-baseline --no-fp32-ref --no-dropin"
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d["kernel_classes"]; print(d["value"], d["ms_per_step"], "conv", k["conv3x3_wino_r64_kernel"]["ms"], "ups", k.get("conv3x3_wino_r64_ups_kernel", {}).get("ms"))'
for rep in 1 2 3; do
  for lib in "$@"; do
    echo "== $lib"; VD_LIB=$lib timeout -k 10 200 python bench.py --steps 20 --warmup 5 $F | python -c "$P" || exit 1
  done
done
