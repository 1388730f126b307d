#!/bin/bash
# Same-box A/B of library builds, alternating, three rounds:   tools/ab.sh <tag> <script + args, quoted> <lib|product> [<lib|product> ...]
#   e.g. (via gpurun)  bash tools/ab.sh r04c "tools/conv_bench.py --quick" product tools/_timing/abl1.so tools/_timing/abl2.so
#        bash tools/ab.sh r04d "bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fp32-ref --no-dropin" product tools/_timing/x.so
# Environment switches of the product library are A/B'd the same way: VD_MATH=bf16x6:product as a lib name sets the variable.
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=$1; CMD=$2; shift; shift; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
for rep in 1 2 3; do
  for lib in "$@"; do
    envs=""; l=$lib
    while [[ "$l" == *=*:* ]]; do envs="$envs ${l%%:*}"; l=${l#*:}; done
    echo "== round $rep: $lib" | tee -a $O/ab.log
    if [ "$l" = product ]; then env $envs timeout -k 10 300 python $CMD 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.log
    else env $envs VD_LIB=$l timeout -k 10 300 python $CMD 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.log; fi
    [ ${PIPESTATUS[0]} -ge 124 ] && exit 124
  done
done
exit 0
