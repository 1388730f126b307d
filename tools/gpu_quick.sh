#!/bin/bash
# One short GPU-box pass while iterating on a kernel: the operator tests, then the headline bench without its side legs.
# A step that had to be killed at its limit ends the pass (nothing else is started on that box).   bash tools/gpu_quick.sh <tag> [pytest args]
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-quick}; shift; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest ${@:-tests/test_gpu_ops.py} -q -x > $O/tests.log 2>&1; rc=$?; tail -15 $O/tests.log
[ $rc -ge 124 ] && exit $rc
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-fp32-ref --no-dropin > $O/bench.json 2> $O/bench.err; rc2=$?
[ $rc2 -ne 0 ] && { tail -20 $O/bench.err; exit $rc2; }
python - <<PY
import json
d = json.load(open("$O/bench.json"))
print("steps/s", d["value"], "ms/step", d["ms_per_step"])
for k, v in sorted(d["kernel_classes"].items(), key=lambda kv: -kv[1]["ms"]): print(f"  {k:34s} {v['launches']:3d} {v['ms']:7.3f} ms  {v['tflops']} TFLOP/s  {v['gbs']} GB/s")
PY
exit $rc
