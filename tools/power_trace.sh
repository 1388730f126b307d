#!/bin/bash
# Board power and shader clock while a command runs (rocm-smi sampled every 0.25 s by a child we start and end ourselves):
#   bash tools/power_trace.sh <tag> <command...>        -> gpurun_out/<tag>/power.log, summary on stdout
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=$1; shift; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
( while true; do rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Package Power|sclk|junction" | tr '\n' ' ' ; echo; sleep 0.25; done ) > $O/power.log 2>&1 &
SAMPLER=$!
sleep 1
"$@" > $O/cmd.log 2>&1; rc=$?
sleep 0.5
kill $SAMPLER 2>/dev/null
python3 - <<PY
import re
P=[];C=[];T=[]
for l in open("$O/power.log"):
    m=re.search(r"Package Power \(W\): ([\d.]+)",l); c=re.search(r"sclk clock level: \d+: \((\d+)Mhz\)",l); t=re.search(r"junction\) \(C\): ([\d.]+)",l)
    if m: P.append(float(m.group(1)))
    if c: C.append(int(c.group(1)))
    if t: T.append(float(t.group(1)))
if P: print("power W: idle(first) %.0f  max %.0f  mean of top half %.0f  (%d samples)"%(P[0], max(P), sum(sorted(P)[len(P)//2:])/max(len(P)-len(P)//2,1), len(P)))
if C: print("sclk MHz: min %d max %d"%(min(C),max(C)), "last", C[-5:])
if T: print("junction C: max %.0f"%max(T))
PY
tail -3 $O/cmd.log | cut -c1-300
exit $rc
