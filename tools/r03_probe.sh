#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r03f; mkdir -p $O; cd $R
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/mfma_slot_probe.hip -o /tmp/p 2>&1 | grep -v warning | head -5
timeout -k 10 120 /tmp/p | tee $O/mfma_slot_probe.txt
timeout -k 10 300 python tools/full_sampler_bench.py > $O/full_sampler.json 2> $O/full_sampler.err; tail -c 900 $O/full_sampler.json
