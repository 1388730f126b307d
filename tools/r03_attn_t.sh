#!/bin/bash
# temporal attention on the matrix pipe: parity of the op tests, then old (VD_ATTN_T=valu) against new at the headline shapes
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -q -x -k "attention_temporal" 2>&1 | tail -15 || exit 1
for T in 16 20; do
  echo "== VALU kernel, T=$T"; VD_ATTN_T=valu timeout -k 10 120 python tools/attn_bench.py --T $T --reps 50 || exit 1
  echo "== matrix-pipe kernel, T=$T"; timeout -k 10 120 python tools/attn_bench.py --T $T --reps 50 || exit 1
done
