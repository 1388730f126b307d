#!/usr/bin/env python3
"""Timing of the two attention cores at the headline shapes (B=8, T=16; 16x16 C=384 and 8x8 C=512, 4 heads).
python tools/attn_bench.py [--reps 20]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_diffusion_amd import _lib  # noqa: E402


def timeit(fn, reps):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / reps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--T", type=int, default=16)
    args = ap.parse_args()
    L = _lib.lib()
    B, T, heads = 8, args.T, 4
    for HW, C in [(256, 384), (64, 512)]:
        qkv = torch.rand(B * T * HW, 3 * C, device="cuda") - 0.5
        R = [torch.rand(B, T, T, C, device="cuda") - 0.5 for _ in range(3)]
        mask = torch.ones(B, T, device="cuda")
        out = torch.empty(B * T * HW, C, device="cuda")
        t_rpe = timeit(lambda: _lib.check(L.vd_op_attn_temporal(_lib.ptr(qkv), _lib.ptr(R[0]), _lib.ptr(R[1]), _lib.ptr(R[2]), _lib.ptr(mask),
                                                                B, T, HW, C, heads, 0, _lib.ptr(out), _lib.current_stream())), args.reps)
        t_no = timeit(lambda: _lib.check(L.vd_op_attn_temporal(_lib.ptr(qkv), None, None, None, _lib.ptr(mask), B, T, HW, C, heads, 0,
                                                               _lib.ptr(out), _lib.current_stream())), args.reps)
        t_sp = timeit(lambda: _lib.check(L.vd_op_attn_spatial(_lib.ptr(qkv), B * T, HW, C, heads, _lib.ptr(out), _lib.current_stream())), args.reps)
        print(f"HW={HW:4d} C={C}: temporal with RPE {t_rpe:7.1f} us   without RPE {t_no:7.1f} us   "
              f"spatial {t_sp:7.1f} us", flush=True)


if __name__ == "__main__":
    main()
