"""Error of the bf16x6 split GEMM (gemm_split.hip) and of the fp32-MFMA GEMM (gemm_frag.hip) against an fp64 product,
over K and with / without a residual in the accumulator.  Usage (GPU box): python tools/split_accuracy.py"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
_lib = importlib.import_module("video-diffusion_amd._lib")


def main():
    L = _lib.lib()
    g = torch.Generator().manual_seed(0)
    M, N = 512, 128
    print(f"{'K':>6} {'res':>3} | {'split mean':>11} {'fp32 mean':>11} {'ratio':>6} | {'split max':>10} {'fp32 max':>10}")
    for K in (32, 96, 288, 864, 1152, 2304, 4608):
        for res in (0, 1):
            a = torch.rand(M, K, generator=g) * 2 - 1
            w = (torch.rand(N, K, generator=g) * 2 - 1) * (3.0 / K) ** 0.5
            r = (torch.rand(M, N, generator=g) * 2 - 1) if res else None
            ws = torch.empty(3 * N * K, dtype=torch.int16)
            _lib.check(L.vd_pack_linear_split(_lib.ptr(w), _lib.ptr(ws), N, K))
            wf = torch.empty(N * K)
            _lib.check(L.vd_pack_linear_frag(_lib.ptr(w), _lib.ptr(wf), N, K))
            ad, wsd, wfd = a.cuda(), ws.cuda(), wf.cuda()
            rd = r.cuda() if res else None
            o_s, o_f = torch.empty(M, N, device="cuda"), torch.empty(M, N, device="cuda")
            st = _lib.current_stream()
            _lib.check(L.vd_op_linear_split(_lib.ptr(ad), M, K, _lib.ptr(wsd), None, _lib.ptr(rd), 0, _lib.ptr(o_s), N, st))
            _lib.check(L.vd_op_conv(_lib.ptr(ad), None, K, K, M, 1, 1, 0, 1, 0, 1, None, _lib.ptr(wfd), None, None, None, None, 0,
                                    _lib.ptr(rd), None, 0, _lib.ptr(o_f), N, st))
            torch.cuda.synchronize()
            ref = a.double() @ w.double().t() + (r.double() if res else 0)
            es, ef = (o_s.cpu().double() - ref).abs(), (o_f.cpu().double() - ref).abs()
            print(f"{K:6d} {res:3d} | {es.mean():11.3e} {ef.mean():11.3e} {es.mean() / ef.mean():6.2f} | {es.max():10.3e} {ef.max():10.3e}")


if __name__ == "__main__":
    main()
