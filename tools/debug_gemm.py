import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from video_diffusion_amd import _lib
L = _lib.lib()
M, K, N = 128, 64, 64
A = torch.arange(M * K, dtype=torch.float32).reshape(M, K) / 100.0
W = torch.zeros(N, K); W[torch.arange(N), torch.arange(N)] = 1.0
wf = torch.empty(N * K)
_lib.check(L.vd_pack_linear_frag(_lib.ptr(W.contiguous()), _lib.ptr(wf), N, K))
out = torch.full((M, N), -7.0, device="cuda")
Ad, wfd = A.cuda(), wf.cuda()
_lib.check(L.vd_op_conv(_lib.ptr(Ad), None, K, K, M, 1, 1, 0, 1, 0, 1, None, _lib.ptr(wfd), None, None, None, None, 0, None, None, 0,
                        _lib.ptr(out), N, _lib.current_stream()))
torch.cuda.synchronize()
o = out.cpu()
print("max err", (o - A[:, :N]).abs().max().item())
print(o[:4, :8]); print(A[:4, :8])
print(o[60:64, 28:36]); print(A[60:64, 28:36])
