"""GPU: every HIP kernel through its C-ABI entry point vs a plain fp32 torch-CPU statement of the
same operator (reference semantics cited per test), seeded inputs, ragged / edge shapes."""
import numpy as np
import os

import pytest
import torch
import torch.nn.functional as F

from helpers import close
from video_diffusion_amd import _lib

pytestmark = pytest.mark.gpu
TOL = dict(atol=1e-4, rtol=1e-4)          # the tier's stated fp32 tolerance


def dev(t):
    return t.to("cuda").contiguous()


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def pack_conv(w):                          # OIHW -> [tap][O][I]   (engine layout, include/vd_amd.h)
    O, I, kh, kw = w.shape
    return w.permute(2, 3, 0, 1).reshape(kh * kw, O, I).contiguous()


def pack_lin_frag(w):
    """[N][K] -> [K/32][N/32][4][64][4] through the library's host packer (+ closed-form check)."""
    N, K = w.shape
    src = w.contiguous().float()
    out = torch.empty(N * K)
    _lib.check(_lib.lib().vd_pack_linear_frag(_lib.ptr(src), _lib.ptr(out), N, K))
    ref = w.reshape(N // 32, 32, K // 32, 4, 2, 4).permute(2, 0, 3, 4, 1, 5).reshape(-1)
    assert torch.equal(out, ref)
    return out


def pack_wino(w):
    """OIHW -> Winograd F(2x2,3x3) image U = G g G^T in [I/16][16][O/32][2][64][4] (library packer + closed form).
    Row 2 of U is stored negated: the kernel takes row 2 of B^T negated as well (one fma per value)."""
    O, I = w.shape[:2]
    src = w.contiguous().float()
    out = torch.empty(16 * O * I)
    _lib.check(_lib.lib().vd_pack_conv3_wino(_lib.ptr(src), _lib.ptr(out), O, I))
    G = torch.tensor([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], dtype=torch.float64)
    U = torch.einsum("ik,ockl,jl->ijoc", G, w.double(), G).float()                            # [i][j][co][ci]
    U[2] = -U[2]
    U = U.reshape(16, O, I)
    ref = U.reshape(16, O // 32, 32, I // 16, 2, 2, 4).permute(3, 0, 1, 4, 5, 2, 6).reshape(-1)
    assert torch.equal(out, ref)
    return out


def run_conv(x0, x1, w, bias, *, ups=0, stride=1, affA=None, affB=None, act=0, res=None, fbias=None, generic=False,
             wino=True):
    """x0/x1 NCHW cpu tensors; returns NCHW cpu tensor computed by the HIP kernel."""
    N, C0, H, W = x0.shape
    Cin = C0 + (x1.shape[1] if x1 is not None else 0)
    O, _, k, _ = w.shape
    pad = 1 if k == 3 else 0
    Ho = ((H << ups) + 2 * pad - k) // stride + 1
    out = torch.empty(N, Ho, Ho, O, device="cuda")
    d = lambda t: None if t is None else dev(t)  # noqa: E731
    bufs = [dev(nhwc(x0)), d(nhwc(x1)) if x1 is not None else None, dev(pack_conv(w)), d(bias), d(affA), d(affB),
            d(nhwc(res)) if res is not None else None, d(fbias)]
    wfrag = wwino = None
    if O % 32 == 0 and Cin % 32 == 0 and not generic:
        if k == 3:
            if O % 64 == 0 and wino:
                wwino = dev(pack_wino(w))
        elif affA is None and fbias is None:
            wfrag = dev(pack_lin_frag(w.reshape(O, Cin)))
    rc = _lib.lib().vd_op_conv(_lib.ptr(bufs[0]), _lib.ptr(bufs[1]), C0, Cin, N, H, W, ups, stride, pad, k,
                               _lib.ptr(bufs[2]), _lib.ptr(wfrag), _lib.ptr(wwino), _lib.ptr(bufs[3]), _lib.ptr(bufs[4]),
                               _lib.ptr(bufs[5]), act,
                               _lib.ptr(bufs[6]), _lib.ptr(bufs[7]), 0 if fbias is None else fbias.shape[1],
                               _lib.ptr(out), O, _lib.current_stream())
    _lib.check(rc)
    torch.cuda.synchronize()
    return out.permute(0, 3, 1, 2).cpu()


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


@pytest.mark.parametrize("N,Cin,Cout,H", [(3, 32, 64, 16), (2, 64, 32, 8), (5, 128, 128, 4), (1, 96, 160, 32),
                                          (2, 32, 3, 8)])
def test_conv3x3_plain(N, Cin, Cout, H):
    x, w, b = rnd(N, Cin, H, H), rnd(Cout, Cin, 3, 3, scale=(3.0 / (9 * Cin)) ** 0.5), rnd(Cout, scale=0.1)
    ref = F.conv2d(x, w, b, padding=1)
    close(run_conv(x, None, w, b), ref, **TOL)                       # fp32-MFMA Winograd kernel where the shape allows
    close(run_conv(x, None, w, b, generic=True), ref, **TOL)         # generic per-tap kernel


@pytest.mark.parametrize("N,C0,C1,Cout,H", [(3, 64, 32, 96, 16), (3, 64, 32, 128, 16), (5, 32, 32, 64, 8), (2, 96, 0, 192, 32)])
def test_conv3x3_fused_norm_film_silu_residual_concat(N, C0, C1, Cout, H):
    """ResBlock operand path fused into the conv's operand load (unet.py:185-198): conv(silu(x*A+B)) + bias + skip,
    input = cat([h, skip]).  Runs on the direct kernels (the Winograd kernel takes one plain tensor; see
    test_conv3x3_winograd_after_activation_pass for the engine's form of the same block)."""
    h, s = rnd(N, C0, H, H), (rnd(N, C1, H, H, seed=1) if C1 else None)
    A, B = rnd(N, C0 + C1, seed=2) + 1.5, rnd(N, C0 + C1, seed=3)
    w, b, res = rnd(Cout, C0 + C1, 3, 3, scale=0.06), rnd(Cout, scale=0.1), rnd(N, Cout, H, H, seed=4)
    x = torch.cat([h, s], 1) if C1 else h
    ref = F.conv2d(F.silu(x * A[:, :, None, None] + B[:, :, None, None]), w, b, padding=1) + res
    close(run_conv(h, s, w, b, affA=A, affB=B, act=1, res=res), ref, **TOL)


def affine_act(x0, x1, A, B, act=1):
    """vd_op_affine_act on NCHW cpu tensors -> NCHW cpu tensor with C0+C1 channels."""
    N, C0, H, W = x0.shape
    C = C0 + (x1.shape[1] if x1 is not None else 0)
    y = torch.empty(N, H, W, C, device="cuda")
    bufs = [dev(nhwc(x0)), dev(nhwc(x1)) if x1 is not None else None, dev(A), dev(B)]
    _lib.check(_lib.lib().vd_op_affine_act(_lib.ptr(bufs[0]), _lib.ptr(bufs[1]), C0, C, _lib.ptr(bufs[2]), _lib.ptr(bufs[3]),
                                           N, H * W, act, _lib.ptr(y), _lib.current_stream()))
    torch.cuda.synchronize()
    return y.permute(0, 3, 1, 2).cpu()


@pytest.mark.parametrize("N,C0,C1,H", [(3, 64, 32, 16), (2, 32, 0, 8), (5, 128, 64, 4)])
def test_affine_act_pass(N, C0, C1, H):
    """GroupNorm(+FiLM) affine + SiLU over the virtual concat, written once (the 3x3 convs' input in the engine)."""
    h, s = rnd(N, C0, H, H), (rnd(N, C1, H, H, seed=1) if C1 else None)
    A, B = rnd(N, C0 + C1, seed=2) + 1.5, rnd(N, C0 + C1, seed=3)
    x = torch.cat([h, s], 1) if C1 else h
    lin = x * A[:, :, None, None] + B[:, :, None, None]
    close(affine_act(h, s, A, B, act=1), F.silu(lin), atol=2e-6, rtol=2e-6)
    close(affine_act(h, s, A, B, act=0), lin, atol=1e-6, rtol=1e-6)


@pytest.mark.parametrize("N,C0,C1,Cout,H", [(3, 64, 32, 128, 16), (5, 32, 32, 64, 8), (2, 96, 0, 192, 32), (9, 64, 0, 64, 8),
                                            (1, 32, 32, 128, 64)])
def test_conv3x3_winograd_after_activation_pass(N, C0, C1, Cout, H):
    """The engine's ResBlock convs: activation pass (affine + SiLU + concat) then the Winograd kernel on the plain
    tensor with bias + residual in its output transform (8x8: four frames per block; 5 and 9 frames -> ragged last
    block whose rows are dropped by the buffer range check)."""
    h, s = rnd(N, C0, H, H), (rnd(N, C1, H, H, seed=1) if C1 else None)
    A, B = rnd(N, C0 + C1, seed=2) + 1.5, rnd(N, C0 + C1, seed=3)
    w, b, res = rnd(Cout, C0 + C1, 3, 3, scale=0.06), rnd(Cout, scale=0.1), rnd(N, Cout, H, H, seed=4)
    x = torch.cat([h, s], 1) if C1 else h
    ref = F.conv2d(F.silu(x * A[:, :, None, None] + B[:, :, None, None]), w, b, padding=1) + res
    act = affine_act(h, s, A, B)
    close(run_conv(act, None, w, b, res=res), ref, **TOL)                      # Winograd
    close(run_conv(act, None, w, b, res=res, wino=False), ref, **TOL)          # direct kernel on the same input
    fb = rnd(N, Cout, seed=5)
    close(run_conv(act, None, w, b, res=res, fbias=fb), ref + fb[:, :, None, None], **TOL)   # + per-frame bias (no FiLM)


@pytest.mark.parametrize("N,Cin,Cout,H,C1", [(3, 64, 128, 32, 0), (5, 32, 64, 8, 0), (2, 96, 192, 16, 64), (6, 64, 64, 8, 32)])
def test_conv3x3_winograd_groupnorm_partials(N, Cin, Cout, H, C1):
    """GroupNorm statistics of the conv OUTPUT from its own epilogue (per frame, block, channel partial sums) folded
    by vd_op_gn_affine -- alone and as the first half of a channel concat whose second half went through the ordinary
    statistics pass -- against vd_op_gn_fold reading the tensors (unet.py:185-198: the next GroupNorm + FiLM)."""
    L = _lib.lib()
    x, w, b = rnd(N, Cin, H, H), rnd(Cout, Cin, 3, 3, scale=0.06), rnd(Cout, scale=0.1)
    res = rnd(N, Cout, H, H, seed=4)
    xd, wd, bd, rd = dev(nhwc(x)), dev(pack_wino(w)), dev(b), dev(nhwc(res))
    out = torch.empty(N, H, H, Cout, device="cuda")
    split = L.vd_conv_stats_split(H)
    part = torch.full((N, split, Cout, 2), float("nan"), dtype=torch.float64, device="cuda")
    _lib.check(L.vd_op_conv_stats(_lib.ptr(xd), Cin, N, H, H, 0, _lib.ptr(wd), _lib.ptr(bd), _lib.ptr(rd), None, 0, _lib.ptr(out),
                                  Cout, _lib.ptr(part), _lib.current_stream()))
    torch.cuda.synchronize()
    ref = F.conv2d(x, w, b, padding=1) + res
    close(out.permute(0, 3, 1, 2).cpu(), ref, **TOL)
    # the table against sums over the stored tensor
    o64 = out.double()
    tot = part.sum(1).cpu()
    close(tot[..., 0], o64.sum((1, 2)).cpu(), atol=1e-3, rtol=1e-5)
    close(tot[..., 1], (o64 * o64).sum((1, 2)).cpu(), atol=1e-3, rtol=1e-5)
    # folded affine: fused table (+ a second source through the statistics pass) == statistics pass over everything
    C = Cout + C1
    gamma, beta, film = dev(rnd(C, seed=6) + 1.0), dev(rnd(C, seed=7)), dev(rnd(N, 2 * C, seed=8))
    s1 = dev(nhwc(rnd(N, C1, H, H, seed=9))) if C1 else None
    A0, B0, A1, B1 = (torch.empty(N, C, device="cuda") for _ in range(4))
    _lib.check(L.vd_op_gn_fold(_lib.ptr(out), _lib.ptr(s1), Cout, C, N, H * H, _lib.ptr(gamma), _lib.ptr(beta), _lib.ptr(film), 2 * C,
                               _lib.ptr(A0), _lib.ptr(B0), _lib.current_stream()))
    part1, split1 = None, 0
    if C1:                       # second table: per-channel sums of s1 in one block per frame (same layout, split 1)
        s64 = s1.double()
        part1 = torch.stack([s64.sum((1, 2)), (s64 * s64).sum((1, 2))], -1).reshape(N, 1, C1, 2).contiguous()
        split1 = 1
    _lib.check(L.vd_op_gn_affine(_lib.ptr(part), split, Cout, _lib.ptr(part1), split1, C, N, H * H, _lib.ptr(gamma), _lib.ptr(beta),
                                 _lib.ptr(film), 2 * C, _lib.ptr(A1), _lib.ptr(B1), _lib.current_stream()))
    torch.cuda.synchronize()
    close(A1.cpu(), A0.cpu(), atol=2e-6, rtol=2e-6)
    close(B1.cpu(), B0.cpu(), atol=2e-6, rtol=2e-6)


def test_conv3x3_upsample_fused():
    """Upsample (unet.py:63-72): nearest x2 read through the gather, zero padding at the UPSAMPLED border."""
    x, w, b = rnd(2, 64, 8, 8), rnd(64, 64, 3, 3, scale=0.07), rnd(64, scale=0.1)
    ref = F.conv2d(F.interpolate(x, scale_factor=2, mode="nearest"), w, b, padding=1)
    close(run_conv(x, None, w, b, ups=1), ref, **TOL)
    close(run_conv(x, None, w, b, ups=1, wino=False), ref, **TOL)


@pytest.mark.parametrize("H", [16, 8, 2])
def test_conv3x3_stride2(H):
    """Downsample (unet.py:88-95)."""
    x, w, b = rnd(3, 32, H, H), rnd(32, 32, 3, 3, scale=0.1), rnd(32, scale=0.1)
    close(run_conv(x, None, w, b, stride=2), F.conv2d(x, w, b, stride=2, padding=1), **TOL)


def test_conv1x1_skip_and_frame_bias():
    x0, x1 = rnd(2, 64, 8, 8), rnd(2, 64, 8, 8, seed=5)
    w, b, fb = rnd(96, 128, 1, 1, scale=0.15), rnd(96, scale=0.1), rnd(2, 200, seed=6)
    ref = F.conv2d(torch.cat([x0, x1], 1), w, b) + fb[:, :96, None, None]
    close(run_conv(x0, x1, w, b, fbias=fb), ref, **TOL)


def test_conv1x1_concat_residual_fragment_path():
    """ResBlock skip connection over cat([h, skip]) (unet.py:171-173,198) on the fragment-major GEMM, and
    proj_out + residual (unet.py:537-538); both against the generic kernel as well."""
    x0, x1 = rnd(3, 96, 16, 16), rnd(3, 32, 16, 16, seed=5)
    w, b, res = rnd(160, 128, 1, 1, scale=0.15), rnd(160, scale=0.1), rnd(3, 160, 16, 16, seed=7)
    ref = F.conv2d(torch.cat([x0, x1], 1), w, b) + res
    close(run_conv(x0, x1, w, b, res=res), ref, **TOL)
    close(run_conv(x0, x1, w, b, res=res, generic=True), ref, **TOL)


@pytest.mark.parametrize("M,K,Nout", [(128, 128, 512), (7, 512, 1500), (4099, 96, 288), (33, 32, 40), (300, 1024, 64)])
def test_linear_ragged(M, K, Nout):
    """nn.Linear as a 1x1 'convolution' over M rows: M, N not multiples of any tile."""
    a, w, b = rnd(M, K), rnd(Nout, K, scale=(3.0 / K) ** 0.5), rnd(Nout, scale=0.1)
    got = run_conv(a.view(M, K, 1, 1), None, w.view(Nout, K, 1, 1), b, act=1).view(M, Nout)
    close(got, F.linear(F.silu(a), w, b), **TOL)


def test_conv_is_deterministic_and_linear_at_scale():
    """Size-independent properties at a BASELINE-sized layer (N=128 frames, 64x64, 128->128):
    bit-identical reruns, and linearity conv(a*x) = a*conv(x) - (a-1)*bias."""
    g = torch.Generator().manual_seed(3)
    x = torch.rand(128, 64, 64, 128, generator=g, device="cpu").cuda() - 0.5        # already NHWC
    w_oihw = rnd(128, 128, 3, 3, scale=0.03)
    w, ww = dev(pack_conv(w_oihw)), dev(pack_wino(w_oihw))
    b = dev(rnd(128, scale=0.1))
    outs = []
    for scale, wino in ((1.0, ww), (1.0, ww), (2.0, ww), (1.0, None)):
        xs = (x * scale).contiguous()
        o = torch.empty(128, 64, 64, 128, device="cuda")
        _lib.check(_lib.lib().vd_op_conv(_lib.ptr(xs), None, 128, 128, 128, 64, 64, 0, 1, 1, 3, _lib.ptr(w), None,
                                         _lib.ptr(wino), _lib.ptr(b), None, None, 0, None, None, 0, _lib.ptr(o), 128,
                                         _lib.current_stream()))
        outs.append(o)
    # the engine's kernel (conv_wino_r64.hip, the process' arithmetic) on the same layer
    ws = dev(pack_wino_split(w_oihw))
    for scale in (1.0, 1.0, 2.0):
        xs = (x * scale).contiguous()
        o = torch.empty(128, 64, 64, 128, device="cuda")
        _lib.check(_lib.lib().vd_op_conv_wino_split(_lib.ptr(xs), 128, 128, 64, 64, 0, _lib.ptr(ws), _lib.ptr(b), None, None, 0, _lib.ptr(o), 128,
                                                    None, _lib.current_stream()))
        outs.append(o)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[4], outs[5])
    lin = 2 * outs[0] - b.view(1, 1, 1, -1)
    assert (outs[2] - lin).abs().max().item() < 2e-5
    # (every operand of the split kernel scales exactly by two: so does its result)
    assert (outs[6] - (2 * outs[4] - b.view(1, 1, 1, -1))).abs().max().item() < 2e-5
    # the fp32-MFMA Winograd kernel, the generic per-tap kernel and the split Winograd kernel are three implementations of the same sum
    assert (outs[0] - outs[3]).abs().max().item() < 5e-5
    assert (outs[4] - outs[3]).abs().max().item() < 5e-5


def _gn_fold(x0, x1, gamma, beta, film):
    N, C0, H, W = x0.shape
    C = C0 + (x1.shape[1] if x1 is not None else 0)
    A = torch.empty(N, C, device="cuda")
    B = torch.empty(N, C, device="cuda")
    bufs = [dev(nhwc(x0)), dev(nhwc(x1)) if x1 is not None else None, dev(gamma), dev(beta),
            dev(film) if film is not None else None]
    _lib.check(_lib.lib().vd_op_gn_fold(_lib.ptr(bufs[0]), _lib.ptr(bufs[1]), C0, C, N, H * W, _lib.ptr(bufs[2]),
                                        _lib.ptr(bufs[3]), _lib.ptr(bufs[4]), 0 if film is None else film.shape[1],
                                        _lib.ptr(A), _lib.ptr(B), _lib.current_stream()))
    torch.cuda.synchronize()
    return A.cpu(), B.cpu()


@pytest.mark.parametrize("N,C0,C1,H", [(4, 32, 0, 32), (3, 64, 32, 8), (2, 128, 96, 4), (5, 384, 0, 16), (2, 512, 384, 8),
                                       (1, 1024, 0, 2)])
def test_groupnorm_fold(N, C0, C1, H):
    """GroupNorm32 (nn.py:15-17,80-86) + FiLM (unet.py:190-194) folded to x*A+B; groups may straddle the concat."""
    C = C0 + C1
    x0 = rnd(N, C0, H, H) * 2 + 0.7
    x1 = rnd(N, C1, H, H, seed=7) if C1 else None
    x = torch.cat([x0, x1], 1) if C1 else x0
    gamma, beta, film = rnd(C, seed=8) + 1, rnd(C, seed=9), rnd(N, 2 * C + 8, seed=10)
    A, B = _gn_fold(x0, x1, gamma, beta, film)
    ref = F.group_norm(x, 32, gamma, beta, eps=1e-5) * (1 + film[:, :C, None, None]) + film[:, C:2 * C, None, None]
    close(x * A[:, :, None, None] + B[:, :, None, None], ref, atol=2e-5, rtol=2e-5)
    A2, B2 = _gn_fold(x0, x1, gamma, beta, None)
    close(x * A2[:, :, None, None] + B2[:, :, None, None], F.group_norm(x, 32, gamma, beta, eps=1e-5), atol=2e-5,
          rtol=2e-5)


def test_groupnorm_fold_large_mean():
    """fp64 accumulation: a large common offset must not destroy the variance."""
    x = rnd(2, 64, 16, 16) * 0.01 + 100.0
    gamma, beta = torch.ones(64), torch.zeros(64)
    A, B = _gn_fold(x, None, gamma, beta, None)
    close(x * A[:, :, None, None] + B[:, :, None, None], F.group_norm(x, 32, gamma, beta, eps=1e-5), atol=5e-3,
          rtol=1e-3)


@pytest.mark.parametrize("B,T,HW,C", [(2, 4, 64, 64), (1, 16, 16, 384), (2, 20, 9, 96), (1, 1, 5, 32), (1, 32, 3, 128)])
def test_groupnorm_temporal(B, T, HW, C):
    """RPEAttention norm on the (B*HW, C, T) view (unet.py:472-475): statistics over (T x C/32)."""
    x = rnd(B, T, HW, C) * 1.5 + 0.3
    gamma, beta = rnd(C, seed=1) + 1, rnd(C, seed=2)
    y = torch.empty(B, T, HW, C, device="cuda")
    bufs = [dev(x), dev(gamma), dev(beta)]
    _lib.check(_lib.lib().vd_op_gn_temporal(_lib.ptr(bufs[0]), _lib.ptr(bufs[1]), _lib.ptr(bufs[2]), B, T, HW, C,
                                            _lib.ptr(y), _lib.current_stream()))
    torch.cuda.synchronize()
    ref = F.group_norm(x.permute(0, 2, 3, 1).reshape(B * HW, C, T), 32, gamma, beta, eps=1e-5)
    close(y.cpu(), ref.view(B, HW, C, T).permute(0, 3, 1, 2), atol=2e-5, rtol=2e-5)


@pytest.mark.parametrize("N,L,C,heads", [(3, 256, 64, 4), (2, 64, 128, 4), (2, 100, 384, 4), (1, 16, 32, 4), (2, 257, 512, 4),
                                         (1, 40, 96, 2), (1, 33, 80, 2), (2, 70, 112, 2), (1, 50, 160, 2), (1, 90, 224, 2),
                                         (1, 24, 24, 1), (1, 1024, 384, 4), (1, 600, 256, 2), (2, 128, 192, 2)])
def test_attention_spatial(N, L, C, heads):
    """softmax(q*s k^T) v per (frame, head) (unet.py:477-489,525-536 with no RPE, no mask); ragged L."""
    qkv = rnd(N, L, 3 * C) * 2
    out = torch.empty(N, L, C, device="cuda")
    buf = dev(qkv)
    _lib.check(_lib.lib().vd_op_attn_spatial(_lib.ptr(buf), N, L, C, heads, _lib.ptr(out), _lib.current_stream()))
    torch.cuda.synchronize()
    Fd = C // heads
    q, k, v = qkv.view(N, L, 3, heads, Fd).permute(2, 0, 3, 1, 4)
    ref = (torch.softmax((q * Fd ** -0.5) @ k.transpose(-1, -2), -1) @ v).permute(0, 2, 1, 3).reshape(N, L, C)
    close(out.cpu(), ref, **TOL)


@pytest.mark.parametrize("qmag,kmag,vmag", [(1.0, 1.0, 1.0), (1e-3, 1.0, 1e-3), (30.0, 0.5, 100.0), (1e-6, 1e-6, 1e3), (200.0, 0.05, 1e-4)])
def test_attention_spatial_accuracy_over_magnitudes(qmag, kmag, vmag):
    """The split arithmetic of csrc/attn_spatial.hip against an fp64 softmax(q k^T) v.  f16x3 scales every query row to the top of
    fp16's range (a power of two, undone on the scores) and the softmax weights by 2^14: tiny, large and mixed operand magnitudes
    must come out at fp32-level accuracy relative to the output's scale -- an un-scaled fp16 remainder would lose the second piece of a
    1e-3-sized q entirely (error 1e-3 relative), an overflow would give inf."""
    N, L, C, heads = 2, 256, 384, 4
    Fd = C // heads
    g = torch.Generator().manual_seed(11)
    q, k, v = (torch.randn(N, L, heads, Fd, generator=g) * m for m in (qmag, kmag, vmag))
    q[0, 3] *= 7.0                                                     # rows of different magnitude inside one query tile
    q[1, 40] *= 1e-2
    qkv = torch.stack([q, k, v], dim=2).reshape(N, L, 3 * C).contiguous()
    out = torch.empty(N, L, C, device="cuda")
    buf = dev(qkv)
    _lib.check(_lib.lib().vd_op_attn_spatial(_lib.ptr(buf), N, L, C, heads, _lib.ptr(out), _lib.current_stream()))
    torch.cuda.synchronize()
    qd, kd, vd = (t.double().permute(0, 2, 1, 3) for t in (q, k, v))
    ref = (torch.softmax((qd * Fd ** -0.5) @ kd.transpose(-1, -2), -1) @ vd).permute(0, 2, 1, 3).reshape(N, L, C)
    ref32 = (torch.softmax((qd.float() * Fd ** -0.5) @ kd.float().transpose(-1, -2), -1) @ vd.float()).permute(0, 2, 1, 3).reshape(N, L, C)
    got = out.cpu().double()
    assert torch.isfinite(got).all()
    scale = ref.abs().max()
    e_got, e_f32 = (got - ref).abs().max() / scale, (ref32.double() - ref).abs().max() / scale
    assert e_got <= max(4.0 * e_f32, 2e-6), (e_got, e_f32)


def test_attention_spatial_peaked_scores():
    """Online-softmax rescale path: one key dominates late in the sequence (guide rule: force the rare branch)."""
    N, L, C, heads = 1, 128, 64, 4
    qkv = rnd(N, L, 3 * C)
    qkv[0, 100, C:2 * C] *= 40.0
    out = torch.empty(N, L, C, device="cuda")
    buf = dev(qkv)
    _lib.check(_lib.lib().vd_op_attn_spatial(_lib.ptr(buf), N, L, C, heads, _lib.ptr(out), _lib.current_stream()))
    torch.cuda.synchronize()
    Fd = C // heads
    q, k, v = qkv.view(N, L, 3, heads, Fd).permute(2, 0, 3, 1, 4)
    ref = (torch.softmax((q * Fd ** -0.5) @ k.transpose(-1, -2), -1) @ v).permute(0, 2, 1, 3).reshape(N, L, C)
    close(out.cpu(), ref, **TOL)


@pytest.mark.parametrize("B,T,HW,C,heads,rpe,mask,allow", [(2, 4, 64, 64, 4, True, False, 1), (1, 16, 16, 384, 4, True, True, 1),
                                                           (2, 20, 9, 96, 4, True, True, 0), (1, 5, 33, 32, 4, False, True, 1),
                                                           (1, 32, 4, 128, 4, True, False, 1), (1, 1, 7, 32, 2, True, False, 1),
                                                           # the matrix-pipe kernel (16-pixel blocks, head dim % 16 == 0): one / two frame tiles, ragged T
                                                           (2, 20, 32, 384, 4, True, True, 0), (1, 16, 64, 512, 4, True, True, 1),
                                                           (2, 7, 16, 128, 2, False, True, 1), (1, 32, 16, 256, 4, True, False, 1),
                                                           (1, 17, 48, 192, 4, True, True, 0), (1, 3, 16, 64, 4, True, True, 1),
                                                           (8, 16, 256, 384, 4, True, False, 1), (2, 20, 64, 512, 4, False, True, 1)])
def test_attention_temporal(B, T, HW, C, heads, rpe, mask, allow):
    """unet.py:486-536 + RPE einsums :357-378 + mask rule :511-524.  The launcher picks the kernel from the per-item shape
    alone: the 16-pixel matrix-pipe kernel where pixels % 16 == 0 and head dim % 16 == 0, 4-pixel VALU blocks otherwise; the
    last assert below holds it to that (a clip alone == the same clip inside a batch, bit for bit)."""
    qkv = rnd(B, T, HW, 3 * C) * 1.5
    Rk, Rq, Rv = (rnd(B, T, T, C, seed=s) for s in (1, 2, 3))
    m = None
    if mask:
        m = torch.ones(B, T)
        m[:, T // 2] = 0
        if T > 2:
            m[0, 0] = 0
    out = torch.empty(B, T, HW, C, device="cuda")
    bufs = [dev(qkv)] + [dev(r) if rpe else None for r in (Rk, Rq, Rv)] + [dev(m) if mask else None]
    _lib.check(_lib.lib().vd_op_attn_temporal(*[_lib.ptr(b) for b in bufs], B, T, HW, C, heads, allow, _lib.ptr(out),
                                              _lib.current_stream()))
    torch.cuda.synchronize()
    Fd = C // heads
    scale = Fd ** -0.5
    x = qkv.permute(0, 2, 1, 3).reshape(B, HW, T, 3, heads, Fd).permute(3, 0, 1, 4, 2, 5)     # t B D H T F
    q, k, v = x[0] * scale, x[1], x[2]
    w = q @ k.transpose(-1, -2)
    if rpe:
        rk, rq, rv = (r.view(B, T, T, heads, Fd) for r in (Rk, Rq, Rv))
        w = w + torch.einsum("bdhtf,btshf->bdhts", q, rk)
        w = w + torch.einsum("bdhtf,btshf->bdhts", k * scale, rq).transpose(-1, -2)
    if mask:
        ok = m.view(B, 1, T) * m.view(B, T, 1)
        if allow:
            ok = ok + (1 - m.view(B, 1, T)) * (1 - m.view(B, T, 1))
        else:
            ok = ok.clone()
            ok[:, range(T), range(T)] = 1.0
        w = w - torch.where(ok == 0, torch.tensor(float("inf")), torch.tensor(0.0)).view(B, 1, 1, T, T)
    a = torch.softmax(w, -1)
    o = a @ v
    if rpe:
        o = o + torch.einsum("bdhts,btshf->bdhtf", a, rv)
    ref = o.permute(0, 3, 1, 2, 4).reshape(B, T, HW, C)
    close(out.cpu(), ref, **TOL)
    if B > 1:                                                        # the last batch item on its own: the same bits
        one = torch.empty(1, T, HW, C, device="cuda")
        b1 = [bufs[0][B - 1:].contiguous()] + [None if r is None else r[B - 1:].contiguous() for r in bufs[1:]]
        _lib.check(_lib.lib().vd_op_attn_temporal(*[_lib.ptr(b) for b in b1], 1, T, HW, C, heads, allow, _lib.ptr(one), _lib.current_stream()))
        torch.cuda.synchronize()
        assert torch.equal(one[0], out[B - 1])


@pytest.mark.parametrize("N,H,C", [(2, 32, 32), (3, 16, 128), (1, 20, 64), (2, 4, 32)])
def test_output_head(N, H, C):
    """out = conv3x3(silu(GN(h))) to 3 channels, written NCHW (unet.py:744-749,838)."""
    x = rnd(N, C, H, H)
    A, B = rnd(N, C, seed=1) + 1.2, rnd(N, C, seed=2)
    w, b = rnd(3, C, 3, 3, scale=0.1), rnd(3, scale=0.1)
    out = torch.empty(N, 3, H, H, device="cuda")
    bufs = [dev(nhwc(x)), dev(A), dev(B), dev(pack_conv(w)), dev(b)]
    _lib.check(_lib.lib().vd_op_out_conv(*[_lib.ptr(t) for t in bufs], N, H, H, C, 3, _lib.ptr(out),
                                         _lib.current_stream()))
    torch.cuda.synchronize()
    ref = F.conv2d(F.silu(x * A[:, :, None, None] + B[:, :, None, None]), w, b, padding=1)
    close(out.cpu(), ref, **TOL)


def test_affine_apply():
    x, A, B = rnd(3, 10, 10, 64), rnd(3, 64, seed=1), rnd(3, 64, seed=2)
    y = torch.empty(3, 10, 10, 64, device="cuda")
    bufs = [dev(x), dev(A), dev(B)]
    _lib.check(_lib.lib().vd_op_affine_apply(*[_lib.ptr(t) for t in bufs], 3, 100, 64, _lib.ptr(y), _lib.current_stream()))
    torch.cuda.synchronize()
    close(y.cpu(), x * A[:, None, None, :] + B[:, None, None, :], atol=1e-6, rtol=1e-6)


def test_randn_moments_and_reproducibility():
    a = torch.empty(1 << 20, device="cuda")
    b = torch.empty(1 << 20, device="cuda")
    L = _lib.lib()
    _lib.check(L.vd_randn(_lib.ptr(a), a.numel(), 5, 0, _lib.current_stream()))
    _lib.check(L.vd_randn(_lib.ptr(b), b.numel(), 5, 0, _lib.current_stream()))
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    assert abs(a.mean().item()) < 5e-3 and abs(a.std().item() - 1) < 5e-3
    assert abs((a ** 4).mean().item() - 3) < 0.05 and torch.isfinite(a).all()
    _lib.check(L.vd_randn(_lib.ptr(b), b.numel(), 6, 0, _lib.current_stream()))
    torch.cuda.synchronize()
    assert not torch.equal(a, b)


MODE = {0: "f16x3", 1: "bf16x6", 2: "fp32"}


def math_mode():
    """The process' arithmetic (VD_MATH, include/vd_amd.h); with fp32 the split entry points run bf16x6."""
    return MODE[_lib.lib().vd_math_mode()]


def split_planes(w2d):
    """[N][K] fp32 -> ([3][N][K] int16 pieces, [N] scale) exactly as csrc/split_pack.hip forms them.
    f16x3: per-row power-of-two scale s with max|w s| in [2^13, 2^14); b0 = f16(w s), b1 = f16(w s - b0), 2^-12 b0.
    bf16x6: the exact three-way split (the sum of the pieces IS the weight)."""
    N = w2d.shape[0]
    if math_mode() == "f16x3":
        mx = w2d.abs().amax(1)
        _, ex = torch.frexp(mx)
        s = torch.where(mx > 0, torch.ldexp(torch.ones(N), 14 - ex), torch.ones(N))
        ws = w2d * s[:, None]
        b0 = ws.half()
        b1 = (ws - b0.float()).half()
        b2 = (b0.float() / 4096).half()
        rec = b0.double() + b1.double()
        assert ((rec - ws.double()).abs() <= ws.double().abs() * 2.0 ** -22 + 2.0 ** -25).all()
        return torch.stack([b0, b1, b2]).view(torch.int16), s
    p1 = w2d.bfloat16()
    r1 = w2d - p1.float()
    p2 = r1.bfloat16()
    p3 = (r1 - p2.float()).bfloat16()
    assert torch.equal(p1.float() + p2.float() + p3.float(), w2d)            # the split is exact
    return torch.stack([p1, p2, p3]).view(torch.int16), torch.ones(N)


def with_trailer(img, s):
    tr = torch.cat([s.float(), 1.0 / s.float()]).view(torch.int16)
    return torch.cat([img, tr])


def pack_lin_split(w):
    """[N][K] -> three 16-bit planes in MFMA fragment order [K/16][N/32][3][64][8] + the trailer of per-row scales
    (library packer + closed form)."""
    N, K = w.shape
    src = w.contiguous().float()
    out = torch.empty(_lib.lib().vd_split_image_u16(N, K), dtype=torch.int16)
    assert out.numel() == 3 * N * K + 4 * N
    _lib.check(_lib.lib().vd_pack_linear_split(_lib.ptr(src), _lib.ptr(out), N, K))
    planes, sc = split_planes(src)                                           # [3][N][K]
    ref = planes.reshape(3, N // 32, 32, K // 16, 2, 8).permute(3, 1, 0, 4, 2, 5).reshape(-1)
    assert torch.equal(out, with_trailer(ref, sc))
    return out


# Error against an fp64 product, relative to the fp32-MFMA kernel's on the same inputs (all kernels accumulate in fp32): the claim
# the bench line and INTEGRATION.md make for the default arithmetic is exactly this bound.  bf16x6 multiplies exact operands,
# f16x3 carries them to 22 bits and drops a1*b1; measured on MI355X (profiles/r04_split_accuracy.json, K = 32 .. 4608, operand
# magnitudes 1e-3 .. 100) BOTH are at least as close to fp64 as the fp32 MFMA: f16x3 max <= 1.06x, mean <= 1.02x; bf16x6 max <=
# 1.09x, mean <= 0.88x (the rounding of the fp32 accumulator dominates, and a 16-deep MFMA rounds K / 16 times, not K / 2).
def err_bounds():
    return (1.5, 1.5) if math_mode() != "f16x3" else (1.5, 1.25)


def check_vs_fp32_kernel(e_split, e_fp32):
    fmax, fmean = err_bounds()
    assert e_split.max() <= fmax * e_fp32.max() + 1e-7, (e_split.max(), e_fp32.max())
    assert e_split.mean() <= fmean * e_fp32.mean() + 1e-8, (e_split.mean(), e_fp32.mean())


@pytest.mark.parametrize("M,K,N,act,res", [(128, 128, 512, 0, 0), (4099, 96, 288, 0, 1), (300, 1024, 64, 1, 0),
                                           (2048, 384, 384, 0, 1), (70, 64, 128, 1, 1),
                                           # many-tile shapes (128x128 / 64x128 tile classes, ragged M)
                                           (8192, 64, 1536, 0, 1), (8135, 96, 1536, 0, 0), (8192, 128, 1536, 1, 0),
                                           (32768, 64, 512, 0, 1), (65536, 32, 320, 0, 0),
                                           # the 128x192 tile (N % 192 == 0, >= 384 of them): 2-slot weight ring, ragged M
                                           (32768, 96, 384, 0, 1), (32700, 64, 1152, 1, 0), (8192, 160, 1536, 0, 0)])
def test_linear_split_is_fp32_accurate(M, K, N, act, res):
    """csrc/gemm_split.hip in the process' arithmetic (f16x3: two fp16 pieces, three piece products; bf16x6: the exact split,
    six), fp32 accumulation.  Held to the op tolerance against torch fp32 AND, against an fp64 product, to the fp32-MFMA
    kernel's error on the same inputs (gemm_frag.hip; check_vs_fp32_kernel states the bound per mode)."""
    L = _lib.lib()
    a, w, b = rnd(M, K), rnd(N, K, scale=(3.0 / K) ** 0.5), rnd(N, scale=0.1)
    r = rnd(M, N, seed=4) if res else None
    ad, bd, rd = dev(a), dev(b), (dev(r) if res else None)
    out_s, out_f = torch.empty(M, N, device="cuda"), torch.empty(M, N, device="cuda")
    ws = dev(pack_lin_split(w))
    _lib.check(L.vd_op_linear_split(_lib.ptr(ad), M, K, _lib.ptr(ws), _lib.ptr(bd), _lib.ptr(rd), act, _lib.ptr(out_s), N,
                                    _lib.current_stream()))
    wf = dev(pack_lin_frag(w))
    _lib.check(L.vd_op_conv(_lib.ptr(ad), None, K, K, M, 1, 1, 0, 1, 0, 1, None, _lib.ptr(wf), None, _lib.ptr(bd), None, None, act,
                            _lib.ptr(rd), None, 0, _lib.ptr(out_f), N, _lib.current_stream()))
    torch.cuda.synchronize()
    x = F.silu(a) if act else a
    ref32 = x @ w.t() + b + (r if res else 0)
    close(out_s.cpu(), ref32, **TOL)
    # the GPU applies its own SiLU (fast exp / rcp): compare the matrix products on the activation the GPU used
    xg = x.double()
    ref64 = xg @ w.double().t() + b.double() + (r.double() if res else 0)
    e_split = (out_s.cpu().double() - ref64).abs()
    e_fp32 = (out_f.cpu().double() - ref64).abs()
    if not act:
        check_vs_fp32_kernel(e_split, e_fp32)


@pytest.mark.parametrize("nfr,HW,K,N,res", [(128, 256, 96, 384, 1), (128, 64, 64, 512, 1), (6, 64, 64, 96, 0), (3, 1024, 32, 128, 0),
                                            (129, 64, 32, 384, 1)])
def test_linear_split_groupnorm_partial_sums(nfr, HW, K, N, res):
    """csrc/gemm_split.hip epilogue: per-(frame, row block, channel) [sum, sum of squares] of the output, the table the next
    GroupNorm's fold reads instead of a statistics pass (proj_out + residual of an attention block followed by the next
    normalization: unet.py:537-538, nn.py:15-17).  All tile classes (128x192, 128x128, 64x128, 64x64), an odd frame count
    whose last 128-row tile is half empty; checked against sums of the stored output in fp64."""
    L = _lib.lib()
    M = nfr * HW
    a, w, b = rnd(M, K), rnd(N, K, scale=(3.0 / K) ** 0.5), rnd(N, scale=0.1)
    r = rnd(M, N, seed=4) if res else None
    ad, bd, rd = dev(a), dev(b), (dev(r) if res else None)
    out = torch.empty(M, N, device="cuda")
    split = L.vd_linear_stats_split(M, N, HW)
    part = torch.full((nfr, split, N, 2), float("nan"), dtype=torch.float64, device="cuda")
    ws = dev(pack_lin_split(w))
    _lib.check(L.vd_op_linear_split_stats(_lib.ptr(ad), M, K, _lib.ptr(ws), _lib.ptr(bd), _lib.ptr(rd), 0, _lib.ptr(out), N, HW,
                                          _lib.ptr(part), _lib.current_stream()))
    torch.cuda.synchronize()
    close(out.cpu(), a @ w.t() + b + (r if res else 0), **TOL)
    o64 = out.double().reshape(nfr, HW, N)
    tot = part.sum(1).cpu()
    close(tot[..., 0], o64.sum(1).cpu(), atol=1e-3, rtol=1e-5)
    close(tot[..., 1], (o64 * o64).sum(1).cpu(), atol=1e-3, rtol=1e-5)
    rows = HW // split                                                # every block of rows on its own
    blk = o64.reshape(nfr, split, rows, N)
    close(part[..., 0].cpu(), blk.sum(2).cpu(), atol=1e-3, rtol=1e-5)


def pack_wino_split(w):
    """OIHW -> U = G g G^T (fp64, rounded once to fp32, row 3 negated) -> [I/16][16][O/32][3][64][8] + trailer (library
    packer + closed form: the per-cout scale runs over all of the cout's (cin, position) entries)."""
    O, I = w.shape[:2]
    out = torch.empty(_lib.lib().vd_split_image_u16(O, 16 * I), dtype=torch.int16)
    _lib.check(_lib.lib().vd_pack_conv3_wino_split(_lib.ptr(w.contiguous().float()), _lib.ptr(out), O, I))
    G = torch.tensor([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], dtype=torch.float64)
    U = torch.einsum("ik,ockl,jl->ocij", G, w.double(), G)                    # [co][ci][i][j]
    U[:, :, 3] = -U[:, :, 3]
    planes, sc = split_planes(U.float().reshape(O, I * 16))                  # [3][O][(ci, xi)]
    ref = planes.reshape(3, O // 32, 32, I // 16, 2, 8, 16).permute(3, 6, 1, 0, 4, 2, 5).reshape(-1)
    assert torch.equal(out, with_trailer(ref, sc))
    return out


@pytest.mark.parametrize("N,Cin,Cout,H,ups", [(3, 64, 128, 16, 0), (5, 32, 64, 8, 0), (2, 96, 160, 32, 0), (9, 64, 32, 8, 0),
                                               (1, 128, 64, 64, 0), (2, 64, 64, 8, 1), (6, 160, 192, 8, 0), (41, 32, 128, 32, 0),
                                               (37, 64, 256, 16, 0), (1100, 32, 64, 8, 0), (8, 64, 128, 32, 0), (64, 32, 128, 32, 0),
                                               (3, 224, 64, 16, 1), (300, 64, 64, 16, 0), (16, 128, 192, 64, 0), (2, 96, 384, 16, 0)])
def test_conv3x3_winograd_split_is_fp32_accurate(N, Cin, Cout, H, ups):
    _check_wino_split(N, Cin, Cout, H, ups)


@pytest.mark.parametrize("N,Cin,Cout,H", [(41, 32, 128, 32), (64, 32, 128, 32), (16, 160, 128, 64), (128, 64, 256, 16), (43, 96, 128, 32),
                                          (64, 64, 512, 16), (16, 320, 128, 64)])
def test_conv3x3_winograd_128_cout_blocks(N, Cin, Cout, H):
    """csrc/conv_wino_z128.hip: the same convolution with the column half of the output transform accumulated by the matrix
    pipe (six (position, column) groups of MFMAs instead of four, 128 couts per block, the third weight piece formed in
    registers).  The library picks it by grid fill (vd_conv_wino_block_couts); these shapes are ones it picks it for under
    the default arithmetic -- one and several trips of the chunk-pair loop, one to four cout blocks, tile-group counts that are and
    are not a multiple of 8 (both item orders).  Same bounds as the 64-cout kernel."""
    L = _lib.lib()
    if L.vd_math_mode() == 0:
        assert L.vd_conv_wino_block_couts(N, H, Cin, Cout) == 128
    _check_wino_split(N, Cin, Cout, H, 0)


def _check_wino_split(N, Cin, Cout, H, ups):
    """csrc/conv_wino_r64.hip (maps >= 8x8; 8x8 maps go four frames to an item, incl. frame counts that are not a multiple of
    four): Winograd F(2x2,3x3) with the element products as piece products of the split fp32 operands (the process'
    arithmetic).  Held to the op tolerance against torch fp32, to the fp32-MFMA Winograd kernel's error against an fp64 conv
    (check_vs_fp32_kernel), and the GroupNorm partial sums checked against the stored output.  Several shapes have more work
    items than the GPU has CUs and take the cout-inner item order (tile blocks a multiple of 8, more than one cout block); the
    cases cover 2..14 channel chunks, the x2 upsampled source and all three tile grids."""
    if Cout % 64:
        pytest.skip("the split Winograd kernel owns 64 couts per block (the engine sends other widths to the generic kernel)")
    L = _lib.lib()
    op = L.vd_op_conv_wino_split
    x, w, b = rnd(N, Cin, H, H), rnd(Cout, Cin, 3, 3, scale=(3.0 / (9 * Cin)) ** 0.5), rnd(Cout, scale=0.1)
    Ho = H << ups
    res = rnd(N, Cout, Ho, Ho, seed=4)
    fb = rnd(N, Cout, seed=5)
    xd, bd, rd, fd = dev(nhwc(x)), dev(b), dev(nhwc(res)), dev(fb)
    out_s = torch.empty(N, Ho, Ho, Cout, device="cuda")
    split = L.vd_conv_stats_split(Ho)
    part = torch.full((N, split, Cout, 2), float("nan"), dtype=torch.float64, device="cuda")
    ws = dev(pack_wino_split(w))
    _lib.check(op(_lib.ptr(xd), Cin, N, H, H, ups, _lib.ptr(ws), _lib.ptr(bd), _lib.ptr(rd), _lib.ptr(fd), Cout,
                  _lib.ptr(out_s), Cout, _lib.ptr(part), _lib.current_stream()))
    torch.cuda.synchronize()
    xin = F.interpolate(x, scale_factor=2, mode="nearest") if ups else x
    ref = F.conv2d(xin, w, b, padding=1) + res + fb[:, :, None, None]
    got = out_s.permute(0, 3, 1, 2).cpu()
    close(got, ref, **TOL)
    ref64 = F.conv2d(xin.double(), w.double(), b.double(), padding=1) + res.double() + fb.double()[:, :, None, None]
    e_split = (got.double() - ref64).abs()
    if Cout % 64 == 0:                                              # the fp32-MFMA Winograd kernel on the same inputs
        out_f = run_conv(x, None, w, b, ups=ups, res=res, fbias=fb)
        e_fp32 = (out_f.double() - ref64).abs()
        check_vs_fp32_kernel(e_split, e_fp32)
    o64 = out_s.double()
    tot = part.sum(1).cpu()
    close(tot[..., 0], o64.sum((1, 2)).cpu(), atol=1e-3, rtol=1e-5)
    close(tot[..., 1], (o64 * o64).sum((1, 2)).cpu(), atol=1e-3, rtol=1e-5)


@pytest.mark.parametrize("N,Cin,Cout,H", [(2, 64, 64, 8), (5, 32, 128, 8), (3, 96, 64, 16), (2, 64, 192, 32), (9, 160, 64, 8),
                                          (1, 32, 64, 64), (128, 64, 64, 16), (3, 512, 128, 16)])
def test_upsample_conv_sub_pixel_form_is_fp32_accurate(N, Cin, Cout, H):
    """csrc/conv_wino_r64.hip, sub-pixel form of Upsample + conv3x3 (unet.py:70-77): four phase kernels over the SOURCE map, one
    Winograd column of each structurally zero and skipped.  Against torch's F.interpolate(nearest) + conv2d at the op
    tolerance; against an fp64 reference no further away than the same layer on the upsampled map (the kernel it replaces);
    GroupNorm partial sums (four table entries per tile group) against the stored output.  Source maps 8 (four frames per
    item, frame counts that are not a multiple of four), 16, 32, 64; 2..32 channel chunks."""
    L = _lib.lib()
    x, w, b = rnd(N, Cin, H, H), rnd(Cout, Cin, 3, 3, scale=(3.0 / (9 * Cin)) ** 0.5), rnd(Cout, scale=0.1)
    xd, bd = dev(nhwc(x)), dev(b)
    Ho = 2 * H
    out_s = torch.full((N, Ho, Ho, Cout), float("nan"), device="cuda")
    split = L.vd_conv_ups_stats_split(H)
    part = torch.full((N, split, Cout, 2), float("nan"), dtype=torch.float64, device="cuda")
    wu = torch.empty(L.vd_split_image_u16(4 * Cout, 16 * Cin), dtype=torch.int16)
    _lib.check(L.vd_pack_conv3_wino_ups(_lib.ptr(w.contiguous().float()), _lib.ptr(wu), Cout, Cin))
    wud = dev(wu)
    _lib.check(L.vd_op_conv_wino_ups(_lib.ptr(xd), Cin, N, H, _lib.ptr(wud), _lib.ptr(bd), _lib.ptr(out_s), Cout, _lib.ptr(part),
                                     _lib.current_stream()))
    torch.cuda.synchronize()
    xin = F.interpolate(x, scale_factor=2, mode="nearest")
    ref = F.conv2d(xin, w, b, padding=1)
    got = out_s.permute(0, 3, 1, 2).cpu()
    close(got, ref, **TOL)
    ref64 = F.conv2d(xin.double(), w.double(), b.double(), padding=1)
    e_new = (got.double() - ref64).abs()
    out_o = torch.empty(N, Ho, Ho, Cout, device="cuda")               # the same layer as F(2x2,3x3) on the upsampled map
    ws = dev(pack_wino_split(w))
    _lib.check(L.vd_op_conv_wino_split(_lib.ptr(xd), Cin, N, H, H, 1, _lib.ptr(ws), _lib.ptr(bd), None, None, 0, _lib.ptr(out_o), Cout,
                                       None, _lib.current_stream()))
    torch.cuda.synchronize()
    e_old = (out_o.permute(0, 3, 1, 2).cpu().double() - ref64).abs()
    assert e_new.max() <= 1.5 * e_old.max() + 1e-7, (e_new.max(), e_old.max())
    assert e_new.mean() <= 1.5 * e_old.mean() + 1e-8, (e_new.mean(), e_old.mean())
    o64 = out_s.double()
    tot = part.sum(1).cpu()
    close(tot[..., 0], o64.sum((1, 2)).cpu(), atol=1e-3, rtol=1e-5)
    close(tot[..., 1], (o64 * o64).sum((1, 2)).cpu(), atol=1e-3, rtol=1e-5)


@pytest.mark.parametrize("H,Cin,Cout,kind", [(8, 128, 128, "ups"), (8, 128, 128, "s1"), (16, 256, 128, "s1"), (16, 128, 128, "ups"),
                                             (32, 64, 64, "s1"), (32, 32, 128, "s1")])
def test_conv_frame_subset_and_order_invariance(H, Cin, Cout, kind):
    """A frame's output AND its GroupNorm partial sums must not depend on which frames share its launch, nor on its place among
    them: the window suffix skip and the prefix cache run layers on gathered frame subsets and promise the full launch's bits.
    (Round 4 found the 8x8 kernel's statistics depending on the frame's place in a four-frame item: hipcc had contracted
    `ss += y * y` into an fma in one unrolled copy of the loop and not in the other.)"""
    L = _lib.lib()
    N = 12
    g = torch.Generator().manual_seed(3)
    x = dev(torch.rand(N, H, H, Cin, generator=g) - 0.5)
    w = (torch.rand(Cout, Cin, 3, 3, generator=g) - 0.5) * (3.0 / (9 * Cin)) ** 0.5
    b = dev(torch.rand(Cout, generator=g))
    Ho = 2 * H if kind == "ups" else H
    if kind == "ups":
        wp = torch.empty(L.vd_split_image_u16(4 * Cout, 16 * Cin), dtype=torch.int16)
        _lib.check(L.vd_pack_conv3_wino_ups(_lib.ptr(w.contiguous()), _lib.ptr(wp), Cout, Cin))
        split = L.vd_conv_ups_stats_split(H)
    else:
        wp, split = pack_wino_split(w), L.vd_conv_stats_split(H)
    wd = dev(wp)

    def run(xx):
        n = xx.shape[0]
        out = torch.empty(n, Ho, Ho, Cout, device="cuda")
        part = torch.zeros(n, split, Cout, 2, dtype=torch.float64, device="cuda")
        if kind == "ups":
            _lib.check(L.vd_op_conv_wino_ups(_lib.ptr(xx), Cin, n, H, _lib.ptr(wd), _lib.ptr(b), _lib.ptr(out), Cout, _lib.ptr(part),
                                             _lib.current_stream()))
        else:
            _lib.check(L.vd_op_conv_wino_split(_lib.ptr(xx), Cin, n, H, H, 0, _lib.ptr(wd), _lib.ptr(b), None, None, 0, _lib.ptr(out), Cout,
                                               _lib.ptr(part), _lib.current_stream()))
        torch.cuda.synchronize()
        return out, part

    of, pf = run(x)
    for sel in ([2, 3, 4, 5, 8, 9, 10, 11], [11, 2, 7, 0, 5], [6], [3, 2, 1, 0, 7, 6, 5, 4, 11, 10, 9]):
        oc, pc = run(x[sel].contiguous())
        assert torch.equal(oc, of[sel]), (sel, (oc - of[sel]).abs().max())
        assert torch.equal(pc, pf[sel]), (sel, (pc - pf[sel]).abs().max())


def pack_conv_split(w):
    O, I = w.shape[:2]
    out = torch.empty(_lib.lib().vd_split_image_u16(O, 9 * I), dtype=torch.int16)
    _lib.check(_lib.lib().vd_pack_conv3_split(_lib.ptr(w.contiguous().float()), _lib.ptr(out), O, I))
    return out


@pytest.mark.parametrize("N,Cin,Cout,H,stride,res", [(3, 64, 64, 16, 2, 0), (2, 128, 128, 64, 2, 0), (5, 32, 96, 8, 2, 1),
                                                     (7, 96, 32, 4, 1, 1), (7, 96, 32, 4, 1, 0), (1, 64, 160, 32, 2, 0),
                                                     (3, 32, 32, 2, 2, 0)])
def test_conv3x3_split_gemm_is_fp32_accurate(N, Cin, Cout, H, stride, res):
    """csrc/gemm_split.hip, CONV mode: the stride-2 Downsample conv (unet.py:98) as the split GEMM over an implicit
    im2col operand.  Held to the op tolerance against torch fp32 and, against an fp64 conv, required to be no further
    away than the generic fp32-MFMA kernel on the same inputs."""
    L = _lib.lib()
    x, w, b = rnd(N, Cin, H, H), rnd(Cout, Cin, 3, 3, scale=(3.0 / (9 * Cin)) ** 0.5), rnd(Cout, scale=0.1)
    Ho = (H - 1) // stride + 1
    r = rnd(N, Cout, Ho, Ho, seed=4) if res else None
    xd, bd, rd = dev(nhwc(x)), dev(b), (dev(nhwc(r)) if res else None)
    out_s = torch.empty(N, Ho, Ho, Cout, device="cuda")
    ws = dev(pack_conv_split(w))
    _lib.check(L.vd_op_conv_split(_lib.ptr(xd), Cin, N, H, H, stride, _lib.ptr(ws), _lib.ptr(bd), _lib.ptr(rd), _lib.ptr(out_s),
                                  Cout, _lib.current_stream()))
    torch.cuda.synchronize()
    ref = F.conv2d(x, w, b, stride=stride, padding=1) + (r if res else 0)
    got = out_s.permute(0, 3, 1, 2).cpu()
    close(got, ref, **TOL)
    if res:          # the split kernel starts its accumulator at bias + residual, the generic one adds them last: the
        return       # rounding of the two differs by the residual's magnitude, so the fp64 comparison is made without
    ref64 = F.conv2d(x.double(), w.double(), b.double(), stride=stride, padding=1)
    out_f = run_conv(x, None, w, b, stride=stride, generic=True)
    e_split, e_fp32 = (got.double() - ref64).abs(), (out_f.double() - ref64).abs()
    check_vs_fp32_kernel(e_split, e_fp32)


@pytest.mark.parametrize("mode", ["f16x3", "bf16x6", "fp32"])
def test_every_arithmetic_mode_end_to_end(mode):
    """Every VD_MATH mode the library ships is held to the SAME bar as the default one: the mode is read once per process, so
    tools/mode_check.py runs as a child process per mode -- a linear layer and a 3x3 conv against fp64 next to the fp32-MFMA
    kernel, and the whole network against the reference's goldens (tiny config: 10 eps cases + p_sample at four timesteps;
    the default 116 M model) at the tier's tolerance 1e-4 + 1e-4 |ref|.  The process' own mode is covered by the rest of
    this suite and skipped here."""
    import json
    import subprocess
    import sys
    if mode == math_mode():
        pytest.skip("the test process' own mode: the whole suite runs in it")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "mode_check.py")], env={**os.environ, "VD_MATH": mode},
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    assert rep["mode"] == mode and mode in rep["version"]
    for net in ("eps_tiny", "psample_tiny", "eps_full64"):
        assert rep[net]["outside_tol"] == 0, (net, rep[net])
    fmax, fmean = (1.5, 1.25) if mode == "f16x3" else (1.5, 1.5)
    for op in ("linear", "conv"):
        k = rep[op]["fp32_kernel"]
        assert rep[op]["max_err"] <= fmax * k["max_err"] + 1e-7 and rep[op]["mean_err"] <= fmean * k["mean_err"] + 1e-8, rep[op]
        # Signed mean error.  The operand split rounds to nearest in both modes (rounds 1-3 truncated in the conv kernel: a bias of the
        # sign of a*b).  What is left is a small NEGATIVE offset of the 16-bit MFMA's own fp32 accumulation -- the same in f16x3 and
        # bf16x6, absent from the fp32 MFMA, growing with K: -2 .. -6 % of the mean |error| at these sizes, -31 % at K = 4608
        # (profiles/r04_split_accuracy.json) -- i.e. 1e-8 .. 1e-7 of the output's RMS.  Bounded here so that a regression of the
        # split (a truncating piece would show +-30 % at K = 512) cannot hide behind it.
        assert abs(rep[op]["signed_mean_err"]) <= 0.2 * rep[op]["mean_err"] + 1e-9, ("systematic bias", rep[op])
        assert abs(k["signed_mean_err"]) <= 0.05 * k["mean_err"] + 1e-9, ("fp32 kernel bias", k)


@pytest.mark.parametrize("N,Cin,Cout,H,C1", [(41, 32, 128, 32, 0), (64, 64, 128, 32, 0), (16, 128, 128, 64, 0), (128, 64, 256, 16, 0), (43, 96, 128, 32, 0),
                                             (16, 320, 128, 64, 0), (32, 256, 256, 32, 0), (17, 160, 128, 64, 0),
                                             (16, 256, 128, 64, 128), (24, 128, 256, 32, 32), (16, 192, 128, 64, 160), (33, 64, 128, 32, 32)])
def test_conv3x3_winograd_with_the_activation_in_its_patch_staging(N, Cin, Cout, H, C1):
    """csrc/conv_wino_z128.hip, ACT form (r05): out = conv3x3(silu(x * A[n][c] + B[n][c])) + bias + res with the GroupNorm(+FiLM) affine and
    the SiLU applied while the kernel stages its patch -- the ResBlock's `GroupNorm -> SiLU -> conv` (unet.py:138-141,185-198) without the
    activation image; C1 > 0: over the virtual concat of two tensors (the decoder's th.cat([h, hs.pop()], 1), unet.py:826-828; source widths
    that are and are not a multiple of 32).  Against torch fp32 at the op tolerance; against the materialising form (vd_op_affine_act +
    vd_op_conv_wino_split, the same arithmetic element for element: the zero padding must be zeros of the ACTIVATED tensor, silu(B) would be
    wrong) to the bit; GroupNorm partial sums of the output as in the plain kernel; odd chunk-pair counts, one and two cout blocks, both item orders."""
    L = _lib.lib()
    if math_mode() != "f16x3" or not L.vd_conv_wino_act_ok(N, H, Cin, Cout):
        pytest.skip("the activating form serves the f16x3 arithmetic on conv_wino_z128.hip's shapes")
    C0 = Cin - C1
    x, w, b = rnd(N, Cin, H, H, scale=2.0), rnd(Cout, Cin, 3, 3, scale=(3.0 / (9 * Cin)) ** 0.5), rnd(Cout, scale=0.1)
    A, B = rnd(N, Cin, seed=2) + 1.3, rnd(N, Cin, seed=3) * 0.7
    res = rnd(N, Cout, H, H, seed=4)
    x0d, x1d = dev(nhwc(x[:, :C0])), (dev(nhwc(x[:, C0:])) if C1 else None)
    bd, rd, Ad, Bd = dev(b), dev(nhwc(res)), dev(A), dev(B)
    ws = dev(pack_wino_split(w))
    split = L.vd_conv_stats_split(H)
    out_a, out_m = torch.empty(N, H, H, Cout, device="cuda"), torch.empty(N, H, H, Cout, device="cuda")
    part_a = torch.full((N, split, Cout, 2), float("nan"), dtype=torch.float64, device="cuda")
    part_m = torch.full((N, split, Cout, 2), float("nan"), dtype=torch.float64, device="cuda")
    _lib.check(L.vd_op_conv_wino_act(_lib.ptr(x0d), _lib.ptr(x1d), C0, Cin, N, H, H, _lib.ptr(ws), _lib.ptr(bd), _lib.ptr(Ad), _lib.ptr(Bd), _lib.ptr(rd),
                                     _lib.ptr(out_a), Cout, _lib.ptr(part_a), _lib.current_stream()))
    act = torch.empty(N, H, H, Cin, device="cuda")
    _lib.check(L.vd_op_affine_act(_lib.ptr(x0d), _lib.ptr(x1d), C0, Cin, _lib.ptr(Ad), _lib.ptr(Bd), N, H * H, 1, _lib.ptr(act), _lib.current_stream()))
    _lib.check(L.vd_op_conv_wino_split(_lib.ptr(act), Cin, N, H, H, 0, _lib.ptr(ws), _lib.ptr(bd), _lib.ptr(rd), None, 0, _lib.ptr(out_m), Cout,
                                       _lib.ptr(part_m), _lib.current_stream()))
    torch.cuda.synchronize()
    ref = F.conv2d(F.silu(x * A[:, :, None, None] + B[:, :, None, None]), w, b, padding=1) + res
    close(out_a.permute(0, 3, 1, 2).cpu(), ref, **TOL)
    assert torch.equal(out_a, out_m), float((out_a - out_m).abs().max())
    assert torch.equal(part_a, part_m)
