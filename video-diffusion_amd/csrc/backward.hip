// Backward-data kernels of the video UNet for `use_gradient_method` (gaussian_diffusion.py:264-271,350-364): the gradient
// of an observation mismatch w.r.t. the network INPUT x_t -- no weight gradients.  The matrix products of the backward
// pass (3x3 convs on 180-degree-rotated transposed weights, linear layers on transposed weights) run on the forward
// kernels over a second packed weight image; this file holds what has no forward counterpart:
//
//   gn_bwd_*            GroupNorm32 (+FiLM, +SiLU) backward over a virtual concat        (nn.py:15-17, unet.py:185-198)
//   gn_temporal_bwd     temporal GroupNorm on the (B*HW, C, T) view                       (unet.py:472-475)
//   attn_temporal_bwd   softmax(q'k + q'Rk + k'Rq) (v + Rv) backward                      (unet.py:486-536, 357-378)
//   attn_sp_bwd_*       spatial softmax(q'k) v backward (dq pass, dk/dv pass)             (unet.py:486-536 without RPE)
//   zero_stuff2 / sumpool2 / add / out_conv_bwd / stem_col2im                            (unet.py:98, 69, 744-749, 951-983)
//   guided_grad / guided_final                                                            (gaussian_diffusion.py:350-364)
//
// Off the benchmarked path (one extra forward + this backward per guided step); written for clarity and coalesced access,
// fp32 with fp64 reductions where the forward uses them.
#include <algorithm>
#include <cmath>

#include "vd_common.h"

namespace vd {

__device__ __forceinline__ float sigmoid_f(float v) { return __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }
// d/dy [y * sigmoid(y)]
__device__ __forceinline__ float silu_grad_f(float y) { const float s = sigmoid_f(y); return s * (1.0f + y * (1.0f - s)); }

// ------------------------------------------------------------------------------------------------ GroupNorm backward
// Forward: y = x*A[n][c] + B[n][c] with A = rstd*gamma' (gamma' = gamma*(1+scale)), a = act(y).  Given da:
//   g = da * act'(y);  T1 = sum_group g*A;  T2 = sum_group g*A*(x - mean)
//   dx = g*A - T1/cnt - (x - mean) * rstd^2 * T2/cnt        (cnt = HW * C/32)
// Pass 1 (gn_bwd_partial): per (frame, pixel range, channel) partial [T1, T2] in fp64.  Pass 2 (gn_bwd_fold): per (frame,
// group) K1 = -rstd^2*T2/cnt, K0 = -T1/cnt - mean*K1.  Pass 3 (gn_bwd_apply): dx = g*A + x*K1 + K0 (+ extra), written to the
// two sources of the virtual concat, each assigned or accumulated.
__global__ __launch_bounds__(256) void gn_bwd_partial_kernel(GnBwdArgs a, int split, int per) {
    const int n = blockIdx.y, sp = blockIdx.x;
    const int C = a.C, tpp = C >> 2, ppi = 256 / tpp, cg = C / 32;
    const int tid = threadIdx.x, pl = tid / tpp, q = tid - pl * tpp, c = q * 4;
    const int p_begin = sp * per, p_end = min(a.HW, p_begin + per);
    double t1[4] = {0, 0, 0, 0}, t2[4] = {0, 0, 0, 0};
    if (pl < ppi) {
        const float* src; int ld;
        if (c < a.C0) { src = a.x0 + (size_t)n * a.HW * a.C0 + c; ld = a.C0; } else { src = a.x1 + (size_t)n * a.HW * (C - a.C0) + (c - a.C0); ld = C - a.C0; }
        const f32x4 A = *reinterpret_cast<const f32x4*>(a.A + (size_t)n * C + c), B = *reinterpret_cast<const f32x4*>(a.B + (size_t)n * C + c);
        float mu[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) mu[e] = a.mr[((size_t)n * 32 + (c + e) / cg) * 2];
        for (int p = p_begin + pl; p < p_end; p += ppi) {
            const f32x4 x = *reinterpret_cast<const f32x4*>(src + (size_t)p * ld);
            const f32x4 d = *reinterpret_cast<const f32x4*>(a.dy + ((size_t)n * a.HW + p) * C + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float y = x[e] * A[e] + B[e];
                const float g = (a.act ? d[e] * silu_grad_f(y) : d[e]) * A[e];
                t1[e] += (double)g; t2[e] += (double)g * (double)(x[e] - mu[e]);
            }
        }
    }
    __shared__ double red[256 * 8];
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[tid * 8 + e] = t1[e]; red[tid * 8 + 4 + e] = t2[e]; }
    __syncthreads();
    if (tid < tpp) {
        double s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int k = 0; k < ppi; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e) s[e] += red[(k * tpp + tid) * 8 + e];
        double* o = a.part + (((size_t)n * split + sp) * C + tid * 4) * 2;
#pragma unroll
        for (int e = 0; e < 4; ++e) { o[e * 2] = s[e]; o[e * 2 + 1] = s[4 + e]; }
    }
}

__global__ __launch_bounds__(256) void gn_bwd_fold_kernel(const double* __restrict__ part, int split, int C, double cnt,
                                                          const float* __restrict__ mr, float* __restrict__ K) {
    const int n = blockIdx.x, g = threadIdx.x >> 3, l = threadIdx.x & 7, cg = C / 32;
    double t1 = 0, t2 = 0;
    for (int ci = l; ci < cg; ci += 8)
        for (int sp = 0; sp < split; ++sp) {
            const double* p = part + (((size_t)n * split + sp) * C + g * cg + ci) * 2;
            t1 += p[0]; t2 += p[1];
        }
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) { t1 += __shfl_xor(t1, o, 8); t2 += __shfl_xor(t2, o, 8); }
    if (l == 0) {
        const double mean = mr[((size_t)n * 32 + g) * 2], rstd = mr[((size_t)n * 32 + g) * 2 + 1];
        const double k1 = -rstd * rstd * t2 / cnt;
        K[((size_t)n * 32 + g) * 2] = (float)(-t1 / cnt - mean * k1);
        K[((size_t)n * 32 + g) * 2 + 1] = (float)k1;
    }
}

__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(GnBwdArgs a, int per) {
    const int n = blockIdx.y;
    const int C = a.C, tpp = C >> 2, ppi = blockDim.x / tpp, cg = C / 32;
    const int tid = threadIdx.x, pl = tid / tpp, c = (tid - pl * tpp) * 4;
    const int p_begin = blockIdx.x * per, p_end = min(a.HW, p_begin + per);
    const bool first = c < a.C0;
    const int ld = first ? a.C0 : C - a.C0, cc = first ? c : c - a.C0;
    const float* src = (first ? a.x0 : a.x1) + (size_t)n * a.HW * ld + cc;
    float* dst = (first ? a.dx0 : a.dx1) + (size_t)n * a.HW * ld + cc;
    const int acc = first ? a.acc0 : a.acc1;
    const f32x4 A = *reinterpret_cast<const f32x4*>(a.A + (size_t)n * C + c), B = *reinterpret_cast<const f32x4*>(a.B + (size_t)n * C + c);
    float k0[4], k1[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { k0[e] = a.K[((size_t)n * 32 + (c + e) / cg) * 2]; k1[e] = a.K[((size_t)n * 32 + (c + e) / cg) * 2 + 1]; }
    for (int p = p_begin + pl; p < p_end; p += ppi) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(src + (size_t)p * ld);
        const f32x4 d = *reinterpret_cast<const f32x4*>(a.dy + ((size_t)n * a.HW + p) * C + c);
        f32x4 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float y = x[e] * A[e] + B[e];
            const float g = a.act ? d[e] * silu_grad_f(y) : d[e];
            r[e] = g * A[e] + x[e] * k1[e] + k0[e];
        }
        if (a.extra) r += *reinterpret_cast<const f32x4*>(a.extra + ((size_t)n * a.HW + p) * C + c);
        if (acc) r += *reinterpret_cast<const f32x4*>(dst + (size_t)p * ld);
        *reinterpret_cast<f32x4*>(dst + (size_t)p * ld) = r;
    }
}

int gn_bwd_split(int nfr, int HW, int C) { return gn_stats_split(nfr, HW, C); }

int launch_gn_bwd(const GnBwdArgs& a, hipStream_t s) {
    VD_REQUIRE(a.C % 32 == 0 && a.C <= 1024 && a.C0 % 4 == 0 && (a.x1 != nullptr || a.C0 == a.C), "GroupNorm backward: channel counts");
    VD_REQUIRE(a.part && a.K && a.mr && a.dx0 && (a.dx1 || a.C0 == a.C), "GroupNorm backward: buffers");
    const int split = gn_bwd_split(a.N, a.HW, a.C), per = (a.HW + split - 1) / split;
    hipLaunchKernelGGL(gn_bwd_partial_kernel, dim3(split, a.N), dim3(256), 0, s, a, split, per);
    hipLaunchKernelGGL(gn_bwd_fold_kernel, dim3(a.N), dim3(256), 0, s, a.part, split, a.C, (double)a.HW * (a.C / 32), a.mr, a.K);
    const int tpp = a.C / 4, ppi = 256 / tpp;
    int sp2 = 1;
    while (a.N * sp2 < 2048 && a.HW / (sp2 * 2) >= ppi * 4) sp2 *= 2;
    hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(sp2, a.N), dim3(ppi * tpp), 0, s, a, (a.HW + sp2 - 1) / sp2);
    VD_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------ temporal GroupNorm backward
// y[b,t,p,c] = (x - mean[b,p,g]) * rstd[b,p,g] * gamma[c] + beta[c], statistics over (T x C/32).  One block = ppb pixels of
// one batch element; thread -> (pixel slot, channel quad).  Three sweeps over the T rows (they stay in L2): statistics,
// the two group sums of dy*gamma and dy*gamma*xhat, the result.
__global__ __launch_bounds__(256) void gn_temporal_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                              const float* __restrict__ dy, int T, int HW, int C, int accumulate,
                                                              float* __restrict__ dx) {
    extern __shared__ __attribute__((aligned(16))) double sred[];   // [ppb][C][2]
    const int tpp = C >> 2, ppb = 256 / tpp, cg = C / 32;
    const int tid = threadIdx.x, pl = tid / tpp, q = tid - pl * tpp;
    const int b = blockIdx.y, p = blockIdx.x * ppb + pl;
    const bool active = pl < ppb && p < HW;
    const double cnt = (double)cg * T;
    auto row = [&](const float* base, int t) { return *reinterpret_cast<const f32x4*>(base + (((size_t)b * T + t) * HW + p) * C + q * 4); };
    double s[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
    if (active) {
        for (int t = 0; t < T; ++t) {
            const f32x4 v = row(x, t);
#pragma unroll
            for (int e = 0; e < 4; ++e) { s[e] += v[e]; ss[e] += (double)v[e] * v[e]; }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) { sred[((size_t)pl * C + q * 4 + e) * 2] = s[e]; sred[((size_t)pl * C + q * 4 + e) * 2 + 1] = ss[e]; }
    }
    __syncthreads();
    float mean[4], rstd[4];
    if (active) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int g = (q * 4 + e) / cg;
            double gs = 0, gss = 0;
            for (int k = 0; k < cg; ++k) { gs += sred[((size_t)pl * C + g * cg + k) * 2]; gss += sred[((size_t)pl * C + g * cg + k) * 2 + 1]; }
            const double m = gs / cnt;
            double var = gss / cnt - m * m;
            if (var < 0) var = 0;
            mean[e] = (float)m; rstd[e] = (float)(1.0 / sqrt(var + 1e-5));
        }
    }
    __syncthreads();
    f32x4 gm = {0.f, 0.f, 0.f, 0.f};
    if (active) {
        gm = *reinterpret_cast<const f32x4*>(gamma + q * 4);
        double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
        for (int t = 0; t < T; ++t) {
            const f32x4 v = row(x, t), d = row(dy, t);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const double g = (double)d[e] * gm[e];
                s1[e] += g; s2[e] += g * (double)((v[e] - mean[e]) * rstd[e]);
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) { sred[((size_t)pl * C + q * 4 + e) * 2] = s1[e]; sred[((size_t)pl * C + q * 4 + e) * 2 + 1] = s2[e]; }
    }
    __syncthreads();
    if (!active) return;
    float m1[4], m2[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int g = (q * 4 + e) / cg;
        double a1 = 0, a2 = 0;
        for (int k = 0; k < cg; ++k) { a1 += sred[((size_t)pl * C + g * cg + k) * 2]; a2 += sred[((size_t)pl * C + g * cg + k) * 2 + 1]; }
        m1[e] = (float)(a1 / cnt); m2[e] = (float)(a2 / cnt);
    }
    for (int t = 0; t < T; ++t) {
        const f32x4 v = row(x, t), d = row(dy, t);
        float* o = dx + (((size_t)b * T + t) * HW + p) * C + q * 4;
        f32x4 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = rstd[e] * (d[e] * gm[e] - m1[e] - (v[e] - mean[e]) * rstd[e] * m2[e]);
        if (accumulate) r += *reinterpret_cast<const f32x4*>(o);
        *reinterpret_cast<f32x4*>(o) = r;
    }
}

int launch_gn_temporal_bwd(const float* x, const float* gamma, const float* dy, int B, int T, int HW, int C, int accumulate,
                           float* dx, hipStream_t s) {
    VD_REQUIRE(C % 32 == 0 && C <= 1024 && T >= 1 && T <= 32, "temporal GroupNorm backward: shape");
    const int ppb = 256 / (C / 4);
    hipLaunchKernelGGL(gn_temporal_bwd_kernel, dim3((HW + ppb - 1) / ppb, B), dim3(256), (size_t)ppb * C * 2 * sizeof(double), s, x, gamma, dy,
                       T, HW, C, accumulate, dx);
    VD_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------ temporal attention backward
// Forward (attn_temporal.hip): w[t,s] = q't.(ks + Rk[t,s]) + scale*ks.Rq[s,t], q' = q*scale; a = softmax_s(w, mask);
// o[t] = sum_s a[t,s] (vs + Rv[t,s]).  Given do:
//   da[t,s] = do[t].(vs + Rv[t,s]);  dw = a*(da - sum_s a*da);  dv[s] = sum_t a[t,s] do[t]
//   dq[t] = scale * sum_s dw[t,s] (ks + Rk[t,s]);  dk[s] = sum_t dw[t,s] (q't + scale*Rq[s,t])
// One block per (pixel, head, batch element); T <= 32, F <= 256.
template <bool RPE>
__global__ __launch_bounds__(256) void attn_temporal_bwd_kernel(AttnTemporalArgs a, const float* __restrict__ dout, float* __restrict__ dqkv) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int T = a.T, C = a.C, HW = a.HW, C3 = 3 * C, F = C / a.heads, FP = F + 4, TS = T + 1;
    float* qs = smem;                    // [T][FP] q*scale
    float* ks = qs + T * FP;             // [T][FP]
    float* vs = ks + T * FP;             // [T][FP]
    float* ds = vs + T * FP;             // [T][FP] do
    float* P = ds + T * FP;              // [T][TS]
    float* W = P + T * TS;               // [T][TS] da, then dw
    const int p = blockIdx.x, h = blockIdx.y, b = blockIdx.z, tid = threadIdx.x;
    const size_t tok0 = (size_t)b * T * HW + p;                       // token (b, t, p) = tok0 + t*HW
    for (int i = tid; i < T * F; i += 256) {
        const int t = i / F, f = i - t * F;
        const float* r = a.qkv + (tok0 + (size_t)t * HW) * C3 + h * F + f;
        qs[t * FP + f] = r[0] * a.scale; ks[t * FP + f] = r[C]; vs[t * FP + f] = r[2 * C];
        ds[t * FP + f] = dout[(tok0 + (size_t)t * HW) * C + h * F + f];
    }
    __syncthreads();
    for (int pr = tid; pr < T * T; pr += 256) {
        const int t = pr / T, s_ = pr - t * T;
        const float* rk = RPE ? a.Rk + (((size_t)b * T + t) * T + s_) * C + h * F : nullptr;
        const float* rq = RPE ? a.Rq + (((size_t)b * T + s_) * T + t) * C + h * F : nullptr;
        const float* rv = RPE ? a.Rv + (((size_t)b * T + t) * T + s_) * C + h * F : nullptr;
        float w = 0.f, da = 0.f;
        for (int f = 0; f < F; ++f) {
            const float k = ks[s_ * FP + f];
            w += qs[t * FP + f] * (k + (RPE ? rk[f] : 0.f)) + (RPE ? a.scale * k * rq[f] : 0.f);
            da += ds[t * FP + f] * (vs[s_ * FP + f] + (RPE ? rv[f] : 0.f));
        }
        bool masked = false;
        if (a.mask) {
            const float mt = a.mask[b * T + t], ms = a.mask[b * T + s_];
            float allowed = mt * ms;
            if (a.allow_pad) allowed += (1.f - mt) * (1.f - ms);
            else if (t == s_) allowed = 1.f;
            masked = allowed == 0.f;
        }
        P[t * TS + s_] = masked ? -INFINITY : w;
        W[t * TS + s_] = da;
    }
    __syncthreads();
    if (tid < T) {
        float* r = P + tid * TS; float* d = W + tid * TS;
        float mx = -INFINITY;
        for (int s_ = 0; s_ < T; ++s_) mx = fmaxf(mx, r[s_]);
        float sum = 0.f;
        for (int s_ = 0; s_ < T; ++s_) { const float e = __expf(r[s_] - mx); r[s_] = e; sum += e; }
        const float inv = 1.0f / sum;
        float dot = 0.f;
        for (int s_ = 0; s_ < T; ++s_) { r[s_] *= inv; dot += r[s_] * d[s_]; }
        for (int s_ = 0; s_ < T; ++s_) d[s_] = r[s_] * (d[s_] - dot);
    }
    __syncthreads();
    for (int i = tid; i < T * F; i += 256) {
        const int t = i / F, f = i - t * F;            // t doubles as the key index s for dk / dv
        float dq = 0.f, dk = 0.f, dv = 0.f;
        for (int u = 0; u < T; ++u) {
            const float wtu = W[t * TS + u], wut = W[u * TS + t];
            dq += wtu * (ks[u * FP + f] + (RPE ? a.Rk[(((size_t)b * T + t) * T + u) * C + h * F + f] : 0.f));
            dk += wut * (qs[u * FP + f] + (RPE ? a.scale * a.Rq[(((size_t)b * T + t) * T + u) * C + h * F + f] : 0.f));
            dv += P[u * TS + t] * ds[u * FP + f];
        }
        float* o = dqkv + (tok0 + (size_t)t * HW) * C3 + h * F + f;
        o[0] = dq * a.scale; o[C] = dk; o[2 * C] = dv;
    }
}

int launch_attn_temporal_bwd(const AttnTemporalArgs& a, const float* dout, float* dqkv, hipStream_t s) {
    VD_REQUIRE(a.T >= 1 && a.T <= 32 && a.C % a.heads == 0, "temporal attention backward: shape");
    const int F = a.C / a.heads;
    const size_t lds = ((size_t)4 * a.T * (F + 4) + (size_t)2 * a.T * (a.T + 1)) * sizeof(float);
    VD_REQUIRE(lds <= 150 * 1024, "temporal attention backward: head dim too large");
    const dim3 grid(a.HW, a.heads, a.B);
    if (a.Rk) {
        VD_RAISE_LDS((&attn_temporal_bwd_kernel<true>), lds);
        hipLaunchKernelGGL(attn_temporal_bwd_kernel<true>, grid, dim3(256), lds, s, a, dout, dqkv);
    } else {
        VD_RAISE_LDS((&attn_temporal_bwd_kernel<false>), lds);
        hipLaunchKernelGGL(attn_temporal_bwd_kernel<false>, grid, dim3(256), lds, s, a, dout, dqkv);
    }
    VD_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------ spatial attention backward
// o[t] = sum_s softmax_s(q't.ks) vs over the L pixels of one (frame, head).  Pass 1, block = 16 queries: the score rows in
// LDS, row statistics, dq, and per row lse = max + log(sum), D = sum_s P dP for pass 2.  Pass 2, block = 16 keys: walks the
// queries in tiles of 16, recomputes P from lse, accumulates dk and dv in registers.
constexpr int SQT = 16;
__global__ __launch_bounds__(256) void attn_sp_bwd_dq_kernel(AttnSpatialArgs a, const float* __restrict__ dout, float* __restrict__ dqkv,
                                                             float* __restrict__ lse, float* __restrict__ Dv) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int L = a.L, C = a.C, C3 = 3 * C, F = C / a.heads, FP = F + 4, LS = L + 1;
    float* qs = smem;                    // [SQT][FP]
    float* ds = qs + SQT * FP;           // [SQT][FP]
    float* S = ds + SQT * FP;            // [SQT][LS] scores -> P
    float* G = S + SQT * LS;             // [SQT][LS] dP -> dS
    const int n = blockIdx.z, h = blockIdx.y, t0 = blockIdx.x * SQT, tid = threadIdx.x;
    const size_t tok0 = (size_t)n * L;
    for (int i = tid; i < SQT * F; i += 256) {
        const int t = i / F, f = i - t * F, tt = min(t0 + t, L - 1);
        qs[t * FP + f] = a.qkv[(tok0 + tt) * C3 + h * F + f] * a.scale;
        ds[t * FP + f] = dout[(tok0 + tt) * C + h * F + f];
    }
    __syncthreads();
    for (int s_ = tid; s_ < L; s_ += 256) {
        float sc[SQT], dp[SQT];
#pragma unroll
        for (int t = 0; t < SQT; ++t) { sc[t] = 0.f; dp[t] = 0.f; }
        const float* kr = a.qkv + (tok0 + s_) * C3 + C + h * F;
        for (int f = 0; f < F; f += 4) {
            const f32x4 k = *reinterpret_cast<const f32x4*>(kr + f), v = *reinterpret_cast<const f32x4*>(kr + C + f);
#pragma unroll
            for (int t = 0; t < SQT; ++t) {
                const f32x4 q = *reinterpret_cast<const f32x4*>(qs + t * FP + f), d = *reinterpret_cast<const f32x4*>(ds + t * FP + f);
                sc[t] += q.x * k.x + q.y * k.y + q.z * k.z + q.w * k.w;
                dp[t] += d.x * v.x + d.y * v.y + d.z * v.z + d.w * v.w;
            }
        }
#pragma unroll
        for (int t = 0; t < SQT; ++t) { S[t * LS + s_] = sc[t]; G[t * LS + s_] = dp[t]; }
    }
    __syncthreads();
    {   // 16 lanes per row: max, sum, dot
        const int t = tid >> 4, l = tid & 15;
        float mx = -INFINITY;
        for (int s_ = l; s_ < L; s_ += 16) mx = fmaxf(mx, S[t * LS + s_]);
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 16));
        float sum = 0.f;
        for (int s_ = l; s_ < L; s_ += 16) { const float e = __expf(S[t * LS + s_] - mx); S[t * LS + s_] = e; sum += e; }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 16);
        const float inv = 1.0f / sum;
        float dot = 0.f;
        for (int s_ = l; s_ < L; s_ += 16) { const float pv = S[t * LS + s_] * inv; S[t * LS + s_] = pv; dot += pv * G[t * LS + s_]; }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) dot += __shfl_xor(dot, o, 16);
        for (int s_ = l; s_ < L; s_ += 16) G[t * LS + s_] = S[t * LS + s_] * (G[t * LS + s_] - dot);
        if (l == 0 && t0 + t < L) {
            lse[((size_t)n * a.heads + h) * L + t0 + t] = mx + __logf(sum);
            Dv[((size_t)n * a.heads + h) * L + t0 + t] = dot;
        }
    }
    __syncthreads();
    for (int i = tid; i < SQT * (F >> 2); i += 256) {
        const int t = i / (F >> 2), f = (i - t * (F >> 2)) * 4;
        if (t0 + t >= L) continue;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int s_ = 0; s_ < L; ++s_) acc += *reinterpret_cast<const f32x4*>(a.qkv + (tok0 + s_) * C3 + C + h * F + f) * G[t * LS + s_];
        *reinterpret_cast<f32x4*>(dqkv + (tok0 + t0 + t) * C3 + h * F + f) = acc * a.scale;
    }
}

__global__ __launch_bounds__(256) void attn_sp_bwd_dkv_kernel(AttnSpatialArgs a, const float* __restrict__ dout, float* __restrict__ dqkv,
                                                              const float* __restrict__ lse, const float* __restrict__ Dv) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int L = a.L, C = a.C, C3 = 3 * C, F = C / a.heads, FP = F + 4, F4 = F >> 2;
    float* ks = smem;                    // [SQT][FP]
    float* vs = ks + SQT * FP;
    float* qs = vs + SQT * FP;           // [SQT][FP] q*scale of the current query tile
    float* ds = qs + SQT * FP;
    float* P = ds + SQT * FP;            // [SQT queries][SQT + 1 keys]
    float* G = P + SQT * (SQT + 1);
    const int n = blockIdx.z, h = blockIdx.y, s0 = blockIdx.x * SQT, tid = threadIdx.x;
    const size_t tok0 = (size_t)n * L;
    for (int i = tid; i < SQT * F; i += 256) {
        const int s_ = i / F, f = i - s_ * F, ss = min(s0 + s_, L - 1);
        ks[s_ * FP + f] = a.qkv[(tok0 + ss) * C3 + C + h * F + f];
        vs[s_ * FP + f] = a.qkv[(tok0 + ss) * C3 + 2 * C + h * F + f];
    }
    // thread -> up to two (key, feature quad) items of the dk / dv accumulators
    const int items = SQT * F4;
    f32x4 dk[2], dv[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) { dk[u] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[u] = dk[u]; }
    for (int t0 = 0; t0 < L; t0 += SQT) {
        __syncthreads();
        for (int i = tid; i < SQT * F; i += 256) {
            const int t = i / F, f = i - t * F, tt = min(t0 + t, L - 1);
            qs[t * FP + f] = a.qkv[(tok0 + tt) * C3 + h * F + f] * a.scale;
            ds[t * FP + f] = dout[(tok0 + tt) * C + h * F + f];
        }
        __syncthreads();
        {
            const int t = tid >> 4, s_ = tid & 15;
            float sc = 0.f, dp = 0.f;
            for (int f = 0; f < F; ++f) { sc += qs[t * FP + f] * ks[s_ * FP + f]; dp += ds[t * FP + f] * vs[s_ * FP + f]; }
            const bool ok = t0 + t < L && s0 + s_ < L;
            const size_t r = ((size_t)n * a.heads + h) * L + min(t0 + t, L - 1);
            const float pv = ok ? __expf(sc - lse[r]) : 0.f;
            P[t * (SQT + 1) + s_] = pv;
            G[t * (SQT + 1) + s_] = ok ? pv * (dp - Dv[r]) : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = tid + u * 256;
            if (i < items) {
                const int s_ = i / F4, f = (i - s_ * F4) * 4;
                for (int t = 0; t < SQT; ++t) {
                    dk[u] += *reinterpret_cast<const f32x4*>(qs + t * FP + f) * G[t * (SQT + 1) + s_];
                    dv[u] += *reinterpret_cast<const f32x4*>(ds + t * FP + f) * P[t * (SQT + 1) + s_];
                }
            }
        }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int i = tid + u * 256;
        if (i < items) {
            const int s_ = i / F4, f = (i - s_ * F4) * 4;
            if (s0 + s_ < L) {
                *reinterpret_cast<f32x4*>(dqkv + (tok0 + s0 + s_) * C3 + C + h * F + f) = dk[u];
                *reinterpret_cast<f32x4*>(dqkv + (tok0 + s0 + s_) * C3 + 2 * C + h * F + f) = dv[u];
            }
        }
    }
}

size_t attn_spatial_bwd_ws_floats(const AttnSpatialArgs& a) { return (size_t)2 * a.nfr * a.heads * a.L; }

int launch_attn_spatial_bwd(const AttnSpatialArgs& a, const float* dout, float* dqkv, float* ws, hipStream_t s) {
    const int F = a.C / a.heads;
    VD_REQUIRE(a.C % a.heads == 0 && F % 4 == 0 && F <= 128 && a.L <= 1024, "spatial attention backward: head dim <= 128, L <= 1024");
    float* lse = ws; float* Dv = ws + (size_t)a.nfr * a.heads * a.L;
    const size_t lds1 = ((size_t)2 * SQT * (F + 4) + (size_t)2 * SQT * (a.L + 1)) * sizeof(float);
    const size_t lds2 = ((size_t)4 * SQT * (F + 4) + (size_t)2 * SQT * (SQT + 1)) * sizeof(float);
    VD_RAISE_LDS((&attn_sp_bwd_dq_kernel), lds1);
    const dim3 grid((a.L + SQT - 1) / SQT, a.heads, a.nfr);
    hipLaunchKernelGGL(attn_sp_bwd_dq_kernel, grid, dim3(256), lds1, s, a, dout, dqkv, lse, Dv);
    hipLaunchKernelGGL(attn_sp_bwd_dkv_kernel, grid, dim3(256), lds2, s, a, dout, dqkv, lse, Dv);
    VD_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------ elementwise pieces
// y[n][2oy][2ox][c] = x[n][oy][ox][c], zeros elsewhere: the stride-2 Downsample conv's backward is a stride-1 conv of this
__global__ __launch_bounds__(256) void zero_stuff2_kernel(const float* __restrict__ x, int Ho, int Wo, int C4, size_t total4, float* __restrict__ y) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C4); size_t r = i / C4;
        const int ix = (int)(r % (2 * Wo)); r /= 2 * Wo;
        const int iy = (int)(r % (2 * Ho)); const size_t n = r / (2 * Ho);
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (!(ix & 1) && !(iy & 1)) v = reinterpret_cast<const f32x4*>(x)[((n * Ho + (iy >> 1)) * Wo + (ix >> 1)) * C4 + c];
        reinterpret_cast<f32x4*>(y)[i] = v;
    }
}
int launch_zero_stuff2(const float* x, int nfr, int Ho, int Wo, int C, float* y, hipStream_t s) {
    const size_t total4 = (size_t)nfr * 4 * Ho * Wo * C / 4;
    hipLaunchKernelGGL(zero_stuff2_kernel, dim3((unsigned)std::min<size_t>((total4 + 255) / 256, 8192)), dim3(256), 0, s, x, Ho, Wo, C / 4, total4, y);
    VD_HIP(hipGetLastError());
    return 0;
}

// y[n][oy][ox][c] (+)= sum of the 2x2 block of x: backward of the nearest x2 upsample folded into the Upsample conv's gather
__global__ __launch_bounds__(256) void sumpool2_kernel(const float* __restrict__ x, int Ho, int Wo, int C4, size_t total4, int accumulate, float* __restrict__ y) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C4); size_t r = i / C4;
        const int ox = (int)(r % Wo); r /= Wo;
        const int oy = (int)(r % Ho); const size_t n = r / Ho;
        const f32x4* b = reinterpret_cast<const f32x4*>(x) + ((n * 2 * Ho + 2 * oy) * 2 * Wo + 2 * ox) * C4 + c;
        f32x4 v = b[0] + b[C4] + b[(size_t)2 * Wo * C4] + b[(size_t)2 * Wo * C4 + C4];
        if (accumulate) v += reinterpret_cast<const f32x4*>(y)[i];
        reinterpret_cast<f32x4*>(y)[i] = v;
    }
}
int launch_sumpool2(const float* x, int nfr, int Ho, int Wo, int C, int accumulate, float* y, hipStream_t s) {
    const size_t total4 = (size_t)nfr * Ho * Wo * C / 4;
    hipLaunchKernelGGL(sumpool2_kernel, dim3((unsigned)std::min<size_t>((total4 + 255) / 256, 8192)), dim3(256), 0, s, x, Ho, Wo, C / 4, total4, accumulate, y);
    VD_HIP(hipGetLastError());
    return 0;
}

// y (+)= x
__global__ __launch_bounds__(256) void add_kernel(const float* __restrict__ x, size_t total4, int accumulate, float* __restrict__ y) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (size_t)gridDim.x * 256) {
        f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
        if (accumulate) v += reinterpret_cast<const f32x4*>(y)[i];
        reinterpret_cast<f32x4*>(y)[i] = v;
    }
}
int launch_add(const float* x, size_t n, int accumulate, float* y, hipStream_t s) {
    VD_REQUIRE(n % 4 == 0, "add: element count multiple of 4");
    hipLaunchKernelGGL(add_kernel, dim3((unsigned)std::min<size_t>((n / 4 + 255) / 256, 8192)), dim3(256), 0, s, x, n / 4, accumulate, y);
    VD_HIP(hipGetLastError());
    return 0;
}

// Output head backward, data part: da[n][y][x][c] = sum_{tap, co} deps[n][co][y+1-ky][x+1-kx] * w[tap][co][c]
// (w as the forward stores it: [tap][Cout][C]); NCHW gradient in, NHWC out.
__global__ __launch_bounds__(256) void out_conv_bwd_kernel(const float* __restrict__ deps, const float* __restrict__ w, int H, int Wd, int C, int Cout,
                                                           float* __restrict__ da) {
    const int n = blockIdx.y;
    const size_t HW = (size_t)H * Wd;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < HW * C; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C); const int p = (int)(i / C);
        const int y = p / Wd, x = p - y * Wd;
        float acc = 0.f;
        for (int tap = 0; tap < 9; ++tap) {
            const int oy = y + 1 - tap / 3, ox = x + 1 - tap % 3;
            if (oy < 0 || oy >= H || ox < 0 || ox >= Wd) continue;
            for (int co = 0; co < Cout; ++co) acc += deps[((size_t)n * Cout + co) * HW + (size_t)oy * Wd + ox] * w[((size_t)tap * Cout + co) * C + c];
        }
        da[((size_t)n * HW + p) * C + c] = acc;
    }
}
int launch_out_conv_bwd(const float* deps, const float* w, int nfr, int H, int Wd, int C, int Cout, float* da, hipStream_t s) {
    hipLaunchKernelGGL(out_conv_bwd_kernel, dim3((unsigned)std::min<size_t>(((size_t)H * Wd * C + 255) / 256, 4096), nfr), dim3(256), 0, s, deps, w, H, Wd, C,
                       Cout, da);
    VD_HIP(hipGetLastError());
    return 0;
}

// Stem backward, after dcols = dy * W (the im2col matrix's gradient [pixels][Kpad], k = tap*5 + channel): fold the taps back
// onto the 3 image channels and apply the factor with which x enters the network input (assemble_kernel:
// x*lat + obs_src*obs + x*(1 - any) -> lat + 1 - any).  dx is NCHW like x.
__global__ __launch_bounds__(256) void stem_col2im_kernel(const float* __restrict__ dcols, const float* __restrict__ obs, const float* __restrict__ lat,
                                                          const float* __restrict__ km, int H, int Wd, int Kpad, int cond_mode, float* __restrict__ dx) {
    const int n = blockIdx.y;
    const size_t HW = (size_t)H * Wd;
    const int Cs = cond_mode == 0 ? 5 : (cond_mode == 1 ? 6 : 3);
    // x enters the first three stem channels as x*lat + x*(1 - any) ('channel', 'duplicate' / 'all') or as x itself ('t=0')
    const float any = fminf(obs[n] + lat[n] + km[n], 1.0f), fac = cond_mode == 2 ? 1.0f : lat[n] + 1.0f - any;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < HW * 3; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i / HW); const int q = (int)(i - (size_t)c * HW);
        const int y = q / Wd, x = q - y * Wd;
        float acc = 0.f;
        for (int tap = 0; tap < 9; ++tap) {                           // output pixel p reads input pixel p + (tap/3 - 1, tap%3 - 1)
            const int py = y - (tap / 3 - 1), px = x - (tap % 3 - 1);
            if (py < 0 || py >= H || px < 0 || px >= Wd) continue;
            acc += dcols[((size_t)n * HW + (size_t)py * Wd + px) * Kpad + tap * Cs + c];
        }
        dx[((size_t)n * 3 + c) * HW + q] = acc * fac;
    }
}
int launch_stem_col2im(const float* dcols, const float* obs, const float* lat, const float* km, int nfr, int H, int Wd, int Kpad, int cond_mode,
                       float* dx, hipStream_t s) {
    hipLaunchKernelGGL(stem_col2im_kernel, dim3((unsigned)std::min<size_t>(((size_t)H * Wd * 3 + 255) / 256, 1024), nfr), dim3(256), 0, s, dcols, obs, lat, km, H,
                       Wd, Kpad, cond_mode, dx);
    VD_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------ the guidance itself
// gaussian_diffusion.py:350-364 on top of :319-343: x0r = a*x - b*eps; x0 = clamp(x0r); mean = c1*x0 + c2*x;
// smp = mean + nz*sigma*z; loss = sum ((smp - xtm1)*obs)^2.  d loss / d eps and the part of d loss / d x that does not go
// through the network:  dmean = 2*(smp - xtm1)*obs^2;  dx0r = c1*dmean*[-1 <= x0r <= 1];  deps = -b*dx0r;  dxd = c2*dmean + a*dx0r.
__global__ __launch_bounds__(256) void guided_grad_kernel(GuidedArgs a) {
    const size_t total = (size_t)a.B * a.per, perf = (size_t)a.per / a.T;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int b = (int)(i / a.per);
        const long long tl = a.t[b];
        const int NT = a.num_timesteps;
        if (tl < 0 || tl >= NT) { a.deps[i] = __builtin_nanf(""); a.dxd[i] = __builtin_nanf(""); continue; }
        const float* tb = a.tab + (int)tl;
        const float x = a.x[i], e = a.eps[i];
        const float x0r = tb[TAB_SQRT_RECIP * NT] * x - tb[TAB_SQRT_RECIPM1 * NT] * e;
        const bool bad = !(fabsf(x0r) <= 3.4028234e38f);              // (posterior_kernel: a non-finite eps must not be clamped into range)
        if (bad && a.err) atomicOr(a.err, VD_ERR_NONFINITE);
        const float x0 = (a.clip && !bad) ? fminf(fmaxf(x0r, -1.0f), 1.0f) : x0r;
        const float mean = tb[TAB_COEF1 * NT] * x0 + tb[TAB_COEF2 * NT] * x;
        const float nz = tl != 0 ? 1.0f : 0.0f;
        const float smp = mean + nz * expf(0.5f * tb[TAB_LOGVAR * NT]) * a.noise[i];
        const float om = a.obs[(size_t)b * a.T + (i - (size_t)b * a.per) / perf];
        const float dmean = 2.0f * (smp - a.xtm1[i]) * om * om;
        const float pass = (!a.clip || (x0r >= -1.0f && x0r <= 1.0f)) ? 1.0f : 0.0f;
        const float dx0r = tb[TAB_COEF1 * NT] * dmean * pass;
        a.deps[i] = -tb[TAB_SQRT_RECIPM1 * NT] * dx0r;
        a.dxd[i] = tb[TAB_COEF2 * NT] * dmean + tb[TAB_SQRT_RECIP * NT] * dx0r;
        if (a.mean) a.mean[i] = mean;
        if (a.xstart) a.xstart[i] = x0;
    }
}
int launch_guided_grad(const GuidedArgs& a, hipStream_t s) {
    const size_t total = (size_t)a.B * a.per;
    hipLaunchKernelGGL(guided_grad_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 4096)), dim3(256), 0, s, a);
    VD_HIP(hipGetLastError());
    return 0;
}

// ---- the backward pass runs on a power-of-two multiple of the loss gradient.  The backward-data pass is LINEAR in d loss / d eps, and its matrix
// products run in the forward's arithmetic (f16x3: an operand below 2^-14 is carried to an absolute 2^-37, not a relative 2^-22 -- tensors that are
// small THROUGHOUT are the one case the split does not serve, and gradients are such tensors: d loss / d eps ~ 1e-3 .. 1e-8 by the timestep).  So
// d eps is multiplied by s = 2^k with max |d eps * s| in [1, 2) before the pass and the result by 1 / s behind it: exact (powers of two), and every
// operand of the pass sits where the split is at its full 22 bits.  gs[0] = s, gs[1] = 1 / s, gs[2] = the bits of max |d eps| (zeroed by the caller).
__global__ __launch_bounds__(256) void grad_absmax_kernel(const float* __restrict__ g, size_t n, unsigned* __restrict__ mx) {
    float m = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) m = fmaxf(m, fabsf(g[i]));     // (fmaxf drops a NaN: it stays in g)
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) atomicMax(mx, __builtin_bit_cast(unsigned, m));           // non-negative floats order like their bit patterns
}
__global__ __launch_bounds__(256) void grad_scale_kernel(float* __restrict__ g, size_t n, float* __restrict__ gs) {
    const float m = __builtin_bit_cast(float, reinterpret_cast<const unsigned*>(gs)[2]);
    float s = 1.f;
    if (m > 0.f && m < 3.0e38f) {
        int ex;
        (void)frexpf(m, &ex);                                           // m = f * 2^ex, f in [0.5, 1)
        s = ldexpf(1.f, max(-100, min(100, 1 - ex)));               // max |d eps| * s in [1, 2): gradients inside the network may grow 2^14-fold before fp16's range ends
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { gs[0] = s; gs[1] = 1.f / s; }
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) g[i] *= s;
}
int launch_grad_rescale(float* deps, size_t n, float* gs, hipStream_t st) {
    VD_HIP(hipMemsetAsync(gs, 0, 4 * sizeof(float), st));
    const unsigned grid = (unsigned)std::min<size_t>((n + 255) / 256, 1024);
    hipLaunchKernelGGL(grad_absmax_kernel, dim3(grid), dim3(256), 0, st, deps, n, reinterpret_cast<unsigned*>(gs) + 2);
    hipLaunchKernelGGL(grad_scale_kernel, dim3(grid), dim3(256), 0, st, deps, n, gs);
    VD_HIP(hipGetLastError());
    return 0;
}

// g = dxd + dx_net / s;  mean' = mean - 10 * alpha_t * g / 2  (alpha_t = 1 - beta_t);  sample = mean' + nz*sigma*z2
__global__ __launch_bounds__(256) void guided_final_kernel(GuidedArgs a, const float* __restrict__ dx_net, const float* __restrict__ noise2,
                                                           float* __restrict__ grad, float* __restrict__ mean_out, float* __restrict__ sample) {
    const float inv_s = a.gscale ? a.gscale[1] : 1.f;
    const size_t total = (size_t)a.B * a.per;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int b = (int)(i / a.per);
        const long long tl = a.t[b];
        const int NT = a.num_timesteps;
        if (tl < 0 || tl >= NT) { if (mean_out) mean_out[i] = __builtin_nanf(""); if (sample) sample[i] = __builtin_nanf(""); continue; }
        const float* tb = a.tab + (int)tl;
        const float g = a.dxd[i] + dx_net[i] * inv_s;
        const float m = a.mean[i] - 10.0f * tb[TAB_ALPHA * NT] * g / 2.0f;
        if (grad) grad[i] = g;
        if (mean_out) mean_out[i] = m;
        if (sample) sample[i] = m + (tl != 0 ? 1.0f : 0.0f) * expf(0.5f * tb[TAB_LOGVAR * NT]) * noise2[i];
    }
}
int launch_guided_final(const GuidedArgs& a, const float* dx_net, const float* noise2, float* grad, float* mean_out, float* sample, hipStream_t s) {
    const size_t total = (size_t)a.B * a.per;
    hipLaunchKernelGGL(guided_final_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 4096)), dim3(256), 0, s, a, dx_net, noise2, grad, mean_out,
                       sample);
    VD_HIP(hipGetLastError());
    return 0;
}

}  // namespace vd

// ================================================================================================ return_attn_weights
// The reference logs, per attention block, the softmax weights averaged over the heads (unet.py:457-466:
// attn.view(B*D, -1, T, T).mean(dim=1).abs()): temporal blocks (B*HW, T, T), spatial blocks (B*T, HW, HW).  The attention
// kernels of the step never materialise them (online softmax / registers), so a caller that asks for them gets these two
// extra passes over q, k -- the visualisation path, not the sampling path.
namespace vd {

template <bool RPE>
__global__ __launch_bounds__(256) void attn_temporal_weights_kernel(AttnTemporalArgs a, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int T = a.T, C = a.C, HW = a.HW, C3 = 3 * C, F = C / a.heads, FP = F + 4, TS = T + 1;
    float* qs = smem;                    // [T][FP]
    float* ks = qs + T * FP;
    float* P = ks + T * FP;              // [T][TS] one head
    float* M = P + T * TS;               // [T][TS] mean over heads
    const int p = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const size_t tok0 = (size_t)b * T * HW + p;
    for (int i = tid; i < T * TS; i += 256) M[i] = 0.f;
    for (int h = 0; h < a.heads; ++h) {
        __syncthreads();
        for (int i = tid; i < T * F; i += 256) {
            const int t = i / F, f = i - t * F;
            const float* r = a.qkv + (tok0 + (size_t)t * HW) * C3 + h * F + f;
            qs[t * FP + f] = r[0] * a.scale; ks[t * FP + f] = r[C];
        }
        __syncthreads();
        for (int pr = tid; pr < T * T; pr += 256) {
            const int t = pr / T, s_ = pr - t * T;
            const float* rk = RPE ? a.Rk + (((size_t)b * T + t) * T + s_) * C + h * F : nullptr;
            const float* rq = RPE ? a.Rq + (((size_t)b * T + s_) * T + t) * C + h * F : nullptr;
            float w = 0.f;
            for (int f = 0; f < F; ++f) {
                const float k = ks[s_ * FP + f];
                w += qs[t * FP + f] * (k + (RPE ? rk[f] : 0.f)) + (RPE ? a.scale * k * rq[f] : 0.f);
            }
            bool masked = false;
            if (a.mask) {
                const float mt = a.mask[b * T + t], ms = a.mask[b * T + s_];
                float allowed = mt * ms;
                if (a.allow_pad) allowed += (1.f - mt) * (1.f - ms);
                else if (t == s_) allowed = 1.f;
                masked = allowed == 0.f;
            }
            P[t * TS + s_] = masked ? -INFINITY : w;
        }
        __syncthreads();
        if (tid < T) {
            float* r = P + tid * TS;
            float mx = -INFINITY;
            for (int s_ = 0; s_ < T; ++s_) mx = fmaxf(mx, r[s_]);
            float sum = 0.f;
            for (int s_ = 0; s_ < T; ++s_) { const float e = __expf(r[s_] - mx); r[s_] = e; sum += e; }
            const float inv = 1.0f / sum;
            for (int s_ = 0; s_ < T; ++s_) M[tid * TS + s_] += r[s_] * inv;
        }
    }
    __syncthreads();
    const float ih = 1.0f / (float)a.heads;
    for (int i = tid; i < T * T; i += 256) out[((size_t)b * HW + p) * T * T + i] = fabsf(M[(i / T) * TS + (i % T)] * ih);
}

int launch_attn_temporal_weights(const AttnTemporalArgs& a, float* out, hipStream_t s) {
    VD_REQUIRE(a.T >= 1 && a.T <= 32 && a.C % a.heads == 0, "temporal attention weights: shape");
    const int F = a.C / a.heads;
    const size_t lds = ((size_t)2 * a.T * (F + 4) + (size_t)2 * a.T * (a.T + 1)) * sizeof(float);
    VD_REQUIRE(lds <= 150 * 1024, "temporal attention weights: head dim too large");
    const dim3 grid(a.HW, a.B);
    if (a.Rk) {
        VD_RAISE_LDS((&attn_temporal_weights_kernel<true>), lds);
        hipLaunchKernelGGL(attn_temporal_weights_kernel<true>, grid, dim3(256), lds, s, a, out);
    } else {
        VD_RAISE_LDS((&attn_temporal_weights_kernel<false>), lds);
        hipLaunchKernelGGL(attn_temporal_weights_kernel<false>, grid, dim3(256), lds, s, a, out);
    }
    VD_HIP(hipGetLastError());
    return 0;
}

// block = 16 queries of one frame; heads in sequence, score rows in LDS
__global__ __launch_bounds__(256) void attn_spatial_weights_kernel(AttnSpatialArgs a, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int L = a.L, C = a.C, C3 = 3 * C, F = C / a.heads, FP = F + 4, LS = L + 1;
    float* qs = smem;                    // [SQT][FP]
    float* S = qs + SQT * FP;            // [SQT][LS]
    float* M = S + SQT * LS;             // [SQT][LS]
    const int n = blockIdx.y, t0 = blockIdx.x * SQT, tid = threadIdx.x;
    const size_t tok0 = (size_t)n * L;
    for (int i = tid; i < SQT * LS; i += 256) M[i] = 0.f;
    for (int h = 0; h < a.heads; ++h) {
        __syncthreads();
        for (int i = tid; i < SQT * F; i += 256) {
            const int t = i / F, f = i - t * F, tt = min(t0 + t, L - 1);
            qs[t * FP + f] = a.qkv[(tok0 + tt) * C3 + h * F + f] * a.scale;
        }
        __syncthreads();
        for (int s_ = tid; s_ < L; s_ += 256) {
            float sc[SQT];
#pragma unroll
            for (int t = 0; t < SQT; ++t) sc[t] = 0.f;
            const float* kr = a.qkv + (tok0 + s_) * C3 + C + h * F;
            for (int f = 0; f < F; f += 4) {
                const f32x4 k = *reinterpret_cast<const f32x4*>(kr + f);
#pragma unroll
                for (int t = 0; t < SQT; ++t) {
                    const f32x4 q = *reinterpret_cast<const f32x4*>(qs + t * FP + f);
                    sc[t] += q.x * k.x + q.y * k.y + q.z * k.z + q.w * k.w;
                }
            }
#pragma unroll
            for (int t = 0; t < SQT; ++t) S[t * LS + s_] = sc[t];
        }
        __syncthreads();
        const int t = tid >> 4, l = tid & 15;
        float mx = -INFINITY;
        for (int s_ = l; s_ < L; s_ += 16) mx = fmaxf(mx, S[t * LS + s_]);
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 16));
        float sum = 0.f;
        for (int s_ = l; s_ < L; s_ += 16) { const float e = __expf(S[t * LS + s_] - mx); S[t * LS + s_] = e; sum += e; }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 16);
        const float inv = 1.0f / sum;
        for (int s_ = l; s_ < L; s_ += 16) M[t * LS + s_] += S[t * LS + s_] * inv;
    }
    __syncthreads();
    const float ih = 1.0f / (float)a.heads;
    for (int i = tid; i < SQT * L; i += 256) {
        const int t = i / L, s_ = i - t * L;
        if (t0 + t < L) out[((size_t)n * L + t0 + t) * L + s_] = fabsf(M[t * LS + s_] * ih);
    }
}

int launch_attn_spatial_weights(const AttnSpatialArgs& a, float* out, hipStream_t s) {
    const int F = a.C / a.heads;
    VD_REQUIRE(a.C % a.heads == 0 && F % 4 == 0 && a.L <= 4096, "spatial attention weights: shape");
    const size_t lds = ((size_t)SQT * (F + 4) + (size_t)2 * SQT * (a.L + 1)) * sizeof(float);
    VD_REQUIRE(lds <= 150 * 1024, "spatial attention weights: sequence too long");
    VD_RAISE_LDS((&attn_spatial_weights_kernel), lds);
    hipLaunchKernelGGL(attn_spatial_weights_kernel, dim3((a.L + SQT - 1) / SQT, a.nfr), dim3(256), lds, s, a, out);
    VD_HIP(hipGetLastError());
    return 0;
}

}  // namespace vd
