// Temporal self-attention core with relative-position terms (RPEAttention._forward, unet.py:486-536;
// RPE.forward_qk / forward_v, unet.py:357-378).  Sequence = the T <= 32 frames of one (batch, pixel).
//
//   w[t,s] = q't.ks + q't.Rk[t,s] + (ks*scale).Rq[s,t]      q' = q*scale
//   w -= inf where the frame mask forbids (t,s)               (unet.py:511-524)
//   a = softmax_s(w);  o[t] = sum_s a[t,s] * (vs + Rv[t,s])
//
// One block = PB pixels of one (batch, head).  q' and k of those pixels are staged in LDS once
// (coalesced 4F-byte rows; rows padded by 4 floats so the 16 distinct key rows of a ds_read_b128 lane
// group fall on 16 distinct 16-byte slots, the query row is a broadcast).  A thread owns a (t,s) pair,
// streams its Rk[t,s,:] / Rq[s,t,:] slices from L2 (they do not depend on the pixel, so they are
// amortised over the PB pixels) and accumulates the PB scores in registers.  Scores round-trip through
// LDS ([PB][T][T+1]) for the row softmax and the value pass, where a thread owns (t, 4 features) and
// keeps Rv[t, :, f4] in registers.  RPE is a template parameter: no branch around a load in any loop.
// VALU kernel: T*T*F per (pixel, head) is too ragged for 32x32 MFMA tiles and is 0.2 % of step FLOPs.
#include "vd_common.h"
#include <cstdlib>
#include <string>

namespace vd {

__device__ __forceinline__ float dot4(f32x4 a, f32x4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }

template <int PB, int TMAX, bool RPE>
__global__ __launch_bounds__(256) void attn_temporal_kernel(AttnTemporalArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int T = a.T, C = a.C, HW = a.HW, C3 = 3 * a.C;
    const int F = C / a.heads, F4 = F >> 2, FP = F + 4, TS = T + 1;
    float* qs = smem;                          // [PB][T][FP]  q * scale
    float* ks = qs + PB * T * FP;              // [PB][T][FP]
    float* sc = ks + PB * T * FP;              // [PB][T][TS]
    const int b = blockIdx.z, h = blockIdx.y, p0 = blockIdx.x * PB;
    const int np = min(PB, HW - p0);
    const int tid = threadIdx.x;
    const float* qkv_b = a.qkv + (size_t)b * T * HW * C3 + h * F;

    // ---- stage q', k (zeros for pixels past the end)
    for (int idx = tid; idx < PB * T * F4; idx += 256) {
        const int f4 = idx % F4, r = idx / F4;
        const int t = r % T, p = r / T;
        const float* src = qkv_b + ((size_t)t * HW + p0 + min(p, np - 1)) * C3 + f4 * 4;
        f32x4 q = *reinterpret_cast<const f32x4*>(src) * a.scale;
        f32x4 k = *reinterpret_cast<const f32x4*>(src + C);
        if (p >= np) { q = f32x4{0.f, 0.f, 0.f, 0.f}; k = q; }
        *reinterpret_cast<f32x4*>(qs + (p * T + t) * FP + f4 * 4) = q;
        *reinterpret_cast<f32x4*>(ks + (p * T + t) * FP + f4 * 4) = k;
    }
    __syncthreads();

    // ---- phase 1: scores
    for (int pr = tid; pr < T * T; pr += 256) {
        const int t = pr / T, s = pr - t * T;
        float acc[PB];
#pragma unroll
        for (int p = 0; p < PB; ++p) acc[p] = 0.f;
        const float* rk = RPE ? a.Rk + (((size_t)b * T + t) * T + s) * C + h * F : nullptr;
        const float* rq = RPE ? a.Rq + (((size_t)b * T + s) * T + t) * C + h * F : nullptr;
        const float* qrow = qs + t * FP;
        const float* krow = ks + s * FP;
#pragma unroll 3
        for (int f0 = 0; f0 < F; f0 += 8) {
            f32x4 rk0, rk1, rq0, rq1;
            if constexpr (RPE) {
                rk0 = *reinterpret_cast<const f32x4*>(rk + f0); rk1 = *reinterpret_cast<const f32x4*>(rk + f0 + 4);
                rq0 = *reinterpret_cast<const f32x4*>(rq + f0) * a.scale; rq1 = *reinterpret_cast<const f32x4*>(rq + f0 + 4) * a.scale;
            }
#pragma unroll
            for (int p = 0; p < PB; ++p) {
                const f32x4 q0 = *reinterpret_cast<const f32x4*>(qrow + p * T * FP + f0);
                const f32x4 q1 = *reinterpret_cast<const f32x4*>(qrow + p * T * FP + f0 + 4);
                const f32x4 k0 = *reinterpret_cast<const f32x4*>(krow + p * T * FP + f0);
                const f32x4 k1 = *reinterpret_cast<const f32x4*>(krow + p * T * FP + f0 + 4);
                if constexpr (RPE)
                    acc[p] += dot4(q0, k0 + rk0) + dot4(q1, k1 + rk1) + dot4(k0, rq0) + dot4(k1, rq1);
                else
                    acc[p] += dot4(q0, k0) + dot4(q1, k1);
            }
        }
        bool masked = false;
        if (a.mask) {
            const float mt = a.mask[b * T + t], ms = a.mask[b * T + s];
            float allowed = mt * ms;
            if (a.allow_pad) allowed += (1.f - mt) * (1.f - ms);
            else if (t == s) allowed = 1.f;
            masked = allowed == 0.f;
        }
#pragma unroll
        for (int p = 0; p < PB; ++p) sc[(p * T + t) * TS + s] = masked ? -INFINITY : acc[p];
    }
    __syncthreads();

    // ---- phase 2: row softmax (fp32, like th.softmax(w.float()))
    for (int row = tid; row < np * T; row += 256) {
        float* r = sc + row * TS;
        float mx = -INFINITY;
        for (int s = 0; s < T; ++s) mx = fmaxf(mx, r[s]);
        float sum = 0.f;
        for (int s = 0; s < T; ++s) { const float e = __expf(r[s] - mx); r[s] = e; sum += e; }
        const float inv = 1.0f / sum;
        for (int s = 0; s < T; ++s) r[s] *= inv;
    }
    __syncthreads();

    // ---- phase 3: o[t] = sum_s a[t,s] (v_s + Rv[t,s])
    for (int item = tid; item < T * F4; item += 256) {
        const int t = item / F4, f4 = item - t * F4;
        f32x4 rv[TMAX];
#pragma unroll
        for (int s = 0; s < TMAX; ++s) {
            rv[s] = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (RPE)
                rv[s] = *reinterpret_cast<const f32x4*>(a.Rv + (((size_t)b * T + t) * T + min(s, T - 1)) * C + h * F + f4 * 4);
        }
        for (int p = 0; p < np; ++p) {
            const float* ar = sc + (p * T + t) * TS;
            const float* vbase = qkv_b + 2 * C + (size_t)(p0 + p) * C3 + f4 * 4;
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < TMAX; ++s) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(vbase + (size_t)min(s, T - 1) * HW * C3);
                const float w = s < T ? ar[min(s, T - 1)] : 0.f;
                o += (v + rv[s]) * w;
            }
            *reinterpret_cast<f32x4*>(a.out + (((size_t)b * T + t) * HW + p0 + p) * C + h * F + f4 * 4) = o;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same operator on the fp32 matrix pipe (v_mfma_f32_16x16x4_f32: exact fp32 FMA chains, 32 cycles per 16x16x4).
//
// The three relative-position contractions are GEMMs once the PIXELS are the M dimension: for a fixed query frame t,
// q'[t, px, :] . Rk[t, s, :] is (16 px x F) x (F x T); for a fixed key frame s, k[s, px, :] . Rq[s, t, :] likewise, and
// for a fixed t  a[px, t, :] . Rv[t, :, f]  is (16 px x T) x (T x F).  q k^T and a v are per-pixel (T x F) x (F x T) /
// (T x T) x (T x F) products.  One block = 16 pixels of one (batch, head), four waves:
//
//   A1  wave w: q' k^T of pixels 4w..4w+3 (M = t, N = s)                       -> LDS  w[px][t][s]
//   A2  wave w: frames t = w, w+4, ..: accumulator preloaded from w[.][t][.] (M = px, N = s), + q'.Rk, stored back
//   A3  wave w: frames s = w, w+4, ..: the same with M = px, N = t, + k.Rq'
//   softmax over s, one (px, t) row per thread (mask rule of unet.py:511-524), probabilities back into w
//   B   wave w: feature tiles w, w+4, .. of 16: a v for all 16 pixels (M = t, N = f; 16 independent accumulators), then the
//       (tile = px, lane group = t/4, register = t%4) image is turned into (tile = t, lane group = px/4, register = px%4)
//       in registers -- a 4x4 transpose between lane rows and registers, v_permlane16_swap + v_permlane32_swap -- which is
//       the accumulator layout of the a.Rv product (M = px) of every frame t; its result goes to HBM (64-byte runs).
//
// Operands are read from L2 straight into fragment layout.  The k index of an MFMA step is free to permute as long as A
// and B agree, so a lane reads 16 bytes (features 16j + 4*(lane/16) .. +3) and feeds one element to each of four steps.
// Frames past T (T not a multiple of 16) are clamped on load and carry probability 0.
// ---------------------------------------------------------------------------------------------------------------------
#ifdef VD_ATT_TIMING
__device__ unsigned long long g_att_stamp[16];
#define ATT_STAMP(i)                                                                                   \
    do {                                                                                               \
        if (threadIdx.x == 0 && blockIdx.x == 3 && blockIdx.y == 1 && blockIdx.z == 2) {               \
            __builtin_amdgcn_sched_barrier(0);                                                         \
            g_att_stamp[i] = __builtin_amdgcn_s_memrealtime();                                         \
            __builtin_amdgcn_sched_barrier(0);                                                         \
        }                                                                                              \
    } while (0)
extern "C" int vd_debug_att_stamps(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_att_stamp), sizeof(g_att_stamp));
}
#else
#define ATT_STAMP(i)
#endif

__device__ __forceinline__ f32x4 mfma4(float x, float y, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, c, 0, 0, 0); }

// X[a] (register a, lane row c) -> X[c] (register c, lane row a): rows of 16 lanes
__device__ __forceinline__ void rows_regs_transpose(float& x0, float& x1, float& x2, float& x3) {
    auto u = [](float f) { return __builtin_bit_cast(unsigned, f); };
    auto f = [](unsigned v) { return __builtin_bit_cast(float, v); };
    auto p01 = __builtin_amdgcn_permlane16_swap(u(x0), u(x1), false, false);
    auto p23 = __builtin_amdgcn_permlane16_swap(u(x2), u(x3), false, false);
    auto q02 = __builtin_amdgcn_permlane32_swap(p01[0], p23[0], false, false);
    auto q13 = __builtin_amdgcn_permlane32_swap(p01[1], p23[1], false, false);
    x0 = f(q02[0]); x2 = f(q02[1]); x1 = f(q13[0]); x3 = f(q13[1]);
}

#ifndef VD_ATT_OCC
#define VD_ATT_OCC 2       // waves per SIMD the compiler must leave room for (4: <= 128 registers = two resident 8-wave blocks): without / with the RPE terms (A/B)
#endif
#ifndef VD_ATT_OCC_RPE
#define VD_ATT_OCC_RPE 2
#endif
template <int NT, int JM, bool RPE, bool EXACT>            // EXACT: F == 16*JM and T == 16*NT -- no guard around any request (a
// conditional request makes hipcc wait for ALL outstanding loads at the next use, which voids the requests made ahead)
__global__ __launch_bounds__(512, (RPE ? VD_ATT_OCC_RPE : VD_ATT_OCC)) void attn_temporal_mfma_kernel(AttnTemporalArgs a) {
    constexpr int TP = 16 * NT, RS = TP + 1, PS = TP * RS + (NT == 1 ? 1 : 17);   // pixel stride = 17 mod 32 banks
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* w = smem;                                        // [16 px][PS]: row (t) stride RS
    const int T = EXACT ? 16 * NT : a.T, C = a.C, HW = a.HW, C3 = 3 * a.C;
    const int F = C / a.heads, NJ = EXACT ? JM : F >> 4;    // <= JM
    const int b = blockIdx.z, h = blockIdx.y, p0 = blockIdx.x * 16;
    const int tid = threadIdx.x, wv = tid >> 6, l = tid & 63, i16 = l & 15, g = l >> 4;
    const float* qb = a.qkv + ((size_t)b * T * HW + p0) * C3 + h * F;
    auto row = [&](int t, int px) { return qb + ((size_t)t * HW + px) * C3; };
    auto ld4 = [](const float* p) { return *reinterpret_cast<const f32x4*>(p); };
    const int tcl[2] = {min(i16, T - 1), min(16 + i16, T - 1)};                    // this lane's frame in tile 0 / 1, clamped
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    // ---- A: the three score terms, all in ONE walk over the features.  Each wave owns pixels 2w, 2w+1 of q' k^T (M = t, N = s)
    //      and frames w, w+8 (, w+16, w+24) of q'.Rk (M = px, N = s) and of k.Rq' (M = px, N = t): q and k are needed in two
    //      fragment layouts, and requesting both inside the same feature group makes the second request an L2 hit (first
    //      build, term after term: 70 % L2 misses, the tensor fetched from HBM / MALL twice).  The operands of a feature
    //      group are requested together and, for one frame tile, one group ahead of their MFMAs.
    ATT_STAMP(0);
    constexpr int JG = NT == 1 ? 2 : 1, NG = JM / JG, NU = 2 * NT;
    constexpr bool AHEAD = NT == 1;
    struct Ops { f32x4 qa1[2][NT][JG], kb1[2][NT][JG], xa2[NU][JG], rb2[NU][NT][JG], xa3[NU][JG], rb3[NU][NT][JG]; };
    f32x4 acc1[2][NT][NT], acc2[NU][NT], acc3[NU][NT];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int m = 0; m < NT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n) acc1[u][m][n] = zero4;
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int n = 0; n < NT; ++n) { acc2[u][n] = zero4; acc3[u][n] = zero4; }
    const float* q1[2][NT];                                  // row (frame tcl[m], pixel 2w+u): q at +0, k at +C
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int m = 0; m < NT; ++m) q1[u][m] = row(tcl[m], 2 * wv + u) + 4 * g;
    const float* rkq[NT];                                    // + t*T*C: row (t, s = tcl[n]) of Rk / (s, t = tcl[n]) of Rq
#pragma unroll
    for (int n = 0; n < NT; ++n) rkq[n] = (size_t)b * T * T * C + (size_t)tcl[n] * C + h * F + 4 * g + (const float*)nullptr;
    auto request = [&](Ops& o, int grp, bool own, bool rel) {   // own: q' k^T operands; rel: those of the two relative-position terms
#pragma unroll
        for (int jj = 0; jj < JG; ++jj) {
            const int j = grp * JG + jj;
            if (EXACT || j < NJ) {
                if (own) {
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int m = 0; m < NT; ++m) { o.qa1[u][m][jj] = ld4(q1[u][m] + 16 * j); o.kb1[u][m][jj] = ld4(q1[u][m] + C + 16 * j); }
                }
                if constexpr (RPE) {
#pragma unroll
                    for (int u = 0; u < (rel ? NU : 0); ++u) {
                        const int t = wv + 8 * u;
                        if (EXACT || t < T) {
                            const float* xr = row(t, i16) + 4 * g + 16 * j;
                            o.xa2[u][jj] = ld4(xr);
                            o.xa3[u][jj] = ld4(xr + C);
#pragma unroll
                            for (int n = 0; n < NT; ++n) {
                                const size_t ro = (size_t)(rkq[n] - (const float*)nullptr) + (size_t)t * T * C + 16 * j;
                                o.rb2[u][n][jj] = ld4(a.Rk + ro);
                                o.rb3[u][n][jj] = ld4(a.Rq + ro);
                            }
                        }
                    }
                }
            }
        }
    };
    auto products = [&](const Ops& o, int grp, bool own, bool rel) {
#pragma unroll
        for (int jj = 0; jj < JG; ++jj) {
            if (EXACT || grp * JG + jj < NJ) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
#pragma unroll
                    for (int u = 0; u < (own ? 2 : 0); ++u)
#pragma unroll
                        for (int m = 0; m < NT; ++m)
#pragma unroll
                            for (int n = 0; n < NT; ++n) acc1[u][m][n] = mfma4(o.qa1[u][m][jj][e] * a.scale, o.kb1[u][n][jj][e], acc1[u][m][n]);
                    if constexpr (RPE) {
#pragma unroll
                        for (int u = 0; u < (rel ? NU : 0); ++u)
                            if (EXACT || wv + 8 * u < T) {
#pragma unroll
                                for (int n = 0; n < NT; ++n) {
                                    acc2[u][n] = mfma4(o.xa2[u][jj][e] * a.scale, o.rb2[u][n][jj][e], acc2[u][n]);
                                    acc3[u][n] = mfma4(o.xa3[u][jj][e], o.rb3[u][n][jj][e] * a.scale, acc3[u][n]);
                                }
                            }
                    }
                }
            }
        }
    };
    if constexpr (AHEAD) {
        Ops o[2];
        request(o[0], 0, true, true);
#pragma unroll
        for (int grp = 0; grp < NG; ++grp) {
            if (grp + 1 < NG) request(o[(grp + 1) & 1], grp + 1, true, true);
            __builtin_amdgcn_sched_barrier(0);               // the scheduler otherwise sinks every request to two loads before its use
            products(o[grp & 1], grp, true, true);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {                                                 // two frame tiles: the registers hold one half of a group's operands at a time
#pragma unroll 1
        for (int grp = 0; grp < NG; ++grp) {
            Ops o;
            request(o, grp, true, false);
            products(o, grp, true, false);
            if constexpr (RPE) {
                request(o, grp, false, true);
                products(o, grp, false, true);
            }
        }
    }
    ATT_STAMP(1);
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int m = 0; m < NT; ++m)
#pragma unroll
            for (int n = 0; n < NT; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r) w[(2 * wv + u) * PS + (16 * m + 4 * g + r) * RS + 16 * n + i16] = acc1[u][m][n][r];
    __syncthreads();
    if constexpr (RPE) {
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int t = wv + 8 * u;
            if (t < T) {
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int r = 0; r < 4; ++r) w[(4 * g + r) * PS + t * RS + 16 * n + i16] += acc2[u][n][r];
            }
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int sk = wv + 8 * u;
            if (sk < T) {
#pragma unroll
                for (int n = 0; n < NT; ++n)
#pragma unroll
                    for (int r = 0; r < 4; ++r) w[(4 * g + r) * PS + (16 * n + i16) * RS + sk] += acc3[u][n][r];
            }
        }
        __syncthreads();
    }

    ATT_STAMP(2);
    // ---- row softmax (fp32, like th.softmax(w.float())); columns T..TP-1 and rows >= T become 0
    for (int rix = tid; rix < 16 * TP; rix += 512) {
        const int px = rix / TP, t = rix - px * TP;
        float* r = w + px * PS + t * RS;
        if (t >= T) {
            for (int s = 0; s < TP; ++s) r[s] = 0.f;
            continue;
        }
        float mx = -INFINITY;
        const float mt = a.mask ? a.mask[b * T + t] : 1.f;
        for (int s = 0; s < T; ++s) {
            float v = r[s];
            if (a.mask) {
                const float ms = a.mask[b * T + s];
                float allowed = mt * ms;
                if (a.allow_pad) allowed += (1.f - mt) * (1.f - ms);
                else if (t == s) allowed = 1.f;
                if (allowed == 0.f) v = -INFINITY;
            }
            r[s] = v;
            mx = fmaxf(mx, v);
        }
        float sum = 0.f;
        for (int s = 0; s < T; ++s) { const float e = __expf(r[s] - mx); r[s] = e; sum += e; }
        const float inv = 1.0f / sum;
        for (int s = 0; s < T; ++s) r[s] *= inv;
        for (int s = T; s < TP; ++s) r[s] = 0.f;
    }
    __syncthreads();

    ATT_STAMP(3);
    // ---- B: o[t] = sum_s a[t,s] (v_s + Rv[t,s]); wave w owns features 16w .. 16w+15 of the head
    const int KS = (T + 3) >> 2;                                                           // k steps of the probability products
    if (EXACT ? wv < JM : wv < NJ) {
        const int f0 = 16 * wv + i16;
        const float* vb = qb + 2 * C + f0;                                                 // + (s*HW + px)*C3
        const float* rvb = RPE ? a.Rv + ((size_t)b * T) * T * C + h * F + f0 : nullptr;    // + (t*T + s)*C
#pragma unroll 1
        for (int m = 0; m < NT; ++m) {
            if (16 * m >= T) break;
            float rr[NT == 1 ? 4 : 1][16];
            f32x4 acc[16];
#pragma unroll
            for (int px = 0; px < 16; ++px) acc[px] = zero4;
#pragma unroll 1
            for (int kc = 0; kc < NT; ++kc) {                                              // four k steps (16 key frames) at a time
                if (16 * kc >= T) break;
                float vr[4][16];
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4)
                    if (EXACT || 4 * kc + k4 < KS) {
                        const float* vs = vb + (size_t)min(16 * kc + 4 * k4 + g, T - 1) * HW * C3;
#pragma unroll
                        for (int px = 0; px < 16; ++px) vr[k4][px] = vs[(size_t)px * C3];
                    }
                if constexpr (RPE && NT == 1) {                                            // the a.Rv operands ride along (registers allow it for one frame tile)
#pragma unroll
                    for (int k4 = 0; k4 < 4; ++k4)
                        if (EXACT || k4 < KS) {
                            const float* rv = rvb + (size_t)min(4 * k4 + g, T - 1) * C;
#pragma unroll
                            for (int tl = 0; tl < 16; ++tl) rr[k4][tl] = rv[(size_t)min(tl, T - 1) * T * C];
                        }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4)
                    if (EXACT || 4 * kc + k4 < KS) {
                        const float* ar = w + (16 * m + i16) * RS + 16 * kc + 4 * k4 + g;
#pragma unroll
                        for (int px = 0; px < 16; ++px) acc[px] = mfma4(ar[px * PS], vr[k4][px], acc[px]);
                    }
            }
            ATT_STAMP(4);
            // (tile px = 4A+Bq, row c, reg d) = o[px][t = 16m + 4c + d]  ->  (tile tl = 4c+d, row A, reg Bq) = o[px = 4A+Bq][t = 16m + tl]
            f32x4 res[16];
#pragma unroll
            for (int bq = 0; bq < 4; ++bq)
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    float x0 = acc[bq][d], x1 = acc[4 + bq][d], x2 = acc[8 + bq][d], x3 = acc[12 + bq][d];
                    rows_regs_transpose(x0, x1, x2, x3);
                    res[d][bq] = x0; res[4 + d][bq] = x1; res[8 + d][bq] = x2; res[12 + d][bq] = x3;
                }
            if constexpr (RPE) {
                if constexpr (NT == 1) {
#pragma unroll
                    for (int k4 = 0; k4 < 4; ++k4)
                        if (EXACT || k4 < KS) {
                            const float* ar = w + i16 * PS + 4 * k4 + g;
#pragma unroll
                            for (int tl = 0; tl < 16; ++tl) res[tl] = mfma4(ar[tl * RS], rr[k4][tl], res[tl]);
                        }
                } else {
#pragma unroll 1
                    for (int kc = 0; kc < NT; ++kc) {
                        if (16 * kc >= T) break;
                        float r4[4][16];
#pragma unroll
                        for (int k4 = 0; k4 < 4; ++k4)
                            if (4 * kc + k4 < KS) {
                                const float* rv = rvb + (size_t)min(16 * kc + 4 * k4 + g, T - 1) * C;
#pragma unroll
                                for (int tl = 0; tl < 16; ++tl) r4[k4][tl] = rv[(size_t)min(16 * m + tl, T - 1) * T * C];
                            }
#pragma unroll
                        for (int k4 = 0; k4 < 4; ++k4)
                            if (4 * kc + k4 < KS) {
                                const float* ar = w + i16 * PS + (16 * m) * RS + 16 * kc + 4 * k4 + g;
#pragma unroll
                                for (int tl = 0; tl < 16; ++tl) res[tl] = mfma4(ar[tl * RS], r4[k4][tl], res[tl]);
                            }
                    }
                }
            }
            ATT_STAMP(5);
#pragma unroll
            for (int tl = 0; tl < 16; ++tl) {
                const int t = 16 * m + tl;
                if (t < T) {
                    float* o = a.out + (((size_t)b * T + t) * HW + p0 + 4 * g) * C + h * F + f0;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[(size_t)r * C] = res[tl][r];
                }
            }
        }
    }
    ATT_STAMP(6);
}

template <int NT, int JM, bool RPE, bool EXACT>
static int launch_tm(const AttnTemporalArgs& a, hipStream_t s) {
    constexpr int TP = 16 * NT, PS = TP * (TP + 1) + (NT == 1 ? 1 : 17);
    constexpr size_t lds = (size_t)16 * PS * sizeof(float);
    if (lds > 48 * 1024) VD_RAISE_LDS((&attn_temporal_mfma_kernel<NT, JM, RPE, EXACT>), lds);
    dim3 grid(a.HW / 16, a.heads, a.B);
    hipLaunchKernelGGL((attn_temporal_mfma_kernel<NT, JM, RPE, EXACT>), grid, dim3(512), lds, s, a);
    VD_HIP(hipGetLastError());
    return 0;
}

template <int PB, int TMAX, bool RPE>
static int launch_tt(const AttnTemporalArgs& a, size_t lds, hipStream_t s) {
    VD_RAISE_LDS((&attn_temporal_kernel<PB, TMAX, RPE>), lds);
    dim3 grid((a.HW + PB - 1) / PB, a.heads, a.B);
    hipLaunchKernelGGL((attn_temporal_kernel<PB, TMAX, RPE>), grid, dim3(256), lds, s, a);
    VD_HIP(hipGetLastError());
    return 0;
}

int launch_attn_temporal(const AttnTemporalArgs& a, hipStream_t s) {
    VD_REQUIRE(a.T >= 1 && a.T <= 32, "temporal window of 1..32 frames");
    VD_REQUIRE(a.C % a.heads == 0 && (a.C / a.heads) % 8 == 0, "head dim multiple of 8");
    VD_REQUIRE((a.Rk == nullptr) == (a.Rq == nullptr) && (a.Rk == nullptr) == (a.Rv == nullptr), "all or no RPE terms");
    const int F = a.C / a.heads;
    // The kernel is chosen by the per-item shape alone (pixels, heads, head dim, T), never by the batch: a clip gives the same
    // bits whether it is sampled alone (a strong-scaling shard) or inside a batch.  (At B = 1 the 4-pixel blocks of the
    // VALU kernel fill the chip slightly better -- 5.87 against 5.95 ms per step -- which is not worth a batch-dependent result.)
    if (a.HW % 16 == 0 && F % 16 == 0 && F <= 128) {           // the matrix-pipe kernel: 16-pixel blocks, 16-feature k steps, one feature tile per wave
        const bool rpe = a.Rk != nullptr;
        if (a.T == 16 && F == 96) return rpe ? launch_tm<1, 6, true, true>(a, s) : launch_tm<1, 6, false, true>(a, s);     // the default models' two shapes
        if (a.T == 16 && F == 128) return rpe ? launch_tm<1, 8, true, true>(a, s) : launch_tm<1, 8, false, true>(a, s);
        if (a.T <= 16) {
            if (F <= 64) return rpe ? launch_tm<1, 4, true, false>(a, s) : launch_tm<1, 4, false, false>(a, s);
            return rpe ? launch_tm<1, 8, true, false>(a, s) : launch_tm<1, 8, false, false>(a, s);
        }
        if (F <= 64) return rpe ? launch_tm<2, 4, true, false>(a, s) : launch_tm<2, 4, false, false>(a, s);
        return rpe ? launch_tm<2, 8, true, false>(a, s) : launch_tm<2, 8, false, false>(a, s);
    }
    auto lds_for = [&](int pb) { return ((size_t)2 * pb * a.T * (F + 4) + (size_t)pb * a.T * (a.T + 1)) * sizeof(float); };
    const bool rpe = a.Rk != nullptr;
    const bool big = lds_for(4) > 96 * 1024;
    VD_REQUIRE(lds_for(2) <= 150 * 1024, "head dim too large for the temporal attention tile");
    if (a.T <= 16) {
        if (big) return rpe ? launch_tt<2, 16, true>(a, lds_for(2), s) : launch_tt<2, 16, false>(a, lds_for(2), s);
        return rpe ? launch_tt<4, 16, true>(a, lds_for(4), s) : launch_tt<4, 16, false>(a, lds_for(4), s);
    }
    if (big) return rpe ? launch_tt<2, 32, true>(a, lds_for(2), s) : launch_tt<2, 32, false>(a, lds_for(2), s);
    return rpe ? launch_tt<4, 32, true>(a, lds_for(4), s) : launch_tt<4, 32, false>(a, lds_for(4), s);
}

}  // namespace vd
