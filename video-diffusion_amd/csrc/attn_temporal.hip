// Temporal self-attention core with relative-position terms (RPEAttention._forward, unet.py:486-536;
// RPE.forward_qk / forward_v, unet.py:357-378).  Sequence = the T <= 32 frames of one (batch, pixel).
//
//   w[t,s] = q't.ks + q't.Rk[t,s] + (ks*scale).Rq[s,t]      q' = q*scale
//   w -= inf where the frame mask forbids (t,s)               (unet.py:511-524)
//   a = softmax_s(w);  o[t] = sum_s a[t,s] * (vs + Rv[t,s])
//
// The R tensors depend on (batch, t, s) but not on the pixel, so one block handles PB pixels of one
// (batch, head): a thread owns a (t,s) pair, keeps its Rk/Rq slice in registers and walks the pixels;
// L2 traffic for R drops by PB and q/k rows are shared by the 16..32 lanes with equal t (or s).
// Scores round-trip through LDS ([PB][T][T+1], padded -> conflict-free row walks) for the softmax and
// the value pass, where a thread owns (t, 4 features) and keeps Rv[t, :, f4] in registers.
// VALU kernel: T*T*F per (pixel, head) is too ragged for 32x32 MFMA tiles and is ~1 % of step FLOPs.
#include "vd_common.h"

namespace vd {

constexpr int PB = 16;

template <int TMAX>
__global__ __launch_bounds__(256) void attn_temporal_kernel(AttnTemporalArgs a) {
    extern __shared__ __attribute__((aligned(16))) float sc[];     // [PB][T][T+1]
    const int T = a.T, C = a.C, HW = a.HW, C3 = 3 * a.C;
    const int F = C / a.heads;
    const int b = blockIdx.z, h = blockIdx.y, p0 = blockIdx.x * PB;
    const int np = min(PB, HW - p0);
    const int tid = threadIdx.x;
    const int TS = T + 1;
    const float* qkv_b = a.qkv + (size_t)b * T * HW * C3 + h * F;
    const bool rpe = a.Rk != nullptr;

    // ---- phase 1: scores
    for (int pr = tid; pr < T * T; pr += 256) {
        const int t = pr / T, s = pr - t * T;
        float acc[PB];
#pragma unroll
        for (int p = 0; p < PB; ++p) acc[p] = 0.f;
        const float* rk = rpe ? a.Rk + (((size_t)b * T + t) * T + s) * C + h * F : nullptr;
        const float* rq = rpe ? a.Rq + (((size_t)b * T + s) * T + t) * C + h * F : nullptr;
        const float* qrow = qkv_b + ((size_t)t * HW + p0) * C3;
        const float* krow = qkv_b + ((size_t)s * HW + p0) * C3 + C;
        for (int f0 = 0; f0 < F; f0 += 32) {
            const int nq = min(8, (F - f0) / 4);
            f32x4 Rkr[8], Rqr[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                Rkr[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                Rqr[i] = Rkr[i];
                if (rpe && i < nq) {
                    Rkr[i] = *reinterpret_cast<const f32x4*>(rk + f0 + i * 4);
                    Rqr[i] = *reinterpret_cast<const f32x4*>(rq + f0 + i * 4);
                }
            }
#pragma unroll
            for (int p = 0; p < PB; ++p) {
                if (p < np) {
                    float d = 0.f;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        if (i < nq) {
                            const f32x4 q = *reinterpret_cast<const f32x4*>(qrow + (size_t)p * C3 + f0 + i * 4) * a.scale;
                            const f32x4 k = *reinterpret_cast<const f32x4*>(krow + (size_t)p * C3 + f0 + i * 4);
                            const f32x4 ks = k * a.scale;
                            const f32x4 kk = k + Rkr[i];
#pragma unroll
                            for (int e = 0; e < 4; ++e) d += q[e] * kk[e] + ks[e] * Rqr[i][e];
                        }
                    }
                    acc[p] += d;
                }
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
        for (int p = 0; p < PB; ++p)
            if (p < np) sc[(p * T + t) * TS + s] = masked ? -INFINITY : acc[p];
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
    const int F4 = F / 4;
    for (int item = tid; item < T * F4; item += 256) {
        const int t = item / F4, f4 = item - t * F4;
        f32x4 rv[TMAX];
#pragma unroll
        for (int s = 0; s < TMAX; ++s) {
            rv[s] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (rpe && s < T) rv[s] = *reinterpret_cast<const f32x4*>(a.Rv + (((size_t)b * T + t) * T + s) * C + h * F + f4 * 4);
        }
        for (int p = 0; p < np; ++p) {
            const float* ar = sc + (p * T + t) * TS;
            const float* vbase = qkv_b + 2 * C + (size_t)(p0 + p) * C3 + f4 * 4;
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < TMAX; ++s) {
                if (s < T) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(vbase + (size_t)s * HW * C3);
                    o += (v + rv[s]) * ar[s];
                }
            }
            *reinterpret_cast<f32x4*>(a.out + (((size_t)b * T + t) * HW + p0 + p) * C + h * F + f4 * 4) = o;
        }
    }
}

int launch_attn_temporal(const AttnTemporalArgs& a, hipStream_t s) {
    VD_REQUIRE(a.T >= 1 && a.T <= 32, "temporal window of 1..32 frames");
    VD_REQUIRE(a.C % a.heads == 0 && (a.C / a.heads) % 4 == 0, "head dim multiple of 4");
    VD_REQUIRE((a.Rk == nullptr) == (a.Rq == nullptr) && (a.Rk == nullptr) == (a.Rv == nullptr), "all or no RPE terms");
    dim3 grid((a.HW + PB - 1) / PB, a.heads, a.B);
    const size_t lds = (size_t)PB * a.T * (a.T + 1) * sizeof(float);
    if (a.T <= 16) {
        hipLaunchKernelGGL(attn_temporal_kernel<16>, grid, dim3(256), lds, s, a);
    } else {
        static bool attr = false;
        if (!attr) {
            VD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_temporal_kernel<32>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, PB * 32 * 33 * 4));
            attr = true;
        }
        hipLaunchKernelGGL(attn_temporal_kernel<32>, grid, dim3(256), lds, s, a);
    }
    VD_HIP(hipGetLastError());
    return 0;
}

}  // namespace vd
