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

template <int PB, int TMAX, bool RPE>
static int launch_tt(const AttnTemporalArgs& a, size_t lds, hipStream_t s) {
    static size_t attr = 0;
    if (lds > attr) {
        VD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_temporal_kernel<PB, TMAX, RPE>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr = lds;
    }
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
