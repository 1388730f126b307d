// Spatial self-attention core (RPEAttention without RPE/mask, unet.py:471-540 as called at :260-266)
// on fp32 MFMA, flash-style: softmax(q*scale . k^T) v for L = H*W tokens of one (frame, head).
//
// Transposed formulation so every per-query quantity is lane-local:
//   S^T[key][query] = K_tile . Q^T          -> lane holds one query column, 16 of 32 keys in registers
//   row max / row sum = in-register over 16 values + one cross-half exchange (lane ^ 32)
//   O^T[f][query]  += V^T . P^T             -> the S^T accumulator registers ARE the B operand of the
//                                              next MFMA (one f32 per lane, k = lane>>5 pairs key r with
//                                              key r+4), so P never leaves registers.
// Per block: 4 waves x 32 queries; K/V tiles of 32 keys staged in LDS (rows padded by 4 floats ->
// conflict-free ds_read_b128 for K fragments and ds_read_b32 for V^T fragments).
#include "vd_common.h"

namespace vd {

template <int F>
__global__ __launch_bounds__(256) void attn_spatial_kernel(AttnSpatialArgs a) {
    constexpr int KG = F / 8;                 // float4 k-groups of the QK^T contraction
    constexpr int FT = (F + 31) / 32;         // 32-wide output tiles over F
    constexpr int LDK = F + 4;                // K tile row stride
    constexpr int LDV = FT * 32 + 4;          // V tile row stride (zero padded to FT*32)
    __shared__ __attribute__((aligned(16))) float Ks[32 * LDK];
    __shared__ __attribute__((aligned(16))) float Vs[32 * LDV];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const int n = blockIdx.z, h = blockIdx.y;
    const int q0 = blockIdx.x * 128 + wave * 32;
    const int C3 = 3 * a.C;
    const float* base = a.qkv + (size_t)n * a.L * C3 + h * F;

    // Q^T fragments: lane (query lr, half lh) keeps q[8kg + 4lh .. +3], pre-scaled.
    f32x4 qf[KG];
    const int qi = q0 + lr;
    const bool qok = qi < a.L;
#pragma unroll
    for (int kg = 0; kg < KG; ++kg) {
        if (qok) {
            f32x4 v = *reinterpret_cast<const f32x4*>(base + (size_t)qi * C3 + kg * 8 + lh * 4);
            qf[kg] = v * a.scale;
        } else {
            qf[kg] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    f32x16 o[FT];
#pragma unroll
    for (int t = 0; t < FT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
    float m = -INFINITY, l = 0.f;

    // zero the V padding columns once (only matters when F is not a multiple of 32)
    if (FT * 32 != F)
        for (int i = tid; i < 32 * LDV; i += 256) Vs[i] = 0.f;

    for (int k0 = 0; k0 < a.L; k0 += 32) {
        __syncthreads();
        // stage K, V tiles: 32 keys x F floats each, float4 granularity
        for (int i = tid; i < 32 * (F / 4); i += 256) {
            const int r = i / (F / 4), c4 = i - r * (F / 4);
            const int key = k0 + r;
            f32x4 kv = f32x4{0.f, 0.f, 0.f, 0.f}, vv = kv;
            if (key < a.L) {
                const float* p = base + (size_t)key * C3 + c4 * 4;
                kv = *reinterpret_cast<const f32x4*>(p + a.C);
                vv = *reinterpret_cast<const f32x4*>(p + 2 * a.C);
            }
            *reinterpret_cast<f32x4*>(Ks + r * LDK + c4 * 4) = kv;
            *reinterpret_cast<f32x4*>(Vs + r * LDV + c4 * 4) = vv;
        }
        __syncthreads();

        // S^T = K . Q^T  (A = K rows, B = Q^T)
        f32x16 st;
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
        for (int kg = 0; kg < KG; ++kg) {
            const f32x4 kf = *reinterpret_cast<const f32x4*>(Ks + lr * LDK + kg * 8 + lh * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) st = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[e], qf[kg][e], st, 0, 0, 0);
        }
        // lane (query lr, half lh) holds keys k0 + (r&3) + 8*(r>>2) + 4*lh
        float mloc = -INFINITY;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (key >= a.L) st[r] = -INFINITY;
            mloc = fmaxf(mloc, st[r]);
        }
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32));
        const float mnew = fmaxf(m, mloc);
        const float alpha = __expf(m - mnew);      // m = -inf on the first tile -> 0
        float ls = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) { st[r] = __expf(st[r] - mnew); ls += st[r]; }
        ls += __shfl_xor(ls, 32);
        l = l * alpha + ls;
        m = mnew;
#pragma unroll
        for (int t = 0; t < FT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[t][r] *= alpha;
        // O^T += V^T . P^T : A[i = f][k] = V[key(r, k)][f],  B[k][j = query] = P^T = st[r] of this lane
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int krow = (r & 3) + 8 * (r >> 2) + 4 * lh;
#pragma unroll
            for (int t = 0; t < FT; ++t) {
                const float vf = Vs[krow * LDV + t * 32 + lr];
                o[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(vf, st[r], o[t], 0, 0, 0);
            }
        }
    }
    if (!qok) return;
    const float inv = 1.0f / l;
    float* op = a.out + ((size_t)n * a.L + qi) * a.C + h * F;
#pragma unroll
    for (int t = 0; t < FT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int f = t * 32 + 8 * g + 4 * lh;     // rows (r&3) of register group g
            if (f < F) {
                f32x4 v = {o[t][4 * g] * inv, o[t][4 * g + 1] * inv, o[t][4 * g + 2] * inv, o[t][4 * g + 3] * inv};
                *reinterpret_cast<f32x4*>(op + f) = v;
            }
        }
}

int launch_attn_spatial(const AttnSpatialArgs& a, hipStream_t s) {
    VD_REQUIRE(a.C % a.heads == 0, "channels divisible by heads");
    const int F = a.C / a.heads;
    dim3 grid((a.L + 127) / 128, a.heads, a.nfr);
    switch (F) {
#define VD_CASE(FV) case FV: hipLaunchKernelGGL((attn_spatial_kernel<FV>), grid, dim3(256), 0, s, a); break;
        VD_CASE(8) VD_CASE(16) VD_CASE(24) VD_CASE(32) VD_CASE(48) VD_CASE(64) VD_CASE(96) VD_CASE(128)
#undef VD_CASE
        default:
            set_error("spatial attention: unsupported head dim " + std::to_string(F));
            return -1;
    }
    VD_HIP(hipGetLastError());
    return 0;
}

}  // namespace vd
