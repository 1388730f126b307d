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


// ------------------------------------------------------------------------------------------------------------------
// The same attention on the bf16 matrix cores at fp32 accuracy (default; VD_MATH=fp32 keeps the kernel above).
// q*scale, k, v and the softmax weights p are each split EXACTLY into three bf16 pieces (vd_common.h: split_a/split_b)
// and every product runs as six piece products of v_mfma_f32_32x32x16_bf16 with fp32 accumulation -- the arithmetic of
// VD_MATH=bf16x6 in gemm_split.hip / conv_wino_r64.hip: 6 x 32 cycles per 16 k instead of 8 x 64.  (VD_MATH=f16x3: below.)
//   S^T = K . Q^T : A = K tile (three bf16 planes in LDS, rows of FK bf16 + 16 bytes: conflict-free ds_read_b128),
//                   B = Q^T pieces, split once per wave and kept in registers
//   O^T += V^T . P^T : B = P^T, the S^T accumulator registers split in place (k order of an accumulator tile:
//                   element j of lane half h is key 16s + 8(j>>2) + 4h + (j&3)); A = V^T in that same key order, read
//                   from the row-major V planes with ds_read_b64_tr_b16 (a 16-lane group receives 4 keys x 16 features
//                   column-major), so V is staged as it streams in, without a transposing write pass.
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));

// gfx940-class hazard: a non-transcendental VALU instruction that reads a VGPR written by a transcendental one (v_exp_f32:
// the softmax weights below) needs one wait state in between.  hipcc inserts it between its own instructions but does
// not look inside inline asm, and split_a's first instruction may be scheduled right behind the v_exp_f32 that produced
// its operand: the 16-lane passes of the transcendental unit that had not retired yet were then read stale -- a fixed
// lane pattern, for one element, in exactly the head dims whose schedule put the two back to back (F = 40, 48: keys 24
// and 28 of a tile counted twice for queries 0-3, 8-11, ...).  The guard below is part of the asm block's own text.
__device__ __forceinline__ void split_f4(f32x4 v, u32x2_t& p1, u32x2_t& p2, u32x2_t& p3) {
    float r0, r1, r2, r3;
    unsigned a, b, c, d, e, f;
    asm volatile("s_nop 1" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));     // operands pinned behind the wait states
    split_a(v.x, v.y, a, r0, r1, 0x07060302u);
    split_a(v.z, v.w, b, r2, r3, 0x07060302u);
    split_b(r0, r1, c, e, 0x07060302u);
    split_b(r2, r3, d, f, 0x07060302u);
    p1 = u32x2_t{a, b}; p2 = u32x2_t{c, d}; p3 = u32x2_t{e, f};
}

// f16x3 (vd_common.h; the default arithmetic): K and V are the "a" side -- a0 = f16(x), a1 = f16((x - a0) * 2^12) -- Q^T and P^T the "b"
// side -- b0 = f16(y), b1 = f16(y - b0), b2 = 2^-12 b0 -- and a product is a1 b2 + a0 b1 + a0 b0 on v_mfma_f32_32x32x16_f16: half the MFMAs,
// two LDS planes instead of three, 2.5 instead of 5.5 vector instructions per split value.  The b side must sit near the top of fp16's
// range for b1 to stay normal (gemm_split.hip scales its weight rows on the host): every query row is scaled by a power of two to
// max |q| in [2^13, 2^14) -- undone on the scores, exactly -- and the softmax weights (<= 1) by 2^14, undone with the final 1 / l.
__device__ __forceinline__ void split_a16(f32x4 v, u32x2_t& p0, u32x2_t& p1) {
    asm volatile("s_nop 1" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));     // (the transcendental hazard above)
    p0 = u32x2_t{f16_pack(v.x, v.y), f16_pack(v.z, v.w)};
    p1 = u32x2_t{f16_pack_scaled(f16_rem_lo(p0.x, v.x), f16_rem_hi(p0.x, v.y)), f16_pack_scaled(f16_rem_lo(p0.y, v.z), f16_rem_hi(p0.y, v.w))};
}
__device__ __forceinline__ void split_b16(f32x4 v, u32x2_t& b0, u32x2_t& b1, u32x2_t& b2) {
    asm volatile("s_nop 1" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w));
    b0 = u32x2_t{f16_pack(v.x, v.y), f16_pack(v.z, v.w)};
    b1 = u32x2_t{f16_pack(f16_rem_lo(b0.x, v.x), f16_rem_hi(b0.x, v.y)), f16_pack(f16_rem_lo(b0.y, v.z), f16_rem_hi(b0.y, v.w))};
    const unsigned two_m12 = 0x0c000c00u;
    asm("v_pk_mul_f16 %0, %2, %4\n\tv_pk_mul_f16 %1, %3, %4" : "=&v"(b2.x), "=&v"(b2.y) : "v"(b0.x), "v"(b0.y), "s"(two_m12));
}
// pieces of eight values in the order an MFMA B operand wants them: [piece] = {lo quad, hi quad}
template <bool F16>
__device__ __forceinline__ void split_b8(f32x4 v0, f32x4 v1, u32x4_t (&out)[3]) {
    u32x2_t a1, a2, a3, b1, b2, b3;
    if constexpr (F16) { split_b16(v0, a1, a2, a3); split_b16(v1, b1, b2, b3); }
    else { split_f4(v0, a1, a2, a3); split_f4(v1, b1, b2, b3); }
    out[0] = u32x4_t{a1.x, a1.y, b1.x, b1.y};
    out[1] = u32x4_t{a2.x, a2.y, b2.x, b2.y};
    out[2] = u32x4_t{a3.x, a3.y, b3.x, b3.y};
}
#ifndef VD_ATTN_ABL
#define VD_ATTN_ABL 0      // timing-only builds (results WRONG; tools/build_variant.sh): bit 0 no K / V loads, 1 no scores / softmax / output MFMAs, 2 no split + LDS stores
#endif
template <bool F16>
__device__ __forceinline__ f32x16 attn_mfma(u32x4_t a, u32x4_t b, f32x16 c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

constexpr int attn_rowv(int F) {           // bytes per V row: >= 64 per 32-feature tile, a multiple of 8, dwords = 16 mod 32
    int b = ((F + 31) / 32) * 64;
    while ((b / 4) % 32 != 16) b += 8;
    return b;
}

// NW waves x 32 queries per block.  L = 256 (16 x 16 tokens): NW = 8, ONE block per (frame, head) -- K and V are fetched and split once (with
// 128 queries per block the two blocks of a (frame, head) were dealt to different XCDs and each fetched K / V from HBM: 300 MB per launch
// for 201 algorithmic).  The key tiles are double buffered in LDS: tile t + 1 is split and stored behind tile t's MFMAs, ONE barrier per tile.
template <int F, bool F16, int NW>
__global__ __launch_bounds__(NW * 64) void attn_spatial_split_kernel(AttnSpatialArgs a) {
    constexpr int NT = NW * 64;
    constexpr int KS = (F + 15) / 16, FK = KS * 16;     // k-steps of the QK^T contraction (zero padded)
    constexpr int FT = (F + 31) / 32;                   // 32-wide output tiles over F
    constexpr int ROWK = FK * 2 + 16, ROWV = attn_rowv(F);
    constexpr int KPL = 32 * ROWK, VPL = 32 * ROWV;
    constexpr int NPL = F16 ? 2 : 3;                    // planes of K and V (the "a" side)
    constexpr int NPP = F16 ? 3 : 6;                    // piece products
    // (a piece, b piece) per product, small terms first: f16x3 a1 b2, a0 b1, a0 b0; bf16x6 the six of gemm_split.hip
    constexpr int PA[6] = {F16 ? 1 : 0, F16 ? 0 : 1, F16 ? 0 : 2, 0, 1, 0}, PB[6] = {2, 1, 0, 1, 0, 0};
    constexpr int STAGE = NPL * (KPL + VPL);           // one key tile: [K planes][V planes]
    extern __shared__ __attribute__((aligned(16))) char attn_sp_lds[];      // [2][STAGE]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    const int n = blockIdx.z, h = blockIdx.y;
    const int q0 = blockIdx.x * (NW * 32) + wave * 32;
    const int C3 = 3 * a.C;
    const float* base = a.qkv + (size_t)n * a.L * C3 + h * F;

    // Q^T pieces: lane (query lr, half lh) holds q[16ks + 8lh .. +7] * scale of k-step ks
    u32x4_t qp[KS][3];
    const int qi = q0 + lr;
    const bool qok = qi < a.L;
    f32x4 qv[KS][2];
    float qmax = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const int f0 = ks * 16 + lh * 8;
        qv[ks][0] = f32x4{0.f, 0.f, 0.f, 0.f}; qv[ks][1] = qv[ks][0];
        if (qok && f0 < F) {
            qv[ks][0] = *reinterpret_cast<const f32x4*>(base + (size_t)qi * C3 + f0) * a.scale;
            qv[ks][1] = *reinterpret_cast<const f32x4*>(base + (size_t)qi * C3 + f0 + 4) * a.scale;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) qmax = fmaxf(qmax, fmaxf(fabsf(qv[ks][0][e]), fabsf(qv[ks][1][e])));
    }
    // f16x3: the query row times 2^e with max |q 2^e| in [2^13, 2^14); the scores leave through 2^-e
    float qs = 1.f, qs_inv = 1.f;
    if constexpr (F16) {
        qmax = fmaxf(qmax, __shfl_xor(qmax, 32));
        if (qmax > 0.f && qmax < INFINITY) {
            const int ex = (int)((__builtin_bit_cast(unsigned, qmax) >> 23) & 0xff) - 126;       // qmax = m 2^ex, m in [0.5, 1) (a subnormal row: a smaller e, harmless)
            const int e = max(-100, min(100, 14 - ex));
            qs = ldexpf(1.f, e); qs_inv = ldexpf(1.f, -e);
        }
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) split_b8<F16>(qv[ks][0] * qs, qv[ks][1] * qs, qp[ks]);
    f32x16 o[FT];
#pragma unroll
    for (int t = 0; t < FT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[t][r] = 0.f;
    float m = -INFINITY, l = 0.f;

    // padding that is read but never staged: K columns F..FK (they enter the contraction) and V columns F..FT*32 (they
    // only feed output rows that are not stored; zeroed so that no NaN pattern meets a zero weight)
    if constexpr (FK != F || FT * 32 != F) {
        for (int i = tid; i < 2 * STAGE / 4; i += NT) reinterpret_cast<unsigned*>(attn_sp_lds)[i] = 0u;
        __syncthreads();
    }

    // transposed-read addresses of this lane: row (lane&15)>>2 of the 4-key block, columns 16*((lane>>4)&1) + 4*(lane&3)
    const int trow = (lane & 15) >> 2, tcol = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    const int voff = (4 * lh + trow) * ROWV + tcol * 2;
    const int koff = lr * ROWK + lh * 16;
    typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4;

    // K / V rows of a key tile: requested one tile ahead into registers (the loads fly under the previous tile's MFMAs),
    // split and stored once the tile before has been consumed
    constexpr int NIT = (32 * (F / 4) + NT - 1) / NT;
    f32x4 pk[NIT], pv[NIT];
    auto prefetch = [&](int k0) {
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            const int i = tid + u * NT;
            const int r = i / (F / 4), c4 = i - r * (F / 4);
            const int key = k0 + r;
            pk[u] = f32x4{0.f, 0.f, 0.f, 0.f}; pv[u] = pk[u];
            if (!(VD_ATTN_ABL & 1) && i < 32 * (F / 4) && key < a.L) {
                const float* p = base + (size_t)key * C3 + c4 * 4;
                pk[u] = *reinterpret_cast<const f32x4*>(p + a.C);
                pv[u] = *reinterpret_cast<const f32x4*>(p + 2 * a.C);
            }
        }
    };
    auto stage = [&](int buf) {                                    // the prefetched tile, split, into buffer buf
        char* Kd = attn_sp_lds + buf * STAGE;
        char* Vd = Kd + NPL * KPL;
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            const int i = tid + u * NT;
            if (!(VD_ATTN_ABL & 4) && i < 32 * (F / 4)) {
                const int r = i / (F / 4), c4 = i - r * (F / 4);
                u32x2_t k1, k2, k3, v1, v2, v3;
                if constexpr (F16) { split_a16(pk[u], k1, k2); split_a16(pv[u], v1, v2); }
                else { split_f4(pk[u], k1, k2, k3); split_f4(pv[u], v1, v2, v3); }
                char* kd = Kd + r * ROWK + c4 * 8;
                char* vd = Vd + r * ROWV + c4 * 8;
                *reinterpret_cast<u32x2_t*>(kd) = k1; *reinterpret_cast<u32x2_t*>(kd + KPL) = k2;
                *reinterpret_cast<u32x2_t*>(vd) = v1; *reinterpret_cast<u32x2_t*>(vd + VPL) = v2;
                if constexpr (!F16) { *reinterpret_cast<u32x2_t*>(kd + 2 * KPL) = k3; *reinterpret_cast<u32x2_t*>(vd + 2 * VPL) = v3; }
            }
        }
    };
    prefetch(0);
    stage(0);
    if (32 < a.L) prefetch(32);
    __syncthreads();
    for (int k0 = 0, tile = 0; k0 < a.L; k0 += 32, ++tile) {
        const char* Ks = attn_sp_lds + (tile & 1) * STAGE;
        const char* Vs = Ks + NPL * KPL;

        if (!(VD_ATTN_ABL & 2)) {
        // S^T = K . Q^T, six piece products per k-step, small terms first
        f32x16 st;
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            u32x4_t kf[NPL];
#pragma unroll
            for (int p = 0; p < NPL; ++p) kf[p] = *reinterpret_cast<const u32x4_t*>(Ks + p * KPL + koff + ks * 32);
#pragma unroll
            for (int q = 0; q < NPP; ++q) st = attn_mfma<F16>(kf[PA[q]], qp[ks][PB[q]], st);
        }
        if constexpr (F16) st *= qs_inv;                            // the query row's scale leaves (exact)
        // lane (query lr, half lh) holds keys k0 + (r&3) + 8*(r>>2) + 4*lh
        float mloc = -INFINITY;
        if (k0 + 32 > a.L) {                                          // (uniform: only a last, partial tile has keys to mask)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (k0 + (r & 3) + 8 * (r >> 2) + 4 * lh >= a.L) st[r] = -INFINITY;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, st[r]);
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
        // P^T pieces of the two k-steps (keys 0..15 and 16..31 of the tile): registers 8s .. 8s+7;  O^T += V^T . P^T
        constexpr float PSC = F16 ? 16384.f : 1.f;                    // f16x3: the weights times 2^14 (undone with 1 / l)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            u32x4_t pp[3];
            split_b8<F16>(f32x4{st[8 * s], st[8 * s + 1], st[8 * s + 2], st[8 * s + 3]} * PSC,
                          f32x4{st[8 * s + 4], st[8 * s + 5], st[8 * s + 6], st[8 * s + 7]} * PSC, pp);
            constexpr int TG = (FT % 2 == 0 && FT > 2) ? 2 : FT;       // feature tiles in flight together (register budget at F = 128)
#pragma unroll
            for (int t0 = 0; t0 < FT; t0 += TG) {
                u32x4_t vf[TG][NPL];
#pragma unroll
                for (int t = 0; t < TG; ++t)
#pragma unroll
                    for (int p = 0; p < NPL; ++p) {
                        const char* vb = Vs + p * VPL + voff + (16 * s) * ROWV + (t0 + t) * 64;
                        const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(vb));
                        const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(vb + 8 * ROWV));
                        typedef short s16x8_t __attribute__((ext_vector_type(8)));
                        vf[t][p] = __builtin_bit_cast(u32x4_t, s16x8_t{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w});
                    }
                // (an accumulator is touched every TG-th MFMA: a dependent one issued back to back waits for its predecessor's write-back)
#pragma unroll
                for (int q = 0; q < NPP; ++q)
#pragma unroll
                    for (int t = 0; t < TG; ++t) o[t0 + t] = attn_mfma<F16>(vf[t][PA[q]], pp[PB[q]], o[t0 + t]);
            }
        }
        }
        // the next tile into the other buffer (last read before the barrier that ended the previous tile), the one behind it requested
        if (k0 + 32 < a.L) {
            stage((tile + 1) & 1);
            if (k0 + 64 < a.L) prefetch(k0 + 64);
        }
        __syncthreads();
    }
    if (!qok) return;
    const float inv = (F16 ? 1.f / 16384.f : 1.f) / l;
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

template <int F, bool F16, int NW>
static int launch_attn_spatial_split(const AttnSpatialArgs& a, hipStream_t s) {
    constexpr int KS = (F + 15) / 16, NPL = F16 ? 2 : 3;
    constexpr int LDS = 2 * NPL * (32 * (KS * 32 + 16) + 32 * attn_rowv(F));
    VD_RAISE_LDS((&attn_spatial_split_kernel<F, F16, NW>), (size_t)LDS);
    dim3 grid((a.L + NW * 32 - 1) / (NW * 32), a.heads, a.nfr);
    hipLaunchKernelGGL((attn_spatial_split_kernel<F, F16, NW>), grid, dim3(NW * 64), LDS, s, a);
    VD_HIP(hipGetLastError());
    return 0;
}
template <int F, bool F16>
static int launch_attn_spatial_nw(const AttnSpatialArgs& a, hipStream_t s) {
    // tokens per block: the whole (frame, head) when it has <= 256 of them
    if (a.L > 128) return launch_attn_spatial_split<F, F16, 8>(a, s);
    if (a.L > 64) return launch_attn_spatial_split<F, F16, 4>(a, s);
    return launch_attn_spatial_split<F, F16, 2>(a, s);
}

int launch_attn_spatial(const AttnSpatialArgs& a, hipStream_t s) {
    VD_REQUIRE(a.C % a.heads == 0, "channels divisible by heads");
    const int F = a.C / a.heads;
    dim3 grid((a.L + 127) / 128, a.heads, a.nfr);
    const int mode = math_mode();
    switch (F) {
#define VD_CASE(FV) case FV: if (mode == MATH_FP32) hipLaunchKernelGGL((attn_spatial_kernel<FV>), grid, dim3(256), 0, s, a); \
                             else if (mode == MATH_F16X3) { if (launch_attn_spatial_nw<FV, true>(a, s)) return -1; } \
                             else { if (launch_attn_spatial_nw<FV, false>(a, s)) return -1; } break;
        VD_CASE(8) VD_CASE(16) VD_CASE(24) VD_CASE(32) VD_CASE(40) VD_CASE(48) VD_CASE(56) VD_CASE(64) VD_CASE(80) VD_CASE(96) VD_CASE(112) VD_CASE(128)
#undef VD_CASE
        default:
            set_error("spatial attention: unsupported head dim " + std::to_string(F));
            return -1;
    }
    VD_HIP(hipGetLastError());
    return 0;
}

}  // namespace vd
