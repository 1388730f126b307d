// Linear layers and 1x1 convolutions at fp32 accuracy on the bf16 matrix cores, gfx950.
//
//   out[m][n] = bias[n] + res[m][n] + sum_k f(A[m][k]) * W[n][k]        f = identity | SiLU
//
// An fp32 value is EXACTLY the sum of three bf16 values (3 x 8 significand bits = 24): x = x1 + x2 + x3 with
// x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2) (each subtraction exact in fp32).  A product a*b is then the
// sum of nine piece products, each exact in fp32; the three smallest (a2 b3, a3 b2, a3 b3 <= 2^-24 |a b|) are below the
// rounding of an fp32 accumulator and are dropped.  The remaining six go through v_mfma_f32_32x32x16_bf16 with fp32
// accumulation: 6 instructions x 32 cycles per 16 k instead of 8 x 64 cycles of v_mfma_f32_32x32x2_f32 -- 2.67x the
// fp32 matrix rate, and, unlike the fp32 MFMA, the bf16 MFMA leaves the vector ALU free for the splitting itself.
// Measured against an fp64 product the result is at least as close as the fp32-MFMA kernel's (tests/test_gpu_ops.py).
//
// Same operand plan as gemm_frag.hip: the A tile [BM][32] of a K-chunk is split once by the staging threads into three
// bf16 planes in LDS (rows padded to 80 bytes: conflict-free ds_read_b128 fragments), the weights are split on the
// host at load time and stream from L2 in fragment order [K/16][N/32][piece 3][lane 64][8 bf16], three k-steps ahead.
//
// F16 (the default arithmetic, VD_MATH=f16x3, vd_common.h): the A tile is split into TWO fp16 planes, a0 = f16(x) and
// a1 = f16((x - a0) * 2^12); three piece products a1 * (2^-12 b0) + a0 * b1 + a0 * b0 on v_mfma_f32_32x32x16_f16; the weight
// rows carry a power-of-two scale (image trailer): the accumulators start at (bias + residual) * s and leave through 1 / s.
#include <cstring>
#include <vector>

#include "vd_common.h"

namespace vd {

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

template <bool F16>
__device__ __forceinline__ f32x16 mfma_piece(u32x4 a, u32x4 b, f32x16 c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

#ifndef VD_GS_PF3
#define VD_GS_PF3 1        // 0: the A operand one chunk ahead for every tile (A/B)
#endif
#ifndef VD_GS_PF3_ALL
#define VD_GS_PF3_ALL 0    // 1: three chunks of the A operand in flight for the 128-row tiles too (A/B)
#endif
#ifndef VD_GS_RING6
#define VD_GS_RING6 1      // 0: three weight slots for the 64x64 tile too (A/B)
#endif
#ifndef VD_GS_B2REG
#define VD_GS_B2REG 1      // f16x3: the third weight piece 2^-12 b0 formed in registers (four v_pk_mul_f16 per fragment) instead of loaded (A/B: 0)
#endif
#ifndef VD_GS_RING3_192
#define VD_GS_RING3_192 0  // 1: 128x192 tile with two k-steps of weights ahead (the registers the un-fetched third piece freed).  Measured r04v,
#endif                     // same box: qkv 8192 x 512 x 1536 41.4 -> 43.3 us, 32768 x 384 x 1152 95.5 -> 97.6, class 4.03 -> 4.08 ms: not a latency problem
#ifndef VD_GS_XCD
#define VD_GS_XCD 1        // XCD-aware block -> tile mapping (A/B: 0)
#endif
#ifndef VD_GS_SKIP
#define VD_GS_SKIP 0       // kernel-experiment builds: bit 0 no weight loads, 1 no A staging, 2 no split + LDS stores, 3 stores of unsplit bits (timing only)
#endif
#ifdef VD_GS_TIMING
// cycle stamps of ONE block (the middle row tile, column tile 0; with two blocks per CU the shader clock counts both: use the 100 MHz clock):
// 0 start, 1 prologue done (first A tile in LDS), 2 K loop done, 3 stores issued; 4 / 5: the 100 MHz clock at 0 / 3
__device__ unsigned long long g_gs_stamp[8];
#define GS_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x == gridDim.x / 2 && blockIdx.y == 0) {                        \
        __builtin_amdgcn_sched_barrier(0); g_gs_stamp[i] = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_sched_barrier(0); } } while (0)
extern "C" int vd_debug_gs_stamps(unsigned long long* host_out) { return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_gs_stamp), sizeof(g_gs_stamp)); }
#else
#define GS_STAMP(i)
#endif
constexpr int SROW = 80;                   // bytes per LDS row of one plane: 32 bf16 + 16 bytes of padding

template <bool F16>
__device__ __forceinline__ void split3(f32x4 v, u32x2& p1, u32x2& p2, u32x2& p3) {
    if constexpr (F16) {
        p1.x = f16_pack(v.x, v.y); p1.y = f16_pack(v.z, v.w);
        p2.x = f16_pack_scaled(f16_rem_lo(p1.x, v.x), f16_rem_hi(p1.x, v.y));
        p2.y = f16_pack_scaled(f16_rem_lo(p1.y, v.z), f16_rem_hi(p1.y, v.w));
        p3 = p2;
    } else {
        const bf16x4 q1 = __builtin_convertvector(v, bf16x4);
        f32x4 r = v - __builtin_convertvector(q1, f32x4);
        const bf16x4 q2 = __builtin_convertvector(r, bf16x4);
        r = r - __builtin_convertvector(q2, f32x4);
        const bf16x4 q3 = __builtin_convertvector(r, bf16x4);
        p1 = __builtin_bit_cast(u32x2, q1); p2 = __builtin_bit_cast(u32x2, q2); p3 = __builtin_bit_cast(u32x2, q3);
    }
}

// CONV: the A operand is the implicit im2col matrix of a 3x3 convolution (any stride, zero padding 1) over one NHWC
// source: K runs tap-major (k = tap*Cin + c, the order pack_conv3_split stores), the row of an output pixel moves with
// the tap, and a tap that falls outside the image reads zeros through the descriptor's range check.
// SIDE: the ResBlock skip convolution (unet.py:159-166, 1x1 over the block input) reads the same tensor the block's first
// GroupNorm+SiLU reads; the column blocks 0 write that activation image from the rows they stage (IgemmArgs::side): one pass
// over the (largest) tensor of the block instead of two.
template <int BM, int BN, bool ACT, bool CONV, bool F16 = true, bool SIDE = false>      // F16: VD_MATH=f16x3 (default) | bf16x6 (vd_common.h)
__global__ __launch_bounds__(256, 2) void gemm_split_kernel(IgemmArgs a) {
    constexpr int MI = BM / 64, NI = BN / 64, AR = BM / 32;
    // weight ring slots (k-steps ahead = RING - 1): 2 for the 128x192 tile (256 registers per wave; 3 under f16x3, whose ring holds two
    // pieces per fragment), 3 for the others -- and 6
    // for the 64x64 tile: its k-step is 6 MFMAs (0.1 us), its launches are the small-M ones (one block per CU, nothing else
    // to hide behind), and two steps ahead left every weight fragment a full L2 round trip short
    constexpr bool B2R = F16 && VD_GS_B2REG;                           // the third weight piece lives in ONE register set (b2), not in the ring
    constexpr int RING = NI >= 3 ? (B2R && VD_GS_RING3_192 ? 3 : 2) : (BM == 64 && BN == 64 && VD_GS_RING6) ? 6 : 3;
    constexpr int BP = B2R ? 2 : 3;                                     // fetched pieces per fragment
    // A-operand prefetch distance in chunks.  A chunk of a 64-row tile is 12 .. 24 MFMAs (0.2 .. 0.4 us): one chunk ahead, the
    // split + store of the next chunk waits a full memory round trip every chunk, and a small-M launch (a B = 1 shard: 60 of
    // them per step) costs ~1 us per chunk whatever its size.  The small tiles have the registers for three chunks in flight.
    constexpr int PF = ((BM == 64 && VD_GS_PF3) || VD_GS_PF3_ALL) ? 3 : 1;      // (three chunks in flight for the HBM-bound 128x128 SIDE launches too: measured, no gain -- r04i)
    constexpr int NPL = F16 ? 2 : 3;                                      // planes of the A tile
    constexpr int PLANE = BM * SROW, ABUF = NPL * PLANE;                  // bytes
    extern __shared__ __attribute__((aligned(16))) char smem_c[];         // [2][3 planes][BM][SROW]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    GS_STAMP(0);
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 31, lh = lane >> 5;
    // block -> tile.  Workgroups go to the 8 XCDs round-robin in dispatch order (x fastest), and each XCD has its own L2: with the
    // plain (x, y) = (row tile, column tile) mapping the N / BN column tiles of one row tile run on different XCDs at different
    // times and every one of them streams the A rows in again (qkv, N = 1152: six passes over A).  Here an XCD keeps its own row
    // tiles and walks their column tiles back to back: A comes in once per XCD, the (small) weight image is what all share.
    // Same box (r04w): 32768 x 384 x 1152 95.5 -> 90 us, 131072 x 640 x 256 205 -> 180, 131072 x 512 x 256 160 -> 148; launches whose
    // row tiles all fit the chip at once (8192 rows: 64 tiles) lose 5 % and keep the plain mapping.
    int bxi = blockIdx.x, byi = blockIdx.y;
    if (VD_GS_XCD && gridDim.y > 1 && (gridDim.x & 7) == 0 && gridDim.x >= 128) {
        const unsigned lin = blockIdx.x + gridDim.x * blockIdx.y, xcd = lin & 7, loc = lin >> 3;
        byi = loc % gridDim.y;
        bxi = (loc / gridDim.y) * 8 + xcd;
    }
    const int m0 = bxi * BM, n0 = byi * BN;
    // staging thread -> (row of the tile, 16-byte quad of the 32-wide chunk).  The two 8-lane groups of a 16-lane ds_write_b64
    // unit take rows FOUR apart (80-byte rows: 4 * 80 B = 16 banks mod 32, i.e. disjoint bank halves); with adjacent rows the
    // second row wrapped onto four banks of the first and a third of the LDS cycles were conflict cycles (PMC r02l / r03p:
    // SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.34; no measurable time, but it was the one conflict left in the step)
    const int lgrp = tid >> 3, lrow = ((lgrp >> 1) & 3) + 8 * (lgrp >> 3) + 4 * (lgrp & 1), lq = tid & 7;
    const int cpt = a.Cin >> 5;                                            // chunks per tap
    const int nchunk = CONV ? 9 * cpt : cpt, ncoblk = a.Cout >> 5;
    const int C1 = a.Cin - a.C0;
    if (a.zcount > 1) {                          // batched problems of one shape
        const int z = blockIdx.z;
        a.src0 += (size_t)z * a.zs_a; a.out += (size_t)z * a.zs_out;
        if (a.ztab) {                            // weights / bias of problem z from a table of offsets (IgemmArgs::ztab)
            a.wfrag = a.zbase + a.ztab[2 * z];
            a.bias = a.zbase + a.ztab[2 * z + 1];
        } else {
            a.wfrag += (size_t)z * a.zs_w;
            if (a.bias) a.bias += (size_t)z * a.zs_bias;
        }
    }

    const auto asrc0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.src0), 0,
                                                         (CONV ? a.nfr * a.Hs * a.Ws : a.M) * a.C0 * 4, 0x00020000);
    const auto asrc1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.src1 ? a.src1 : a.src0), 0,
                                                         a.src1 ? a.M * C1 * 4 : 0, 0x00020000);
    unsigned ao0[AR], ao1[AR];
    int iy0[AR], ix0[AR];                        // CONV: top-left tap of the row's output pixel; ao0 = its byte offset
#pragma unroll
    for (int j = 0; j < AR; ++j) {
        const unsigned row = (unsigned)min(m0 + lrow + 32 * j, a.M - 1);
        if constexpr (CONV) {
            const unsigned ox = row % (unsigned)a.Wo, t = row / (unsigned)a.Wo;
            const unsigned oy = t % (unsigned)a.Ho, n = t / (unsigned)a.Ho;
            iy0[j] = (int)oy * a.stride - 1;
            ix0[j] = (int)ox * a.stride - 1;
            ao0[j] = (unsigned)(((int)n * a.Hs + iy0[j]) * a.Ws + ix0[j]) * (unsigned)(a.Cin * 4) + lq * 16u;   // may wrap: only used when valid
            ao1[j] = 0;
        } else {
            ao0[j] = row * (unsigned)(a.C0 * 4) + lq * 16u;
            ao1[j] = row * (unsigned)(C1 * 4) + lq * 16u;
        }
    }
    // weights: [K/16][N/32][piece][lane][8 bf16] = 3072 bytes per (k-step, column block)
    const auto bsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wfrag), 0, (CONV ? 9 : 1) * a.Cin * a.Cout * 6, 0x00020000);
    unsigned bo[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) bo[j] = (unsigned)min(byi * (BN / 32) + wn * NI + j, ncoblk - 1) * 3072u + lane * 16u;

    f32x4 ra[PF][AR];
    u32x4 bfr[RING][NI][BP], b2[NI], afr[2][MI][NPL];     // [ring slot][tile][piece]; b2: 2^-12 x piece 0 of the k-step being multiplied
    // SIDE: this block's frame, the affine pair of the chunk in flight, and the image rows of this thread
    static_assert(!SIDE || (BM == 128 && !CONV && !ACT), "side output: 128-row tiles of a plain 1x1");
    const bool side_on = SIDE && byi == 0;
    const float* sAp = SIDE ? a.sideA + (size_t)(m0 / (SIDE ? a.side_hw : 1)) * a.Cin + lq * 4 : nullptr;
    const float* sBp = SIDE ? a.sideB + (size_t)(m0 / (SIDE ? a.side_hw : 1)) * a.Cin + lq * 4 : nullptr;
    f32x4 sA = {0.f, 0.f, 0.f, 0.f}, sB = sA;
    auto side_affine = [&](int chunk) {
        if constexpr (SIDE) {
            if (side_on) { sA = *reinterpret_cast<const f32x4*>(sAp + chunk * 32); sB = *reinterpret_cast<const f32x4*>(sBp + chunk * 32); }
        }
    };
    auto side_store = [&](f32x4 v, int chunk, int j) {
        if constexpr (SIDE) {
            const int row = m0 + lrow + 32 * j;
            if (side_on && row < a.M) {
                f32x4 r = v * sA + sB;
                r.x = silu_f(r.x); r.y = silu_f(r.y); r.z = silu_f(r.z); r.w = silu_f(r.w);
                *reinterpret_cast<f32x4*>(a.side + (size_t)row * a.Cin + chunk * 32 + lq * 4) = r;
            }
        }
    };
    auto a_prefetch = [&](int chunk, int rs) {
        if constexpr (CONV) {
            const int tap = chunk / cpt, c = (chunk - tap * cpt) * 32;
            const int kh = tap / 3, kw = tap - 3 * kh;
            const unsigned shift = (unsigned)((kh * a.Ws + kw) * a.Cin * 4);
#pragma unroll
            for (int j = 0; j < AR; ++j) {
                const bool ok = (unsigned)(iy0[j] + kh) < (unsigned)a.Hs && (unsigned)(ix0[j] + kw) < (unsigned)a.Ws;
                ra[rs][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(asrc0, ok ? ao0[j] + shift : 0x80000000u, c * 4, 0));
            }
            return;
        }
        const int c = chunk * 32;
        if (c < a.C0) {
#pragma unroll
            for (int j = 0; j < AR; ++j)
                ra[rs][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(asrc0, ao0[j], c * 4, 0));
        } else {
#pragma unroll
            for (int j = 0; j < AR; ++j)
                ra[rs][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(asrc1, ao1[j], (c - a.C0) * 4, 0));
        }
    };
    auto a_store = [&](char* Ad, int rs) {       // split once per element, three planes
#pragma unroll
        for (int j = 0; j < AR; ++j) {
            f32x4 v = ra[rs][j];
            side_store(v, 0, j);                 // (the prologue's chunk)
            if constexpr (ACT) { v.x = silu_f(v.x); v.y = silu_f(v.y); v.z = silu_f(v.z); v.w = silu_f(v.w); }
            u32x2 p1, p2, p3;
            split3<F16>(v, p1, p2, p3);
            char* d = Ad + (lrow + 32 * j) * SROW + lq * 8;
            *reinterpret_cast<u32x2*>(d) = p1;
            *reinterpret_cast<u32x2*>(d + PLANE) = p2;
            if constexpr (!F16) *reinterpret_cast<u32x2*>(d + 2 * PLANE) = p3;
        }
    };
    // f16x3: piece 2 of a weight fragment is 2^-12 x piece 0 (split_pack.hip) -- exact in fp16 arithmetic, subnormals included -- so it
    // is not fetched: a third less weight traffic from the L2 (the ablation builds of r04o: the weight loads are 20 % of this kernel)
    const unsigned two_m12 = 0x0c000c00u;
    auto b_third = [&](int slot, int j) {
        if constexpr (B2R)
            asm("v_pk_mul_f16 %0, %4, %8\n\tv_pk_mul_f16 %1, %5, %8\n\tv_pk_mul_f16 %2, %6, %8\n\tv_pk_mul_f16 %3, %7, %8"
                : "=&v"(b2[j][0]), "=&v"(b2[j][1]), "=&v"(b2[j][2]), "=&v"(b2[j][3])
                : "v"(bfr[slot][j][0][0]), "v"(bfr[slot][j][0][1]), "v"(bfr[slot][j][0][2]), "v"(bfr[slot][j][0][3]), "s"(two_m12));
    };
    auto b_load = [&](int slot, int kstep) {      // kstep = global 16-wide k step
        const int so = kstep * ncoblk * 3072;
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int p = 0; p < BP; ++p)
                bfr[slot][j][p] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(bsrc, bo[j] + p * 1024, so, 0));
    };
    const int aoff = (wm * (BM / 2) + lr) * SROW + lh * 16;
    auto a_frags = [&](int slot, const char* Ab, int ks) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int p = 0; p < NPL; ++p)
                afr[slot][i][p] = *reinterpret_cast<const u32x4*>(Ab + p * PLANE + aoff + i * 32 * SROW + ks * 32);
    };

    const auto osrc = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, a.M * a.ldo * 4, 0x00020000);
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.res ? a.res : a.out), 0,
                                                        a.res ? a.M * a.ldo * 4 : 0, 0x00020000);
    unsigned vb[MI][NI];
    float bv[NI], wsc[NI], winv[NI];             // F16: the weight rows' power-of-two scale and its reciprocal (image trailer)
    const float* trailer = a.wfrag + (size_t)(CONV ? 9 : 1) * a.Cin * a.Cout * 3 / 2;
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int co = n0 + wn * (BN / 2) + j * 32 + lr;
        bv[j] = a.bias && co < a.Cout ? a.bias[co] : 0.f;
        wsc[j] = F16 ? trailer[min(co, a.Cout - 1)] : 1.f;
        winv[j] = F16 ? trailer[a.Cout + min(co, a.Cout - 1)] : 1.f;
#pragma unroll
        for (int i = 0; i < MI; ++i)
            vb[i][j] = co < a.Cout ? (unsigned)((m0 + wm * (BM / 2) + i * 32 + 4 * lh) * a.ldo + co) * 4u : 0x80000000u;
    }
    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = bv[j];
    if (a.res) {                                 // (uniform) 16 residual requests per tile only where there is a residual
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int srow = ((r & 3) + 8 * (r >> 2)) * a.ldo * 4;
                    acc[i][j][r] += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, vb[i][j] + srow, 0, 0));
                }
    }
    if constexpr (F16) {                         // (bias + residual) * s: exact (s is a power of two), undone by 1 / s at the end
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] *= wsc[j];
    }

    const int nks = 2 * nchunk;
    a_prefetch(0, 0);
    side_affine(0);
    if constexpr (PF == 3) { a_prefetch(min(1, nchunk - 1), 1); a_prefetch(min(2, nchunk - 1), 2); }
#pragma unroll
    for (int r = 0; r < RING - 1; ++r) b_load(r, min(r, nks - 1));
    a_store(smem_c, 0);
    __syncthreads();
    a_frags(0, smem_c, 0);
    GS_STAMP(1);

    // k-steps in a ring of 3 weight slots (two steps ahead; 2 slots, one step ahead, for the 128x192 tile) and 2 fragment slots (one step ahead); the six piece
    // products of a tile go into its accumulator back to back, small terms first
    // Every memory request of a k-step goes out behind its own tile's MFMAs, never in bursts (conv_wino_r64.hip: a burst of
    // requests blocks the wave's issue and starves the matrix pipe): the weight loads of the step RING - 1 ahead, and in the
    // first k-step of a chunk the A prefetch and the fragment reads of the second k-step; the second k-step carries the
    // split + store of the next chunk's rows, one row slice per tile.  (r03n, same-box A/B against the compiler-scheduled loop
    // with its bursts of 9 + 4 loads: qkv 163 -> 154 us, proj 56.6 -> 52.7, class 6.59 -> 6.30 ms; the 128x192 tile also
    // stops spilling: 256 registers + 44 bytes of scratch -> 236, none.)
    auto b_load_one = [&](int slot, int kstep, int idx) {                // idx = j * 3 + p
        const int j = idx / 3, p2 = idx - 3 * j;
        if (p2 >= BP) return;                                             // formed from piece 0 by b_third
        bfr[slot][j][p2] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(bsrc, bo[j] + p2 * 1024, kstep * ncoblk * 3072, 0));
    };
    auto a_prefetch_one = [&](int chunk, int rs, int j) {
        if constexpr (CONV) {
            const int tap = chunk / cpt, c = (chunk - tap * cpt) * 32;
            const int kh = tap / 3, kw = tap - 3 * kh;
            const unsigned shift = (unsigned)((kh * a.Ws + kw) * a.Cin * 4);
            const bool ok = (unsigned)(iy0[j] + kh) < (unsigned)a.Hs && (unsigned)(ix0[j] + kw) < (unsigned)a.Ws;
            ra[rs][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(asrc0, ok ? ao0[j] + shift : 0x80000000u, c * 4, 0));
        } else {
            const int c = chunk * 32;
            ra[rs][j] = c < a.C0 ? __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(asrc0, ao0[j], c * 4, 0))
                                 : __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(asrc1, ao1[j], (c - a.C0) * 4, 0));
        }
    };
    auto a_frag_one = [&](int slot, const char* Ab, int ks, int idx) {   // idx = i * 3 + p
        const int i = idx / 3, p2 = idx - 3 * i;
        if (p2 >= NPL) return;
        afr[slot][i][p2] = *reinterpret_cast<const u32x4*>(Ab + p2 * PLANE + aoff + i * 32 * SROW + ks * 32);
    };
    auto a_store_one = [&](char* Ad, int rs, int j, int data_chunk) {
        f32x4 v = ra[rs][j];
        if (data_chunk >= 0) side_store(v, data_chunk, j);
        if constexpr (ACT) { v.x = silu_f(v.x); v.y = silu_f(v.y); v.z = silu_f(v.z); v.w = silu_f(v.w); }
        u32x2 p1, p2, p3;
        if (VD_GS_SKIP & 8) {                    // timing only: the raw bits instead of the pieces -- what an operand that arrives split would cost
            p1 = u32x2{__builtin_bit_cast(unsigned, v.x), __builtin_bit_cast(unsigned, v.y)};
            p2 = u32x2{__builtin_bit_cast(unsigned, v.z), __builtin_bit_cast(unsigned, v.w)};
            p3 = p2;
        } else split3<F16>(v, p1, p2, p3);
        char* d = Ad + (lrow + 32 * j) * SROW + lq * 8;
        *reinterpret_cast<u32x2*>(d) = p1;
        *reinterpret_cast<u32x2*>(d + PLANE) = p2;
        if constexpr (!F16) *reinterpret_cast<u32x2*>(d + 2 * PLANE) = p3;
    };
    constexpr int NT = MI * NI;                                            // tiles = slots per k-step
    auto kstep = [&](int chunk, int ks, int gslot, int aslot, int pf_chunk, int pf_slot, int st_slot) {
        const int g = 2 * chunk + ks;
        const char* Acur = smem_c + (chunk & 1) * ABUF;
        char* Anext = smem_c + ((chunk + 1) & 1) * ABUF;
#pragma unroll
        for (int j = 0; j < NI; ++j) b_third(gslot, j);
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int t = i * NI + j;
                f32x16 c = acc[i][j];
                if constexpr (F16) {             // a1 * (2^-12 b0) + a0 * b1 + a0 * b0
                    c = mfma_piece<true>(afr[aslot][i][1], B2R ? b2[j] : bfr[gslot][j][BP - 1], c);
                    c = mfma_piece<true>(afr[aslot][i][0], bfr[gslot][j][1], c);
                    c = mfma_piece<true>(afr[aslot][i][0], bfr[gslot][j][0], c);
                } else {
                    c = mfma_piece<false>(afr[aslot][i][0], bfr[gslot][j][2], c);
                    c = mfma_piece<false>(afr[aslot][i][1], bfr[gslot][j][1], c);
                    c = mfma_piece<false>(afr[aslot][i][2], bfr[gslot][j][0], c);
                    c = mfma_piece<false>(afr[aslot][i][0], bfr[gslot][j][1], c);
                    c = mfma_piece<false>(afr[aslot][i][1], bfr[gslot][j][0], c);
                    c = mfma_piece<false>(afr[aslot][i][0], bfr[gslot][j][0], c);
                }
                acc[i][j] = c;
                // this tile's share of the step's requests
#pragma unroll
                for (int q = t; q < NI * 3; q += NT)
                    if (!(VD_GS_SKIP & 1)) b_load_one((gslot + RING - 1) % RING, min(g + RING - 1, nks - 1), q);
                if (ks == 0) {
#pragma unroll
                    for (int q = t; q < AR; q += NT) if (!(VD_GS_SKIP & 2)) a_prefetch_one(pf_chunk, pf_slot, q);
                    if (SIDE && t == NT - 1 && chunk + 1 < nchunk) side_affine(chunk + 1);   // PF = 1: the chunk stored in the second k-step
#pragma unroll
                    for (int q = t; q < MI * 3; q += NT) a_frag_one(aslot ^ 1, Acur, 1, q);
                } else {
#pragma unroll
                    for (int q = t; q < AR; q += NT) if (!(VD_GS_SKIP & 6)) a_store_one(Anext, st_slot, q, SIDE && chunk + 1 < nchunk ? chunk + 1 : -1);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
    };
    for (int chunk0 = 0; chunk0 < nchunk; chunk0 += 3) {
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) {
            const int chunk = chunk0 + cc;
            if (chunk < nchunk) {
                const int nxt = min(chunk + PF, nchunk - 1);
                kstep(chunk, 0, RING == 2 ? 0 : (2 * cc) % RING, 0, nxt, PF == 3 ? cc : 0, 0);
                kstep(chunk, 1, RING == 2 ? 1 : (2 * cc + 1) % RING, 1, 0, 0, PF == 3 ? (cc + 1) % 3 : 0);
                __syncthreads();
                a_frags(0, smem_c + ((chunk + 1) & 1) * ABUF, 0);
            }
        }
    }

    GS_STAMP(2);
    if constexpr (F16) {
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] *= winv[j];
    }
    // GroupNorm partial sums of the output (the layout gn_stats_partial / the Winograd epilogue write): a wave tile is BM/2
    // rows of ONE frame (a.stats_hw % (BM/2) == 0), a lane holds 16 rows of its column per M-tile, the other 16 sit in lane ^ 32
    if (a.stats) {
        const int row0 = m0 + wm * (BM / 2);
        if (row0 < a.M) {
            const int fr = row0 / a.stats_hw, sp = (row0 - fr * a.stats_hw) / (BM / 2);
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                // every term through fp64, as in gn_stats_partial: the stem's output carries per-frame offsets (mask channels)
                // far above its spread, and E[x^2] - E[x]^2 amplifies an fp32 rounding of the partial sums by that ratio
                double d0 = 0.0, d1 = 0.0;
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) { const double v = (double)acc[i][j][r]; d0 += v; d1 += v * v; }
                d0 += __shfl_xor(d0, 32);
                d1 += __shfl_xor(d1, 32);
                const int co = n0 + wn * (BN / 2) + j * 32 + lr;
                if (lh == 0 && co < a.Cout) {
                    double* o = a.stats + (((size_t)fr * a.stats_split + sp) * a.Cout + co) * 2;
                    o[0] = d0; o[1] = d1;
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            unsigned vbe = vb[i][j];
            asm volatile("" : "+v"(vbe));
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int srow = ((r & 3) + 8 * (r >> 2)) * a.ldo * 4;
                const float val = acc[i][j][r];
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), osrc, vbe + srow, 0, 0);
            }
        }
#ifdef VD_GS_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    GS_STAMP(3);
}


// 3x3 convolutions whose weights came from pack_conv3_split (the stride-2 Downsample convs, unet.py:98)
bool conv_split_supported(const IgemmArgs& a) {
    return a.wsplit && a.wfrag != nullptr && a.ksz == 3 && a.pad == 1 && (a.stride == 1 || a.stride == 2) && a.ups == 0 &&
           a.Cout % 32 == 0 && a.Cin % 32 == 0 && a.C0 == a.Cin && a.affA == nullptr && a.fbias == nullptr && a.zcount <= 1 &&
           (a.res == nullptr || a.res_ld == a.ldo) && (size_t)a.nfr * a.Hs * a.Ws * a.Cin < (1u << 28) &&
           (size_t)a.M * a.ldo < (1u << 28) && (size_t)9 * a.Cin * a.Cout * 6 < (1u << 31);
}

bool gemm_split_supported(const IgemmArgs& a) {
    return a.wsplit && a.wfrag != nullptr && a.ksz == 1 && a.stride == 1 && a.pad == 0 && a.ups == 0 && a.Cout % 32 == 0 &&
           a.Cin % 32 == 0 && a.C0 % 32 == 0 && a.affA == nullptr && a.fbias == nullptr && (a.res == nullptr || a.res_ld == a.ldo) &&
           (size_t)a.M * std::max(std::max(a.C0, a.Cin - a.C0), a.ldo) < (1u << 28) && (size_t)a.Cin * a.Cout * 6 < (1u << 31);
}

bool gemm_split_side_supported(const IgemmArgs& a) {
    return gemm_split_supported(a) && a.act == 0 && a.zcount <= 1 && gemm_split_tile_class(igemm_sel_M(a), a.Cout) == 0 &&
           a.side_hw > 0 && a.side_hw % 128 == 0 && a.M % a.side_hw == 0;
}

template <int BM, int BN, bool F16>
static int launch_gs_m(const IgemmArgs& a, hipStream_t s) {
    const size_t lds = (size_t)2 * (F16 ? 2 : 3) * BM * SROW;
    dim3 grid((a.M + BM - 1) / BM, (a.Cout + BN - 1) / BN, a.zcount > 1 ? a.zcount : 1);
    if (a.side) {
        if constexpr (BM == 128 && BN == 128) {
            VD_REQUIRE(gemm_split_side_supported(a) && a.sideA && a.sideB, "side output: plain 1x1 on the 128x128 tile, whole frames of a multiple of 128 rows");
            hipLaunchKernelGGL((gemm_split_kernel<128, 128, false, false, F16, true>), grid, dim3(256), lds, s, a);
            VD_HIP(hipGetLastError());
            return 0;
        } else {
            VD_REQUIRE(false, "side output: 128x128 tile only");
        }
    }
    if (a.ksz == 3) {
        if (a.act) hipLaunchKernelGGL((gemm_split_kernel<BM, BN, true, true, F16>), grid, dim3(256), lds, s, a);
        else hipLaunchKernelGGL((gemm_split_kernel<BM, BN, false, true, F16>), grid, dim3(256), lds, s, a);
    } else if (a.act) hipLaunchKernelGGL((gemm_split_kernel<BM, BN, true, false, F16>), grid, dim3(256), lds, s, a);
    else hipLaunchKernelGGL((gemm_split_kernel<BM, BN, false, false, F16>), grid, dim3(256), lds, s, a);
    VD_HIP(hipGetLastError());
    return 0;
}
template <int BM, int BN>
static int launch_gs(const IgemmArgs& a, hipStream_t s) {
    return f16_math() ? launch_gs_m<BM, BN, true>(a, s) : launch_gs_m<BM, BN, false>(a, s);
}

// 128x192 tile for the wide projections (qkv: N = 1152, 1536; proj: N = 384): the A tile is staged and split once per 192
// instead of 128 columns, and their grids come out in whole rounds of 2 blocks per CU (N = 1536 at M = 8192: 512 blocks
// instead of 768; N = 384 at M = 32768: 512 instead of 768)
int gemm_split_tile_class(int M, int Cout) {
    const int base = igemm_tile_class(M, Cout);
    if (base != 0 || Cout % 192) return base;
    return (long)((M + 127) / 128) * (Cout / 192) >= 384 ? 4 : base;
}

int gemm_split_stats_rows(int M, int Cout) {
    const int c = gemm_split_tile_class(M, Cout);
    return c == 2 || c == 3 ? 32 : 64;
}

int launch_gemm_split(const IgemmArgs& a, int tile_class, hipStream_t s) {
    if (tile_class == 0) tile_class = gemm_split_tile_class(igemm_sel_M(a), a.Cout);
    if (a.stats) {
        const int rows = tile_class == 2 || tile_class == 3 ? 32 : 64;
        VD_REQUIRE(a.zcount <= 1 && a.stats_hw > 0 && a.stats_hw % rows == 0 && a.M % a.stats_hw == 0 &&
                   a.stats_split == a.stats_hw / rows, "GroupNorm partial sums from the GEMM epilogue: whole frames of a multiple of the wave tile's rows");
    }
    switch (tile_class) {
        case 4: return launch_gs<128, 192>(a, s);
        case 0: return launch_gs<128, 128>(a, s);
        case 1: return launch_gs<128, 64>(a, s);
        case 2: return launch_gs<64, 128>(a, s);
        default: return launch_gs<64, 64>(a, s);
    }
}

}  // namespace vd
