// 3x3 stride-1 convolution by Winograd F(2x2,3x3) on fp32 MFMA, gfx950.
//
//   Y(2x2) = A^T [ (G g G^T) .* (B^T d B) ] A        per (tile, cin, cout); d = 4x4 input patch, g = 3x3 filter
//
// 16 multiplies per 2x2 outputs instead of 36: 2.25x fewer MFMAs than the direct kernel (conv_halo.hip), all
// arithmetic still fp32 (the transforms are +-1 additions; G g G^T is formed in fp64 on the host, rounded once).
//
// Block = 256 threads = 4 waves, ONE wave per SIMD with the whole 512-entry register file:
//   * a block owns 64 tiles (16x16 output pixels of one frame, or 8x8 pixels of four frames) x 64 couts;
//   * wave i (0..3) owns Winograd row i: the 4 positions xi = (i, j), j = 0..3, for 2 M-tiles x 2 N-tiles
//     -> 16 accumulator tiles = 256 VGPRs;
//   * the activated input patch of a 16-channel chunk ((2T+2)^2 pixels, GroupNorm/FiLM affine + SiLU applied,
//     zero padded, read through upsample / virtual concat) is double-buffered in LDS -- the ONLY LDS operand;
//   * a lane (tile r, k-half h) builds its A fragments in registers: row i of B^T touches 2 patch rows, so
//     8 ds_read_b128 + 4 adds give t[0..3], 4 more adds give V[i][0..3] = the fragments of 4 positions;
//   * B fragments (transformed weights) stream from L2 in fragment order [chunk][xi][cout/32][kg][lane][4],
//     one coalesced 1 KiB load each, one k-group ahead;
//   * one barrier per chunk (128 MFMAs per wave).
//   Output: Z[i][q] = sum_j M[i][j] A[j][q] is wave-local; the sum over i crosses waves through LDS once per block;
//   wave (p,q) then owns output pixel (p,q) of every tile and adds bias + residual.
#include "vd_common.h"

namespace vd {

constexpr int WKC = 16;          // channels per chunk
constexpr int WLD = 20;          // LDS pixel stride (floats)

struct WinoGeom {
    int TF;                      // frames per block (1 or 4)
    int tiles_x, tiles_y;        // blocks per frame
};

#ifdef VD_WINO_TIMING
// kernel-experiment builds only (tools/wino_timing.py): shader-clock stamps of the first and the last block, wave 0
__device__ unsigned long long g_wino_stamp[8];
#define WINO_STAMP(i)                                                                                                  \
    do {                                                                                                               \
        if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) g_wino_stamp[i] = __builtin_readcyclecounter();    \
        if (threadIdx.x == 0 && blockIdx.x == gridDim.x - 1 && blockIdx.y == gridDim.y - 1)                             \
            g_wino_stamp[4 + i] = __builtin_readcyclecounter();                                                        \
    } while (0)
extern "C" int vd_debug_wino_stamps(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_wino_stamp), sizeof(g_wino_stamp));
}
#else
#define WINO_STAMP(i)
#endif
#ifndef VD_WINO_SKIP
#define VD_WINO_SKIP 0     // kernel-experiment builds: bit 0 no patch staging, 1 no fragment transform, 2 no weight loads,
#endif                     // 3 no barrier, 4 no patch loads, 5 no fragment reads (results are then wrong; timing only)

__device__ __forceinline__ void pin(f32x4& v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }
__device__ __forceinline__ void pin1(float& v) { asm volatile("" : "+v"(v)); }

// PRO: operand prologue (affine + SiLU) compiled in/out; TF4: four 8x8 frames per block (else one frame, 16x16 pixels).
template <bool PRO, bool TF4>
__global__ __launch_bounds__(256, 1) void conv3x3_wino_kernel(IgemmArgs a, WinoGeom g) {
    constexpr int TTL = TF4 ? 2 : 3, TT = 1 << TTL, P = 2 * TT + 2;   // tiles per dim per frame, patch width
    constexpr int NX = TF4 ? 7 : 6;            // patch float4 per thread
    constexpr int SS = TF4 ? 16 : 64;          // patch-pixel step between a thread's staged elements
    constexpr int FS = TF4 ? 112 : 384;        // LDS pixels per frame slot (>= P*P, = NX * SS)
    constexpr int XBUF = (TF4 ? 4 : 1) * FS * WLD;
    extern __shared__ __attribute__((aligned(16))) float smem[];      // [2][XBUF]; reused as Z exchange at the end
    const int tid = threadIdx.x, lane = tid & 63;
    const int wi = __builtin_amdgcn_readfirstlane(tid >> 6);          // Winograd row owned by this wave (scalar)
    const int lr = lane & 31, lh = lane >> 5;
    WINO_STAMP(0);
    int bx = blockIdx.x;
    const int bxx = bx % g.tiles_x; bx /= g.tiles_x;
    const int byy = bx % g.tiles_y; bx /= g.tiles_y;
    const int f0 = bx * (TF4 ? 4 : 1);
    const int ox0 = bxx * 2 * TT, oy0 = byy * 2 * TT;               // output-pixel origin of the block
    const int Hl = a.Hs << a.ups, Wl = a.Ws << a.ups;
    const int C1 = a.Cin - a.C0;
    const int nchunk = a.Cin / WKC, ncoblk = a.Cout >> 5;
    const int cob0 = blockIdx.y * 2;                                 // BN = 64 = 2 cout blocks

    // ---- patch staging.  TF4: wave f stages frame f (one affine pair per thread); else the block's one frame.
    // thread -> patch pixels sp0 + SS*e of its frame slot, channel quad lq
    const int lq = tid & 3;
    const int sf = TF4 ? wi : 0;
    const int sp0 = (TF4 ? lane : tid) >> 2;
    const int n_st = min(f0 + sf, a.nfr - 1);
    int so[NX];                                                       // source pixel index (clamped)
    unsigned zmask = 0;                                               // bit e: element is zero padding
#pragma unroll
    for (int e = 0; e < NX; ++e) {
        const int pl = sp0 + SS * e;
        const int py = pl / P, px = pl - py * P;
        const int ly = oy0 + py - 1, lx = ox0 + px - 1;
        const bool in = pl < P * P && f0 + sf < a.nfr && ly >= 0 && ly < Hl && lx >= 0 && lx < Wl;
        so[e] = in ? (n_st * a.Hs + (ly >> a.ups)) * a.Ws + (lx >> a.ups) : 0;
        if (!in) zmask |= 1u << e;
    }
    const int xw = (sf * FS + sp0) * WLD + lq * 4;                   // LDS float offset of element 0
    f32x4 rx[NX], affa, affb;
    auto x_load = [&](int chunk) {
        const int c = chunk * WKC;                                    // uniform: scalar base + 32-bit lane offsets
        const bool second = c >= a.C0;
        const char* base = reinterpret_cast<const char*>(second ? a.src1 + (c - a.C0) : a.src0 + c);
        const int ld = second ? C1 : a.C0;
#pragma unroll
        for (int e = 0; e < NX; ++e)
            rx[e] = *reinterpret_cast<const f32x4*>(base + (unsigned)(so[e] * ld + lq * 4) * 4u);
        if constexpr (PRO) {
            const unsigned ao = (unsigned)(n_st * a.Cin + lq * 4) * 4u;
            affa = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(a.affA + c) + ao);
            affb = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(a.affB + c) + ao);
        }
    };
    // whole elements (prologue of the block) ...
    auto x_store = [&](float* Xd) {
#pragma unroll
        for (int e = 0; e < NX; ++e) {
            f32x4 v = rx[e];                           // no pixel-range branch: the buffers are padded to NX*SS pixels
            if constexpr (PRO) {
                v = v * affa + affb;
                v.x = silu_f(v.x); v.y = silu_f(v.y); v.z = silu_f(v.z); v.w = silu_f(v.w);
            }
            if ((zmask >> e) & 1) v = f32x4{0.f, 0.f, 0.f, 0.f};      // zero padding AFTER norm + activation
            *reinterpret_cast<f32x4*>(Xd + xw + e * SS * WLD) = v;
        }
    };
    // ... and ONE component of one element, the unit of staging work placed in an MFMA gap of the main loop (issue
    // cost ~44 cycles of the 56 a 64-cycle MFMA leaves: fma 4, mul 4, exp 8, add 4, rcp 8, mul 4, select 4, + hazards).
    // pin() on the way in and out nails the arithmetic to this point of the program: pure VALU work is otherwise
    // hoisted to where its inputs are born (guide 5.7 item 3).
    f32x4 sv;
    auto x_piece = [&](float* Xd, int e, int c) {
        if constexpr (PRO) {
            float x = rx[e][c];
            pin1(x);
            x = silu_f(x * affa[c] + affb[c]);
            if ((zmask >> e) & 1) x = 0.f;
            pin1(x);
            sv[c] = x;
            if (c == 3) *reinterpret_cast<f32x4*>(Xd + xw + e * SS * WLD) = sv;
        } else if (c == 0) {                           // pre-activated input (the engine's path): select + store only
            f32x4 v = rx[e];
            pin(v);
            if ((zmask >> e) & 1) v = f32x4{0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4*>(Xd + xw + e * SS * WLD) = v;
        }
    };

    // ---- A fragments: lane (tile m*32+lr, k-half lh); row wi of B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]].
    // Row 2 is taken negated (d1 - d2; pack_conv3_wino negates U's row 2 to match) so that every row is
    // d[r0] + sg * d[r1]: one fma per value
    const int r0 = wi == 0 ? 0 : 1, r1 = wi == 3 ? 3 : 2;
    const float sg = wi == 1 ? 1.f : -1.f;
    int xb[2];                                                      // LDS float offset of the tile's patch origin
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int t = m * 32 + lr;
        const int tx = t & (TT - 1), ty = (t >> TTL) & (TT - 1), f = t >> (2 * TTL);
        xb[m] = (f * FS + 2 * ty * P + 2 * tx) * WLD + lh * 4;
    }
    const int rowo0 = r0 * P * WLD, rowo1 = r1 * P * WLD;
    f32x4 raw[2][8], frag[2][4];
    auto a_read = [&](int slot, const float* X, int m, int kg) {
        const float* p = X + xb[m] + kg * 8;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            raw[slot][j] = *reinterpret_cast<const f32x4*>(p + rowo0 + j * WLD);
            raw[slot][4 + j] = *reinterpret_cast<const f32x4*>(p + rowo1 + j * WLD);
        }
    };
    auto a_transform = [&](int slot) {          // raw[slot] -> frag[slot]
        f32x4 t[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] = raw[slot][4 + j] * sg + raw[slot][j];
        frag[slot][0] = t[0] - t[2]; frag[slot][1] = t[1] + t[2]; frag[slot][2] = t[2] - t[1]; frag[slot][3] = t[1] - t[3];
    };
    // the same in six pieces (4 + 4 + 4 + 4 + 8 + 8 VALU): 0..3 row sums, 4..5 the fragments
    f32x4 tcol[4];
    auto a_piece = [&](int slot, int i) {
        if (i < 4) {
            pin(raw[slot][i]); pin(raw[slot][4 + i]);
            tcol[i] = raw[slot][4 + i] * sg + raw[slot][i];
            pin(tcol[i]);
        } else if (i == 4) {
            frag[slot][0] = tcol[0] - tcol[2]; frag[slot][1] = tcol[1] + tcol[2];
            pin(frag[slot][0]); pin(frag[slot][1]);
        } else {
            frag[slot][2] = tcol[2] - tcol[1]; frag[slot][3] = tcol[1] - tcol[3];
            pin(frag[slot][2]); pin(frag[slot][3]);
        }
    };

    // ---- B fragments: U[chunk][xi = 4*wi + j][cob][kg][lane][4]: scalar base per (chunk, j, n) + lane offset
    const int cobn[2] = {cob0, min(cob0 + 1, ncoblk - 1)};
    const unsigned ulane = lane * 16u;
    f32x4 bfr[2][4][2];
    auto b_load = [&](int slot, int chunk, int kg) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const float* ub = a.wwino + ((((size_t)chunk * 16 + wi * 4 + j) * ncoblk + cobn[n]) * 2 + kg) * 256;
                bfr[slot][j][n] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(ub) + ulane);
            }
    };

    f32x16 acc[2][4][2];                                            // [m][j][n]
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][j][n][r] = 0.f;

    // ---- software pipeline over the 4 MFMA groups (kg, m) = (0,0) (0,1) (1,0) (1,1) of each chunk; with one wave per
    // SIMD nothing else hides a stall, so every operand is requested >= 2 groups (>= 4096 matrix-pipe cycles) ahead:
    //   patch rows (HBM)      x_load(c+2)   before group 2 of chunk c   -> staged to LDS in groups 0,1 of chunk c+1
    //   weights (L2)          b_load        two groups before use
    //   A fragments (LDS)     a_read        two groups before use, transformed one group before use
    // The one barrier per chunk sits between groups 1 and 2: the next patch is complete and nobody reads the current
    // one any more, so groups 2,3 already fetch the next chunk's first fragments.
    x_load(0);
    b_load(0, 0, 0);
    x_store(smem);
    x_load(min(1, nchunk - 1));
    __syncthreads();
    a_read(0, smem, 0, 0);
    a_read(1, smem, 1, 0);
    a_transform(0);
    WINO_STAMP(1);

    for (int chunk = 0; chunk < nchunk; ++chunk) {
        const int n1 = min(chunk + 1, nchunk - 1), n2 = min(chunk + 2, nchunk - 1);
        const float* Xc = smem + (chunk & 1) * XBUF;
        float* Xn = smem + ((chunk + 1) & 1) * XBUF;
#pragma unroll
        for (int gi = 0; gi < 4; ++gi) {
            const int kg = gi >> 1, m = gi & 1;
            if (gi == 0 && !(VD_WINO_SKIP & 4)) b_load(1, chunk, 1);
            if (gi == 2) {
                if (!(VD_WINO_SKIP & 8)) __syncthreads();
                if (!(VD_WINO_SKIP & 4)) b_load(0, n1, 0);
                if (!(VD_WINO_SKIP & 16)) x_load(n2);
            }
            if (!(VD_WINO_SKIP & 32)) {
                if (gi < 2) a_read(gi & 1, Xc, gi & 1, 1);         // group gi+2 of this chunk
                else a_read(gi & 1, Xn, gi & 1, 0);                // group gi-2 of the next chunk
            }
            __builtin_amdgcn_sched_barrier(0);
            // this group's 32 MFMAs, each followed by one hand-placed piece of VALU work and a full scheduling barrier:
            // with one wave per SIMD nothing else fills the matrix pipe, so the next group's fragment transform (even
            // gaps 2..12) and, in groups 0,1, half of the next chunk's patch staging (odd gaps) sit in the MFMA shadows
            constexpr int XH = (NX + 1) / 2;                        // patch elements staged in group 0; the rest in group 1
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                const int e = k >> 3, j = (k >> 1) & 3, n = k & 1;
                acc[m][j][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(frag[gi & 1][j][e], bfr[kg][j][n][e], acc[m][j][n], 0, 0, 0);
                if ((k & 1) == 0 && k >= 2 && k <= 12 && !(VD_WINO_SKIP & 2)) a_piece((gi + 1) & 1, (k - 2) >> 1);
                if ((k & 1) == 1 && gi < 2) {
                    const int i = k >> 1;                           // 0..15: component i&3 of element i>>2 of this half
                    const int el = (gi == 0 ? 0 : XH) + (i >> 2);
                    if (el < (gi == 0 ? XH : NX) && !(VD_WINO_SKIP & 1)) x_piece(Xn, el, i & 3);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    WINO_STAMP(2);

    // ---- output transform.  A^T = [[1,1,1,0],[0,1,-1,-1]].  Wave-local: Z[q] = sum_j M[wi][j] A[j][q]; the sum over
    // i crosses waves through LDS; wave (p, q) then owns output pixel (p, q) of every tile.
    const int p = wi >> 1, q = wi & 1;
    const int co0 = blockIdx.y * 64 + lr;
    int opix[2][16];
    unsigned okm[2] = {0, 0};
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int t = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;        // C/D row = tile
            const int tx = t & (TT - 1), ty = (t >> TTL) & (TT - 1), f = t >> (2 * TTL);
            const int nf = f0 + f;
            if (nf < a.nfr) okm[m] |= 1u << r;
            opix[m][r] = (min(nf, a.nfr - 1) * Hl + oy0 + 2 * ty + p) * Wl + ox0 + 2 * tx + q;
        }
    // the residual rows are requested before the exchange so their latency hides behind it
    f32x16 rv[2][2];
    if (a.res) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    rv[m][n][r] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.res) +
                                                                  (unsigned)(opix[m][r] * a.res_ld + co0 + n * 32) * 4u);
    }
    __syncthreads();                                                 // every wave is done with the patch buffers
    // Z exchange layout in LDS: [i 4][q 2][m 2][n 2][reg16/4][lane 64][4]  (128 KB)
    float* Zs = smem;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const f32x16 z0 = acc[m][0][n] + acc[m][1][n] + acc[m][2][n];
            const f32x16 z1 = acc[m][1][n] - acc[m][2][n] - acc[m][3][n];
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) {
                float* d0 = Zs + (((((wi * 2 + 0) * 2 + m) * 2 + n) * 4 + c4) * 64 + lane) * 4;
                float* d1 = Zs + (((((wi * 2 + 1) * 2 + m) * 2 + n) * 4 + c4) * 64 + lane) * 4;
                *reinterpret_cast<f32x4*>(d0) = f32x4{z0[4 * c4], z0[4 * c4 + 1], z0[4 * c4 + 2], z0[4 * c4 + 3]};
                *reinterpret_cast<f32x4*>(d1) = f32x4{z1[4 * c4], z1[4 * c4 + 1], z1[4 * c4 + 2], z1[4 * c4 + 3]};
            }
        }
    __syncthreads();
    // wave (p, q): Y[p][q] = sum_i A^T[p][i] Z[i][q]; p = 0: Z0+Z1+Z2, p = 1: Z1-Z2-Z3
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            f32x16 y;
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) {
                auto zl = [&](int i) {
                    return *reinterpret_cast<const f32x4*>(Zs + (((((i * 2 + q) * 2 + m) * 2 + n) * 4 + c4) * 64 + lane) * 4);
                };
                const f32x4 v = p == 0 ? zl(0) + zl(1) + zl(2) : zl(1) - zl(2) - zl(3);
                y[4 * c4] = v.x; y[4 * c4 + 1] = v.y; y[4 * c4 + 2] = v.z; y[4 * c4 + 3] = v.w;
            }
            const int co = co0 + n * 32;
            const float bv = a.bias ? a.bias[co] : 0.f;
            if (a.res) y += rv[m][n];
            if (a.fbias) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int t = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    y[r] += a.fbias[(size_t)min(f0 + (t >> (2 * TTL)), a.nfr - 1) * a.fbias_ld + co];
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if ((okm[m] >> r) & 1)
                    *reinterpret_cast<float*>(reinterpret_cast<char*>(a.out) + (unsigned)(opix[m][r] * a.ldo + co) * 4u) = y[r] + bv;
        }
    WINO_STAMP(3);
}

static bool wino_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

bool conv_wino_supported(const IgemmArgs& a) {
    const int Hl = a.Hs << a.ups, Wl = a.Ws << a.ups;
    return a.wwino != nullptr && a.ksz == 3 && a.stride == 1 && a.pad == 1 && Hl == Wl && wino_pow2(Hl) && Hl >= 8 &&
           a.Cout % 64 == 0 && a.Cin % WKC == 0 && a.C0 % WKC == 0 && (a.affA != nullptr) == (a.act != 0) &&
           // 32-bit byte offsets into the sources / residual / output
           (size_t)a.nfr * a.Hs * a.Ws * std::max(a.C0, a.Cin - a.C0) < (1u << 30) && (size_t)a.nfr * a.Cin < (1u << 30) &&
           (size_t)a.nfr * Hl * Wl * std::max(a.ldo, a.res ? a.res_ld : 0) < (1u << 30);
}

int launch_conv_wino(const IgemmArgs& a, hipStream_t s) {
    const int Hl = a.Hs << a.ups;
    WinoGeom g;
    const int TT = Hl >= 16 ? 8 : 4;                   // tiles per dim per frame in a block
    g.TF = 64 / (TT * TT);                             // 1 or 4 frames
    g.tiles_x = Hl / (2 * TT); g.tiles_y = Hl / (2 * TT);
    const size_t lds = std::max((size_t)2 * (g.TF == 4 ? 7 : 6) * 64 * WLD * sizeof(float), (size_t)4 * 2 * 2 * 2 * 4 * 64 * 4 * sizeof(float));
    static bool attr = false;
    if (!attr) {
        const void* fns[4] = {reinterpret_cast<const void*>(&conv3x3_wino_kernel<true, true>),
                              reinterpret_cast<const void*>(&conv3x3_wino_kernel<true, false>),
                              reinterpret_cast<const void*>(&conv3x3_wino_kernel<false, true>),
                              reinterpret_cast<const void*>(&conv3x3_wino_kernel<false, false>)};
        for (const void* f : fns) VD_HIP(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = true;
    }
    const int fgroups = (a.nfr + g.TF - 1) / g.TF;
    dim3 grid(g.tiles_x * g.tiles_y * fgroups, a.Cout / 64);
    const bool tf4 = g.TF == 4;
    if (a.affA) {
        if (tf4) hipLaunchKernelGGL((conv3x3_wino_kernel<true, true>), grid, dim3(256), lds, s, a, g);
        else hipLaunchKernelGGL((conv3x3_wino_kernel<true, false>), grid, dim3(256), lds, s, a, g);
    } else {
        if (tf4) hipLaunchKernelGGL((conv3x3_wino_kernel<false, true>), grid, dim3(256), lds, s, a, g);
        else hipLaunchKernelGGL((conv3x3_wino_kernel<false, false>), grid, dim3(256), lds, s, a, g);
    }
    VD_HIP(hipGetLastError());
    return 0;
}

// host: U = G g G^T (fp64, rounded once), packed [Cin/16][xi 16][Cout/32][kg 2][lane 64][4]:
//   lane 32h+r of k-group kg holds U[xi][co = 32*blk + r][ci = 16*chunk + 8*kg + 4*h + e]
void pack_conv3_wino(const float* oihw, float* out, int O, int I) {
    static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    const int nchunk = I / WKC, ncoblk = O / 32;
    for (int co = 0; co < O; ++co)
        for (int ci = 0; ci < I; ++ci) {
            const float* gk = oihw + ((size_t)co * I + ci) * 9;
            double tmp[4][3], U[4][4];
            for (int i = 0; i < 4; ++i)
                for (int c = 0; c < 3; ++c) tmp[i][c] = G[i][0] * gk[0 * 3 + c] + G[i][1] * gk[1 * 3 + c] + G[i][2] * gk[2 * 3 + c];
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) U[i][j] = tmp[i][0] * G[j][0] + tmp[i][1] * G[j][1] + tmp[i][2] * G[j][2];
            const int ch = ci / WKC, k = ci % WKC, kg = k >> 3, h = (k >> 2) & 1, e = k & 3;
            const int cb = co >> 5, r = co & 31;
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j)
                    out[((((((size_t)ch * 16 + i * 4 + j) * ncoblk + cb) * 2 + kg) * 64) + h * 32 + r) * 4 + e] =
                        (float)(i == 2 ? -U[i][j] : U[i][j]);       // row 2 negated: the kernel's B^T row 2 is negated too
        }
}

}  // namespace vd
