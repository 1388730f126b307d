// 3x3 stride-1 convolution by Winograd F(2x2,3x3) at fp32 accuracy on the bf16 matrix cores, gfx950.
//
//   Y(2x2) = A^T [ (G g G^T) .* (B^T d B) ] A        per (tile, cin, cout)                     (see conv_wino.hip)
//
// The element products of the Winograd domain are matrix products over (tile, cout) with K = cin.  Both operands are
// fp32: V = B^T d B is formed in fp32 registers from the fp32 patch, U = G g G^T in fp64 on the host.  Each is then
// split EXACTLY into three bf16 pieces (3 x 8 significand bits = 24; gemm_split.hip) and six of the nine piece
// products go through v_mfma_f32_32x32x16_bf16 with fp32 accumulation -- fp32 accuracy at 2.67x the rate of
// v_mfma_f32_32x32x2_f32, and the bf16 MFMA leaves the vector ALU free for the transform and the splitting.
//
// Block = 256 threads = 4 waves, one wave per SIMD; a block owns 64 tiles (16x16 output pixels of one frame, or 8x8
// pixels of four frames) x 32 couts; wave i owns Winograd row i: 4 positions x 2 M-tiles = 8 accumulator tiles.
// Per 16-channel chunk (= ONE MFMA k-step) and wave: 48 MFMAs in two groups (one per M-tile).  While a group's 24 MFMAs
// run, the vector ALU turns the NEXT group's raw patch values into fragments: row combination d[r0] + sg d[r1], column
// combination, three-way split (~176 VALU per group; this, not the matrix pipe, paces the loop).  The raw patch is the
// only LDS operand (double-buffered, zero padding by the buffer range check); the split weights stream from L2 as
// [Cin/16][16 positions][Cout/32][3 pieces][64 lanes][8 bf16], one coalesced 1 KiB load each, one chunk ahead;
// requests are issued one per MFMA slot (never in bursts: see conv_wino.hip).  Output transform as in conv_wino.hip.
#include <cstring>

#include "vd_common.h"

namespace vd {

constexpr int SKC = 16;          // channels per chunk = k of one bf16 MFMA
constexpr int SLD = 20;          // LDS pixel stride (floats)

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

struct WinoSGeom { int TF, tiles_x, tiles_y; };

#ifdef VD_WINO_TIMING
// kernel-experiment builds only (tools/wino_timing.py --split): shader-clock stamps of block 0, wave 0
__device__ unsigned long long g_winos_stamp[10];
#define WINOS_STAMP(i)                                                                                                 \
    do {                                                                                                               \
        if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) {                                                  \
            g_winos_stamp[i] = __builtin_readcyclecounter();                                                           \
            if (i == 0) g_winos_stamp[8] = __builtin_amdgcn_s_memrealtime();                                           \
            if (i == 3) g_winos_stamp[9] = __builtin_amdgcn_s_memrealtime();                                           \
        }                                                                                                              \
    } while (0)
extern "C" int vd_debug_winos_stamps(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_winos_stamp), sizeof(g_winos_stamp));
}
#else
#define WINOS_STAMP(i)
#endif
#ifndef VD_WINOS_SKIP
#define VD_WINOS_SKIP 0    // kernel-experiment builds: bit 0 no VALU pieces, 1 no weight loads, 2 no fragment reads, 3 no patch
#endif                     // loads/stores, 4 no barrier (results are then wrong; timing only)

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// x = p1 + p2 + p3 exactly, per pair of fp32 values, by truncation: p1 = top 16 bits of x (a bf16 value), r = x - p1
// (exact: the low 16 significand bits), p2 = top 16 bits of r, p3 = r - p2 (at most 8 significant bits: a bf16 value).
// The bf16 pieces of a pair are the high halves of two registers, packed by one v_perm_b32: 11 VALU per pair.
// Only PLAIN vector instructions may appear in this loop: v_pk_*_f32 and v_dot2c_f32_bf16 do not overlap the bf16 MFMA
// (each costs its 4 cycles plus a ~20-cycle bubble once per slot), plain fp32 / integer / v_cvt_pk_bf16_f32 ones do, up
// to 6 per MFMA for free and 4 cycles each beyond (tools/mfma_bf16_coissue.hip).  v_cvt_pk_bf16_f32 + v_dot2c would
// split a pair in 7 instructions (tools/dot2_probe.hip) but is slower here for that reason.
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned& p1, unsigned& p2, unsigned& p3, unsigned sel) {
    float h0, h1, r0, r1;
    asm("v_and_b32 %3, 0xffff0000, %7\n\t"
        "v_and_b32 %4, 0xffff0000, %8\n\t"
        "v_perm_b32 %0, %8, %7, %9\n\t"
        "v_sub_f32 %5, %7, %3\n\t"
        "v_sub_f32 %6, %8, %4\n\t"
        "v_and_b32 %3, 0xffff0000, %5\n\t"
        "v_and_b32 %4, 0xffff0000, %6\n\t"
        "v_perm_b32 %1, %6, %5, %9\n\t"
        "v_sub_f32 %5, %5, %3\n\t"
        "v_sub_f32 %6, %6, %4\n\t"
        "v_perm_b32 %2, %6, %5, %9"
        : "=&v"(p1), "=&v"(p2), "=&v"(p3), "=&v"(h0), "=&v"(h1), "=&v"(r0), "=&v"(r1)
        : "v"(x0), "v"(x1), "s"(sel));
}
template <bool TF4>
__global__ __launch_bounds__(256, 1) void conv3x3_wino_split_kernel(IgemmArgs a, WinoSGeom g) {
    constexpr int TTL = TF4 ? 2 : 3, TT = 1 << TTL, P = 2 * TT + 2;   // tiles per dim per frame, patch width
    constexpr int NX = TF4 ? 7 : 6;            // patch float4 per thread
    constexpr int SS = TF4 ? 16 : 64;          // patch-pixel step between a thread's staged elements
    // LDS image of a patch (bytes): pixels of even and odd x in two planes per row, 80 bytes per pixel (16 channels + 16
    // bytes of padding), rows RSB apart, frames FSB apart.  A fragment read takes the same (row, column) of 16 tiles per
    // lane group: neighbouring tiles are neighbouring pixels of ONE plane (5 slots of 16 bytes apart), and RSB / FSB are
    // chosen so that the tile rows / frames of a group land on the remaining slot classes: every ds_read_b128 of the
    // loop is bank-conflict free (the straightforward [pixel][20 floats] image was 2- to 4-way conflicted, and at the
    // bf16 MFMA rate the LDS was as busy as the matrix pipe).
    constexpr int PSB = 80, POB = (P / 2) * PSB;
    constexpr int RSB = TF4 ? 800 : 1472;      // >= 2*POB; = 2 (TF4) / 4 (else) mod 8 slots
    constexpr int FSB = TF4 ? 8192 : 0;
    constexpr int XBUF = (TF4 ? 4 * FSB : P * RSB) / 4;              // floats per patch buffer
    extern __shared__ __attribute__((aligned(16))) float smem[];      // [2][XBUF] + a dump slot; reused as Z exchange
    const int tid = threadIdx.x, lane = tid & 63;
    const int wi = __builtin_amdgcn_readfirstlane(tid >> 6);          // Winograd row owned by this wave (scalar)
    const int lr = lane & 31, lh = lane >> 5;
    WINOS_STAMP(0);
    int bx = blockIdx.x;
    const int bxx = bx % g.tiles_x; bx /= g.tiles_x;
    const int byy = bx % g.tiles_y; bx /= g.tiles_y;
    const int f0 = bx * (TF4 ? 4 : 1);
    const int ox0 = bxx * 2 * TT, oy0 = byy * 2 * TT;               // output-pixel origin of the block
    const int Hl = a.Hs << a.ups, Wl = a.Ws << a.ups;
    const int nchunk = a.Cin / SKC, ncoblk = a.Cout >> 5;
    const int cob = blockIdx.y;                                      // BN = 32 = one cout block

    // ---- patch staging (as conv_wino.hip): thread -> patch pixels sp0 + SS*e of its frame slot, channel quad lq
    const int lq = tid & 3;
    const int sf = TF4 ? wi : 0;
    const int sp0 = (TF4 ? lane : tid) >> 2;
    unsigned xo[NX];
#pragma unroll
    for (int e = 0; e < NX; ++e) {
        const int pl = sp0 + SS * e;
        const int py = pl / P, px = pl - py * P;
        const int ly = oy0 + py - 1, lx = ox0 + px - 1;
        const bool in = pl < P * P && f0 + sf < a.nfr && ly >= 0 && ly < Hl && lx >= 0 && lx < Wl;
        xo[e] = in ? (unsigned)(((f0 + sf) * a.Hs + (ly >> a.ups)) * a.Ws + (lx >> a.ups)) * (unsigned)(a.Cin * 4) + lq * 16u
                   : 0x80000000u;
    }
    const auto xsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.src0), 0, a.nfr * a.Hs * a.Ws * a.Cin * 4, 0x00020000);
    int xw[NX];                                                       // LDS float offset of each staged element
#pragma unroll
    for (int e = 0; e < NX; ++e) {
        const int pl = sp0 + SS * e;
        const int py = pl / P, px = pl - py * P;
        // in 16-byte units: the f32x4 indexing below keeps every LDS access provably aligned (a float-offset pointer
        // made hipcc fall back to ds_read2_b32 pairs)
        xw[e] = pl < P * P ? (sf * FSB + py * RSB + (px & 1) * POB + (px >> 1) * PSB) / 16 + lq
                           : 2 * (XBUF / 4) + (sf * (NX * SS - P * P) + pl - P * P) * 4 + lq;   // past the patch: own dump slot
    }
    f32x4 rx[2][NX];
    auto x_load_one = [&](int set, int chunk, int e) {
        rx[set][e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xsrc, xo[e], chunk * (SKC * 4), 0));
    };
    auto x_store = [&](int set, int buf, int e0, int e1) {
#pragma unroll
        for (int e = 0; e < NX; ++e)
            if (e >= e0 && e < e1) reinterpret_cast<f32x4*>(smem)[xw[e] + buf * (XBUF / 4)] = rx[set][e];
    };

    // ---- A fragments: lane (tile m*32+lr, k-half lh) holds 8 channels (8*lh ..) of 4 positions, each as 3 bf16 pieces.
    // Row wi of B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]], row 2 negated (and U's row 2): d[r0] + sg*d[r1]
    const int r0 = wi == 0 ? 0 : 1, r1 = wi == 3 ? 3 : 2;
    const float sg = wi == 1 ? 1.f : -1.f;
    int ab[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int t = m * 32 + lr;
        const int tx = t & (TT - 1), ty = (t >> TTL) & (TT - 1), f = t >> (2 * TTL);
        const int xb = (f * FSB + 2 * ty * RSB + tx * PSB) / 16 + lh * 2;          // 16-byte units
        ab[m][0] = xb + r0 * (RSB / 16); ab[m][1] = xb + r1 * (RSB / 16);
    }
    f32x4 raw[16];                                                   // [row r0/r1][column 4][channel half 2]
    auto a_read_one = [&](int buf, int m, int i) {
        const int c = (i >> 1) & 3;                                  // patch column 2*tx + c: plane c&1, pixel tx + (c>>1)
        raw[i] = reinterpret_cast<const f32x4*>(smem)[ab[m][i >> 3] + buf * (XBUF / 4) + ((c & 1) * POB + (c >> 1) * PSB) / 16 + (i & 1)];
    };
    // raw -> three bf16 pieces of the four fragments, in 24 pieces of work, one per MFMA slot of the running group:
    //   pieces 0..3    row combination of patch column c:  t[c] = d[r1][c]*sg + d[r0][c]             (4 packed fma)
    //   then per position j: the column combination (4 packed add) and the split of its four value pairs (11 VALU each)
    u32x4 apc[2][4][3];                                              // [slot][position j][piece] = 8 bf16
    float tc[4][8], fcur[8];                                         // [column][channel]
    auto a_piece = [&](int slot, int k) {
        if (k < 4) {                                                 // asm: hipcc would pair these into v_pk_fma_f32
            const f32x4 lo0 = raw[k * 2], hi0 = raw[k * 2 + 1], lo1 = raw[8 + k * 2], hi1 = raw[8 + k * 2 + 1];
            const float d0[8] = {lo0.x, lo0.y, lo0.z, lo0.w, hi0.x, hi0.y, hi0.z, hi0.w};
            const float d1[8] = {lo1.x, lo1.y, lo1.z, lo1.w, hi1.x, hi1.y, hi1.z, hi1.w};
#pragma unroll
            for (int h = 0; h < 8; ++h) asm("v_fma_f32 %0, %1, %2, %3" : "=v"(tc[k][h]) : "v"(d1[h]), "v"(sg), "v"(d0[h]));
        } else {
            const int j = (k - 4) / 5, s_ = (k - 4) % 5;
            if (s_ == 0) {
                // j0: t0 - t2   j1: t1 + t2   j2: t2 - t1   j3: t1 - t3
                const int ca = j == 0 ? 0 : j == 2 ? 2 : 1, cb = j == 3 ? 3 : j == 2 ? 1 : 2;
#pragma unroll
                for (int h = 0; h < 8; ++h) {
                    if (j == 1) asm("v_add_f32 %0, %1, %2" : "=v"(fcur[h]) : "v"(tc[ca][h]), "v"(tc[cb][h]));
                    else asm("v_sub_f32 %0, %1, %2" : "=v"(fcur[h]) : "v"(tc[ca][h]), "v"(tc[cb][h]));
                }
            } else {
                unsigned p1, p2, p3;
                split_pair(fcur[2 * (s_ - 1)], fcur[2 * (s_ - 1) + 1], p1, p2, p3, 0x07060302u);
                apc[slot][j][0][s_ - 1] = p1; apc[slot][j][1][s_ - 1] = p2; apc[slot][j][2][s_ - 1] = p3;
            }
        }
    };

    // ---- B fragments: U[chunk][xi = 4*wi + j][cob][piece][lane][8 bf16] = 1 KiB per (chunk, xi, cob, piece)
    const auto usrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wwino), 0, 16 * a.Cout * a.Cin * 6, 0x00020000);
    const int ustride = 16 * ncoblk * 3072, uwave = wi * 4 * ncoblk * 3072;
    unsigned bo[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) bo[j] = (unsigned)(j * ncoblk + cob) * 3072u + lane * 16u;
    bf16x8 bfr[2][4][3];                                             // [slot = chunk parity][j][piece]
    auto b_load_one = [&](int slot, int chunk, int i) {              // i = 0..11: (j, piece)
        bfr[slot][i / 3][i % 3] = __builtin_bit_cast(
            bf16x8, __builtin_amdgcn_raw_buffer_load_b128(usrc, bo[i / 3] + (i % 3) * 1024, chunk * ustride + uwave, 0));
    };

    f32x16 acc[2][4];                                                // [m][j]
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][j][r] = 0.f;

    // ---- pipeline.  Patch k lives in LDS buffer k&1 and reaches it through register set k&1:
    //   chunk c, group 0:  request patch c+3 (HBM -> registers)                      3 groups ahead of its LDS write
    //   chunk c, group 1:  write patch c+2 to LDS at the end of the group            one barrier later it is readable
    //   chunk c          :  every fragment read targets patch c+1: group (c,m) reads the raw values of group (c+1,m)
    //                       and turns those of the group after itself into bf16 pieces while its 24 MFMAs run
    //   weights of chunk c+1 arrive during chunk c.  One barrier per chunk, at its start.
    // prologue: patches 0, 1 in LDS, patch 2 in flight, weights 0, pieces of group (0,0), raw values of group (0,1)
#pragma unroll
    for (int e = 0; e < NX; ++e) { x_load_one(0, 0, e); x_load_one(1, min(1, nchunk - 1), e); }
#pragma unroll
    for (int i = 0; i < 12; ++i) b_load_one(0, 0, i);
    x_store(0, 0, 0, NX);
    x_store(1, 1, 0, NX);
#pragma unroll
    for (int e = 0; e < NX; ++e) x_load_one(0, min(2, nchunk - 1), e);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) a_read_one(0, 0, i);
#pragma unroll
    for (int k = 0; k < 24; ++k) a_piece(0, k);
#pragma unroll
    for (int i = 0; i < 16; ++i) a_read_one(0, 1, i);

    WINOS_STAMP(1);
    auto chunk_body = [&](int chunk, int buf) {                     // buf = chunk & 1
        const int n1 = min(chunk + 1, nchunk - 1), n3 = min(chunk + 3, nchunk - 1);
        if (!(VD_WINOS_SKIP & 16)) __syncthreads();                     // patch c+1 complete and visible; nobody reads buffer `buf` (patch c) any more
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x16 c = acc[m][j];
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    constexpr int PA[6] = {0, 1, 2, 0, 1, 0}, PB[6] = {2, 1, 0, 1, 0, 0};     // small terms first
                    const int k = j * 6 + q;                         // MFMA slot 0..23
                    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, apc[m][j][PA[q]]), bfr[buf][j][PB[q]], c, 0, 0, 0);
                    // one request per slot: weights of the next chunk (slots 0..3, 20, 21), raw values of the group after
                    // next (slots 4..19, after pieces 0..3 have consumed the previous ones), patch c+3 (group 0, 4..)
                    if (k < 4) { if (!(VD_WINOS_SKIP & 2)) b_load_one(buf ^ 1, n1, m * 6 + k); }
                    else if (k < 20) { if (!(VD_WINOS_SKIP & 4)) a_read_one(buf ^ 1, m, k - 4); }
                    else if (k < 22) { if (!(VD_WINOS_SKIP & 2)) b_load_one(buf ^ 1, n1, m * 6 + 4 + (k - 20)); }
                    if (m == 0 && k >= 4 && k < 4 + NX && !(VD_WINOS_SKIP & 8)) x_load_one(buf ^ 1, n3, k - 4);
                    if (!(VD_WINOS_SKIP & 1)) a_piece(m ^ 1, k);    // pieces of the next group
                    __builtin_amdgcn_sched_barrier(0);
                }
                acc[m][j] = c;
            }
            if (m == 1 && !(VD_WINOS_SKIP & 8)) x_store(buf, buf, 0, NX);   // patch c+2 -> LDS
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    for (int chunk = 0; chunk < nchunk; chunk += 2) {               // nchunk is even (conv_wino_split_supported)
        chunk_body(chunk, 0);
        chunk_body(chunk + 1, 1);
    }

    WINOS_STAMP(2);
    // ---- output transform (conv_wino.hip): Z[q] = sum_j M[wi][j] A[j][q] wave-local, sum over i through LDS, wave
    // (p, q) owns output pixel (p, q) of every tile; branch-free via buffer range checks
    const int p = wi >> 1, q = wi & 1;
    const int co = blockIdx.y * 32 + lr;
    unsigned oo[2][16];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int t = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int tx = t & (TT - 1), ty = (t >> TTL) & (TT - 1), f = t >> (2 * TTL);
            const int nf = f0 + f;
            const unsigned o = (unsigned)(((nf * Hl + oy0 + 2 * ty + p) * Wl + ox0 + 2 * tx + q) * a.ldo + co) * 4u;
            oo[m][r] = nf < a.nfr ? o : 0x80000000u;
        }
    const int obytes = a.nfr * Hl * Wl * a.ldo * 4;
    const auto osrc = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, obytes, 0x00020000);
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.res ? a.res : a.out), 0, a.res ? obytes : 0, 0x00020000);
    f32x16 rv[2];
    const float bv = a.bias ? a.bias[co] : 0.f;
    __syncthreads();                                                 // every wave is done with the patch buffers
    // Z exchange layout in LDS: [plane = 2i + q][m 2][reg16/4][lane 64][4]  (8 planes of 8 KB)
    float* Zs = smem;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
#pragma unroll
        for (int r = 0; r < 16; ++r) rv[m][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, oo[m][r], 0, 0));
        const f32x16 z0 = acc[m][0] + acc[m][1] + acc[m][2];
        const f32x16 z1 = acc[m][1] - acc[m][2] - acc[m][3];
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
            float* d0 = Zs + ((((wi * 2 + 0) * 2 + m) * 4 + c4) * 64 + lane) * 4;
            float* d1 = Zs + ((((wi * 2 + 1) * 2 + m) * 4 + c4) * 64 + lane) * 4;
            *reinterpret_cast<f32x4*>(d0) = f32x4{z0[4 * c4], z0[4 * c4 + 1], z0[4 * c4 + 2], z0[4 * c4 + 3]};
            *reinterpret_cast<f32x4*>(d1) = f32x4{z1[4 * c4], z1[4 * c4 + 1], z1[4 * c4 + 2], z1[4 * c4 + 3]};
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    const float sgn = p ? -1.f : 1.f;
    const float* zw = Zs + wi * 2048 + lane * 4;                     // Z[p + k][q] is plane wi + 2k
    float gsum[TF4 ? 4 : 1][2] = {};
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        f32x16 y;
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
            const float* zp = zw + (m * 4 + c4) * 256;
            const f32x4 v = *reinterpret_cast<const f32x4*>(zp) +
                            (*reinterpret_cast<const f32x4*>(zp + 2 * 2048) + *reinterpret_cast<const f32x4*>(zp + 4 * 2048)) * sgn;
            y[4 * c4] = v.x; y[4 * c4 + 1] = v.y; y[4 * c4 + 2] = v.z; y[4 * c4 + 3] = v.w;
        }
        y += rv[m];
        if (a.fbias) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int t = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                y[r] += a.fbias[(size_t)min(f0 + (t >> (2 * TTL)), a.nfr - 1) * a.fbias_ld + co];
            }
        }
        y += bv;
#pragma unroll
        for (int r = 0; r < 16; ++r) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, (float)y[r]), osrc, oo[m][r], 0, 0);
        if (a.stats) {
#pragma unroll
            for (int h = 0; h < (TF4 ? 2 : 1); ++h) {
                float s = 0.f, ss = 0.f;
#pragma unroll
                for (int r = h * (TF4 ? 8 : 0); r < (TF4 ? 8 * h + 8 : 16); ++r) { s += y[r]; ss += y[r] * y[r]; }
                const int fs = TF4 ? 2 * m + h : 0;
                gsum[fs][0] += s; gsum[fs][1] += ss;
            }
        }
    }
    if (a.stats) {                                                   // GroupNorm partial sums of the output (conv_wino.hip)
        constexpr int NFS = TF4 ? 4 : 1;
        __syncthreads();
        double* red = reinterpret_cast<double*>(smem);               // [wave 4][lh 2][fs][lr 32][2]
#pragma unroll
        for (int fs = 0; fs < NFS; ++fs) {
            double* d = red + ((((wi * 2 + lh) * NFS + fs) * 32 + lr) * 2);
            d[0] = (double)gsum[fs][0]; d[1] = (double)gsum[fs][1];
        }
        __syncthreads();
        if (tid < NFS * 32) {
            const int fs = tid >> 5, c = tid & 31;
            double s = 0.0, ss = 0.0;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const double* d = red + (((k * NFS + fs) * 32 + c) * 2);
                s += d[0]; ss += d[1];
            }
            const int nf = f0 + fs;
            const int sp = TF4 ? 0 : byy * g.tiles_x + bxx;
            if (nf < a.nfr) {
                double* o = a.stats + (((size_t)nf * a.stats_split + sp) * a.Cout + blockIdx.y * 32 + c) * 2;
                o[0] = s; o[1] = ss;
            }
        }
    }
    WINOS_STAMP(3);
}

static bool wsp_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

bool conv_wino_split_supported(const IgemmArgs& a) {
    const int Hl = a.Hs << a.ups, Wl = a.Ws << a.ups;
    return a.wsplit == 1 && a.wwino != nullptr && a.ksz == 3 && a.stride == 1 && a.pad == 1 && Hl == Wl && wsp_pow2(Hl) && Hl >= 8 &&
           a.Cout % 32 == 0 && a.Cin % (2 * SKC) == 0 && a.src1 == nullptr && a.C0 == a.Cin && a.affA == nullptr && a.act == 0 &&
           (size_t)a.nfr * a.Hs * a.Ws * a.Cin < (1u << 29) && (size_t)a.Cin * a.Cout * 96 < (1u << 31) &&
           (size_t)a.nfr * Hl * Wl * a.ldo < (1u << 29) && (a.res == nullptr || a.res_ld == a.ldo);
}

int launch_conv_wino_split(const IgemmArgs& a, hipStream_t s) {
    const int Hl = a.Hs << a.ups;
    VD_REQUIRE(a.stats == nullptr || a.stats_split == conv_wino_stats_split(Hl), "GroupNorm partial table: split");
    WinoSGeom g;
    const int TT = Hl >= 16 ? 8 : 4;
    g.TF = 64 / (TT * TT);
    g.tiles_x = Hl / (2 * TT); g.tiles_y = Hl / (2 * TT);
    const size_t lds = std::max((size_t)3 * (g.TF == 4 ? 4 * 8192 : 18 * 1472) + 64, (size_t)8 * 2048 * sizeof(float));   // 2 buffers + dump
    static bool attr = false;
    if (!attr) {
        const void* fns[2] = {reinterpret_cast<const void*>(&conv3x3_wino_split_kernel<true>),
                              reinterpret_cast<const void*>(&conv3x3_wino_split_kernel<false>)};
        for (const void* f : fns) VD_HIP(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = true;
    }
    const int fgroups = (a.nfr + g.TF - 1) / g.TF;
    dim3 grid(g.tiles_x * g.tiles_y * fgroups, a.Cout / 32);
    if (g.TF == 4) hipLaunchKernelGGL((conv3x3_wino_split_kernel<true>), grid, dim3(256), lds, s, a, g);
    else hipLaunchKernelGGL((conv3x3_wino_split_kernel<false>), grid, dim3(256), lds, s, a, g);
    VD_HIP(hipGetLastError());
    return 0;
}

// host: U = G g G^T (fp64, rounded once to fp32, row 2 negated), split into three bf16 pieces, packed
// [Cin/16][xi 16][Cout/32][piece 3][lane 64][8]: lane 32h+r holds U[xi][co = 32*blk + r][ci = 16*chunk + 8*h + e]
void split3_host(float v, unsigned short out[3]);
void pack_conv3_wino_split(const float* oihw, unsigned short* out, int O, int I) {
    static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    const int ncoblk = O / 32;
    for (int co = 0; co < O; ++co)
        for (int ci = 0; ci < I; ++ci) {
            const float* gk = oihw + ((size_t)co * I + ci) * 9;
            double tmp[4][3], U[4][4];
            for (int i = 0; i < 4; ++i)
                for (int c = 0; c < 3; ++c) tmp[i][c] = G[i][0] * gk[0 * 3 + c] + G[i][1] * gk[1 * 3 + c] + G[i][2] * gk[2 * 3 + c];
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) U[i][j] = tmp[i][0] * G[j][0] + tmp[i][1] * G[j][1] + tmp[i][2] * G[j][2];
            const int ch = ci / SKC, k = ci % SKC, h = k >> 3, e = k & 7;
            const int cb = co >> 5, r = co & 31;
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) {
                    unsigned short pc[3];
                    split3_host((float)(i == 2 ? -U[i][j] : U[i][j]), pc);
                    for (int q3 = 0; q3 < 3; ++q3)
                        out[(((((size_t)ch * 16 + i * 4 + j) * ncoblk + cb) * 3 + q3) * 64 + h * 32 + r) * 8 + e] = pc[q3];
                }
        }
}

}  // namespace vd
