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
// kernel-experiment builds only (tools/wino_timing.py): shader-clock stamps of block 0, wave 0
__device__ unsigned long long g_wino_stamp[10];         // [8], [9]: constant-rate (100 MHz) clock at stamps 0 and 3
#define WINO_STAMP(i)                                                                                                  \
    do {                                                                                                               \
        if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) {                                                  \
            g_wino_stamp[i] = __builtin_readcyclecounter();                                                            \
            if (i == 0) g_wino_stamp[8] = __builtin_amdgcn_s_memrealtime();                                            \
            if (i == 3) g_wino_stamp[9] = __builtin_amdgcn_s_memrealtime();                                            \
        }                                                                                                              \
    } while (0)
extern "C" int vd_debug_wino_stamps(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_wino_stamp), sizeof(g_wino_stamp));
}
#else
#define WINO_STAMP(i)
#endif
#if defined(VD_WINO_TIMING) && VD_WINO_TIMING == 2
// per-group split of the main loop (block 0, wave 0): [issue + 32 MFMAs] and [transform + patch writes] of groups 0..3
__device__ unsigned long long g_wino_seg[32];          // [wave][8], a block in the middle of the grid
extern "C" int vd_debug_wino_segments(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_wino_seg), sizeof(g_wino_seg));
}
#define WINO_SEG_DECL unsigned long long seg_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, segt_ = 0
#define WINO_SEG_START segt_ = __builtin_readcyclecounter()
#define WINO_SEG(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); seg_[i] += now_ - segt_; segt_ = now_; } while (0)
#define WINO_SEG_FLUSH do { if ((threadIdx.x & 63) == 0 && blockIdx.x == gridDim.x / 2 + 3 && blockIdx.y == 0) for (int i_ = 0; i_ < 8; ++i_) g_wino_seg[(threadIdx.x >> 6) * 8 + i_] = seg_[i_]; } while (0)
#else
#define WINO_SEG_DECL
#define WINO_SEG_START
#define WINO_SEG(i)
#define WINO_SEG_FLUSH
#endif
#ifndef VD_WINO_SKIP
#define VD_WINO_SKIP 0     // kernel-experiment builds: bit 0 no patch staging, 1 no fragment transform, 2 no weight loads,
#endif                     // 3 no barrier, 4 no patch loads, 5 no fragment reads (results are then wrong; timing only)

typedef float f32x2 __attribute__((ext_vector_type(2)));

// TF4: four 8x8 frames per block (else one frame, 16x16 pixels).  The input is plain activations of ONE tensor: the
// engine materialises GroupNorm+SiLU (and the skip concat) first (norm.hip affine_act_kernel), because on gfx950 VALU
// work does not hide behind fp32 MFMAs (tools/mfma_peak.hip: every VALU instruction beside v_mfma_f32_32x32x2_f32 costs
// its 4 cycles; SALU, LDS and memory issue are free) -- so the loop below is written to contain as few VALU
// instructions as possible: 16 packed ops per MFMA group for the fragment transform and nothing else.
template <bool TF4>
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
    const int nchunk = a.Cin / WKC, ncoblk = a.Cout >> 5;
    const int cob0 = blockIdx.y * 2;                                 // BN = 64 = 2 cout blocks

    // ---- patch staging.  TF4: wave f stages frame f; else the block's one frame.  Thread -> patch pixels sp0 + SS*e of
    // its frame slot, channel quad lq.  The source is read through a buffer descriptor: elements of the zero padding
    // (and of frames past the end) get an out-of-range offset and the load itself returns zeros -- no select.
    const int lq = tid & 3;
    const int sf = TF4 ? wi : 0;
    const int sp0 = (TF4 ? lane : tid) >> 2;
    unsigned xo[NX];                                                  // byte offset of (pixel, channel lq*4) in the source
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
    const int xw = (sf * FS + sp0) * WLD + lq * 4;                   // LDS float offset of element 0
    f32x4 rx[2][NX];                                                  // two sets: a patch is requested 4 MFMA groups ahead
    auto x_load = [&](int set, int chunk) {
#pragma unroll
        for (int e = 0; e < NX; ++e)
            rx[set][e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xsrc, xo[e], chunk * (WKC * 4), 0));
    };
    auto x_load_one = [&](int set, int chunk, int e) {
        rx[set][e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xsrc, xo[e], chunk * (WKC * 4), 0));
    };
    auto x_store = [&](int set, int buf, int e0, int e1) {            // the buffers are padded to NX*SS pixels: no range branch
#pragma unroll
        for (int e = 0; e < NX; ++e)
            if (e >= e0 && e < e1) *reinterpret_cast<f32x4*>(smem + xw + buf * XBUF + e * SS * WLD) = rx[set][e];
    };

    // ---- A fragments: lane (tile m*32+lr, k-half lh); row wi of B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]].
    // Row 2 is taken negated (d1 - d2; pack_conv3_wino negates U's row 2 to match) so that every row is
    // d[r0] + sg * d[r1]: one (packed) fma per value pair
    const int r0 = wi == 0 ? 0 : 1, r1 = wi == 3 ? 3 : 2;
    const float sg = wi == 1 ? 1.f : -1.f;
    int xb[2];                                                      // LDS float offset of the tile's patch origin ...
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int t = m * 32 + lr;
        const int tx = t & (TT - 1), ty = (t >> TTL) & (TT - 1), f = t >> (2 * TTL);
        xb[m] = (f * FS + 2 * ty * P + 2 * tx) * WLD + lh * 4;
    }
    int ab[2][2];                                                    // ... + the two patch rows this wave combines
#pragma unroll
    for (int m = 0; m < 2; ++m) { ab[m][0] = xb[m] + r0 * P * WLD; ab[m][1] = xb[m] + r1 * P * WLD; }
    f32x4 raw[8];
    f32x2 fragl[2][4], fragh[2][4];                                  // channel pairs (0,1) and (2,3) of the k-quad
    const f32x2 sg2 = {sg, sg};
    // buf: which LDS patch buffer (compile-time after the 2x unroll of the chunk loop, so every address below is one
    // of four lane-invariant VGPRs plus an instruction offset)
    auto a_read = [&](int buf, int m, int kg) {
        const float* p0 = smem + ab[m][0] + buf * XBUF + kg * 8;
        const float* p1 = smem + ab[m][1] + buf * XBUF + kg * 8;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            raw[j] = *reinterpret_cast<const f32x4*>(p0 + j * WLD);
            raw[4 + j] = *reinterpret_cast<const f32x4*>(p1 + j * WLD);
        }
    };
    auto a_read_one = [&](int buf, int m, int kg, int i) {           // i = 0..7: patch row r0 / r1, column j
        raw[i] = *reinterpret_cast<const f32x4*>(smem + ab[m][i >> 2] + buf * XBUF + kg * 8 + (i & 3) * WLD);
    };
    // raw -> frag[slot]: t_j = d[r1][j]*sg + d[r0][j], then the column combination of B: 8 packed fma + 8 packed add for
    // 32 MFMAs' worth of A operands.  Written as asm because hipcc scalarises the <4 x float> form (each value is consumed
    // by one MFMA as a scalar) and only partly re-packs it; the two channel halves are independent, one block each.
    // The results feed the next group's first MFMA directly: the closing s_nop covers the VALU-write -> MFMA-read wait
    // states that hipcc does not pad for asm.
    auto half_transform = [&](f32x2& f0, f32x2& f1, f32x2& f2, f32x2& f3, f32x2 a0, f32x2 a1, f32x2 a2, f32x2 a3, f32x2 b0,
                              f32x2 b1, f32x2 b2, f32x2 b3) {
        asm volatile(
            "v_pk_fma_f32 %4, %8, %12, %4\n\t"
            "v_pk_fma_f32 %5, %9, %12, %5\n\t"
            "v_pk_fma_f32 %6, %10, %12, %6\n\t"
            "v_pk_fma_f32 %7, %11, %12, %7\n\t"
            "v_pk_add_f32 %0, %4, %6 neg_lo:[0,1] neg_hi:[0,1]\n\t"
            "v_pk_add_f32 %1, %5, %6\n\t"
            "v_pk_add_f32 %2, %6, %5 neg_lo:[0,1] neg_hi:[0,1]\n\t"
            "v_pk_add_f32 %3, %5, %7 neg_lo:[0,1] neg_hi:[0,1]\n\t"
            "s_nop 1"
            : "=&v"(f0), "=&v"(f1), "=&v"(f2), "=&v"(f3), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)
            : "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(sg2));
    };
    auto lo2 = [](const f32x4& v) { return f32x2{v.x, v.y}; };
    auto hi2 = [](const f32x4& v) { return f32x2{v.z, v.w}; };
    auto a_transform = [&](int s_) {
        half_transform(fragl[s_][0], fragl[s_][1], fragl[s_][2], fragl[s_][3], lo2(raw[0]), lo2(raw[1]), lo2(raw[2]), lo2(raw[3]),
                       lo2(raw[4]), lo2(raw[5]), lo2(raw[6]), lo2(raw[7]));
        half_transform(fragh[s_][0], fragh[s_][1], fragh[s_][2], fragh[s_][3], hi2(raw[0]), hi2(raw[1]), hi2(raw[2]), hi2(raw[3]),
                       hi2(raw[4]), hi2(raw[5]), hi2(raw[6]), hi2(raw[7]));
    };

    // ---- B fragments: U[chunk][xi = 4*wi + j][cob][kg][lane][4] through a buffer descriptor: one lane offset per
    // (j, n), everything else (chunk, wave row, k-group) in the scalar offset -- no address arithmetic in the loop
    const auto usrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wwino), 0, 16 * a.Cout * a.Cin * 4, 0x00020000);
    const int ustride = 16 * ncoblk * 2048, uwave = wi * 4 * ncoblk * 2048;
    unsigned bo[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int n = 0; n < 2; ++n) bo[j][n] = (unsigned)(j * ncoblk + min(cob0 + n, ncoblk - 1)) * 2048u + lane * 16u;
    f32x4 bfr[2][4][2];
    auto b_load = [&](int slot, int chunk, int kg) {
        const int so = chunk * ustride + uwave + kg * 1024;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int n = 0; n < 2; ++n)
                bfr[slot][j][n] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(usrc, bo[j][n], so, 0));
    };

    auto b_load_one = [&](int slot, int chunk, int kg, int i) {      // i = 0..7: (j, n)
        bfr[slot][i >> 1][i & 1] = __builtin_bit_cast(
            f32x4, __builtin_amdgcn_raw_buffer_load_b128(usrc, bo[i >> 1][i & 1], chunk * ustride + uwave + kg * 1024, 0));
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

    // ---- software pipeline over the 4 MFMA groups (kg, m) = (0,0) (0,1) (1,0) (1,1) of each chunk.  With one wave per
    // SIMD nothing else hides a stall, so every operand is requested well ahead of its use:
    //   patch rows (HBM)      chunk c+2's in groups 1 and 3 of chunk c  -> written to LDS in groups 0,1 of chunk c+1
    //   weights (L2)          one group ahead, issued in the group before that
    //   A fragments (LDS)     read behind the previous group's first MFMAs, transformed right after its last
    // The one barrier per chunk sits before group 3: the next patch is complete and nobody reads the current one any
    // more, so group 3 already fetches the next chunk's first fragments.
    x_load(0, 0);
    b_load(0, 0, 0);
    x_store(0, 0, 0, NX);
    x_load(1, min(1, nchunk - 1));
    __syncthreads();
    a_read(0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    a_transform(0);
    WINO_STAMP(1);

    WINO_SEG_DECL;
    WINO_SEG_START;
    // one chunk: buf = the patch buffer it reads = the register set that is free to receive the patch after next.
    // Requests are issued ONE per MFMA slot, never in bursts: the four waves run in lockstep, and 4 x 14 one-KiB loads
    // issued together overflow the CU's memory-instruction queue -- the last wave then stalls ~1000 cycles at issue and
    // the others wait for it at the barrier (tools/wino_timing.py, VD_WINO_TIMING=2).
    auto chunk_body = [&](int chunk, int buf) {
        const int n1 = min(chunk + 1, nchunk - 1), n2 = min(chunk + 2, nchunk - 1);
#pragma unroll
        for (int gi = 0; gi < 4; ++gi) {
            const int kg = gi >> 1, m = gi & 1;
            if (gi == 3 && !(VD_WINO_SKIP & 8)) __syncthreads();
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                const int e = k >> 3, j = (k >> 1) & 3, n = k & 1;
                const float av = e < 2 ? fragl[gi & 1][j][e] : fragh[gi & 1][j][e - 2];
                acc[m][j][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bfr[kg][j][n][e], acc[m][j][n], 0, 0, 0);
                bool fence = false;
                if (k < 8 && !(VD_WINO_SKIP & 32)) {                // fragments of the NEXT group, behind MFMAs 0..7
                    if (gi < 3) a_read_one(buf, (gi + 1) & 1, (gi + 1) >> 1, k);
                    else a_read_one(buf ^ 1, 0, 0, k);
                    fence = true;
                }
                if (k >= 8 && k < 24 && (k & 1) == 0) {             // one global request behind every other MFMA 8..22
                    const int i = (k - 8) >> 1;                     // 0..7
                    // weights: slot kg^1 is free once the previous group is done with it
                    if (gi == 0 && !(VD_WINO_SKIP & 4)) { b_load_one(1, chunk, 1, i); fence = true; }
                    if (gi == 2 && !(VD_WINO_SKIP & 4)) { b_load_one(0, n1, 0, i); fence = true; }
                    // patch after next: first half of its rows in group 1, the rest in group 3
                    if (gi == 1 && i < (NX + 1) / 2 && !(VD_WINO_SKIP & 16)) { x_load_one(buf, n2, i); fence = true; }
                    if (gi == 3 && i < NX / 2 && !(VD_WINO_SKIP & 16)) { x_load_one(buf, n2, (NX + 1) / 2 + i); fence = true; }
                }
                if (fence) __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_sched_barrier(0);
            WINO_SEG(2 * gi);
            // after the MFMAs (which cover the LDS latency of the fragment reads): the next group's fragment transform in
            // one VALU burst -- VALU work does not overlap fp32 MFMAs and every MFMA<->VALU switch costs ~8 cycles -- and
            // (groups 0,1) half of the next patch's LDS writes
            if (!(VD_WINO_SKIP & 2)) a_transform((gi + 1) & 1);
            if (!(VD_WINO_SKIP & 1)) {
                if (gi == 0) x_store(buf ^ 1, buf ^ 1, 0, (NX + 1) / 2);
                if (gi == 1) x_store(buf ^ 1, buf ^ 1, (NX + 1) / 2, NX);
            }
            __builtin_amdgcn_sched_barrier(0);
            WINO_SEG(2 * gi + 1);
        }
    };
    for (int chunk = 0; chunk < nchunk; chunk += 2) {               // nchunk is even (conv_wino_supported)
        chunk_body(chunk, 0);
        chunk_body(chunk + 1, 1);
    }
    WINO_SEG_FLUSH;
    WINO_STAMP(2);

    // ---- output transform.  A^T = [[1,1,1,0],[0,1,-1,-1]].  Wave-local: Z[q] = sum_j M[wi][j] A[j][q]; the sum over
    // i crosses waves through LDS; wave (p, q) then owns output pixel (p, q) of every tile.  Everything below is
    // branch-free: rows of frames past the end get an out-of-range buffer offset (loads return 0, stores are dropped).
    const int p = wi >> 1, q = wi & 1;
    const int co0 = blockIdx.y * 64 + lr;
    unsigned oo[2][16];                                              // byte offset of (pixel, cout co0) in out / res
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int t = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;        // C/D row = tile
            const int tx = t & (TT - 1), ty = (t >> TTL) & (TT - 1), f = t >> (2 * TTL);
            const int nf = f0 + f;
            const unsigned o = (unsigned)(((nf * Hl + oy0 + 2 * ty + p) * Wl + ox0 + 2 * tx + q) * a.ldo + co0) * 4u;
            oo[m][r] = nf < a.nfr ? o : 0x80000000u;
        }
    const int obytes = a.nfr * Hl * Wl * a.ldo * 4;
    const auto osrc = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, obytes, 0x00020000);
    // the residual rows (same layout as the output: res_ld == ldo) are requested a quarter at a time between the pieces
    // of the exchange: their latency hides behind it, and 4 waves x 64 requests in one burst would overflow the CU's
    // memory-instruction queue (same effect as in the main loop)
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.res ? a.res : a.out), 0, a.res ? obytes : 0, 0x00020000);
    f32x16 rv[2][2];
    float bv[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) bv[n] = a.bias ? a.bias[co0 + n * 32] : 0.f;
    WINO_STAMP(4);
    __syncthreads();                                                 // every wave is done with the patch buffers
    WINO_STAMP(5);
    // Z exchange layout in LDS: [plane = 2i + q][m 2][n 2][reg16/4][lane 64][4]  (8 planes of 16 KB)
    float* Zs = smem;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                rv[m][n][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, oo[m][r], n * 128, 0));
            const f32x16 z0 = acc[m][0][n] + acc[m][1][n] + acc[m][2][n];
            const f32x16 z1 = acc[m][1][n] - acc[m][2][n] - acc[m][3][n];
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) {
                float* d0 = Zs + (((((wi * 2 + 0) * 2 + m) * 2 + n) * 4 + c4) * 64 + lane) * 4;
                float* d1 = Zs + (((((wi * 2 + 1) * 2 + m) * 2 + n) * 4 + c4) * 64 + lane) * 4;
                *reinterpret_cast<f32x4*>(d0) = f32x4{z0[4 * c4], z0[4 * c4 + 1], z0[4 * c4 + 2], z0[4 * c4 + 3]};
                *reinterpret_cast<f32x4*>(d1) = f32x4{z1[4 * c4], z1[4 * c4 + 1], z1[4 * c4 + 2], z1[4 * c4 + 3]};
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    WINO_STAMP(6);
    __syncthreads();
    WINO_STAMP(7);
    // wave (p, q): Y[p][q] = sum_i A^T[p][i] Z[i][q] = Z[p] + sgn * (Z[p+1] + Z[p+2]), sgn = +1 (p = 0) / -1 (p = 1);
    // Z[p + k][q] is plane wi + 2k
    const float sgn = p ? -1.f : 1.f;
    const float* zw = Zs + wi * 4096 + lane * 4;
    float gsum[TF4 ? 4 : 1][2][2] = {};
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            f32x16 y;
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) {
                const float* zp = zw + ((m * 2 + n) * 4 + c4) * 256;
                const f32x4 v = *reinterpret_cast<const f32x4*>(zp) +
                                (*reinterpret_cast<const f32x4*>(zp + 2 * 4096) + *reinterpret_cast<const f32x4*>(zp + 4 * 4096)) * sgn;
                y[4 * c4] = v.x; y[4 * c4 + 1] = v.y; y[4 * c4 + 2] = v.z; y[4 * c4 + 3] = v.w;
            }
            y += rv[m][n];
            if (a.fbias) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int t = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    y[r] += a.fbias[(size_t)min(f0 + (t >> (2 * TTL)), a.nfr - 1) * a.fbias_ld + co0 + n * 32];
                }
            }
            y += bv[n];
#pragma unroll
            for (int r = 0; r < 16; ++r) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, (float)y[r]), osrc, oo[m][r], n * 128, 0);
            if (a.stats) {                                           // GroupNorm partial sums of what was just stored
#pragma unroll
                for (int h = 0; h < (TF4 ? 2 : 1); ++h) {            // TF4: rows 0..7 / 8..15 are two frames
                    float s = 0.f, ss = 0.f;
#pragma unroll
                    for (int r = h * (TF4 ? 8 : 0); r < (TF4 ? 8 * h + 8 : 16); ++r) { s += y[r]; ss = __builtin_fmaf(y[r], y[r], ss); }     // (explicit: see conv_wino_r64.hip)
                    const int fs = TF4 ? 2 * m + h : 0;              // frame slot of the block
                    gsum[fs][n][0] += s; gsum[fs][n][1] += ss;
                }
            }
        }
    if (a.stats) {
        // per lane: [frame slot][n][sum, sumsq] over its 16 or 32 rows (fp32 over <= 32 values), then doubles: 8 partials
        // per (frame, channel) -- 4 waves (the 4 pixels of a tile) x 2 k-halves -- meet in LDS and one thread per
        // (frame, channel) adds them in a fixed order and writes the block's entry of the partial table
        constexpr int NFS = TF4 ? 4 : 1;
        __syncthreads();                                             // the Z planes are dead
        double* red = reinterpret_cast<double*>(smem);               // [wave 4][lh 2][fs][n 2][lr 32][2]
#pragma unroll
        for (int fs = 0; fs < NFS; ++fs)
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                double* d = red + (((((wi * 2 + lh) * NFS + fs) * 2 + n) * 32 + lr) * 2);
                d[0] = (double)gsum[fs][n][0]; d[1] = (double)gsum[fs][n][1];
            }
        __syncthreads();
        if (tid < NFS * 64) {
            const int fs = tid >> 6, n = (tid >> 5) & 1, c = tid & 31;
            double s = 0.0, ss = 0.0;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const double* d = red + ((((k * NFS + fs) * 2 + n) * 32 + c) * 2);
                s += d[0]; ss += d[1];
            }
            const int nf = f0 + fs;
            const int sp = TF4 ? 0 : byy * g.tiles_x + bxx;
            if (nf < a.nfr) {
                double* o = a.stats + (((size_t)nf * a.stats_split + sp) * a.Cout + blockIdx.y * 64 + n * 32 + c) * 2;
                o[0] = s; o[1] = ss;
            }
        }
    }
    WINO_STAMP(3);
}

static bool wino_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

bool conv_wino_supported(const IgemmArgs& a) {
    const int Hl = a.Hs << a.ups, Wl = a.Ws << a.ups;
    return a.wwino != nullptr && !a.wsplit && a.ksz == 3 && a.stride == 1 && a.pad == 1 && Hl == Wl && wino_pow2(Hl) && Hl >= 8 &&
           a.Cout % 64 == 0 && a.Cin % (2 * WKC) == 0 && a.src1 == nullptr && a.C0 == a.Cin && a.affA == nullptr && a.act == 0 &&
           // 32-bit byte offsets into the source (bit 31 marks the padding) / residual / output
           (size_t)a.nfr * a.Hs * a.Ws * a.Cin < (1u << 29) &&
           (size_t)a.nfr * Hl * Wl * a.ldo < (1u << 29) && (a.res == nullptr || a.res_ld == a.ldo);
}

int conv_wino_stats_split(int Hout) { return Hout >= 16 ? (Hout / 16) * (Hout / 16) : 1; }

int launch_conv_wino(const IgemmArgs& a, hipStream_t s) {
    const int Hl = a.Hs << a.ups;
    VD_REQUIRE(a.stats == nullptr || a.stats_split == conv_wino_stats_split(Hl), "GroupNorm partial table: split");
    WinoGeom g;
    const int TT = Hl >= 16 ? 8 : 4;                   // tiles per dim per frame in a block
    g.TF = 64 / (TT * TT);                             // 1 or 4 frames
    g.tiles_x = Hl / (2 * TT); g.tiles_y = Hl / (2 * TT);
    const size_t lds = std::max((size_t)2 * (g.TF == 4 ? 7 : 6) * 64 * WLD * sizeof(float), (size_t)4 * 2 * 2 * 2 * 4 * 64 * 4 * sizeof(float));
    VD_RAISE_LDS((&conv3x3_wino_kernel<true>), (size_t)160 * 1024);
    VD_RAISE_LDS((&conv3x3_wino_kernel<false>), (size_t)160 * 1024);
    const int fgroups = (a.nfr + g.TF - 1) / g.TF;
    dim3 grid(g.tiles_x * g.tiles_y * fgroups, a.Cout / 64);
    if (g.TF == 4) hipLaunchKernelGGL((conv3x3_wino_kernel<true>), grid, dim3(256), lds, s, a, g);
    else hipLaunchKernelGGL((conv3x3_wino_kernel<false>), grid, dim3(256), lds, s, a, g);
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
