// 3x3 stride-1 convolution by Winograd F(2x2,3x3) at fp32 accuracy on the bf16 matrix cores, 64 couts per block, gfx950.
//
// Same arithmetic as conv_wino_split.hip (V = B^T d B in fp32, U = G g G^T in fp64 on the host, both split EXACTLY into
// three bf16 pieces, six piece products per element through v_mfma_f32_32x32x16_bf16 with fp32 accumulation), different
// work split.  What conv_wino_split.hip showed: a wave that transforms and splits its own fragments for 32 couts spends
// ~10 plain VALU per MFMA slot, and only 6 hide behind a bf16 MFMA (tools/mfma_bf16_coissue.hip).  Here a block owns
// 64 tiles x 64 couts, so every split fragment feeds 12 MFMAs instead of 6, and the input transform is taken out of the
// waves' private work altogether:
//
//   * the 256 threads transform the raw patch ONCE per block: thread (tile, channel quad, row pair) reads its 4x4 patch
//     rows from the raw image in LDS, forms t = d[X] + s*d[S] and the four column combinations in fp32 and writes
//     V[position][tile][16 channels] (fp32, 16-byte slots XOR-swizzled so that both these writes and the fragment reads
//     are bank-conflict free);
//   * wave i owns Winograd row i (positions 4i..4i+3) for all 64 tiles and 64 couts: 4 x 2 x 2 = 16 accumulator tiles
//     (256 AGPRs).  Per position and M-tile it reads one V fragment (2 x ds_read_b128 = 8 channels of its tile), splits
//     it into three bf16 pieces (44 plain VALU) and issues 2 N-tiles x 6 = 12 MFMAs with it: 4-6 VALU per MFMA slot,
//     all hidden.  One wave per SIMD; the split of position p+1 runs under the MFMAs of position p.
//
// Time is cut in "positions" of 12 MFMA slots; a "group" = 4 positions = one M-tile half (32 tiles) of one 16-channel
// chunk.  V is double-buffered per HALF (2 x 32 KB): group g reads Vh[g&1] while the block transforms group g+1 into the
// other half-buffer during positions 2,3 of group g-1 (row pair member 0) and positions 0,1 of group g (member 1); one
// barrier per group, after position 1, is both "V of group g+1 complete" and "V of group g no longer read" (fragment
// reads run two positions ahead).  The raw patch is double-buffered per chunk and staged HBM -> registers -> LDS one
// chunk ahead; weights [Cin/16][16 positions][Cout/32][3 pieces][64 lanes][8 bf16] stream from L2 into a single register
// set, each fragment reloaded for the next chunk right after its last MFMA of this one.
// Sign convention: rows 2 and 3 of B^T are taken as d2 - d1 and d3 - d1 (so that every row is d[X] + s*d[S] with a shared
// row S per row pair); row 3 of U is negated on the host to match.
#include <cstring>

#include "vd_common.h"

namespace vd {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// nbx = tiles_x*tiles_y*frame groups, ncb = Cout/64, nitems = nbx*ncb.  cob_inner: the cout blocks of one tile block run on
// blocks 8 apart (same XCD, same L2) at the same time, so the input is read from HBM once instead of ncb times; chosen
// when the whole layer's weights fit an XCD's L2 next to it (launch_conv_wino_s64)
struct WinoS64Geom { int TF, tiles_x, tiles_y, nbx, ncb, nitems, cob_inner; };

#ifdef VD_WINO_TIMING
__device__ unsigned long long g_s64_stamp[10];
#define S64_STAMP(i)                                                                                                   \
    do {                                                                                                               \
        if (threadIdx.x == 0 && blockIdx.x == 7 && s64_first) {                                                        \
            g_s64_stamp[i] = __builtin_readcyclecounter();                                                             \
            if (i == 0) g_s64_stamp[8] = __builtin_amdgcn_s_memrealtime();                                             \
            if (i == 3) g_s64_stamp[9] = __builtin_amdgcn_s_memrealtime();                                             \
        }                                                                                                              \
    } while (0)
extern "C" int vd_debug_s64_stamps(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_s64_stamp), sizeof(g_s64_stamp));
}
#else
#define S64_STAMP(i)
#endif
#ifndef VD_S64_ILV
#define VD_S64_ILV 1       // A/B builds: 0 = the six piece products of a tile back to back on one accumulator
#endif
#ifndef VD_S64_SKIP
#define VD_S64_SKIP 0      // kernel-experiment builds (timing only, results wrong): bit 0 no split VALU, 1 no weight loads,
#endif                     // 2 no fragment reads, 3 no patch loads/stores, 4 no transform

// workgroup barrier that waits for this wave's LDS traffic only (a __syncthreads() would also drain the weight and
// patch loads in flight)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <bool TF4>
__global__ __launch_bounds__(256, 1) void conv3x3_wino_s64_kernel(IgemmArgs a, WinoS64Geom g) {
    constexpr int TTL = TF4 ? 2 : 3, TT = 1 << TTL, P = 2 * TT + 2;   // tiles per dim per frame, patch width
    // raw patch image (bytes): 64 B per pixel, pixels of even and odd x in two planes per row (neighbouring tiles read
    // neighbouring 64 B of one plane; a plane row is padded by one pixel slot).  It is filled by LDS-DMA
    // (buffer_load_dwordx4 ... lds): lane l of a wave instruction writes 16 bytes at M0 + 16*l, whatever global address it
    // gathers from, and zeros where that address fails the descriptor's range check (tools/lds_dma_probe.hip) -- so the
    // image is a linear array of 16-byte slots, thread tid owns slots e*256 + tid, and padding, halo pixels outside the
    // picture and frames past the end are simply out-of-range offsets.  No staging registers, no ds_write.
    constexpr int SPP = TF4 ? 6 : 10;          // 64-byte pixel slots per plane row (P/2 pixels + 1 pad)
    constexpr int PLB = SPP * 64, RSB = 2 * PLB, FSB = TF4 ? P * RSB : 0;
    constexpr int NX = TF4 ? 8 : 6;            // DMA instructions per thread and patch: ceil(slots / 256)
    constexpr int XBUF = NX * 4096;            // 32768 / 24576 >= 4 * FSB / P * RSB
    constexpr int VH = 32768;                                         // one V half-buffer: [16 positions][32 tiles][64 B]
    constexpr int RAW0 = 2 * VH;
    constexpr int XON = RAW0 + 2 * XBUF;                              // 8 KB: the next item's patch offsets
    constexpr int HALF = TF4 ? 2 * FSB : 8 * RSB;                    // raw offset of tile t + 32 relative to tile t
    extern __shared__ __attribute__((aligned(16))) float smem[];      // [Vh 2][raw 2][8 KB]; Vh reused as Z exchange
    f32x4* const lds4 = reinterpret_cast<f32x4*>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wi = __builtin_amdgcn_readfirstlane(tid >> 6);          // Winograd row owned by this wave (scalar)
    const int lr = lane & 31, lh = lane >> 5;
    // timing builds stamp the phases of ONE steady-state item: the (VD_S64_STAMP_ITEM+1)-th item of block 7
#ifndef VD_S64_STAMP_ITEM
#define VD_S64_STAMP_ITEM 3
#endif
    [[maybe_unused]] bool s64_first = false;
    [[maybe_unused]] int s64_item = 0;
    const int Hl = a.Hs << a.ups, Wl = a.Ws << a.ups;
    const int nchunk = a.Cin >> 4, ncoblk = a.Cout >> 5;
    // work items: (tile block, frame group) fastest, cout block slowest; a block walks items blockIdx.x, +gridDim.x, ...
    struct Item { int bxx, byy, f0, cob0, ox0, oy0; };
    auto decode = [&](int it) {
        Item r;
        int bx = it % g.nbx, cb = it / g.nbx;                         // tile block fastest: concurrent items share one cout block's weights
        if (g.cob_inner) {                                            // (nbx % 8 == 0) items it, it+8, .., it+8*(ncb-1): one tile block, all cout blocks
            const int grp = it / (8 * g.ncb), rem = it - grp * (8 * g.ncb);
            bx = grp * 8 + (rem & 7); cb = rem >> 3;
        }
        r.cob0 = cb * 2;
        r.bxx = bx % g.tiles_x; bx /= g.tiles_x;
        r.byy = bx % g.tiles_y; bx /= g.tiles_y;
        r.f0 = bx * (TF4 ? 4 : 1);
        r.ox0 = r.bxx * 2 * TT; r.oy0 = r.byy * 2 * TT;             // output-pixel origin of the item
        return r;
    };

    // ---- patch staging: thread -> 16-byte slots e*256 + tid of the image = (pixel slot, channel quad)
    unsigned xo[NX];
    auto set_xo = [&](unsigned (&xo)[NX], const Item& w, int tid) {
#pragma unroll
        for (int e = 0; e < NX; ++e) {
            const int gs = e * 256 + tid, lq = gs & 3, ps = gs >> 2;
            const int fr = TF4 ? ps / (P * 2 * SPP) : 0, rr = TF4 ? ps % (P * 2 * SPP) : ps;
            const int py = rr / (2 * SPP), r = rr % (2 * SPP), pxh = r % SPP, px = 2 * pxh + r / SPP;
            const int ly = w.oy0 + py - 1, lx = w.ox0 + px - 1;
            const bool in = fr < (TF4 ? 4 : 1) && py < P && pxh < P / 2 && w.f0 + fr < a.nfr && ly >= 0 && ly < Hl && lx >= 0 && lx < Wl;
            xo[e] = in ? (unsigned)(((w.f0 + fr) * a.Hs + (ly >> a.ups)) * a.Ws + (lx >> a.ups)) * (unsigned)(a.Cin * 4) + lq * 16u
                       : 0x80000000u;
        }
    };
    const Item first = decode(blockIdx.x);
    set_xo(xo, first, tid);
    const auto xsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.src0), 0, a.nfr * a.Hs * a.Ws * a.Cin * 4, 0x00020000);
    typedef __attribute__((address_space(3))) void* lds_ptr;
    auto x_dma_one = [&](int chunk, int buf, int e) {               // 1 KiB of the image per wave instruction; chunk >= nchunk:
        chunk = chunk >= nchunk ? chunk - nchunk : chunk;           // chunk - nchunk of the next item (xo has been swapped by then)
#if defined(__HIP_DEVICE_COMPILE__)     // (the host pass of hipcc drops the whole kernel, stub included, if it sees this builtin)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xsrc, (lds_ptr)(smem + (RAW0 + buf * XBUF + e * 4096) / 4 + wi * 256), 16, xo[e],
                                                 chunk * 64, 0, 0);
#endif
    };

    // ---- block transform: wave w -> row pair ih = w >> 1, tiles (w & 1)*16 + lane/4 of the half, channel quad lane & 3
    const int ih = wi >> 1;
    const int tl = (wi & 1) * 16 + (lane >> 2), kq = lane & 3;
    const int ttx = tl & (TT - 1), tty = (tl >> TTL) & (TT - 1), ttf = tl >> (2 * TTL);
    const int rbase = (RAW0 + ttf * FSB + 2 * tty * RSB + ttx * 64 + kq * 16) / 16;
    // rows: pair 0 = {d0 - d2, d1 + d2} (S = 2), pair 1 = {d2 - d1, d3 - d1} (S = 1)
    const int trS = rbase + (ih ? 1 : 2) * (RSB / 16);
    const int trX[2] = {rbase + (ih ? 2 : 0) * (RSB / 16), rbase + (ih ? 3 : 1) * (RSB / 16)};
    const float tsg[2] = {-1.f, ih ? -1.f : 1.f};
    const int vwb = (tl * 64 + ((kq ^ ((tl >> 2) & 3)) * 16) + ih * 8 * 2048) / 16;     // + (il*4 + j)*128 + half*2048
    float tS[4][4], tX[4][4];                                        // [column][channel]
    // one transform "half window" = row il of the pair for the 32 tiles of `half` of the patch in raw buffer `buf`, streamed
    // by patch column so that few values are live at a time (the main loop has no register to spare):
    //   t_read(c): the two fragments of column c;  t_unit(u), u = 0..7 =
    //   t2, t0, V0 = t0 - t2, t1, V1 = t1 + t2, V2 = t2 - t1, t3, V3 = t1 - t3      with t[c] = dX[c] + s*dS[c], V -> Vh[half]
    auto t_read = [&](int buf, int half, int il, int c) {
        const int o = buf * (XBUF / 16) + half * (HALF / 16) + ((c & 1) * PLB + (c >> 1) * 64) / 16;
        const f32x4 vs = lds4[trS + o], vx = lds4[trX[il] + o];
        tS[c][0] = vs.x; tS[c][1] = vs.y; tS[c][2] = vs.z; tS[c][3] = vs.w;
        tX[c][0] = vx.x; tX[c][1] = vx.y; tX[c][2] = vx.z; tX[c][3] = vx.w;
    };
    auto t_unit = [&](int half, int il, int u) {
        constexpr int ROWC[8] = {2, 0, -1, 1, -1, -1, 3, -1};       // row-combination units: the column they form
        constexpr int VJ[8] = {-1, -1, 0, -1, 1, 2, -1, 3};         // column-combination units: the position they write
        if (ROWC[u] >= 0) {
            const int c = ROWC[u];
            const float s = tsg[il];
#pragma unroll
            for (int h = 0; h < 4; ++h) asm("v_fma_f32 %0, %1, %2, %0" : "+v"(tX[c][h]) : "v"(tS[c][h]), "v"(s));
        } else {
            const int j = VJ[u];
            const int ca = j == 0 ? 0 : j == 2 ? 2 : 1, cb = j == 3 ? 3 : j == 2 ? 1 : 2;
            float v[4];
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                if (j == 1) asm("v_add_f32 %0, %1, %2" : "=v"(v[h]) : "v"(tX[ca][h]), "v"(tX[cb][h]));
                else asm("v_sub_f32 %0, %1, %2" : "=v"(v[h]) : "v"(tX[ca][h]), "v"(tX[cb][h]));
            }
            lds4[vwb + (il * 4 + j) * 128 + half * (VH / 16)] = f32x4{v[0], v[1], v[2], v[3]};
        }
    };

    // ---- V fragments: lane (tile lr of the half, k-half lh) reads channels 8*lh .. 8*lh+7 of position 4*wi + j
    const int vr0 = (wi * 4 * 2048 + lr * 64 + (((2 * lh) ^ ((lr >> 2) & 3)) * 16)) / 16;
    const int vr1 = (wi * 4 * 2048 + lr * 64 + (((2 * lh + 1) ^ ((lr >> 2) & 3)) * 16)) / 16;
    f32x4 vf[2][2];                                                   // [ring slot][channel quad]
    auto v_read = [&](int slot, int half, int j) {
        vf[slot][0] = lds4[vr0 + j * 128 + half * (VH / 16)];
        vf[slot][1] = lds4[vr1 + j * 128 + half * (VH / 16)];
    };
    u32x4 apc[2][3];                                                  // [ring slot][piece] = 8 bf16
    float sr0, sr1;
    auto s_split = [&](int slot, int pr, int stage) {                 // pair pr = channels 2pr, 2pr+1 of vf[slot]
        const f32x4 v = vf[slot][pr >> 1];
        if (stage == 0) {
            unsigned p1;
            split_a((pr & 1) ? v.z : v.x, (pr & 1) ? v.w : v.y, p1, sr0, sr1, 0x07060302u);
            apc[slot][0][pr] = p1;
        } else {
            unsigned p2, p3;
            split_b(sr0, sr1, p2, p3, 0x07060302u);
            apc[slot][1][pr] = p2; apc[slot][2][pr] = p3;
        }
    };

    // ---- B fragments: U[chunk][xi = 4*wi + j][cob][piece][lane][8 bf16] = 1 KiB per (chunk, xi, cob, piece)
    const auto usrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wwino), 0, 16 * a.Cout * a.Cin * 6, 0x00020000);
    const int ustride = 16 * ncoblk * 3072;
    const unsigned blane = lane * 16u;
    // scalar byte offset of (j, n) within a chunk = base(item) + j * bstep + n * 3072
    const int bstep = ncoblk * 3072;
    int bsb = 0;
    bf16x8 bfr[4][2][3];                                              // [j][n][piece], single set
    auto b_load_one = [&](int chunk, int j, int n, int p) {         // past the last chunk: the last one again (unused)
        const int so = min(chunk, nchunk - 1) * ustride + bsb + j * bstep + n * 3072;
        bfr[j][n][p] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(usrc, blane, so + p * 1024, 0));
    };

    f32x16 acc[2][4][2];                                             // [m][j][n]

    // ---- first item: raw[0], raw[1] requested.  Later items find exactly this state: the last two chunks of an item request the
    // first two patches of the NEXT one (patch offsets swapped with the chunk index folded in); they land under the last
    // groups and the output transform.
#pragma unroll
    for (int e = 0; e < NX; ++e) x_dma_one(0, 0, e);
#pragma unroll
    for (int e = 0; e < NX; ++e) x_dma_one(1, 1, e);

    // one position = 12 MFMA slots.  (chunk, m, j) with chunk parity cp compile-time; everything else immediate.
    auto position = [&](int chunk, int cp, int m, int j) {
        const int pi = (m * 4 + j) & 1;                               // ring slot of this position's pieces / fragment
        // what the other work of this position refers to
        const int nj = (j + 1) & 3, nm = j == 3 ? m ^ 1 : m;          // position p+1 (split target)
        const int rj = (j + 2) & 3, rm = j >= 2 ? m ^ 1 : m;          // position p+2 (fragment read)
        // transform: positions 2,3 of (chunk, m) serve row 0 of the group after next, positions 0,1 row 1 of the next group
        //   group after next of (c, 0) is (c+1, 0): half 0 of raw[c+1];  of (c, 1) it is (c+1, 1): half 1 of raw[c+1]
        //   next group of (c, 0) is (c, 1): half 1 of raw[c];            of (c, 1) it is (c+1, 0): half 0 of raw[c+1]
        const int til = j >= 2 ? 0 : 1;
        const int thalf = j >= 2 ? m : m ^ 1;
        const int tbuf = (j >= 2 || m == 1) ? cp ^ 1 : cp;
        const int ubase = (j & 1) * 4;
        const int n1 = chunk + 1;                                     // past the last chunk: the next item's (b_load_one, patch offsets)
        f32x16 cc[2] = {acc[m][j][0], acc[m][j][1]};
#pragma unroll
        for (int k = 0; k < 12; ++k) {                               // slot 0..11 of the position
            {
                constexpr int PA[6] = {0, 1, 2, 0, 1, 0}, PB[6] = {2, 1, 0, 1, 0, 0};     // small terms first
                // VD_S64_ILV: the two cout tiles alternate, so that consecutive MFMAs never share an accumulator
                const int n = VD_S64_ILV ? (k & 1) : k / 6, q = VD_S64_ILV ? (k >> 1) : k % 6;
                cc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, apc[pi][PA[q]]), bfr[j][n][PB[q]], cc[n], 0, 0, 0);
                // split of position p+1: pairs at slots (0,1) (3,4) (6,7) (9,10)
                if (!(VD_S64_SKIP & 1) && k % 3 != 2) s_split(pi ^ 1, k / 3, k % 3);
                // transform unit at slots 2, 5, 8, 11
                if (!(VD_S64_SKIP & 16)) {
                    // column fragments two to five slots ahead of the unit that combines them
                    if ((j & 1) == 0 && k == 0) t_read(tbuf, thalf, til, 2);
                    if ((j & 1) == 0 && k == 1) t_read(tbuf, thalf, til, 0);
                    if ((j & 1) == 0 && k == 6) t_read(tbuf, thalf, til, 1);
                    if ((j & 1) == 1 && k == 3) t_read(tbuf, thalf, til, 3);
                    if (k % 3 == 2) t_unit(thalf, til, ubase + k / 3);
                }
                // fragment of position p+2 (its ring slot is free: position p's was consumed during p-1)
                if (!(VD_S64_SKIP & 4) && k == 3) v_read(pi, rm, rj);
                // weights of the next chunk: each fragment right after its last use (second M-tile only)
                if (!(VD_S64_SKIP & 2) && m == 1) {
                    if (VD_S64_ILV) {
                        if (k == 2 || k == 3) b_load_one(n1, j, k - 2, 2);
                        if (k == 8 || k == 9) b_load_one(n1, j, k - 8, 1);
                        if (k < 2 && j > 0) b_load_one(n1, j - 1, k, 0);
                    } else {
                        if (n == 1 && q < 3) b_load_one(n1, j, 0, q);
                        if (n == 0 && q >= 3 && j > 0) b_load_one(n1, j - 1, 1, q - 3);
                    }
                }
                // patch: raw[chunk+2] by LDS-DMA into the buffer of raw[chunk] (free since this group's barrier), one request
                // every other slot of positions 2,3 of (chunk, 0); it has a whole chunk to land (read from (chunk+1, 0) 2,3 on)
                if (!(VD_S64_SKIP & 8) && m == 0 && j >= 2) {
                    const int xe = (j & 1) * 6 + (k >> 1);            // 0..11
                    if ((k & 1) == 1 && xe < NX) x_dma_one(chunk + 2, cp, xe);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        acc[m][j][0] = cc[0]; acc[m][j][1] = cc[1];
        if (!(VD_S64_SKIP & 2) && m == 1 && j == 3) {
            if (VD_S64_ILV) { b_load_one(n1, 3, 0, 0); b_load_one(n1, 3, 1, 0); }
            else {
#pragma unroll
                for (int p = 0; p < 3; ++p) b_load_one(n1, 3, 1, p);
            }
        }
        if (j == 1) {
            // the patch requested in (chunk-1, 0) must have landed: only the 24 weight loads of (chunk-1, 1) are younger
            if (m == 0) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
            lds_barrier();
        }
        __builtin_amdgcn_sched_barrier(0);
        (void)nj; (void)nm;
    };
    for (int it = blockIdx.x; it < g.nitems; it += gridDim.x) {
#ifdef VD_WINO_TIMING
        s64_first = s64_item++ == VD_S64_STAMP_ITEM;
        S64_STAMP(0);
#endif
        const int itn = it + (int)gridDim.x < g.nitems ? it + (int)gridDim.x : it;   // no next item: its own again (unused)
        // weights of chunk 0: requested here, not under the previous item's output transform -- 96 live registers there
        // made the compiler spill; they come from L2
        bsb = (wi * 4 * ncoblk + decode(it).cob0) * 3072;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int p = 0; p < 3; ++p) b_load_one(0, j, n, p);
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[m][j][n][r] = 0.f;
        {   // the next item's patch offsets, computed while few registers are live and parked in LDS until the last chunks
            unsigned xn[NX];
            int tidv = tid;                          // re-materialised: the lane-dependent parts of set_xo must not be hoisted
            asm volatile("" : "+v"(tidv));           // out of the item loop (they would be spilled and reloaded one by one)
            set_xo(xn, decode(itn), tidv);
            lds4[XON / 16 + tid * 2] = __builtin_bit_cast(f32x4, u32x4{xn[0], xn[1], xn[2], xn[3]});
            lds4[XON / 16 + tid * 2 + 1] = __builtin_bit_cast(f32x4, u32x4{xn[4], xn[5], NX > 6 ? xn[6] : 0u, NX > 6 ? xn[NX - 1] : 0u});
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // raw[0] of this item (LDS-DMA) and the weights of chunk 0 have landed
        __syncthreads();                             // ... for every wave; the previous item is done with the LDS
        S64_STAMP(4);
        // V of groups 0 and 1, pieces of position 0
#pragma unroll
        for (int half = 0; half < 2; ++half)
#pragma unroll
            for (int il = 0; il < 2; ++il) {
#pragma unroll
                for (int c = 0; c < 4; ++c) t_read(0, half, il, c);
#pragma unroll
                for (int u = 0; u < 8; ++u) t_unit(half, il, u);
            }
        __syncthreads();
        S64_STAMP(5);
        v_read(0, 0, 0);
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) { s_split(0, pr, 0); s_split(0, pr, 1); }
        v_read(1, 0, 1);
        S64_STAMP(1);
        for (int chunk = 0; chunk < nchunk; chunk += 2) {           // nchunk is even (conv_wino_s64_supported)
#pragma unroll
            for (int cp = 0; cp < 2; ++cp)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (cp == 0 && m == 0 && j == 2 && chunk + 2 == nchunk) {   // from here on the patch requests are the next item's
                            const u32x4 n0 = __builtin_bit_cast(u32x4, lds4[XON / 16 + tid * 2]), n1 = __builtin_bit_cast(u32x4, lds4[XON / 16 + tid * 2 + 1]);
                            xo[0] = n0.x; xo[1] = n0.y; xo[2] = n0.z; xo[3] = n0.w; xo[4] = n1.x; xo[5] = n1.y;
                            if constexpr (NX > 6) { xo[6] = n1.z; xo[NX - 1] = n1.w; }
                        }
                        position(chunk + cp, cp, m, j);
                    }
        }
        S64_STAMP(2);
        const Item cur = decode(it);
        // lane-derived values re-materialised here: otherwise hipcc hoists the epilogue's lane-dependent address terms out of
        // the item loop and they sit in VGPRs through the main loop, which has none to spare
        int e_lr = lr, e_lh = lh, e_lane = lane, e_tid = tid;
        asm volatile("" : "+v"(e_lr), "+v"(e_lh), "+v"(e_lane), "+v"(e_tid));
        // ---- output transform (conv_wino.hip), one cout block (n) at a time: Z[q] = sum_j M[wi][j] A[j][q] wave-local, sum
        // over i through LDS, wave (p, q) owns output pixel (p, q) of every tile; branch-free via buffer range checks
        const int p = wi >> 1, q = wi & 1;
        unsigned oo[2][16];
    #pragma unroll
        for (int m = 0; m < 2; ++m)
    #pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int t = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * e_lh;
                const int tx = t & (TT - 1), ty = (t >> TTL) & (TT - 1), f = t >> (2 * TTL);
                const int nf = cur.f0 + f;
                const unsigned o = (unsigned)(((nf * Hl + cur.oy0 + 2 * ty + p) * Wl + cur.ox0 + 2 * tx + q) * a.ldo + cur.cob0 * 32 + e_lr) * 4u;
                oo[m][r] = nf < a.nfr ? o : 0x80000000u;
            }
        const int obytes = a.nfr * Hl * Wl * a.ldo * 4;
        const auto osrc = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, obytes, 0x00020000);
        const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.res ? a.res : a.out), 0, a.res ? obytes : 0, 0x00020000);
        float* Zs = smem;                                                // [plane = 2i + q][m 2][reg16/4][e_lane 64][4]  (8 planes of 8 KB)
        const float sgn = p ? -1.f : 1.f;
        // the true M rows: row 3 of both U and V carries a flipped sign, their product does not
    #pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int co = (cur.cob0 + n) * 32 + e_lr;
            const float bv = a.bias ? a.bias[co] : 0.f;
            f32x16 rv[2];
            __syncthreads();                                             // every wave is done with the LDS (V / previous Z)
    #pragma unroll
            for (int m = 0; m < 2; ++m) {
    #pragma unroll
                for (int r = 0; r < 16; ++r) rv[m][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, oo[m][r], n * 128, 0));
                const f32x16 z0 = acc[m][0][n] + acc[m][1][n] + acc[m][2][n];
                const f32x16 z1 = acc[m][1][n] - acc[m][2][n] - acc[m][3][n];
    #pragma unroll
                for (int c4 = 0; c4 < 4; ++c4) {
                    float* d0 = Zs + ((((wi * 2 + 0) * 2 + m) * 4 + c4) * 64 + e_lane) * 4;
                    float* d1 = Zs + ((((wi * 2 + 1) * 2 + m) * 4 + c4) * 64 + e_lane) * 4;
                    *reinterpret_cast<f32x4*>(d0) = f32x4{z0[4 * c4], z0[4 * c4 + 1], z0[4 * c4 + 2], z0[4 * c4 + 3]};
                    *reinterpret_cast<f32x4*>(d1) = f32x4{z1[4 * c4], z1[4 * c4 + 1], z1[4 * c4 + 2], z1[4 * c4 + 3]};
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            __syncthreads();
            const float* zw = Zs + wi * 2048 + e_lane * 4;                 // Z[p + k][q] is plane wi + 2k
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
                        const int t = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * e_lh;
                        y[r] += a.fbias[(size_t)min(cur.f0 + (t >> (2 * TTL)), a.nfr - 1) * a.fbias_ld + co];
                    }
                }
                y += bv;
    #pragma unroll
                for (int r = 0; r < 16; ++r) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, (float)y[r]), osrc, oo[m][r], n * 128, 0);
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
            if (a.stats) {                                               // GroupNorm partial sums of the output (conv_wino.hip)
                constexpr int NFS = TF4 ? 4 : 1;
                __syncthreads();
                double* red = reinterpret_cast<double*>(smem);           // [wave 4][e_lh 2][fs][e_lr 32][2]
    #pragma unroll
                for (int fs = 0; fs < NFS; ++fs) {
                    double* d = red + ((((wi * 2 + e_lh) * NFS + fs) * 32 + e_lr) * 2);
                    d[0] = (double)gsum[fs][0]; d[1] = (double)gsum[fs][1];
                }
                __syncthreads();
                if (e_tid < NFS * 32) {
                    const int fs = e_tid >> 5, c = e_tid & 31;
                    double s = 0.0, ss = 0.0;
    #pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const double* d = red + (((k * NFS + fs) * 32 + c) * 2);
                        s += d[0]; ss += d[1];
                    }
                    const int nf = cur.f0 + fs;
                    const int sp = TF4 ? 0 : cur.byy * g.tiles_x + cur.bxx;
                    if (nf < a.nfr) {
                        double* o = a.stats + (((size_t)nf * a.stats_split + sp) * a.Cout + (cur.cob0 + n) * 32 + c) * 2;
                        o[0] = s; o[1] = ss;
                    }
                }
            }
            if (n == 0) S64_STAMP(6);
        }
        S64_STAMP(3);
    }
}

static bool s64_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

bool conv_wino_s64_supported(const IgemmArgs& a) {
    const int Hl = a.Hs << a.ups, Wl = a.Ws << a.ups;
    return a.wsplit == 2 && a.wwino != nullptr && a.ksz == 3 && a.stride == 1 && a.pad == 1 && Hl == Wl && s64_pow2(Hl) && Hl >= 8 &&
           a.Cout % 64 == 0 && a.Cin % 32 == 0 && a.src1 == nullptr && a.C0 == a.Cin && a.affA == nullptr && a.act == 0 &&
           (size_t)a.nfr * a.Hs * a.Ws * a.Cin < (1u << 29) && (size_t)a.Cin * a.Cout * 96 < (1u << 31) &&
           (size_t)a.nfr * Hl * Wl * a.ldo < (1u << 29) && (a.res == nullptr || a.res_ld == a.ldo);
}

int launch_conv_wino_s64(const IgemmArgs& a, hipStream_t s) {
    const int Hl = a.Hs << a.ups;
    VD_REQUIRE(a.stats == nullptr || a.stats_split == conv_wino_stats_split(Hl), "GroupNorm partial table: split");
    WinoS64Geom g;
    const int TT = Hl >= 16 ? 8 : 4;
    g.TF = 64 / (TT * TT);
    g.tiles_x = Hl / (2 * TT); g.tiles_y = Hl / (2 * TT);
    const size_t lds = 2 * 32768 + 2 * (g.TF == 4 ? 8 : 6) * 4096 + 8192;
    static bool attr = false;
    if (!attr) {
        const void* fns[2] = {reinterpret_cast<const void*>(&conv3x3_wino_s64_kernel<true>),
                              reinterpret_cast<const void*>(&conv3x3_wino_s64_kernel<false>)};
        for (const void* f : fns) VD_HIP(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr = true;
    }
    const int fgroups = (a.nfr + g.TF - 1) / g.TF;
    g.nbx = g.tiles_x * g.tiles_y * fgroups;
    g.ncb = a.Cout / 64;
    g.nitems = g.nbx * g.ncb;
    // weights of the whole layer (96 bytes per (cin, cout)) next to the streaming input in a 4 MB L2
    static const bool no_inner = getenv("VD_S64_NO_COB_INNER") != nullptr;      // A/B switch
    g.cob_inner = !no_inner && g.ncb > 1 && g.nbx % 8 == 0 && (size_t)a.Cin * a.Cout * 96 <= (size_t)(getenv("VD_S64_COB_MB") ? atoi(getenv("VD_S64_COB_MB")) : 3) << 20;
    static int ncu = 0;                               // one block per CU (512 registers per lane, > 110 KB of LDS): a persistent grid
    if (!ncu) {
        int dev = 0;
        VD_HIP(hipGetDevice(&dev));
        VD_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
    }
    dim3 grid(std::min(g.nitems, ncu));
    if (g.TF == 4) hipLaunchKernelGGL((conv3x3_wino_s64_kernel<true>), grid, dim3(256), lds, s, a, g);
    else hipLaunchKernelGGL((conv3x3_wino_s64_kernel<false>), grid, dim3(256), lds, s, a, g);
    VD_HIP(hipGetLastError());
    return 0;
}

// host: U = G g G^T (fp64, rounded once to fp32, row 3 negated), split into three bf16 pieces, packed
// [Cin/16][xi 16][Cout/32][piece 3][lane 64][8]: lane 32h+r holds U[xi][co = 32*blk + r][ci = 16*chunk + 8*h + e]
void split3_host(float v, unsigned short out[3]);
// one (cout co of O, cin ci of I) kernel gk[9] (fp64) -> its 16 x 3 pieces in the image
static void pack_wino_one(const double* gk, unsigned short* out, int O, int I, int co, int ci) {
    static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    const int ncoblk = O / 32;
    double tmp[4][3], U[4][4];
    for (int i = 0; i < 4; ++i)
        for (int c = 0; c < 3; ++c) tmp[i][c] = G[i][0] * gk[0 * 3 + c] + G[i][1] * gk[1 * 3 + c] + G[i][2] * gk[2 * 3 + c];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) U[i][j] = tmp[i][0] * G[j][0] + tmp[i][1] * G[j][1] + tmp[i][2] * G[j][2];
    const int ch = ci / 16, k = ci % 16, h = k >> 3, e = k & 7;
    const int cb = co >> 5, r = co & 31;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            unsigned short pc[3];
            split3_host((float)(i == 3 ? -U[i][j] : U[i][j]), pc);
            for (int q3 = 0; q3 < 3; ++q3)
                out[(((((size_t)ch * 16 + i * 4 + j) * ncoblk + cb) * 3 + q3) * 64 + h * 32 + r) * 8 + e] = pc[q3];
        }
}

void pack_conv3_wino_s64(const float* oihw, unsigned short* out, int O, int I) {
    for (int co = 0; co < O; ++co)
        for (int ci = 0; ci < I; ++ci) {
            double gk[9];
            for (int t = 0; t < 9; ++t) gk[t] = oihw[((size_t)co * I + ci) * 9 + t];
            pack_wino_one(gk, out, O, I, co, ci);
        }
}

// Upsample (nearest x2) + conv3x3 as four 3x3 kernels over the SOURCE map (conv_wino_r64.hip): output pixel (2y + a, 2x + b)
// reads source rows (y-1, y, y+1) with (w0, w1 + w2, 0) for a = 0 and (0, w0 + w1, w2) for a = 1, columns alike with b; the
// sums are formed in fp64.  Image of 4*O couts: cout ((b * O/32 + cb) * 2 + a) * 32 + r is phase (a, b) of real cout 32*cb + r.
void pack_conv3_wino_ups(const float* oihw, unsigned short* out, int O, int I) {
    const int ncb = O / 32;
    for (int co = 0; co < O; ++co)
        for (int ci = 0; ci < I; ++ci) {
            const float* w = oihw + ((size_t)co * I + ci) * 9;
            for (int a = 0; a < 2; ++a)
                for (int b = 0; b < 2; ++b) {
                    double rows[3][3], gk[9];
                    for (int c = 0; c < 3; ++c) {                     // vertical combination, per kernel column
                        const double w0 = w[0 * 3 + c], w1 = w[1 * 3 + c], w2 = w[2 * 3 + c];
                        rows[0][c] = a == 0 ? w0 : 0.0; rows[1][c] = a == 0 ? w1 + w2 : w0 + w1; rows[2][c] = a == 0 ? 0.0 : w2;
                    }
                    for (int r = 0; r < 3; ++r) {
                        gk[r * 3 + 0] = b == 0 ? rows[r][0] : 0.0;
                        gk[r * 3 + 1] = b == 0 ? rows[r][1] + rows[r][2] : rows[r][0] + rows[r][1];
                        gk[r * 3 + 2] = b == 0 ? 0.0 : rows[r][2];
                    }
                    const int cb = co >> 5, r32 = co & 31;
                    pack_wino_one(gk, out, 4 * O, I, ((b * ncb + cb) * 2 + a) * 32 + r32, ci);
                }
        }
}

}  // namespace vd
