// 3x3 stride-1 convolution by Winograd F(2x2,3x3) at fp32 accuracy on the 16-bit matrix cores -- the input transform and the
// operand split run IN THE FRAGMENT LAYOUT, in registers (gfx950).
//
// Block = 4 waves (one per SIMD, 512 registers each) = 64 tiles x 64 couts; wave i owns Winograd row i (4 positions x 2
// M-tiles x 2 cout tiles = 16 accumulator tiles).  V = B^T d B is formed in fp32 and split into 16-bit pieces (vd_common.h:
// two fp16 pieces and three piece products per element by default, F16; three bf16 pieces and six products with
// VD_MATH=bf16x6); U = G g G^T is formed in fp64 on the host and split there (split_pack.hip).  Lane (tile r, k-half h) of
// wave i reads the two patch rows its Winograd row combines -- 4 columns x 8 channels, sixteen ds_read_b128 per M-tile and
// chunk, straight from the raw patch -- and forms t = d[X] + s*d[S], the four column combinations and their split in its own
// registers: the result IS the A fragment.  No V image in LDS, ONE barrier per TWO chunks (patch hand-over), four patch
// buffers: a patch is requested three or four chunks ahead.
//
// Patch image (LDS-DMA: buffer_load_dwordx4 ... lds, lane l of a request writes 16 bytes at M0 + 16 l whatever address it
// gathers from, zeros where that address fails the descriptor's range check): [row 18][x parity 2][slot 10][64 B = 16
// channels of one pixel]; the four 16-byte quads of a pixel are stored at quad ^ ((row >> 1) & 3), so that the 16 lanes of
// a ds_read_b128 group (four tile rows x four tile columns) fall on 16 different bank quads.
#include <cstdlib>
#include <cstring>

#include "vd_common.h"


namespace vd {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <bool F16>
__device__ __forceinline__ f32x16 r64_mfma(u32x4 a, u32x4 b, f32x16 c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

struct WinoR64Geom { int tiles_x, tiles_y, nbx, ncb, nitems, xcd_order, ksplit; int phase_cb = 0; int cgroup = 0; };   // ksplit > 1: blockIdx.y = the block's slice of the channel chunks
// phase_cb > 0 (= real Cout / 32): the sub-pixel form of Upsample + conv (see conv3x3_wino_r64_ups_kernel)

// Geometry of an item's patch image.  TF4 = false: one frame, 8 x 8 tiles (maps >= 16 x 16).  TF4 = true: FOUR frames of an
// 8 x 8 map, 4 x 4 tiles each, a 10 x 10 patch per frame at a frame stride of FSB bytes (every stride a multiple of 256 bytes,
// so the bank argument for the swizzle holds across tile rows and frames alike).
template <bool TF4> struct R64G {
    static constexpr int P = TF4 ? 10 : 18;                 // patch width
    static constexpr int SPP = TF4 ? 6 : 10;                // 64-byte pixel slots per plane row (P / 2 pixels + 1 pad)
    static constexpr int PLB = SPP * 64, RSB = 2 * PLB;
    static constexpr int FSB = TF4 ? P * RSB : 0;           // frame stride
    static constexpr int NX = TF4 ? 8 : 6;                  // DMA instructions per thread and patch (256 threads x 16 B each)
    static constexpr int XBUF = NX * 4096;
    static constexpr int MOFF = TF4 ? 2 * FSB : 8 * RSB;    // second M-tile: two frames / four tile rows further
};
#ifndef VD_R64_ABL
#define VD_R64_ABL 0       // timing-only builds of the main loop, see below
#endif
namespace r64 {
constexpr int NB = 4;                       // patch buffers
// 4 patch buffers (98304 | 131072 bytes; the Z image, 64 KB, overlays them) + one more that only ever receives the requests
// past the item's last chunk (see x_dma): 122880 | 163840
template <bool TF4> constexpr int lds_bytes() { return (VD_R64_ABL & 32) ? 163840 : (NB + 1) * R64G<TF4>::XBUF; }
}  // namespace r64

#ifdef VD_WINO_TIMING
// cycle stamps of ONE work item (wave 0 of block 7): 0 start, 1 loop start, 2 loop end; per cout tile n (3 + 4n ..): Z image
// written, past the exchange barrier, output stored, statistics done; 14 / 15: the 100 MHz clock at start / end
__device__ unsigned long long g_r64_stamp[16];
#define R64_STAMP(i)                                                                                  \
    do {                                                                                              \
        if (threadIdx.x == 0 && blockIdx.x == 7) {                                                    \
            __builtin_amdgcn_sched_barrier(0);                                                        \
            g_r64_stamp[i] = (i) >= 14 ? __builtin_amdgcn_s_memrealtime() : __builtin_readcyclecounter(); \
            __builtin_amdgcn_sched_barrier(0);                                                        \
        }                                                                                             \
    } while (0)
extern "C" int vd_debug_r64_stamps(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_r64_stamp), sizeof(g_r64_stamp));
}
#else
#define R64_STAMP(i)
#endif
#ifndef VD_R64_B2REG
#define VD_R64_B2REG 1     // f16x3: the third weight piece 2^-12 b0 formed in registers (conv_wino_z128.hip) instead of loaded (A/B: 0)
#endif
// Variants that were built, measured and taken out again (the code lives in the history, commit aa1331c): the a1 piece by v_fma_mixlo/hi_f16 (10.31
// instead of 10.98 instructions per MFMA, same bits: 0 .. 5 % slower per layer, r05b); the patch staged through registers instead of LDS-DMA (worth 8 % in
// conv_wino_z128.hip, 2 .. 9 % SLOWER here, r05t: this loop has no issue slot left for twelve more requests per chunk pair); a uniform branch around the
// residual requests of a conv without residual (step 20.01 -> 20.14 ms, r05u: hipcc waits for everything in flight behind a conditional request).
// VD_R64_ABL: timing-only builds of the main loop (results WRONG; tools/build_variant.sh): bit 0 no weight reloads, 1 no
// transform / split, 2 no patch requests, 3 no patch reads, 4 no MFMA -- never set in the product library
// bit 5 (with bit 1): the loop as it would be if the A operand arrived ALREADY transformed and split (review r5 item 1a: V = B^T d B as
// (a0, a1) fp16 pairs written once per layer by a pre-pass): no transform / split, HALF the LDS reads (two ds_read_b128 per position instead
// of four), and the staging traffic of the V image -- 64 tiles x 16 positions x 16 channels x 4 B = 64 KB per chunk instead of the 24 / 32 KB
// patch: 16 - NX more LDS-DMA requests per thread and chunk, one per slot, read from the output buffer (one 4 KB line per request, the same
// lines for the cout blocks of a patch, as V would be shared) into LDS beyond the patch buffers.  A best case: no pre-pass is timed.

// block -> (tile group, first cout tile): blocks are dealt to the 8 XCDs round-robin; inside an XCD the cout blocks of one
// patch are neighbours
__device__ __forceinline__ void r64_item(const WinoR64Geom& g, int& bx, int& cob0) {
    if (g.xcd_order && g.cgroup > 0) {
        // sub-pixel form: 4 x the cout blocks (16 .. 32 weight slices of 1 - 2 MB against 4 MB of L2 per XCD).  With the cout
        // block as the fast index every slice had two concurrent readers per XCD (one at 512 couts) and the loop waited on
        // weights from beyond the L2: 1452 -> 1252 us only for a quarter fewer MFMAs, 338 -> 351 at 512 couts.  Here an XCD
        // walks ALL its patches with four cout blocks before it takes the next four.
        const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3, px = g.nbx >> 3;
        const int c_lo = loc % g.cgroup, rest = loc / g.cgroup;
        cob0 = ((rest / px) * g.cgroup + c_lo) * 2;
        bx = (rest % px) * 8 + xcd;
    } else if (g.xcd_order) {
        const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
        cob0 = (loc % g.ncb) * 2;
        bx = (loc / g.ncb) * 8 + xcd;
    } else {
        bx = blockIdx.x % g.nbx;
        cob0 = (blockIdx.x / g.nbx) * 2;
    }
}

// JS >= 0: the sub-pixel form of nearest-x2 Upsample + conv3x3 (unet.py:70-77).  Output pixel (2y + a, 2x + b) only sees a 2 x 2
// neighbourhood of the SOURCE map: rows (y-1, y) with weights (w0, w1 + w2) for a = 0, (y, y+1) with (w0 + w1, w2) for a = 1,
// the same along x -- four 3x3 kernels with one zero row and one zero column each, convolved with the low-resolution map
// (a.Cout = 4 x the real Cout; cout tile 2*blk + a of block blk = b * phase_cb + cb is phase (a, b) of real couts 32*cb ..).
// In the Winograd domain U = G g G^T of such a kernel has a zero ROW (3 for a = 0, 0 for a = 1) and a zero COLUMN JS (3 for
// b = 0, 0 for b = 1): the column is the same for both cout tiles of a block and is not computed at all -- three positions
// per group instead of four, three patch columns transformed instead of four; the zero row costs nothing to keep.  Same
// products as F(2x2,3x3) on the upsampled map would form, a quarter of them skipped.
template <bool TF4, bool F16, int JS>
__device__ __forceinline__ void r64_body(const IgemmArgs& a, const WinoR64Geom& g) {
    using namespace r64;
    constexpr int NP = JS < 0 ? 4 : 3;                               // positions of a group
    constexpr int JLa[4] = {JS == 0 ? 1 : 0, JS == 0 ? 2 : 1, JS == 0 ? 3 : 2, 3};     // position li of a group -> column j of the row
    constexpr int ORDa[4] = {JS == 0 ? 2 : 0, JS == 0 ? 1 : 2, JS == 0 ? 3 : 1, 3};    // the patch column position li recomputes for the next group
    constexpr bool PH = JS >= 0;
    constexpr int NSLOT = F16 ? 6 : 12;                              // MFMAs of a position: piece products x 2 cout tiles
    using G = R64G<TF4>;
    constexpr int P = G::P, SPP = G::SPP, PLB = G::PLB, RSB = G::RSB, NX = G::NX, XBUF = G::XBUF;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* const lds = reinterpret_cast<char*>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wi = __builtin_amdgcn_readfirstlane(tid >> 6);         // Winograd row of this wave
    const int lr = lane & 31, lh = lane >> 5;
    const int Hl = a.Hs << a.ups, Wl = a.Ws << a.ups;
    const int ncoblk = a.Cout >> 5;
    // split-K (small grids: a B = 1 shard leaves 32 .. 96 items for 256 CUs): block (item, ks) walks channel chunks
    // [ks * nchunk, (ks + 1) * nchunk) and writes its partial output -- no bias, residual or statistics -- to slab ks of a.out,
    // which is then the scratch [ksplit][pixels][Cout]; wino_r64_reduce_kernel sums the slabs
    const int nchunk = (a.Cin >> 4) / g.ksplit;
    const int ks = g.ksplit > 1 ? (int)blockIdx.y : 0;
    const int c_begin = ks * nchunk;

    int bx, cob0;
    r64_item(g, bx, cob0);
    const int bxx = bx % g.tiles_x; bx /= g.tiles_x;
    const int byy = bx % g.tiles_y; bx /= g.tiles_y;
    const int f0 = TF4 ? bx * 4 : bx;                                 // first frame of the item
    const int ox0 = bxx * 16, oy0 = byy * 16;                        // TF4: the whole 8 x 8 map (tiles_x = tiles_y = 1)
#if VD_R64_ABL & 32
    const int patch_id = (f0 * g.tiles_y + byy) * g.tiles_x + bxx;
#endif

    // ---- patch staging: thread -> 16-byte LDS slots e*256 + tid; the slot at quad position lq of patch row py holds the
    // pixel's quad lq ^ ((py >> 1) & 3)
    unsigned xo[NX];
#pragma unroll
    for (int e = 0; e < NX; ++e) {
        const int gs = e * 256 + tid, lq = gs & 3, ps0 = gs >> 2;
        const int fl = TF4 ? ps0 / (P * 2 * SPP) : 0, ps = TF4 ? ps0 % (P * 2 * SPP) : ps0;   // frame of the item, slot inside its image
        const int py = ps / (2 * SPP), r = ps % (2 * SPP), pxh = r % SPP, px = 2 * pxh + r / SPP;
        const int ly = oy0 + py - 1, lx = ox0 + px - 1;
        const bool in = fl < (TF4 ? 4 : 1) && f0 + fl < a.nfr && py < P && pxh < P / 2 && ly >= 0 && ly < Hl && lx >= 0 && lx < Wl;
        xo[e] = in ? (unsigned)(((f0 + fl) * a.Hs + (ly >> a.ups)) * a.Ws + (lx >> a.ups)) * (unsigned)(a.Cin * 4) + (unsigned)((lq ^ ((py >> 1) & 3)) * 16) +
                         (unsigned)(c_begin * 64)
                   : 0x80000000u;
    }
    const auto xsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.src0), 0, a.nfr * a.Hs * a.Ws * a.Cin * 4, 0x00020000);
    typedef __attribute__((address_space(3))) void* lds_ptr;
    // A request past the item's last chunk must not land in the output transform's Z image.  It is NOT skipped by a branch:
    // hipcc's s_waitcnt insertion merges the two paths of a conditional request to the one with FEWER loads in flight, i.e.
    // every wait for a weight fragment behind it becomes a wait for the patch itself.  The request always issues; when it is
    // late it goes through a descriptor of zero records (no memory access, zeros) into the spare fifth buffer.
    const auto xnull = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.src0), 0, 0, 0x00020000);
    auto x_dma_one = [&](int chunk, int e) {
#if defined(__HIP_DEVICE_COMPILE__)      // (hipcc's host pass drops a kernel whose body names this builtin)
        const bool live = chunk < nchunk;
        const int bufi = live ? (chunk & (NB - 1)) : NB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(live ? xsrc : xnull, (lds_ptr)(lds + bufi * XBUF + e * 4096 + wi * 1024), 16, xo[e], chunk * 64, 0, 0);
#endif
    };
    auto x_dma = [&](int chunk) {
#pragma unroll
        for (int e = 0; e < NX; ++e) x_dma_one(chunk, e);
    };
#if VD_R64_ABL & 32
    const unsigned xtra_bytes = (unsigned)(a.nfr * Hl * Wl * a.ldo * 4);
    const auto xtra_src = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, xtra_bytes, 0x00020000);
    const unsigned xtra_span = (unsigned)((nchunk + 4) * (16 - NX) * 4096);                    // the PATCH's lines (shared by its cout blocks)
    const unsigned xtra_base = ((unsigned)patch_id * xtra_span) % (xtra_bytes - xtra_span) + (unsigned)(tid * 16);
    auto x_extra = [&](int chunk, int i) {
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xtra_src, (lds_ptr)(lds + NB * XBUF + i * 4096 + wi * 1024), 16, xtra_base, (chunk * (16 - NX) + i) * 4096, 0, 0);
#endif
    };
#endif


    // ---- transform in the fragment layout.  Lane (tile lr of the M-tile, k-half lh): tile column lr & 7, tile row 4m + (lr >> 3);
    // rows of B^T as d[X] + s*d[S]: 0: d0 - d2, 1: d1 + d2, 2: d2 - d1, 3: d3 - d1 (row 3 of U is negated on the host)
    const int rowX = wi, rowS = wi < 2 ? 2 : 1;
    const float tsg = wi == 1 ? 1.f : -1.f;
    // TF4: tile column lr & 3, tile row (lr >> 2) & 3, frame 2m + (lr >> 4)
    const int ttx = TF4 ? lr & 3 : lr & 7, ttyl = TF4 ? (lr >> 2) & 3 : lr >> 3, tfl = TF4 ? lr >> 4 : 0;
    int adr[2][2];                                                    // [X | S][quad h of the lane's eight channels]
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        adr[0][h] = tfl * G::FSB + (2 * ttyl + rowX) * RSB + ttx * 64 + (((2 * lh + h) ^ ((ttyl + (rowX >> 1)) & 3)) * 16);
        adr[1][h] = tfl * G::FSB + (2 * ttyl + rowS) * RSB + ttx * 64 + (((2 * lh + h) ^ ((ttyl + (rowS >> 1)) & 3)) * 16);
    }
    float t[4][8];                                                    // t[column][channel] of the group being transformed
    f32x4 stx[2], sts[2];                                             // one column of the patch rows X and S, in flight
    float tv[2][8], rr[8];                                            // V of a position (F16: per fragment buffer, its remainder is formed one position later)
    u32x4 af[2][3];                                                   // A fragments [buffer][piece]
    // column c of group (chunk, m): four reads
    auto t_read = [&](int chunk, int m, int c) {
        const char* rb = lds + (chunk & (NB - 1)) * XBUF + m * G::MOFF + (c & 1) * PLB + (c >> 1) * 64;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            stx[h] = *reinterpret_cast<const f32x4*>(rb + adr[0][h]);
            if (!(VD_R64_ABL & 32)) sts[h] = *reinterpret_cast<const f32x4*>(rb + adr[1][h]);
        }
    };
    auto t_fma1 = [&](int c, int e) {          // plain v_fma_f32: hipcc pairs these into v_pk_fma_f32, which does not co-issue with the MFMA
        asm("v_fma_f32 %0, %1, %2, %3" : "=v"(t[c][e]) : "v"(tsg), "v"(sts[e >> 2][e & 3]), "v"(stx[e >> 2][e & 3]));
    };
    auto t_fma = [&](int c, int h) {
#pragma unroll
        for (int e = 4 * h; e < 4 * h + 4; ++e) t_fma1(c, e);
    };
    auto t_comb1 = [&](int buf, int j, int e) {                      // column combination of position j, channel e
        // (asm: hipcc otherwise pairs neighbouring channels into v_pk_add_f32)
        if (j == 0) asm("v_sub_f32 %0, %1, %2" : "=v"(tv[buf][e]) : "v"(t[0][e]), "v"(t[2][e]));
        else if (j == 1) asm("v_add_f32 %0, %1, %2" : "=v"(tv[buf][e]) : "v"(t[1][e]), "v"(t[2][e]));
        else if (j == 2) asm("v_sub_f32 %0, %1, %2" : "=v"(tv[buf][e]) : "v"(t[2][e]), "v"(t[1][e]));
        else asm("v_sub_f32 %0, %1, %2" : "=v"(tv[buf][e]) : "v"(t[1][e]), "v"(t[3][e]));
    };
    auto t_comb = [&](int buf, int j, int h) {
#pragma unroll
        for (int e = 4 * h; e < 4 * h + 4; ++e) t_comb1(buf, j, e);
    };
    // bf16x6: exact three-way split of channels 2pr, 2pr+1 (vd_common.h: split_a / split_b)
    auto t_split_a = [&](int buf, int pr) {
        unsigned p1;
        split_a(tv[buf][2 * pr], tv[buf][2 * pr + 1], p1, rr[2 * pr], rr[2 * pr + 1], 0x07060302u);
        af[buf][0][pr] = p1;
    };
    auto t_split_b = [&](int buf, int pr) {
        unsigned p2, p3;
        split_b(rr[2 * pr], rr[2 * pr + 1], p2, p3, 0x07060302u);
        af[buf][1][pr] = p2;
        af[buf][2][pr] = p3;
    };
    // f16x3: a0 of channels 2pr, 2pr+1 (vd_common.h); the prologue's form
    auto t_f16_a0 = [&](int buf, int pr) { af[buf][0][pr] = f16_pack(tv[buf][2 * pr], tv[buf][2 * pr + 1]); };
    // The slot bodies of the f16x3 loop are ONE asm statement each.  Single-instruction statements would let hipcc interleave them,
    // but it pads every asm output that the next instruction reads with an s_nop (it cannot know the producer is a plain VALU
    // instruction), and with one wave per SIMD an s_nop is an issue slot like any other: the first f16x3 loop of round 4 carried
    // 9 of them per position next to 36 vector instructions, and its six scalar adds for the weight offsets on top.
    // Slot k < 4: the a1 piece of channel pair k of THIS position's fragment -- r = x - a0 (v_fma_mix_f32 reads the fp16 half),
    // r * 2^12, round -- and channels 2k, 2k+1 of the NEXT position's V (column combination X -/+ Y).  `early`: the piece is read
    // by the MFMA of the next slot, and a VALU write needs two wait states before an MFMA reads it as A / B: the combination
    // goes behind the conversion.
#define VD_R64_A1 "v_fma_mix_f32 %3, %5, -1.0, %6 op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %4, %5, -1.0, %7 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t" \
                  "v_ldexp_f32 %3, %3, 12\n\tv_ldexp_f32 %4, %4, 12\n\t"
#define VD_R64_A1_OPS : "=&v"(af[cur][1][k]), "=&v"(tv[nxt][2 * k]), "=&v"(tv[nxt][2 * k + 1]), "=&v"(r0), "=&v"(r1) \
                      : "v"(af[cur][0][k]), "v"(tv[cur][2 * k]), "v"(tv[cur][2 * k + 1]), "v"(t[cx][2 * k]), "v"(t[cy][2 * k]), "v"(t[cx][2 * k + 1]), "v"(t[cy][2 * k + 1])
    auto f16_slot_a = [&](int cur, int nxt, int k, int jn, bool early) {
        const int cx = jn == 0 ? 0 : jn == 2 ? 2 : 1, cy = jn == 0 ? 2 : jn == 1 ? 2 : jn == 2 ? 1 : 3;      // V[jn] = t[cx] - t[cy] (jn = 1: +)
        float r0, r1;
        if (jn == 1) {
            if (early) asm(VD_R64_A1 "v_cvt_pk_f16_f32 %0, %3, %4\n\tv_add_f32 %1, %8, %9\n\tv_add_f32 %2, %10, %11" VD_R64_A1_OPS);
            else asm(VD_R64_A1 "v_add_f32 %1, %8, %9\n\tv_add_f32 %2, %10, %11\n\tv_cvt_pk_f16_f32 %0, %3, %4" VD_R64_A1_OPS);
        } else {
            if (early) asm(VD_R64_A1 "v_cvt_pk_f16_f32 %0, %3, %4\n\tv_sub_f32 %1, %8, %9\n\tv_sub_f32 %2, %10, %11" VD_R64_A1_OPS);
            else asm(VD_R64_A1 "v_sub_f32 %1, %8, %9\n\tv_sub_f32 %2, %10, %11\n\tv_cvt_pk_f16_f32 %0, %3, %4" VD_R64_A1_OPS);
        }
    };
#undef VD_R64_A1
#undef VD_R64_A1_OPS
    // Slot 4: the a0 piece of the next position's fragment (four conversions) + channels 0, 1 of the next group's t column c
    auto f16_slot_b = [&](int nxt, int c) {
        asm("v_cvt_pk_f16_f32 %0, %6, %7\n\tv_cvt_pk_f16_f32 %1, %8, %9\n\tv_cvt_pk_f16_f32 %2, %10, %11\n\tv_cvt_pk_f16_f32 %3, %12, %13\n\t"
            "v_fma_f32 %4, %14, %15, %16\n\tv_fma_f32 %5, %14, %17, %18"
            : "=&v"(af[nxt][0][0]), "=&v"(af[nxt][0][1]), "=&v"(af[nxt][0][2]), "=&v"(af[nxt][0][3]), "=&v"(t[c][0]), "=&v"(t[c][1])
            : "v"(tv[nxt][0]), "v"(tv[nxt][1]), "v"(tv[nxt][2]), "v"(tv[nxt][3]), "v"(tv[nxt][4]), "v"(tv[nxt][5]), "v"(tv[nxt][6]), "v"(tv[nxt][7]),
              "v"(tsg), "v"(sts[0][0]), "v"(stx[0][0]), "v"(sts[0][1]), "v"(stx[0][1]));
    };
    // Slot 5: channels 2 .. 7 of that column
    auto f16_slot_c = [&](int c) {
        asm("v_fma_f32 %0, %6, %7, %8\n\tv_fma_f32 %1, %6, %9, %10\n\tv_fma_f32 %2, %6, %11, %12\n\tv_fma_f32 %3, %6, %13, %14\n\t"
            "v_fma_f32 %4, %6, %15, %16\n\tv_fma_f32 %5, %6, %17, %18"
            : "=&v"(t[c][2]), "=&v"(t[c][3]), "=&v"(t[c][4]), "=&v"(t[c][5]), "=&v"(t[c][6]), "=&v"(t[c][7])
            : "v"(tsg), "v"(sts[0][2]), "v"(stx[0][2]), "v"(sts[0][3]), "v"(stx[0][3]), "v"(sts[1][0]), "v"(stx[1][0]), "v"(sts[1][1]), "v"(stx[1][1]),
              "v"(sts[1][2]), "v"(stx[1][2]), "v"(sts[1][3]), "v"(stx[1][3]));
    };

    // ---- B fragments: U[chunk][xi = 4*wi + j][cob][piece][lane][8 x 16 bit] = 1 KiB per (chunk, xi, cob, piece)
    const auto usrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wwino), 0, 16 * a.Cout * a.Cin * 6, 0x00020000);
    const int ustride = 16 * ncoblk * 3072, bstep = ncoblk * 3072;
    const int bsb = (wi * 4 * ncoblk + cob0) * 3072 + c_begin * ustride;
    const unsigned blane = lane * 16u;
    u32x4 bfr[4][2][3];
    // (n, piece) offsets 0 .. 3072 ride in the instruction's 12-bit immediate, the last two behind a second scalar base: two
    // scalar adds per position instead of six (every instruction of a one-wave-per-SIMD stream is an issue slot)
    constexpr bool B2R = F16 && VD_R64_B2REG;                        // piece 2 = 2^-12 x piece 0: four v_pk_mul_f16 instead of a 1 KiB load
    const unsigned two_m12 = 0x0c000c00u;
    auto b_third = [&](int j, int n) {
        asm("v_pk_mul_f16 %0, %4, %8\n\tv_pk_mul_f16 %1, %5, %8\n\tv_pk_mul_f16 %2, %6, %8\n\tv_pk_mul_f16 %3, %7, %8"
            : "=&v"(bfr[j][n][2][0]), "=&v"(bfr[j][n][2][1]), "=&v"(bfr[j][n][2][2]), "=&v"(bfr[j][n][2][3])
            : "v"(bfr[j][n][0][0]), "v"(bfr[j][n][0][1]), "v"(bfr[j][n][0][2]), "v"(bfr[j][n][0][3]), "s"(two_m12));
    };
    auto b_load_one = [&](int chunk, int j, int n, int p) {
        if (B2R && p == 2) return;
        const int idx = n * 3 + p, so = chunk * ustride + bsb + j * bstep;
        bfr[j][n][p] = idx < 4 ? __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(usrc, blane + idx * 1024u, so, 0))
                               : __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(usrc, blane + (idx - 4) * 1024u, so + 4096, 0));
    };
    auto b_load = [&](int chunk, int j, int n) {
#pragma unroll
        for (int p = 0; p < 3; ++p) b_load_one(chunk, j, n, p);
    };

    f32x16 acc[2][4][2];                                              // [m][j][n]; zeroed behind the prologue's requests
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int p = 0; p < 3; ++p) af[b][p] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int p = 0; p < 3; ++p) bfr[j][n][p] = u32x4{0u, 0u, 0u, 0u};

    R64_STAMP(0); R64_STAMP(14);
    // ---- prologue: three patches and the weights of chunk 0 requested; group (0, 0) transformed whole, position 0 split,
    // column 0 of group (0, 1) in flight -- the state the loop expects at the top of a group
#pragma unroll
    for (int c = 0; c < 3; ++c) x_dma(c);
#pragma unroll
    for (int li = 0; li < NP - 1; ++li) { b_load(0, JLa[li], 0); b_load(0, JLa[li], 1); }
    // the 256 accumulator writes (1 k cycles of issue) go under the wait for the first patch instead of behind it
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (j != JS) asm volatile("v_accvgpr_write_b32 %0, 0" : "=a"(acc[m][j][n][r]));
    __builtin_amdgcn_sched_barrier(0);
    // patch 0 has landed in every wave; patches 1, 2 (first read behind the loop's first barrier, which waits for them: 2 * NX requests)
    // and the (NP - 1) * 6 (4: two loads per fragment) weight loads may be in flight
    {
        constexpr int W = 2 * NX + (NP - 1) * (B2R ? 4 : 6);
        static_assert(W == 20 || W == 24 || W == 28 || W == 30 || W == 34, "wait count of the prologue");
        if constexpr (W == 20) asm volatile("s_waitcnt vmcnt(20)\n\ts_barrier" ::: "memory");
        else if constexpr (W == 24) asm volatile("s_waitcnt vmcnt(24)\n\ts_barrier" ::: "memory");
        else if constexpr (W == 28) asm volatile("s_waitcnt vmcnt(28)\n\ts_barrier" ::: "memory");
        else if constexpr (W == 30) asm volatile("s_waitcnt vmcnt(30)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(34)\n\ts_barrier" ::: "memory");
    }
#pragma unroll
    for (int c = 0; c < 4; ++c)
        if (JS < 0 || (JS == 3 ? c < 3 : c > 0)) { t_read(0, 0, c); t_fma(c, 0); t_fma(c, 1); }
    t_comb(0, JLa[0], 0); t_comb(0, JLa[0], 1);
    if constexpr (F16) {
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) t_f16_a0(0, pr);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_nop 1");                 // the first MFMA reads af[0][0]
        __builtin_amdgcn_sched_barrier(0);
    } else {
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) t_split_a(0, pr);
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) t_split_b(0, pr);
    }
    t_read(0, 1, ORDa[0]);

    R64_STAMP(1);
    // ---- main loop.  Group (chunk, m) = NP positions x NSLOT slots; a slot = one MFMA + a share of the vector work + at most
    // one memory request (requests in bursts block the wave's issue and starve the matrix pipe: one per slot, never more).
    //   bf16x6 (12 slots): slot k = product k >> 1 of (A2,B0) (A1,B1) (A1,B0) (A0,B2) (A0,B1) (A0,B0), cout tile k & 1, + <= 6
    //     vector instructions of the NEXT position's fragment (k 0, 3: column combination; 1, 2, 4, 5: first split step of a
    //     channel pair; 6..9: second), k 10, 11: the next group's t (column ord[j], in place: its last reader ran earlier in
    //     this group) + the reads of the column after it.
    //   f16x3 (6 slots): slot k = product k >> 1 of (A0,B0) (A0,B1) (A1,B2), cout tile k & 1; the fragment's a0 piece is
    //     complete when its position starts, its a1 piece is formed in slots 0..3 (one channel pair each: 4 instructions) and
    //     first read in slot 4; beside it slots 0..3 form the NEXT position's V (2 channels each), slot 4 packs its a0 and
    //     starts the next group's t column, slot 5 finishes it and issues the reads of the column after it: 6 per slot.
    // Weights: one register set, fragment (j, n, piece) reloaded in the position after its last use, one load per slot.
    // Patch of chunk c: first read in position 3 of group (c - 1, 0), last read in position 2 of group (c, 0); the block's only
    // barrier sits in front of position NP - 1 of the EVEN groups (c, 0): patches c + 1, c + 2 (requested two chunks ago) have
    // landed in every wave, and the buffers of c - 1, c are free for c + 3, c + 4.  Those are requested in the odd chunk,
    // BEHIND its last weight loads (position 0 of group (c, 0)): loads return in order, and a weight fragment requested
    // behind a patch would wait for the patch's HBM round trip.
    constexpr int PA6[6] = {2, 1, 1, 0, 0, 0}, PB6[6] = {0, 1, 0, 2, 1, 0};
    constexpr int PA3[3] = {0, 0, 1}, PB3[3] = {0, 1, 2};
    // two chunks per trip: the chunk's parity (which decides the barrier and the patch requests) is a compile-time constant,
    // so the wait counts hipcc derives for the weight fragments are exact on every path
    for (int chunk0 = 0; chunk0 < nchunk; chunk0 += 2) {
#pragma unroll
      for (int cpar = 0; cpar < 2; ++cpar) {
        const int chunk = chunk0 + cpar;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
#pragma unroll
            for (int li = 0; li < NP; ++li) {
                const int j = JLa[li];
                const int cur = ((cpar * 2 + m) * NP + li) & 1, nxt = cur ^ 1;      // the A fragment buffers alternate per position
                const int jn = JLa[(li + 1) % NP];                                  // the next position: li + 1 of this group, or 0 of the next one
#pragma unroll
                for (int k = 0; k < NSLOT; ++k) {
                    const int q = k >> 1, n = k & 1;
                    if (m == 0 && li == NP - 1 && k == 0 && cpar == 0) {
                        // the 18 (12: two loads per fragment) youngest requests are weight loads
                        if constexpr (B2R) asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                        else asm volatile("s_waitcnt vmcnt(18) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                    }
                    // the patch requests of the chunk pair, one per slot
                    if (cpar == 1 && !(VD_R64_ABL & 4)) {
                        if constexpr (F16) {
                            const int ord = m == 0 ? (li - 1) * 6 + k : (li == 0 ? (NP - 1) * 6 + k : 99);      // slots behind position 0 of group (c, 0)
                            if ((m == 1 || li > 0) && ord < 2 * NX) x_dma_one(chunk + 2 + ord / NX, ord % NX);
                        } else {
                            if (m == 0 && (li == 0 || li == 1) && k >= 12 - NX) x_dma_one(chunk + 2 + li, k - (12 - NX));   // the last NX slots of positions 0, 1
                        }
                    }
#if VD_R64_ABL & 32
                    if (cpar == 0 && F16) {                          // the V image's extra staging requests: 2 * (16 - NX) per chunk pair, one per slot
                        const int slot = m == 0 ? (li - 1) * 6 + k : (li == 0 ? (NP - 1) * 6 + k : 99);
                        if ((m == 1 || li > 0) && slot < 2 * (16 - NX)) x_extra(chunk + slot / (16 - NX), slot % (16 - NX));
                    }
#endif
                    if (VD_R64_ABL & 16) {}
                    else if constexpr (F16) acc[m][j][n] = r64_mfma<true>(af[cur][PA3[q]], bfr[j][n][PB3[q]], acc[m][j][n]);
                    else acc[m][j][n] = r64_mfma<false>(af[cur][PA6[q]], bfr[j][n][PB6[q]], acc[m][j][n]);
                    if (B2R && m == 0 && (k == 1 || k == 2)) b_third(j, k - 1);      // first read in slot 4 / 5; the M-tile group m = 1 finds it in place
                    if (VD_R64_ABL & 32) {                          // the two ds_read_b128 of a position ARE its fragment's pieces
                        if (k == 4) { af[nxt][0] = __builtin_bit_cast(u32x4, stx[0]); af[nxt][1] = __builtin_bit_cast(u32x4, stx[1]); }
                    } else if (VD_R64_ABL & 2) {}
                    else if constexpr (F16) {
                        if (k < 4) f16_slot_a(cur, nxt, k, jn, k == 3);
                        else if (k == 4) f16_slot_b(nxt, ORDa[li]);
                        else f16_slot_c(ORDa[li]);
                    } else {
                        if (k == 0) t_comb(nxt, jn, 0);
                        else if (k == 3) t_comb(nxt, jn, 1);
                        else if (k == 1 || k == 2) t_split_a(nxt, k - 1);
                        else if (k == 4 || k == 5) t_split_a(nxt, k - 2);
                        else if (k >= 6 && k < 10) t_split_b(nxt, k - 6);
                        else t_fma(ORDa[li], k - 10);
                    }
                    if (k == NSLOT - 1 && !(VD_R64_ABL & 8)) {
                        // the reads of the column after the one just transformed: of the group after this one -- (chunk, 1) or
                        // (chunk + 1, 0) -- or, from the last position, of the one after that
                        if (li < NP - 1) t_read(m == 0 ? chunk : chunk + 1, m ^ 1, ORDa[li + 1]);
                        else t_read(chunk + 1, m, ORDa[0]);
                    }
                    // weights: (j - 1, n) of the next chunk once its last product has issued; (3, n) in position 0 of the next group
                    if (k < 6 && !(VD_R64_ABL & 1)) {                 // one weight load per slot: (n, piece) = (k / 3, k % 3)
                        if (m == 1 && li > 0) b_load_one(chunk + 1, JLa[li - 1], k / 3, k % 3);
                        if (m == 0 && li == 0) b_load_one(chunk, JLa[NP - 1], k / 3, k % 3);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

    R64_STAMP(2);
    // ---- output transform, one cout tile at a time: Z[q] = sum_j M[wi][j] A[j][q] is wave-local, the sum over the rows
    // crosses the waves through LDS; wave (p, q) = (wi >> 1, wi & 1) then owns output pixel (p, q) of every tile.
    // Z image: [plane 2*i + q 8][m 2][c4 4][lane 64][4 floats] = 64 KB over the patch buffers.
    const int p = wi >> 1, q = wi & 1;
    // sub-pixel form: block blk = cob0 / 2 = pb * phase_cb + cb; cout tile n is phase (n, pb) of real couts 32*cb .. 32*cb + 31, its
    // pixel (y, x) of the low-resolution map goes to (2y + n, 2x + pb) of the output
    const int pcb = PH ? (cob0 >> 1) % g.phase_cb : 0, ppb = PH ? (cob0 >> 1) / g.phase_cb : 0;
    const int Ho = PH ? 2 * Hl : Hl, Wo = PH ? 2 * Wl : Wl;
    const int obytes = a.nfr * Ho * Wo * a.ldo * 4;
    const auto osrc = __builtin_amdgcn_make_buffer_rsrc(a.out + (size_t)ks * (obytes >> 2), 0, obytes, 0x00020000);
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.res ? a.res : a.out), 0, a.res ? obytes : 0, 0x00020000);
    const float sgn = p ? -1.f : 1.f;
    float* Zs = smem;
    unsigned oo[2][16];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int tt = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int tx = TF4 ? tt & 3 : tt & 7, ty = TF4 ? (tt >> 2) & 3 : tt >> 3, nf = f0 + (TF4 ? tt >> 4 : 0);
            const unsigned o = PH ? (unsigned)(((nf * Ho + 2 * (oy0 + 2 * ty + p)) * Wo + 2 * (ox0 + 2 * tx + q) + ppb) * a.ldo + pcb * 32 + lr) * 4u
                                  : (unsigned)(((nf * Hl + oy0 + 2 * ty + p) * Wl + ox0 + 2 * tx + q) * a.ldo + cob0 * 32 + lr) * 4u;
            oo[m][r] = nf < a.nfr ? o : 0x80000000u;                 // TF4: a frame past the end is neither read nor stored
        }
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int co = PH ? pcb * 32 + lr : (cob0 + n) * 32 + lr;
        const int nso = PH ? n * Wo * a.ldo * 4 : n * 128;             // byte offset of cout tile n: one output row down | 32 channels on
        const float bias = a.bias ? a.bias[co] : 0.f;
        // F16: the weight row's power-of-two scale leaves here (image trailer: [Cout] s, [Cout] 1 / s; Cout = the image's 4 x real in the sub-pixel form)
        const float winv = F16 ? (a.wwino + (size_t)24 * a.Cout * a.Cin)[a.Cout + (cob0 + n) * 32 + lr] : 1.f;
        // per-frame bias: TF1 one frame; TF4 registers 0..7 of M-tile m belong to frame 2m, 8..15 to frame 2m + 1
        float bvf[2][2];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
                bvf[m][hf] = bias + (a.fbias ? a.fbias[(size_t)min(f0 + (TF4 ? 2 * m + hf : 0), a.nfr - 1) * a.fbias_ld + co] : 0.f);
        f32x16 rv[2];
        if (n) __syncthreads();                                      // the previous Z is no longer read
#pragma unroll
        for (int m = 0; m < 2; ++m) {
#pragma unroll
            for (int r = 0; r < 16; ++r) rv[m][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, oo[m][r], nso, 0));
            // Z = M A: columns (1, 1, 1, 0) and (0, 1, -1, -1); the column the sub-pixel form never computed is zero
            const f32x16 z0 = JS == 0 ? acc[m][1][n] + acc[m][2][n] : acc[m][0][n] + acc[m][1][n] + acc[m][2][n];
            const f32x16 z1 = JS == 3 ? acc[m][1][n] - acc[m][2][n] : acc[m][1][n] - acc[m][2][n] - acc[m][3][n];
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) {
                *reinterpret_cast<f32x4*>(Zs + ((((wi * 2 + 0) * 2 + m) * 4 + c4) * 64 + lane) * 4) = f32x4{z0[4 * c4], z0[4 * c4 + 1], z0[4 * c4 + 2], z0[4 * c4 + 3]};
                *reinterpret_cast<f32x4*>(Zs + ((((wi * 2 + 1) * 2 + m) * 4 + c4) * 64 + lane) * 4) = f32x4{z1[4 * c4], z1[4 * c4 + 1], z1[4 * c4 + 2], z1[4 * c4 + 3]};
            }
        }
        R64_STAMP(3 + 4 * n);
        __syncthreads();
        R64_STAMP(4 + 4 * n);
        const float* zw = Zs + wi * 2048 + lane * 4;                 // Z[p + k][q] is plane wi + 2k
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
            if constexpr (F16) y = y * winv + rv[m];
            else y += rv[m];
#pragma unroll
            for (int r = 0; r < 16; ++r) y[r] += bvf[m][r >> 3];
#pragma unroll
            for (int r = 0; r < 16; ++r) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, (float)y[r]), osrc, oo[m][r], nso, 0);
            if (a.stats) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int fs = TF4 ? 2 * m + (r >> 3) : 0;
                    // explicit fma: left to -ffp-contract, hipcc fused the square into the sum in one unrolled copy of this loop
                    // and not in the other -- a frame's statistics then depended on its place in a four-frame item (r04m)
                    gsum[fs][0] += y[r]; gsum[fs][1] = __builtin_fmaf(y[r], y[r], gsum[fs][1]);
                }
            }
        }
        R64_STAMP(5 + 4 * n);
        if (a.stats) {                                               // GroupNorm partial sums of the output
            constexpr int NFS = TF4 ? 4 : 1;
            __syncthreads();
            double* red = reinterpret_cast<double*>(smem);           // [wave 4][lh 2][frame NFS][lr 32][2]
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
                for (int k = 0; k < 8; ++k) { s += red[((k * NFS + fs) * 32 + c) * 2]; ss += red[((k * NFS + fs) * 32 + c) * 2 + 1]; }
                const int nf = f0 + fs, sp = TF4 ? 0 : byy * g.tiles_x + bxx;
                if (nf < a.nfr) {
                    // sub-pixel form: four table entries per tile group, one per phase, over the real couts
                    double* o = PH ? a.stats + (((size_t)nf * a.stats_split + sp * 4 + 2 * n + ppb) * (g.phase_cb * 32) + pcb * 32 + c) * 2
                                   : a.stats + (((size_t)nf * a.stats_split + sp) * a.Cout + (cob0 + n) * 32 + c) * 2;
                    o[0] = s; o[1] = ss;
                }
            }
        }
        R64_STAMP(6 + 4 * n);
    }
    R64_STAMP(15);
}

template <bool TF4, bool F16>
__global__ __launch_bounds__(256, 1) void conv3x3_wino_r64_kernel(IgemmArgs a, WinoR64Geom g) { r64_body<TF4, F16, -1>(a, g); }

template <bool TF4, bool F16>
__global__ __launch_bounds__(256, 1) void conv3x3_wino_r64_ups_kernel(IgemmArgs a, WinoR64Geom g) {
    int bx, cob0;
    r64_item(g, bx, cob0);
    if ((cob0 >> 1) < g.phase_cb) r64_body<TF4, F16, 3>(a, g);       // phases (., 0): column 3 of U is zero
    else r64_body<TF4, F16, 0>(a, g);                                // phases (., 1): column 0
}

template <auto KERN>
static int r64_launch(dim3 grid, size_t lds, hipStream_t s, const IgemmArgs& k, const WinoR64Geom& g) {
    VD_RAISE_LDS(KERN, (size_t)160 * 1024);      // per kernel and device: each one that runs raises its own LDS limit once
    hipLaunchKernelGGL(KERN, grid, dim3(256), lds, s, k, g);
    VD_HIP(hipGetLastError());
    return 0;
}

static bool r64_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

bool conv_wino_r64_supported(const IgemmArgs& a) {
    const int Hl = a.Hs << a.ups, Wl = a.Ws << a.ups;
    return a.wsplit == 2 && a.wwino != nullptr && a.ksz == 3 && a.stride == 1 && a.pad == 1 && Hl == Wl && r64_pow2(Hl) && Hl >= 8 &&
           a.Cout % 64 == 0 && a.Cin % 32 == 0 && a.src1 == nullptr && a.C0 == a.Cin && a.affA == nullptr && a.act == 0 &&
           (size_t)a.nfr * a.Hs * a.Ws * a.Cin < (1u << 29) && (size_t)a.Cin * a.Cout * 96 < (1u << 31) &&
           (size_t)a.nfr * Hl * Wl * a.ldo < (1u << 29) && (a.res == nullptr || a.res_ld == a.ldo);
}

// ---- split-K for small grids ------------------------------------------------------------------------------------
// out[m][c] = sum_s part[s][m][c] + bias[c] + fbias[frame][c] + res[m][c], and the GroupNorm partial sums of the result in
// the layout the one-launch epilogue writes ([frame][split = 1][Cout][2] doubles: maps <= 16 x 16 only).  Block = one frame
// x 16 channels: thread (channel quad, pixel phase of 64) runs HW / 64 <= 4 iterations of independent 16-byte loads (the
// first version, one frame x 64 channels per block with 64 dependent iterations, took longer than the convolution).
__global__ __launch_bounds__(256) void wino_r64_reduce_kernel(const float* __restrict__ part, int S, size_t slab, const float* __restrict__ bias,
                                                                  const float* __restrict__ fbias, int fbias_ld, const float* __restrict__ res, int res_ld,
                                                                  float* __restrict__ out, int ldo, int HW, int Cout, double* __restrict__ stats) {
    const int f = blockIdx.x, q = threadIdx.x & 3, ph = threadIdx.x >> 2, c = blockIdx.y * 16 + q * 4;
    f32x4 b = bias ? *reinterpret_cast<const f32x4*>(bias + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    if (fbias) b += *reinterpret_cast<const f32x4*>(fbias + (size_t)f * fbias_ld + c);
    double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    for (int p = ph; p < HW; p += 64) {
        const size_t m = (size_t)f * HW + p;
        f32x4 v = *reinterpret_cast<const f32x4*>(part + m * Cout + c);
        for (int k = 1; k < S; ++k) v += *reinterpret_cast<const f32x4*>(part + k * slab + m * Cout + c);
        v += b;
        if (res) v += *reinterpret_cast<const f32x4*>(res + m * res_ld + c);
        *reinterpret_cast<f32x4*>(out + m * ldo + c) = v;
#pragma unroll
        for (int e = 0; e < 4; ++e) { s1[e] += (double)v[e]; s2[e] += (double)v[e] * (double)v[e]; }
    }
    if (stats) {
        __shared__ double red[64][16][2];
#pragma unroll
        for (int e = 0; e < 4; ++e) { red[ph][q * 4 + e][0] = s1[e]; red[ph][q * 4 + e][1] = s2[e]; }
        __syncthreads();
        if (threadIdx.x < 32) {
            const int ch = threadIdx.x >> 1, w = threadIdx.x & 1;
            double t = 0.0;
            for (int k = 0; k < 64; ++k) t += red[k][ch][w];
            stats[((size_t)f * Cout + blockIdx.y * 16 + ch) * 2 + w] = t;
        }
    }
}

// Slices of the channel loop by shape alone: 1 when the grid fills the chip anyway (every launch of the headline window)
// or the map is larger than 16 x 16; else the largest count that keeps >= 4 chunks (an even number) per block and the
// grid within 288 blocks.  The engine sizes the scratch from this (conv_wino_r64_ksplit_floats) in its dry run too.
int conv_wino_r64_ksplit(int nfr, int Hl, int Cin, int Cout) {
    if (Hl > 16 || Hl < 8 || Cout % 64 || Cin % 32) return 1;
    const int items = (Hl == 8 ? (nfr + 3) / 4 : (Hl / 16) * (Hl / 16) * nfr) * (Cout / 64), nchunk = Cin / 16;
    if (items >= 160) return 1;
    int best = 1;
    for (int s2 = 2; s2 <= 16; ++s2)
        if (nchunk % s2 == 0 && (nchunk / s2) % 2 == 0 && nchunk / s2 >= 4 && items * s2 <= 288) best = s2;
    return best;
}

size_t conv_wino_r64_ksplit_floats(int nfr, int Hl, int Cin, int Cout, int nfr_sel) {
    const int S = conv_wino_r64_ksplit(nfr_sel ? nfr_sel : nfr, Hl, Cin, Cout);
    return S > 1 ? (size_t)S * nfr * Hl * Hl * Cout : 0;
}

int conv_wino_ups_stats_split(int Hs) { return 4 * conv_wino_stats_split(Hs); }

// Upsample + conv3x3 in its sub-pixel form: a.wwino is the image of pack_conv3_wino_ups (4 x Cout phase kernels), the kernel
// convolves the LOW-resolution map and writes the interleaved output.
static int launch_conv_wino_r64_ups(const IgemmArgs& a, hipStream_t s) {
    VD_REQUIRE(a.ups == 1 && a.res == nullptr && a.fbias == nullptr && a.Cout % 64 == 0, "sub-pixel Upsample conv: no residual, no per-frame bias");
    VD_REQUIRE(a.stats == nullptr || a.stats_split == conv_wino_ups_stats_split(a.Hs), "GroupNorm partial table: split (sub-pixel form)");
    VD_REQUIRE((size_t)a.Cin * a.Cout * 4 * 96 < ((size_t)1 << 31), "sub-pixel weight image beyond 2 GiB");
    WinoR64Geom g;
    const int Hl = a.Hs;
    const bool tf4 = Hl == 8;
    g.tiles_x = tf4 ? 1 : Hl / 16; g.tiles_y = g.tiles_x;
    g.nbx = g.tiles_x * g.tiles_y * (tf4 ? (a.nfr + 3) / 4 : a.nfr);
    g.ncb = 4 * a.Cout / 64;
    g.nitems = g.nbx * g.ncb;
    g.xcd_order = g.nbx % 8 == 0;
    g.ksplit = 1;
    g.phase_cb = a.Cout / 32;
    g.cgroup = g.ncb % 4 == 0 ? 4 : 2;                                // groups of 1 / 2 / 4 / 8 measured: 1 loses the patch reuse (1347 us at 256 couts), 2 .. 8 within noise
    IgemmArgs k = a;
    k.ups = 0; k.Cout = 4 * a.Cout;                                  // the kernel's view: a stride-1 conv of the source map with 4 x Cout outputs
    const dim3 grid(g.nitems, 1);
    const bool f16 = f16_math();
    if (tf4) return f16 ? r64_launch<&conv3x3_wino_r64_ups_kernel<true, true>>(grid, r64::lds_bytes<true>(), s, k, g)
                        : r64_launch<&conv3x3_wino_r64_ups_kernel<true, false>>(grid, r64::lds_bytes<true>(), s, k, g);
    return f16 ? r64_launch<&conv3x3_wino_r64_ups_kernel<false, true>>(grid, r64::lds_bytes<false>(), s, k, g)
               : r64_launch<&conv3x3_wino_r64_ups_kernel<false, false>>(grid, r64::lds_bytes<false>(), s, k, g);
}

int launch_conv_wino_r64(const IgemmArgs& a, hipStream_t s) {
    if (a.ups_phase) return launch_conv_wino_r64_ups(a, s);
    const int Hl = a.Hs << a.ups;
    VD_REQUIRE(a.stats == nullptr || a.stats_split == conv_wino_stats_split(Hl), "GroupNorm partial table: split");
    WinoR64Geom g;
    const bool tf4 = Hl == 8;                       // four frames of 4 x 4 tiles per item
    g.tiles_x = tf4 ? 1 : Hl / 16; g.tiles_y = g.tiles_x;
    g.nbx = g.tiles_x * g.tiles_y * (tf4 ? (a.nfr + 3) / 4 : a.nfr);
    g.ncb = a.Cout / 64;
    g.nitems = g.nbx * g.ncb;
    g.xcd_order = g.nbx % 8 == 0;
    // (the grouped cout walk of the sub-pixel form changes nothing here: 2 .. 8 cout blocks per patch, headline 27.55 ms with groups of 0 / 2 / 4;
    // re-measured in round 6 on the f16x3 kernel through VD_R64_CGROUP, LAB_NOTES R6)
    static const int env_cgroup = getenv("VD_R64_CGROUP") ? atoi(getenv("VD_R64_CGROUP")) : 0;
    if (env_cgroup > 0 && g.xcd_order && g.ncb % env_cgroup == 0) g.cgroup = env_cgroup;
    // split-K only with scratch from the caller (the engine's arena; the single-operator entry points run one slice)
    g.ksplit = a.ksplit_ws && a.ksplit_ws_floats >= conv_wino_r64_ksplit_floats(a.nfr, Hl, a.Cin, a.Cout, a.nfr_sel)
                   ? conv_wino_r64_ksplit(a.nfr_sel ? a.nfr_sel : a.nfr, Hl, a.Cin, a.Cout) : 1;
    IgemmArgs k = a;
    if (g.ksplit > 1) {
        k.out = a.ksplit_ws; k.ldo = a.Cout; k.bias = nullptr; k.fbias = nullptr; k.res = nullptr; k.stats = nullptr;
    }
    const dim3 grid(g.nitems, g.ksplit);
    const bool f16 = f16_math();
    int rc;
    if (tf4) rc = f16 ? r64_launch<&conv3x3_wino_r64_kernel<true, true>>(grid, r64::lds_bytes<true>(), s, k, g)
                      : r64_launch<&conv3x3_wino_r64_kernel<true, false>>(grid, r64::lds_bytes<true>(), s, k, g);
    else rc = f16 ? r64_launch<&conv3x3_wino_r64_kernel<false, true>>(grid, r64::lds_bytes<false>(), s, k, g)
                  : r64_launch<&conv3x3_wino_r64_kernel<false, false>>(grid, r64::lds_bytes<false>(), s, k, g);
    if (rc) return rc;
    if (g.ksplit > 1) {
        const int HW = Hl * Hl;
        hipLaunchKernelGGL(wino_r64_reduce_kernel, dim3(a.nfr, a.Cout / 16), dim3(256), 0, s, a.ksplit_ws, g.ksplit,
                           (size_t)a.nfr * HW * a.Cout, a.bias, a.fbias, a.fbias_ld, a.res, a.res_ld, a.out, a.ldo, HW, a.Cout, a.stats);
        VD_HIP(hipGetLastError());
    }
    return 0;
}

}  // namespace vd
