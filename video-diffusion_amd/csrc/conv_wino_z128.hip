// 3x3 stride-1 convolution by Winograd F(2x2,3x3) in the f16x3 arithmetic, 64 tiles x 128 couts per block: the COLUMN half of
// the output transform is accumulated by the matrix pipe (gfx950).
//
// conv_wino_r64.hip keeps one accumulator tile per Winograd position -- wave i owns row i: 4 positions x 2 M-tiles x 2 cout
// tiles = all 256 accumulator registers -- and forms Z = M A (columns (1,1,1,0) and (0,1,-1,-1) of A) from them in its
// epilogue.  Its loop is bound by everything BUT the matrix pipe (round 4, docs/LAB_NOTES.md R4.2: MFMA 30 % busy; the vector
// work of transform + split, the weight loads and the patch reads each cost what they issue) and its per-item prologue +
// epilogue by ~17 k cycles whatever the channel count.  Both are costs per 64-cout block of a tile group.
//
// Here the accumulators ARE Z: z0 += V0 U0 + V1 U1 + V2 U2 and z1 += V1 U1 - V2 U2 - V3 U3 per channel chunk -- six (position,
// column) MFMA groups instead of four, 1.5 x the matrix work -- which halves the accumulator tiles per cout and lets a block own
// 128 couts: every transformed and split A fragment feeds 12 (positions 0, 3) or 24 (positions 1, 2) MFMAs instead of 6, the
// patch is staged, read and transformed once per 128 couts, the prologue and the exchange of the output transform are paid once
// per 128 couts, and the epilogue no longer sums position tiles.  The minus signs cost nothing for V3 (its column combination is
// taken with swapped operands) and sixteen v_xor for V2 (the fragment is negated in place between its two uses).
//
// Everything else is conv_wino_r64.hip's: the LDS-DMA patch image and its swizzle, wave i = Winograd row i, the weight image
// (split_pack.hip), one barrier per two chunks, requests one per MFMA slot.  A position is 24 or 48 slots here, so the weight
// fragments of the NEXT position (12 KiB per wave: 4 cout tiles x 3 pieces) have a whole position to arrive in a two-slot
// register ring, and the vector work of the next position's two fragments (8 blocks of 12 instructions) is 2 - 4 per slot.
// Served: f16x3, maps >= 16 x 16, Cout % 128 == 0, no split-K, not the sub-pixel Upsample form (conv_wino_r64.hip keeps those).
#include <cstdlib>
#include <cstring>

#include "vd_common.h"

namespace vd {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct WinoZ128Geom { int tiles_x, tiles_y, nbx, ncb, nitems, xcd_order; };

namespace z128 {
constexpr int NB = 4;                                          // patch buffers
constexpr int P = 18, SPP = 10, PLB = SPP * 64, RSB = 2 * PLB, NX = 6, XBUF = NX * 4096, MOFF = 8 * RSB;
constexpr int LDS_BYTES = (NB + 1) * XBUF;                     // + the spare buffer for requests past the last chunk
}  // namespace z128

// Variants built, measured and taken out (history: commit aa1331c): v_fma_mixlo/hi_f16 for the a1 piece (no faster, r05b), a uniform branch around
// the residual requests (slower, r05u), register staging without the activation as a timing build (r05s: what the staging method alone is worth).
#ifdef VD_WINO_TIMING
__device__ unsigned long long g_z128_stamp[16];
#define Z128_STAMP(i)                                                                                 \
    do {                                                                                              \
        if (threadIdx.x == 0 && blockIdx.x == 7) {                                                    \
            __builtin_amdgcn_sched_barrier(0);                                                        \
            g_z128_stamp[i] = (i) >= 14 ? __builtin_amdgcn_s_memrealtime() : __builtin_readcyclecounter(); \
            __builtin_amdgcn_sched_barrier(0);                                                        \
        }                                                                                             \
    } while (0)
extern "C" int vd_debug_z128_stamps(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_z128_stamp), sizeof(g_z128_stamp));
}
#else
#define Z128_STAMP(i)
#endif

// ACT: the input is NOT an activated image: the kernel reads the raw tensor and applies the folded GroupNorm(+FiLM) affine and the SiLU
// itself -- x -> silu(x * A[frame][c] + B[frame][c]), zero outside the picture -- while it stages the patch (through registers instead of
// LDS-DMA).  What it replaces is a whole pass over the tensor (norm.hip: affine_act, read + write at ~4.6 TB/s: 117 us for 128 channels at
// 64 x 64 x 128 frames) per convolution; what it costs was measured beforehand with a timing-only build carrying the same vector work
// and LDS stores in the same slots (r05j): +6 % of the kernel (423 -> 451 us at 128 -> 128, 709 -> 748 at 256 -> 128).  It pays only where
// ONE or two cout blocks share a patch (every block that stages a patch activates it): Cout <= 256.  Per chunk and thread: 6 loads (one
// patch of 16 channels = 6 x 16 bytes per thread), 24 elements x (fma, select, mul, exp, add, rcp, mul) in the 24 slots of position 2 that
// carry nothing else, 6 ds_write_b128; the patch of chunk c + 2 is activated during chunk c from registers loaded at the end of chunk
// c - 1, published by the barrier that stands in front of its first reader anyway.
template <bool ACT>
__global__ __launch_bounds__(256, 1) void conv3x3_wino_z128_kernel(IgemmArgs a, WinoZ128Geom g) {
    using namespace z128;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* const lds = reinterpret_cast<char*>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wi = __builtin_amdgcn_readfirstlane(tid >> 6);         // Winograd row of this wave
    const int lr = lane & 31, lh = lane >> 5;
    const int Hl = a.Hs, Wl = a.Ws;
    const int ncoblk = a.Cout >> 5, nchunk = a.Cin >> 4;

    // block -> (tile group, first cout tile): dealt to the 8 XCDs round-robin, the cout blocks of one patch neighbours inside an XCD
    int bx, cob0;
    if (g.xcd_order) {
        const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
        cob0 = (loc % g.ncb) * 4;
        bx = (loc / g.ncb) * 8 + xcd;
    } else {
        bx = blockIdx.x % g.nbx;
        cob0 = (blockIdx.x / g.nbx) * 4;
    }
    const int bxx = bx % g.tiles_x; bx /= g.tiles_x;
    const int byy = bx % g.tiles_y; bx /= g.tiles_y;
    const int f0 = bx;                                               // the item's frame
    const int ox0 = bxx * 16, oy0 = byy * 16;

    // ---- patch staging (conv_wino_r64.hip): thread -> 16-byte LDS slots e*256 + tid; the slot at quad position lq of patch row py
    // holds the pixel's quad lq ^ ((py >> 1) & 3)
    unsigned xo[NX];
    unsigned ldo[ACT ? NX : 1];                                       // ACT: byte offset of the thread's slot e inside a patch buffer
    unsigned long long inm[ACT ? NX : 1];                             // ACT: lanes whose slot e lies inside the picture
#pragma unroll
    for (int e = 0; e < NX; ++e) {
        const int gs = e * 256 + tid, lq = gs & 3, ps = gs >> 2;
        const int py = ps / (2 * SPP), r = ps % (2 * SPP), pxh = r % SPP, px = 2 * pxh + r / SPP;
        const int ly = oy0 + py - 1, lx = ox0 + px - 1;
        const bool in = py < P && pxh < P / 2 && ly >= 0 && ly < Hl && lx >= 0 && lx < Wl;
        if constexpr (ACT) {
            // the thread keeps ONE channel quad (lq = tid & 3: its (A, B) are four registers per chunk) and writes it to the place the
            // DMA image gives that quad: position lq ^ ((py >> 1) & 3) of the pixel's four 16-byte slots.  xo = the PIXEL index: the
            // input may be a virtual concat of two tensors of different widths (unet.py:826-828), the byte offset is formed per source
            xo[e] = in ? (unsigned)((f0 * a.Hs + ly) * a.Ws + lx) : 0x80000000u;
            ldo[e] = (unsigned)(((gs & ~3) | (lq ^ ((py >> 1) & 3))) * 16);
            inm[e] = __ballot(in);
        } else {
            xo[e] = in ? (unsigned)((f0 * a.Hs + ly) * a.Ws + lx) * (unsigned)(a.Cin * 4) + (unsigned)((lq ^ ((py >> 1) & 3)) * 16) : 0x80000000u;
        }
    }
    const auto xsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.src0), 0, a.nfr * a.Hs * a.Ws * (ACT ? a.C0 : a.Cin) * 4, 0x00020000);
    const auto xnull = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.src0), 0, 0, 0x00020000);
    typedef __attribute__((address_space(3))) void* lds_ptr;
    auto x_dma_one = [&](int chunk, int e) {                          // (a request past the last chunk always issues: conv_wino_r64.hip)
#if defined(__HIP_DEVICE_COMPILE__)
        const bool live = chunk < nchunk;
        const int bufi = live ? (chunk & (NB - 1)) : NB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(live ? xsrc : xnull, (lds_ptr)(lds + bufi * XBUF + e * 4096 + wi * 1024), 16, xo[e], chunk * 64, 0, 0);
#endif
    };

    // ---- ACT: register staging.  (A, B) of the item's frame sit in LDS behind the patch buffers: [Cin] A, [Cin] B.
    float* const sab = reinterpret_cast<float*>(lds + LDS_BYTES);
    const int cq4 = (tid & 3) * 4;
    f32x4 stg[ACT ? NX : 1];                                          // the patch being staged: loaded raw, activated in place
    f32x4 Aq = {0.f, 0.f, 0.f, 0.f}, Bq = Aq;                         // (A, B) of the thread's channel quad for that patch
    float tq[4];                                                      // in flight between the parts of one slot e
    // The input may be a virtual concat of two tensors (unet.py:826-828): chunks [0, nch0) come from src0 (C0 channels per pixel), the rest
    // from src1.  Byte offsets of the thread's six slots for either source; the descriptor of a patch is formed from SCALAR selects (a
    // select between two descriptor VALUES goes through vector registers and hipcc wraps every load in a waterfall loop)
    const int nch0 = a.C0 >> 4;
    unsigned xo1[ACT ? NX : 1];
    if constexpr (ACT) {
#pragma unroll
        for (int e = 0; e < NX; ++e) {
            const unsigned pix = xo[e];
            xo[e] = pix == 0x80000000u ? pix : pix * (unsigned)(a.C0 * 4) + (unsigned)((tid & 3) * 16);
            xo1[e] = pix == 0x80000000u ? pix : pix * (unsigned)((a.Cin - a.C0) * 4) + (unsigned)((tid & 3) * 16);
        }
    }
    auto p_fetch = [&](int patch, int e) -> f32x4 {                  // (a request past the last chunk always issues, through an empty descriptor)
        if constexpr (ACT) {
            const bool first = patch < nch0;
            const float* base = first ? a.src0 : a.src1;
            const int bytes = patch >= nchunk ? 0 : a.nfr * a.Hs * a.Ws * (first ? a.C0 : a.Cin - a.C0) * 4;
            const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base ? base : a.src0), 0, bytes, 0x00020000);
            return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, first ? xo[e] : xo1[e], (first ? patch : patch - nch0) * 64, 0));
        } else {
            return f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto p_load = [&](int patch, int e) { if constexpr (ACT) stg[e] = p_fetch(patch, e); };
    auto p_coef = [&](int patch) {
        if constexpr (ACT) {
            const int pc = min(patch, nchunk - 1) * 16 + cq4;
            Aq = *reinterpret_cast<const f32x4*>(sab + pc);
            Bq = *reinterpret_cast<const f32x4*>(sab + a.Cin + pc);
        }
    };
    // slot e in EIGHT parts of 2 - 4 vector instructions, consecutive instructions on different elements; the arithmetic of norm.hip's pass to
    // the bit: fma(x, A, B), silu(v) = v * rcp(1 + exp2(-log2(e) v)).  v_exp_f32 / v_rcp_f32 run at a quarter of the vector rate (16 cycles
    // per wave instruction): the first version -- four parts of 7 with two of them each -- put 52 cycles of vector work into a 32-cycle MFMA
    // slot (cycle stamps: 7.2 k cycles per chunk against 5.7 k of the plain kernel); eight parts over the odd slots of positions 1 AND 2
    // keep a part at <= 2 transcendental + 2 plain instructions.
    auto p_act_on = [&](f32x4 (&stg)[ACT ? NX : 1], int e, int part) {
        if constexpr (ACT) {
            if (part == 0)
                asm("v_fma_f32 %0, %0, %4, %8\n\tv_fma_f32 %1, %1, %5, %9\n\tv_fma_f32 %2, %2, %6, %10\n\tv_fma_f32 %3, %3, %7, %11"
                    : "+v"(stg[e][0]), "+v"(stg[e][1]), "+v"(stg[e][2]), "+v"(stg[e][3])
                    : "v"(Aq[0]), "v"(Aq[1]), "v"(Aq[2]), "v"(Aq[3]), "v"(Bq[0]), "v"(Bq[1]), "v"(Bq[2]), "v"(Bq[3]));
            else if (part == 1)
                asm("v_cndmask_b32_e64 %0, 0, %0, %4\n\tv_cndmask_b32_e64 %1, 0, %1, %4\n\tv_cndmask_b32_e64 %2, 0, %2, %4\n\tv_cndmask_b32_e64 %3, 0, %3, %4"
                    : "+v"(stg[e][0]), "+v"(stg[e][1]), "+v"(stg[e][2]), "+v"(stg[e][3]) : "s"(inm[e]));
            else if (part == 2)
                asm("v_mul_f32 %0, 0xbfb8aa3b, %4\n\tv_mul_f32 %1, 0xbfb8aa3b, %5\n\tv_mul_f32 %2, 0xbfb8aa3b, %6\n\tv_mul_f32 %3, 0xbfb8aa3b, %7"
                    : "=&v"(tq[0]), "=&v"(tq[1]), "=&v"(tq[2]), "=&v"(tq[3]) : "v"(stg[e][0]), "v"(stg[e][1]), "v"(stg[e][2]), "v"(stg[e][3]));
            else if (part == 3)
                asm("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1" : "+v"(tq[0]), "+v"(tq[1]));
            else if (part == 4)
                asm("v_exp_f32 %2, %2\n\tv_exp_f32 %3, %3\n\tv_add_f32 %0, 1.0, %0\n\tv_add_f32 %1, 1.0, %1" : "+v"(tq[0]), "+v"(tq[1]), "+v"(tq[2]), "+v"(tq[3]));
            else if (part == 5)
                asm("v_rcp_f32 %0, %0\n\tv_rcp_f32 %1, %1\n\tv_add_f32 %2, 1.0, %2\n\tv_add_f32 %3, 1.0, %3" : "+v"(tq[0]), "+v"(tq[1]), "+v"(tq[2]), "+v"(tq[3]));
            else if (part == 6)
                asm("v_rcp_f32 %4, %4\n\tv_rcp_f32 %5, %5\n\tv_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %3"
                    : "+v"(stg[e][0]), "+v"(stg[e][1]), "+v"(tq[0]), "+v"(tq[1]), "+v"(tq[2]), "+v"(tq[3]));
            else
                asm("v_mul_f32 %0, %0, %2\n\tv_mul_f32 %1, %1, %3" : "+v"(stg[e][2]), "+v"(stg[e][3]) : "v"(tq[2]), "v"(tq[3]));
        }
    };
    auto p_act = [&](int e, int part) { p_act_on(stg, e, part); };
    auto p_store = [&](int patch, int e) {
        if constexpr (ACT) *reinterpret_cast<f32x4*>(lds + (patch < nchunk ? (patch & (NB - 1)) : NB) * XBUF + ldo[e]) = stg[e];
    };

    // ---- the lane's patch addresses: lane (tile lr of the M-tile, k-half lh): tile column lr & 7, tile row 4m + (lr >> 3); rows of
    // B^T as d[X] + s*d[S]: 0: d0 - d2, 1: d1 + d2, 2: d2 - d1, 3: d3 - d1 (row 3 of U is negated on the host)
    const int rowX = wi, rowS = wi < 2 ? 2 : 1;
    const float tsg = wi == 1 ? 1.f : -1.f;
    const int ttx = lr & 7, ttyl = lr >> 3;
    int adr[2][2];                                                    // [X | S][quad h of the lane's eight channels]
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        adr[0][h] = (2 * ttyl + rowX) * RSB + ttx * 64 + (((2 * lh + h) ^ ((ttyl + (rowX >> 1)) & 3)) * 16);
        adr[1][h] = (2 * ttyl + rowS) * RSB + ttx * 64 + (((2 * lh + h) ^ ((ttyl + (rowS >> 1)) & 3)) * 16);
    }
    // V of position j = t[ca] -/+ t[cb], t[c] = X[c] + tsg * S[c]; position 3 is taken NEGATED (t3 - t1): z1 subtracts it
    //   j:      0        1        2        3
    //   ca, cb: 0, 2     1, 2     2, 1     3, 1          op: - + - -
    f32x4 raw[2][2][2][2];                                            // [m][column a | b][row X | S][quad h]: one fragment pair in flight
    u32x4 af[2][2][2];                                                // A fragments [buffer][m][piece a0 | a1]
    auto frag_read = [&](int chunk, int jn, int idx) {               // read idx of 16: m = idx >> 3, quad (idx >> 2) & 1, column (idx >> 1) & 1, row idx & 1
        const int m = idx >> 3, h = (idx >> 2) & 1, cs = (idx >> 1) & 1, rs = idx & 1;   // (four reads feed two frag_pair blocks)
        const int c = cs == 0 ? (jn == 0 ? 0 : jn == 1 ? 1 : jn == 2 ? 2 : 3) : (jn == 0 ? 2 : jn == 1 ? 2 : 1);
        const char* rb = lds + (chunk & (NB - 1)) * XBUF + m * MOFF + (c & 1) * PLB + (c >> 1) * 64;
        raw[m][cs][rs][h] = *reinterpret_cast<const f32x4*>(rb + adr[rs][h]);
    };
    // channel pair pr (quad pr >> 1, elements 2 (pr & 1), +1) of fragment (m, jn) -> af[nxt][m], in three parts that go to three
    // different MFMA slots (4 - 6 vector instructions hide behind an MFMA, a seventh and later do not: tools/probes/mfma_f16_coissue.hip):
    //   0: row combination of both columns (4 v_fma_f32)   1: column combination, a0 = f16 (3)   2: a1 = f16((x - a0) * 2^12) (5)
    // Each part is ONE asm statement: hipcc pads every asm output that the next instruction reads with an s_nop, and an s_nop is
    // an issue slot like any other (conv_wino_r64.hip).
    float pa0, pa1, pb0, pb1, pv0, pv1;                              // in flight between the parts of one block
    auto frag_part = [&](int nxt, int jn, int b, int part) {         // block b of 8: m = b >> 2, pair b & 3
        const int m = b >> 2, pr = b & 3, h = pr >> 1, e0 = 2 * (pr & 1);
        if (part == 0) {
            asm("v_fma_f32 %0, %4, %6, %5\n\tv_fma_f32 %1, %4, %8, %7\n\tv_fma_f32 %2, %4, %10, %9\n\tv_fma_f32 %3, %4, %12, %11"
                : "=&v"(pa0), "=&v"(pa1), "=&v"(pb0), "=&v"(pb1)
                : "v"(tsg), "v"(raw[m][0][0][h][e0]), "v"(raw[m][0][1][h][e0]), "v"(raw[m][0][0][h][e0 + 1]), "v"(raw[m][0][1][h][e0 + 1]),
                  "v"(raw[m][1][0][h][e0]), "v"(raw[m][1][1][h][e0]), "v"(raw[m][1][0][h][e0 + 1]), "v"(raw[m][1][1][h][e0 + 1]));
        } else if (part == 1) {
            if (jn == 1)
                asm("v_add_f32 %1, %3, %5\n\tv_add_f32 %2, %4, %6\n\tv_cvt_pk_f16_f32 %0, %1, %2"
                    : "=&v"(af[nxt][m][0][pr]), "=&v"(pv0), "=&v"(pv1) : "v"(pa0), "v"(pa1), "v"(pb0), "v"(pb1));
            else
                asm("v_sub_f32 %1, %3, %5\n\tv_sub_f32 %2, %4, %6\n\tv_cvt_pk_f16_f32 %0, %1, %2"
                    : "=&v"(af[nxt][m][0][pr]), "=&v"(pv0), "=&v"(pv1) : "v"(pa0), "v"(pa1), "v"(pb0), "v"(pb1));
        } else if (part == 3) {                                      // parts 0 + 1 in one statement
#define VD_Z128_P01(OP)                                                                                                                   \
    asm("v_fma_f32 %3, %7, %9, %8\n\tv_fma_f32 %4, %7, %11, %10\n\tv_fma_f32 %5, %7, %13, %12\n\tv_fma_f32 %6, %7, %15, %14\n\t"          \
        OP " %1, %3, %5\n\t" OP " %2, %4, %6\n\tv_cvt_pk_f16_f32 %0, %1, %2"                                                             \
        : "=&v"(af[nxt][m][0][pr]), "=&v"(pv0), "=&v"(pv1), "=&v"(pa0), "=&v"(pa1), "=&v"(pb0), "=&v"(pb1)                                  \
        : "v"(tsg), "v"(raw[m][0][0][h][e0]), "v"(raw[m][0][1][h][e0]), "v"(raw[m][0][0][h][e0 + 1]), "v"(raw[m][0][1][h][e0 + 1]),       \
          "v"(raw[m][1][0][h][e0]), "v"(raw[m][1][1][h][e0]), "v"(raw[m][1][0][h][e0 + 1]), "v"(raw[m][1][1][h][e0 + 1]))
            if (jn == 1) VD_Z128_P01("v_add_f32");
            else VD_Z128_P01("v_sub_f32");
#undef VD_Z128_P01
        } else {
            float r0, r1;
            asm("v_fma_mix_f32 %1, %3, -1.0, %4 op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %2, %3, -1.0, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
                "v_ldexp_f32 %1, %1, 12\n\tv_ldexp_f32 %2, %2, 12\n\tv_cvt_pk_f16_f32 %0, %1, %2"
                : "=&v"(af[nxt][m][1][pr]), "=&v"(r0), "=&v"(r1) : "v"(af[nxt][m][0][pr]), "v"(pv0), "v"(pv1));
        }
    };
    // The 24 parts of a position run in its steps 3 .. 23 (reads: two per step in steps 0 .. 7, the four of blocks 2g, 2g + 1 in
    // steps 2g, 2g + 1): blocks 0 .. 2 in two steps each (parts 0 + 1 together), the others in three.

    // ---- B fragments: U[chunk][xi = 4*wi + j][cob][piece][lane][8 x 16 bit] = 1 KiB per (chunk, xi, cob, piece); the twelve of a
    // position (4 cout tiles x 3 pieces) are contiguous: offsets 0 .. 3072 in the instruction's immediate, three scalar bases
    const auto usrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wwino), 0, 16 * a.Cout * a.Cin * 6, 0x00020000);
    const int ustride = 16 * ncoblk * 3072, bstep = ncoblk * 3072;
    const int bsb = (wi * 4 * ncoblk + cob0) * 3072;
    const unsigned blane = lane * 16u;
    u32x4 bfr[2][4][3];                                               // [ring slot = position & 1][cout tile][piece]
    // Pieces 0 and 1 only: piece 2 = 2^-12 b0 (split_pack.hip) is formed here, four v_pk_mul_f16 per fragment (exact: a power of two,
    // fp16 subnormals honoured like the host's conversion) -- a third less weight traffic from the L2 for 16 vector instructions per
    // position, in slots that carry none
    auto b_load_one = [&](int chunk, int j, int k) {                 // k-th load of a position (8), in the order the MFMAs want them: piece k >> 2 of cout tile k & 3
        const int idx = (k & 3) * 3 + (k >> 2);
        const int so = chunk * ustride + bsb + j * bstep + (idx >> 2) * 4096;
        bfr[j & 1][idx / 3][idx % 3] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(usrc, blane + (idx & 3) * 1024u, so, 0));
    };

    const unsigned two_m12 = 0x0c000c00u;                             // (2^-12, 2^-12) in fp16
    auto b_third = [&](int slot, int n) {
        asm("v_pk_mul_f16 %0, %4, %8\n\tv_pk_mul_f16 %1, %5, %8\n\tv_pk_mul_f16 %2, %6, %8\n\tv_pk_mul_f16 %3, %7, %8"
            : "=&v"(bfr[slot][n][2][0]), "=&v"(bfr[slot][n][2][1]), "=&v"(bfr[slot][n][2][2]), "=&v"(bfr[slot][n][2][3])
            : "v"(bfr[slot][n][0][0]), "v"(bfr[slot][n][0][1]), "v"(bfr[slot][n][0][2]), "v"(bfr[slot][n][0][3]), "s"(two_m12));
    };

    f32x16 acc[2][2][4];                                              // [z][m][n]
    Z128_STAMP(0); Z128_STAMP(14);
    // ---- prologue: three patches and the weights of (chunk 0, position 0) requested; the 256 accumulator writes go under the wait
    auto zero_acc = [&]() {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int z = 0; z < 2; ++z)
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n)
#pragma unroll
                    for (int r = 0; r < 16; ++r) asm volatile("v_accvgpr_write_b32 %0, 0" : "=a"(acc[z][m][n][r]));
        __builtin_amdgcn_sched_barrier(0);
    };
    if constexpr (ACT) {
        // Patches 0 and 1 requested together with the weights of (chunk 0, position 0) -- one memory round trip at the head of the item --,
        // the accumulator writes and the frame's (A, B) table (LDS) under the wait; both patches activated and stored whole; patch 2 is left
        // in flight in the staging registers: the state the loop expects (chunk c activates patch c + 2 in its position 2)
        f32x4 stg1[NX];
#pragma unroll
        for (int e = 0; e < NX; ++e) stg[e] = p_fetch(0, e);
#pragma unroll
        for (int e = 0; e < NX; ++e) stg1[e] = p_fetch(1, e);
#pragma unroll
        for (int idx = 0; idx < 8; ++idx) b_load_one(0, 0, idx);
        for (int c = tid; c < a.Cin; c += 256) {
            sab[c] = a.affA[(size_t)f0 * a.Cin + c];
            sab[a.Cin + c] = a.affB[(size_t)f0 * a.Cin + c];
        }
        zero_acc();
        __syncthreads();
        p_coef(0);
#pragma unroll
        for (int e = 0; e < NX; ++e) {
#pragma unroll
            for (int part = 0; part < 8; ++part) p_act(e, part);
            p_store(0, e);
        }
        p_coef(1);
#pragma unroll
        for (int e = 0; e < NX; ++e) {
#pragma unroll
            for (int part = 0; part < 8; ++part) p_act_on(stg1, e, part);
            *reinterpret_cast<f32x4*>(lds + (1 < nchunk ? 1 : NB) * XBUF + ldo[e]) = stg1[e];
        }
#pragma unroll
        for (int e = 0; e < NX; ++e) p_load(2, e);
    } else {
#pragma unroll
        for (int c = 0; c < 3; ++c)
#pragma unroll
            for (int e = 0; e < NX; ++e) x_dma_one(c, e);
#pragma unroll
        for (int idx = 0; idx < 8; ++idx) b_load_one(0, 0, idx);
        zero_acc();
    }
    // patch 0 has landed in every wave: the 2 * NX + 8 youngest requests are patches 1, 2 (first read behind the loop's first barrier,
    // which waits for them) and the weights
    if constexpr (ACT) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // patches 0, 1 are written; the loads in flight land in registers
    else asm volatile("s_waitcnt vmcnt(20)\n\ts_barrier" ::: "memory");
    static_assert(NX == 6, "the wait count above");
#pragma unroll
    for (int idx = 0; idx < 16; ++idx) frag_read(0, 0, idx);
#pragma unroll
    for (int t = 0; t < 24; ++t) frag_part(0, 0, t / 3, t % 3);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 1");                                         // (a VALU write needs two wait states before an MFMA reads it as A / B)
    __builtin_amdgcn_sched_barrier(0);

    Z128_STAMP(1);
    // ---- main loop.  Position j of a chunk: for z in (z0 if j < 3) (z1 if j > 0): for m: for n: three piece products (A0,B0)
    // (A0,B1) (A1,B2) into acc[z][m][n] -- 24 or 48 slots.  Beside them everything the NEXT position (j + 1, or 0 of the next
    // chunk) needs, SPREAD over the position in 24 steps (every slot, or every second one): a request to the texture path or to
    // LDS is served at 16 / 8 cycles per wave instruction, the four waves of the block issue theirs at the same time, and a burst
    // of them blocks the wave's issue behind the unit's queue -- the first version, with the twelve weight loads and the sixteen
    // patch reads in the first 12 / 8 slots and the fragment blocks whole, ran 6.2 k cycles per chunk against 4.6 k of MFMA.  Step v:
    // even v < 16: weight load v / 2 (ring slot of position j - 1, free); v < 8: patch reads 2v, 2v + 1; v >= 3: the parts of the
    // fragment blocks (part_step).  Odd chunks: the patch requests of chunks c + 2 (position 1) and c + 3 (position 2) in the
    // steps' gaps.  Patch c is read from position 3 of chunk c - 1 through position 2 of chunk c; the block's only barrier stands
    // in front of position 3 of the EVEN chunks: patches c + 1, c + 2 have landed in every wave (requested a chunk ago: only weight
    // loads are among the 8 youngest requests), the buffers of c - 1, c are free.
    constexpr int PA3[3] = {0, 0, 1}, PB3[3] = {0, 1, 2};
    for (int chunk0 = 0; chunk0 < nchunk; chunk0 += 2) {
#pragma unroll
      for (int cpar = 0; cpar < 2; ++cpar) {
        const int chunk = chunk0 + cpar;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int cur = j & 1, nxt = cur ^ 1, jn = (j + 1) & 3, cn = j == 3 ? chunk + 1 : chunk;
            const int NS = (j == 0 || j == 3) ? 24 : 48, sp = NS / 24;
            if (cpar == 0 && j == 3) {
                if constexpr (ACT) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // (the patches are ds_writes here)
                else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            }
            int s = 0;
#pragma unroll
            for (int zi = 0; zi < 2; ++zi) {
                const int z = j == 0 ? 0 : j == 3 ? 1 : zi;
                if ((j == 0 || j == 3) && zi == 1) continue;
                if (j == 2 && zi == 1) {
                    // -V2: the fragment negated in place between its two uses (sixteen sign flips of fp16 pairs)
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int p = 0; p < 2; ++p)
                            asm volatile("v_xor_b32 %0, 0x80008000, %0\n\tv_xor_b32 %1, 0x80008000, %1\n\tv_xor_b32 %2, 0x80008000, %2\n\tv_xor_b32 %3, 0x80008000, %3"
                                         : "+v"(af[cur][m][p][0]), "+v"(af[cur][m][p][1]), "+v"(af[cur][m][p][2]), "+v"(af[cur][m][p][3]));
                    asm volatile("s_nop 1");
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int q = 0; q < 3; ++q)                      // (an accumulator tile is touched every fourth MFMA: a dependent one issued
#pragma unroll
                        for (int n = 0; n < 4; ++n) {                //  back to back waits for the write-back of its predecessor, 43 cycles instead of 32)
                            acc[z][m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[cur][m][PA3[q]]),
                                                                                 __builtin_bit_cast(f16x8, bfr[cur][n][PB3[q]]), acc[z][m][n], 0, 0, 0);
                            if (s < 4) b_third(cur, s);                 // first read by slot 8 + s
                            if (s % sp == 0) {
                                const int vs = s / sp;
                                if (!(vs & 1) && vs < 16) b_load_one(cn, jn, vs >> 1);
                                if (vs < 8) { frag_read(cn, jn, 2 * vs); frag_read(cn, jn, 2 * vs + 1); }
                                if (vs >= 9) frag_part(nxt, jn, 3 + (vs - 9) / 3, (vs - 9) % 3);
                                else if (vs >= 3 && ((vs - 3) & 1)) frag_part(nxt, jn, (vs - 3) >> 1, 2);
                                else if (vs >= 3) frag_part(nxt, jn, (vs - 3) >> 1, 3);
                            } else if (!ACT && cpar == 1 && (j == 1 || j == 2) && (s & 7) == 1) x_dma_one(chunk + 1 + j, s >> 3);
                            if constexpr (ACT) {
                                // end of position 0: (A, B) of patch chunk + 2; positions 1 and 2, odd slots: its 24 elements, one part of a slot e each, the
                                // store behind the last part; position 3: the six loads of patch chunk + 3 into the registers just freed
                                if (j == 0 && s == 23) p_coef(chunk + 2);
                                if ((j == 1 || j == 2) && (s & 1)) {
                                    const int u = (j - 1) * 24 + (s >> 1);             // 48 parts: slot e = u / 8, part u % 8
                                    p_act(u >> 3, u & 7);
                                    if ((u & 7) == 7) p_store(chunk + 2, u >> 3);
                                }
                                if (j == 3 && s >= 17 && s < 23) p_load(chunk + 3, s - 17);
                            }

                            ++s;
                            __builtin_amdgcn_sched_barrier(0);
                        }
            }
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

    Z128_STAMP(2);
    // ---- output transform, one cout tile at a time: the accumulators are Z[q] = sum_j M[wi][j] A[j][q] already; the sum over the
    // rows crosses the waves through LDS; wave (p, q) = (wi >> 1, wi & 1) then owns output pixel (p, q) of every tile.
    // Z image: [plane 2*i + q 8][m 2][c4 4][lane 64][4 floats] = 64 KB over the patch buffers.
    const int p = wi >> 1, q = wi & 1;
    const int obytes = a.nfr * Hl * Wl * a.ldo * 4;
    const auto osrc = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, obytes, 0x00020000);
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.res ? a.res : a.out), 0, a.res ? obytes : 0, 0x00020000);
    const float sgn = p ? -1.f : 1.f;
    float* Zs = smem;
    unsigned oo[2][16];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int tt = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int tx = tt & 7, ty = tt >> 3;
            oo[m][r] = (unsigned)(((f0 * Hl + oy0 + 2 * ty + p) * Wl + ox0 + 2 * tx + q) * a.ldo + cob0 * 32 + lr) * 4u;
        }
    const float* trailer = a.wwino + (size_t)24 * a.Cout * a.Cin;     // [Cout] s, [Cout] 1 / s (split_pack.hip)
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int co = (cob0 + n) * 32 + lr;
        const int nso = n * 128;                                       // byte offset of cout tile n
        const float winv = trailer[a.Cout + co];
        const float bvf = (a.bias ? a.bias[co] : 0.f) + (a.fbias ? a.fbias[(size_t)f0 * a.fbias_ld + co] : 0.f);
        f32x16 rv[2];
        if (n) __syncthreads();                                      // the previous Z is no longer read
#pragma unroll
        for (int m = 0; m < 2; ++m) {
#pragma unroll
            for (int r = 0; r < 16; ++r) rv[m][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, oo[m][r], nso, 0));
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) {
                *reinterpret_cast<f32x4*>(Zs + ((((wi * 2 + 0) * 2 + m) * 4 + c4) * 64 + lane) * 4) =
                    f32x4{acc[0][m][n][4 * c4], acc[0][m][n][4 * c4 + 1], acc[0][m][n][4 * c4 + 2], acc[0][m][n][4 * c4 + 3]};
                *reinterpret_cast<f32x4*>(Zs + ((((wi * 2 + 1) * 2 + m) * 4 + c4) * 64 + lane) * 4) =
                    f32x4{acc[1][m][n][4 * c4], acc[1][m][n][4 * c4 + 1], acc[1][m][n][4 * c4 + 2], acc[1][m][n][4 * c4 + 3]};
            }
        }
        Z128_STAMP(3 + 2 * n);
        __syncthreads();
        const float* zw = Zs + wi * 2048 + lane * 4;                 // Z[p + k][q] is plane wi + 2k
        float gsum[2] = {0.f, 0.f};
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
            y = y * winv + rv[m];
#pragma unroll
            for (int r = 0; r < 16; ++r) y[r] += bvf;
#pragma unroll
            for (int r = 0; r < 16; ++r) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, (float)y[r]), osrc, oo[m][r], nso, 0);
            if (a.stats) {
#pragma unroll
                for (int r = 0; r < 16; ++r) { gsum[0] += y[r]; gsum[1] = __builtin_fmaf(y[r], y[r], gsum[1]); }
            }
        }
        if (a.stats) {                                               // GroupNorm partial sums of the output (conv_wino_r64.hip)
            __syncthreads();
            double* red = reinterpret_cast<double*>(smem);           // [wave 4][lh 2][lr 32][2]
            double* d = red + (((wi * 2 + lh) * 32 + lr) * 2);
            d[0] = (double)gsum[0]; d[1] = (double)gsum[1];
            __syncthreads();
            if (tid < 32) {
                double s = 0.0, ss = 0.0;
#pragma unroll
                for (int k = 0; k < 8; ++k) { s += red[(k * 32 + tid) * 2]; ss += red[(k * 32 + tid) * 2 + 1]; }
                double* o = a.stats + (((size_t)f0 * a.stats_split + byy * g.tiles_x + bxx) * a.Cout + (cob0 + n) * 32 + tid) * 2;
                o[0] = s; o[1] = ss;
            }
        }
        Z128_STAMP(4 + 2 * n);
    }
    Z128_STAMP(15);
}

// Which of the two kernels a shape gets.  An item here is worth two of conv_wino_r64.hip's (loop 5.5 k cycles per chunk and 128
// couts against 2 x 2.6 - 3.0 k, per-item overhead 21 - 25 k cycles against 2 x 15 - 17 k), so it can win where its grid fills
// the chip's last round as well as the other one's does -- and loses where halving the item count leaves CUs idle (384 couts at
// 16^2, 128 frames: 384 items = 1.5 rounds of 256 CUs against 3.0; 200 | 161 us): decided by the fill of the last round.
static double z128_fill(int items, int cus) { return (double)items / ((double)cus * ((items + cus - 1) / cus)); }

static bool z128_shape(int nfr, int H, int Cin, int Cout, int max_cin);
bool conv_wino_z128_shape(int nfr, int H, int Cin, int Cout) {
#ifndef VD_Z128_MAX_CIN
#define VD_Z128_MAX_CIN 320
#endif
    return z128_shape(nfr, H, Cin, Cout, VD_Z128_MAX_CIN);
}

static bool z128_shape(int nfr, int H, int Cin, int Cout, int max_cin) {
#ifdef VD_Z128_OFF                                   // kernel-experiment builds (tools/build_variant.sh): every shape on conv_wino_r64.hip
    return false;
#endif
    if (!f16_math() || H < 16 || (H & (H - 1)) || Cout % 128 || Cin % 32 || conv_wino_r64_ksplit(nfr, H, Cin, Cout) != 1) return false;
    // 1.5 x the MFMAs for half the per-item overhead and half the vector work: pays while the channel loop is short.  Same box,
    // us per launch, this kernel | conv_wino_r64.hip (r04q, after the latter stopped loading the third weight piece):
    // 128 -> 128 @ 64^2 411 - 428 | 432 - 451, 256 -> 256 @ 32^2 344 - 356 | 361 - 370, 640 -> 256 @ 32^2 805 | 758 - 774
    if (Cin > max_cin) return false;
    // CU count of the CURRENT device (cached per device id: a process that drives several devices must not inherit the first one's)
    static int cu_of[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (cu_of[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cu_of[dev] = n;
    }
    const int cus = cu_of[dev];
    const int items = (H / 16) * (H / 16) * nfr * (Cout / 128);
    return z128_fill(items, cus) >= 0.95 * z128_fill(2 * items, cus);
}

bool conv_wino_z128_supported(const IgemmArgs& a) {
    return conv_wino_r64_supported(a) && a.ups == 0 && !a.ups_phase && conv_wino_z128_shape(a.nfr_sel ? a.nfr_sel : a.nfr, a.Hs, a.Cin, a.Cout);
}

// The same kernel with the GroupNorm(+FiLM) affine + SiLU of its input applied while it stages the patch (ACT): shapes the plain kernel
// takes, with at most two cout blocks per patch (each of them activates it).  The shape half is what the engine asks before it decides
// not to materialise the activation image (engine.hip: res_block).
bool conv_wino_z128_act_shape(int nfr, int H, int Cin, int Cout) {
#ifdef VD_Z128_NO_ACT
    return false;
#endif
    static const bool off = getenv("VD_NO_CONV_ACT") != nullptr;       // A/B switch: the activation pass + the plain kernel
    static const int max_cout = getenv("VD_CONV_ACT_MAX_COUT") ? atoi(getenv("VD_CONV_ACT_MAX_COUT")) : 256;
    // (longer channel loops than the plain kernel takes -- the decoder's 384 .. 640 -> 128 | 256 convs, whose image the skip convolution writes --
    // measured in the step, r05q: 320 | 384 | 640 -> 20.155 | 20.148 | 20.22 ms: what the image costs is what conv_wino_r64.hip's lead there is worth)
    static const int max_cin = getenv("VD_CONV_ACT_MAX_CIN") ? atoi(getenv("VD_CONV_ACT_MAX_CIN")) : VD_Z128_MAX_CIN;
    return !off && Cout <= max_cout && z128_shape(nfr, H, Cin, Cout, max_cin);
}

bool conv_wino_z128_act_supported(const IgemmArgs& a) {
    if (!(a.affA && a.affB && a.act == 1 && (a.src1 != nullptr || a.C0 == a.Cin) && a.C0 % 16 == 0 && a.C0 > 0)) return false;
    IgemmArgs b = a; b.affA = b.affB = nullptr; b.act = 0; b.src1 = nullptr; b.C0 = a.Cin;     // conv_wino_r64.hip's conditions on everything else
    return conv_wino_r64_supported(b) && a.ups == 0 && !a.ups_phase && conv_wino_z128_act_shape(a.nfr_sel ? a.nfr_sel : a.nfr, a.Hs, a.Cin, a.Cout);
}

int launch_conv_wino_z128(const IgemmArgs& a, hipStream_t s) {
    const int Hl = a.Hs;
    VD_REQUIRE(a.stats == nullptr || a.stats_split == conv_wino_stats_split(Hl), "GroupNorm partial table: split");
    WinoZ128Geom g;
    g.tiles_x = Hl / 16; g.tiles_y = g.tiles_x;
    g.nbx = g.tiles_x * g.tiles_y * a.nfr;
    g.ncb = a.Cout / 128;
    g.nitems = g.nbx * g.ncb;
    g.xcd_order = g.nbx % 8 == 0;
    VD_RAISE_LDS((&conv3x3_wino_z128_kernel<false>), (size_t)160 * 1024);
    VD_RAISE_LDS((&conv3x3_wino_z128_kernel<true>), (size_t)160 * 1024);
    if (a.affA) {
        VD_REQUIRE(conv_wino_z128_act_supported(a), "conv_wino_z128 with the activation in its patch staging: shape not covered");
        hipLaunchKernelGGL(conv3x3_wino_z128_kernel<true>, dim3(g.nitems), dim3(256), z128::LDS_BYTES + 2 * a.Cin * sizeof(float), s, a, g);
    } else {
        hipLaunchKernelGGL(conv3x3_wino_z128_kernel<false>, dim3(g.nitems), dim3(256), z128::LDS_BYTES, s, a, g);
    }
    VD_HIP(hipGetLastError());
    return 0;
}

}  // namespace vd
