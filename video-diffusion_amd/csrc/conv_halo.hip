// 3x3 stride-1 convolution, fp32 MFMA, gfx950: LDS-staged input halo + fragment-major weights.
//
// The generic kernel (igemm.hip) gathers and transforms the A operand once per (tap, channel chunk)
// and stages a weight tile through LDS behind a barrier every K-step.  This kernel removes both:
//
//  * A operand: a block owns BM output pixels shaped TF frames x TH x TW.  Per 32-channel chunk the
//    (TH+2)x(TW+2) input patch of each frame is staged ONCE -- GroupNorm/FiLM affine + SiLU applied,
//    zero-padded at the image border, read through the optional nearest-x2 upsample and the virtual
//    concat -- and the nine taps read their fragments from that tile at shifted pixel offsets.
//    The tile is double-buffered: the next chunk's patch is loaded at the start of a chunk and
//    transformed/written one element per tap in the shadow of the MFMAs.
//  * B operand: weights are pre-packed in MFMA-fragment order
//        [tap][chunk][cout/32][kgroup 4][lane 64][4 floats]
//    (lane = 32*h + r holds w[co = 32*blk + r][ci = 32*chunk + 8*kg + 4*h + e]), so a wave fetches
//    the fragment of a k-group with ONE fully coalesced 1 KiB load straight from L2 into the
//    registers the MFMAs read -- no LDS round trip, and no barrier inside a chunk.
//    Loads run two k-groups (32 MFMAs, ~2k cycles) ahead in a 3-deep register ring.
//  => one barrier per 576 MFMAs per wave; the waves of the two co-resident blocks free-run.
#include "vd_common.h"

namespace vd {

constexpr int HLD = 36;      // halo row stride in floats (32 channels + 4 pad: conflict-free-ish b128 reads)

struct HaloGeom {
    int tw_log, th_log;          // tile width / height (powers of two)
    int TF;                      // frames per tile (1 or 2)
    int HPW, HPH, HP;            // halo patch width / height, total halo pixels
    int tiles_x, tiles_y;
};

// PRO: the operand prologue (affine + SiLU) is compiled in or out -- no runtime branch in the K loop: hipcc answers a
// branch around a load with s_waitcnt vmcnt(0) at the join, which would drain the weight ring (guide 5, trap (c)).
// For the same reason every load below is unconditional (clamped address, value selected afterwards).
template <int BM, int BN, bool PRO>
__global__ __launch_bounds__(256, 2) void conv3x3_frag_kernel(IgemmArgs a, HaloGeom g) {
    constexpr int MI = BM / 64, NI = BN / 64;
    constexpr int NEL = BM == 128 ? 7 : 4;                  // halo float4 per thread: HP <= 32*NEL (tiles >= 8x8)
    extern __shared__ __attribute__((aligned(16))) float smem[];      // [2][HP][HLD]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 31, lh = lane >> 5;
    const int TW = 1 << g.tw_log, TH = 1 << g.th_log;
    int bx = blockIdx.x;
    const int tx0 = (bx % g.tiles_x) << g.tw_log; bx /= g.tiles_x;
    const int ty0 = (bx % g.tiles_y) << g.th_log; bx /= g.tiles_y;
    const int f0 = bx * g.TF;
    const int n0 = blockIdx.y * BN;
    const int Hl = a.Hs << a.ups, Wl = a.Ws << a.ups;       // logical (post-upsample) input == output dims
    const int C1 = a.Cin - a.C0;
    const int lq = tid & 7, lrow = tid >> 3;                 // channel quad (fixed per thread), first halo pixel
    const int nchunk = a.Cin >> 5;
    const int ncoblk = a.Cout >> 5;

    // ---- halo elements owned by this thread: source pixel (or -1); frame slot (0/1) in bit 30
    int soff[NEL];
#pragma unroll
    for (int e = 0; e < NEL; ++e) {
        const int pix = lrow + e * 32;
        soff[e] = -1;
        if (pix < g.HP) {
            const int per = g.HPH * g.HPW;
            const int f = pix / per, rem = pix - f * per;
            const int hy = rem / g.HPW, hx = rem - hy * g.HPW;
            const int ly = ty0 + hy - 1, lx = tx0 + hx - 1, n = f0 + f;
            if (n < a.nfr && ly >= 0 && ly < Hl && lx >= 0 && lx < Wl)
                soff[e] = ((n * a.Hs + (ly >> a.ups)) * a.Ws + (lx >> a.ups)) | (f << 30);
        }
    }
    int hb[MI];                                              // halo pixel of this lane's output rows at tap (0,0)
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = wm * (BM / 2) + i * 32 + lr;
        const int x = m & (TW - 1), y = (m >> g.tw_log) & (TH - 1), f = m >> (g.tw_log + g.th_log);
        hb[i] = ((f * g.HPH + y) * g.HPW + x) * HLD + lh * 4;
    }
    // ---- weight fragment addressing: group (tap, chunk, kg), block j -> one float4 per lane
    int cob[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) cob[j] = min(blockIdx.y * (BN / 32) + wn * NI + j, ncoblk - 1);   // clamp: masked at store
    const size_t tap_stride = (size_t)nchunk * ncoblk * 1024;
    const float* wl = a.wfrag + lane * 4;

    f32x4 rh[NEL], aff[2][2];
    auto halo_prefetch = [&](int chunk) {
        const int c = chunk * 32 + lq * 4;
        const float* base; int cc, ld;
        if (c < a.C0) { base = a.src0; cc = c; ld = a.C0; } else { base = a.src1; cc = c - a.C0; ld = C1; }
#pragma unroll
        for (int e = 0; e < NEL; ++e)      // invalid elements read pixel 0 (always mapped) and are zeroed at store time
            rh[e] = *reinterpret_cast<const f32x4*>(base + (size_t)(soff[e] < 0 ? 0 : (soff[e] & 0x3fffffff)) * ld + cc);
        if constexpr (PRO) {
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                const int n = min(f0 + (f < g.TF ? f : 0), a.nfr - 1);
                aff[f][0] = *reinterpret_cast<const f32x4*>(a.affA + (size_t)n * a.Cin + c);
                aff[f][1] = *reinterpret_cast<const f32x4*>(a.affB + (size_t)n * a.Cin + c);
            }
        }
    };
    auto halo_store_one = [&](int e, float* Hd) {
        const int pix = lrow + e * 32;
        if (pix < g.HP) {
            f32x4 v = rh[e];
            if constexpr (PRO) {
                const bool second = (soff[e] >> 30) & 1;
                v = v * (second ? aff[1][0] : aff[0][0]) + (second ? aff[1][1] : aff[0][1]);
                v.x = silu_f(v.x); v.y = silu_f(v.y); v.z = silu_f(v.z); v.w = silu_f(v.w);
            }
            if (soff[e] < 0) v = f32x4{0.f, 0.f, 0.f, 0.f};      // zero padding AFTER norm + activation
            *reinterpret_cast<f32x4*>(Hd + pix * HLD + lq * 4) = v;
        }
    };

    // Accumulators start at bias + residual (+ per-frame bias): the residual tile is fetched while the first halo is
    // being staged instead of behind a drained wait in the epilogue, and costs no extra registers.
    auto out_frame = [&](int i, int r) {
        const int m = wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        return min(f0 + (m >> (g.tw_log + g.th_log)), a.nfr - 1);
    };
    auto out_pixel = [&](int i, int r, bool& ok) -> size_t {
        const int m = wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int x = m & (TW - 1), y = (m >> g.tw_log) & (TH - 1), f = m >> (g.tw_log + g.th_log);
        const int n = f0 + f;
        ok = n < a.nfr;
        return ((size_t)min(n, a.nfr - 1) * Hl + ty0 + y) * Wl + tx0 + x;
    };
    f32x16 acc[MI][NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int co = min(n0 + wn * (BN / 2) + j * 32 + lr, a.Cout - 1);
        const float bv = a.bias ? a.bias[co] : 0.f;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = bv;
            if (a.res) {          // one branch around all 16 loads (a branch per load would drain the queue each time)
#pragma unroll
                for (int r = 0; r < 16; ++r) { bool ok; acc[i][j][r] += a.res[out_pixel(i, r, ok) * a.res_ld + co]; }
            }
            if (a.fbias) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] += a.fbias[(size_t)out_frame(i, r) * a.fbias_ld + co];
            }
        }
    }

    f32x4 bfr[3][NI], afr[2][MI];
    auto b_load = [&](int slot, int chunk, int tap, int kg) {
        const float* p = wl + (size_t)tap * tap_stride + ((size_t)chunk * ncoblk) * 1024 + kg * 256;
#pragma unroll
        for (int j = 0; j < NI; ++j) bfr[slot][j] = *reinterpret_cast<const f32x4*>(p + (size_t)cob[j] * 1024);
    };

    // ---- prologue: chunk 0's halo (all elements), first two weight groups
    halo_prefetch(0);
    b_load(0, 0, 0, 0);
    b_load(1, 0, 0, 1);
#pragma unroll
    for (int e = 0; e < NEL; ++e) halo_store_one(e, smem);
    __syncthreads();

    for (int chunk = 0; chunk < nchunk; ++chunk) {
        const int nxt = min(chunk + 1, nchunk - 1);             // last chunk: harmless redundant prefetch, no branch
        const float* Hcur = smem + (chunk & 1) * g.HP * HLD;
        float* Hnext = smem + ((chunk + 1) & 1) * g.HP * HLD;
        halo_prefetch(nxt);
#pragma unroll
        for (int i = 0; i < MI; ++i) afr[0][i] = *reinterpret_cast<const f32x4*>(Hcur + hb[i]);
#pragma unroll
        for (int gi = 0; gi < 36; ++gi) {
            const int tap = gi >> 2, kg = gi & 3;
            {   // weights two groups ahead (rolls into the next chunk's first two groups)
                const int g2 = gi + 2;
                if (g2 < 36) b_load(g2 % 3, chunk, g2 >> 2, g2 & 3);
                else b_load(g2 % 3, nxt, 0, g2 - 36);
            }
            if (gi + 1 < 36) {   // A fragments one group ahead
                const int t1 = (gi + 1) >> 2, k1 = (gi + 1) & 3;
                const int off = ((t1 / 3) * g.HPW + (t1 % 3)) * HLD + k1 * 8;
#pragma unroll
                for (int i = 0; i < MI; ++i) afr[(gi + 1) & 1][i] = *reinterpret_cast<const f32x4*>(Hcur + hb[i] + off);
            }
            // Pin the software pipeline: without this fence hipcc's scheduler sinks the prefetch loads down to their
            // first use (register-pressure heuristic) and every k-group becomes load -> vmcnt(0) -> MFMA.
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(afr[gi & 1][i][e], bfr[gi % 3][j][e], acc[i][j], 0, 0, 0);
            if (kg == 3 && tap >= 2 && tap - 2 < NEL) halo_store_one(tap - 2, Hnext);   // last chunk: writes the idle buffer
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }

    // ---- epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int co = n0 + wn * (BN / 2) + j * 32 + lr;
        if (co >= a.Cout) continue;
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                bool ok;
                const size_t pix = out_pixel(i, r, ok);
                if (ok) a.out[pix * a.ldo + co] = acc[i][j][r];
            }
    }
}

static int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }
static bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

bool conv_halo_supported(const IgemmArgs& a) {
    const int Hl = a.Hs << a.ups, Wl = a.Ws << a.ups;
    return a.wfrag != nullptr && a.ksz == 3 && a.stride == 1 && a.pad == 1 && is_pow2(Hl) && is_pow2(Wl) && Hl >= 8 &&
           Wl >= 8 && a.Cout % 32 == 0 && (a.affA != nullptr) == (a.act != 0);
}

template <int BM, int BN, bool PRO>
static int launch_halo_p(const IgemmArgs& a, hipStream_t s) {
    const int Hl = a.Hs << a.ups, Wl = a.Ws << a.ups;
    HaloGeom g;
    const int TW = std::min(Wl, BM == 128 ? 16 : 8);
    const int TH = std::min(Hl, 8);
    g.tw_log = ilog2(TW); g.th_log = ilog2(TH);
    g.TF = BM / (TW * TH);
    g.HPW = TW + 2; g.HPH = TH + 2; g.HP = g.TF * g.HPW * g.HPH;
    g.tiles_x = Wl / TW; g.tiles_y = Hl / TH;
    VD_REQUIRE(g.TF >= 1 && g.TF <= 2 && g.HP <= 32 * (BM == 128 ? 7 : 4), "halo tile geometry");
    const size_t lds = (size_t)2 * g.HP * HLD * sizeof(float);
    static size_t attr_lds = 0;
    if (lds > attr_lds) {
        VD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_frag_kernel<BM, BN, PRO>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_lds = lds;
    }
    const int fgroups = (a.nfr + g.TF - 1) / g.TF;
    dim3 grid(g.tiles_x * g.tiles_y * fgroups, (a.Cout + BN - 1) / BN);
    hipLaunchKernelGGL((conv3x3_frag_kernel<BM, BN, PRO>), grid, dim3(256), lds, s, a, g);
    VD_HIP(hipGetLastError());
    return 0;
}

template <int BM, int BN>
static int launch_halo_t(const IgemmArgs& a, hipStream_t s) {
    return a.affA ? launch_halo_p<BM, BN, true>(a, s) : launch_halo_p<BM, BN, false>(a, s);
}

int launch_conv_halo(const IgemmArgs& a, int tile_class, hipStream_t s) {
    switch (tile_class) {
        case 0: return launch_halo_t<128, 128>(a, s);
        case 1: return launch_halo_t<128, 64>(a, s);
        case 2: return launch_halo_t<64, 128>(a, s);
        default: return launch_halo_t<64, 64>(a, s);
    }
}

// host-side repack: OIHW -> [tap][chunk][cout/32][kg][h][r][e]   (see the header comment)
void pack_conv3_frag(const float* oihw, float* out, int O, int I) {
    const int nchunk = I / 32, ncoblk = O / 32;
    for (int tap = 0; tap < 9; ++tap)
        for (int ch = 0; ch < nchunk; ++ch)
            for (int cb = 0; cb < ncoblk; ++cb)
                for (int kg = 0; kg < 4; ++kg)
                    for (int h = 0; h < 2; ++h)
                        for (int r = 0; r < 32; ++r)
                            for (int e = 0; e < 4; ++e) {
                                const int co = cb * 32 + r, ci = ch * 32 + kg * 8 + h * 4 + e;
                                out[(((((size_t)tap * nchunk + ch) * ncoblk + cb) * 4 + kg) * 64 + h * 32 + r) * 4 + e] =
                                    oihw[((size_t)co * I + ci) * 9 + tap];
                            }
}

}  // namespace vd
