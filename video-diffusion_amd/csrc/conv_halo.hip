// 3x3 stride-1 convolution with an LDS-staged input halo tile, fp32 MFMA, gfx950.
//
// The generic kernel (igemm.hip) gathers and transforms the A operand once per (tap, channel chunk):
// 9x the global loads and 9x the GroupNorm-affine + SiLU VALU work of what the math needs.  Here a
// block owns BM output pixels shaped TF frames x TH x TW; per 32-channel chunk it stages the
// (TH+2)x(TW+2) input patch of each frame ONCE -- normalised, activated, zero-padded at the image
// border, optionally read through the nearest-x2 upsample and the virtual concat -- and the nine taps
// read their A fragments from that tile at shifted pixel offsets.  Per K-step (tap, chunk) only the
// [BN][32] weight tile moves (register-staged, double-buffered); the halo is single-buffered and
// swapped once per chunk (its global loads are issued 6 taps ahead).
//
//   LDS: halo HP x 36 floats (<= 41 KB) + weights 2 x BN x 36 floats  -> 2 blocks per CU.
//   Everything else (MFMA mapping, operand k-order trick, epilogue) as in igemm.hip.
#include "vd_common.h"

namespace vd {

constexpr int HBK = 32;
constexpr int HLD = 36;

struct HaloGeom {
    int tw_log, th_log;          // tile width / height (powers of two)
    int TF;                      // frames per tile
    int HPW, HPH, HP;            // halo patch width / height, total halo pixels
    int tiles_x, tiles_y;
};

template <int BM, int BN>
__global__ __launch_bounds__(256, 2) void conv3x3_halo_kernel(IgemmArgs a, HaloGeom g) {
    constexpr int MI = BM / 64, NI = BN / 64, BR = BN / 32;
    constexpr int NEL = BM == 128 ? 7 : 4;                  // halo float4 per thread: HP <= 32*NEL (tiles >= 8x8)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Hs = smem;                         // [HP][HLD]
    float* Bs = smem + g.HP * HLD;            // [2][BN][HLD]

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
    const int lq = tid & 7;                                  // channel quad of the chunk: fixed per thread

    // ---- halo elements owned by this thread: source pixel (or -1) and frame
    int soff[NEL], sfr[NEL];
#pragma unroll
    for (int e = 0; e < NEL; ++e) {
        const int pix = (tid >> 3) + e * 32;
        soff[e] = -1; sfr[e] = 0;
        if (pix < g.HP) {
            const int per = g.HPH * g.HPW;
            const int f = pix / per, rem = pix - f * per;
            const int hy = rem / g.HPW, hx = rem - hy * g.HPW;
            const int ly = ty0 + hy - 1, lx = tx0 + hx - 1, n = f0 + f;
            if (n < a.nfr && ly >= 0 && ly < Hl && lx >= 0 && lx < Wl) {
                soff[e] = (n * a.Hs + (ly >> a.ups)) * a.Ws + (lx >> a.ups);
                sfr[e] = n;
            }
        }
    }
    // ---- A fragment base (halo pixel of this lane's output rows at tap (0,0))
    int hb[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) {
        const int m = wm * (BM / 2) + i * 32 + lr;
        const int x = m & (TW - 1), y = (m >> g.tw_log) & (TH - 1), f = m >> (g.tw_log + g.th_log);
        hb[i] = (f * g.HPH + y) * g.HPW + x;
    }
    const int lrow = tid >> 3;
    const int nchunk = a.Cin / HBK;
    const int nsteps = nchunk * 9;

    f32x4 rh[NEL], rb[BR];
    auto halo_prefetch = [&](int chunk) {
        const int c = chunk * HBK + lq * 4;
        const float* base; int cc, ld;
        if (c < a.C0) { base = a.src0; cc = c; ld = a.C0; } else { base = a.src1; cc = c - a.C0; ld = C1; }
#pragma unroll
        for (int e = 0; e < NEL; ++e)
            rh[e] = soff[e] >= 0 ? *reinterpret_cast<const f32x4*>(base + (size_t)soff[e] * ld + cc)
                                 : f32x4{0.f, 0.f, 0.f, 0.f};
    };
    auto halo_store = [&](int chunk) {
        const int c = chunk * HBK + lq * 4;
#pragma unroll
        for (int e = 0; e < NEL; ++e) {
            const int pix = lrow + e * 32;
            if (pix < g.HP) {
                f32x4 v = rh[e];
                if (soff[e] >= 0) {            // zero padding AFTER norm + activation
                    if (a.affA) {
                        const f32x4 sa = *reinterpret_cast<const f32x4*>(a.affA + (size_t)sfr[e] * a.Cin + c);
                        const f32x4 sb = *reinterpret_cast<const f32x4*>(a.affB + (size_t)sfr[e] * a.Cin + c);
                        v = v * sa + sb;
                    }
                    if (a.act) { v.x = silu_f(v.x); v.y = silu_f(v.y); v.z = silu_f(v.z); v.w = silu_f(v.w); }
                }
                *reinterpret_cast<f32x4*>(Hs + pix * HLD + lq * 4) = v;
            }
        }
    };
    auto b_prefetch = [&](int s) {
        const int chunk = s / 9, tap = s - chunk * 9;
        const float* wt = a.w + ((size_t)tap * a.Cout) * a.Cin + chunk * HBK + lq * 4;
#pragma unroll
        for (int j = 0; j < BR; ++j) {
            const int co = n0 + lrow + 32 * j;
            rb[j] = co < a.Cout ? *reinterpret_cast<const f32x4*>(wt + (size_t)co * a.Cin) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto b_store = [&](int buf) {
        float* Bd = Bs + buf * BN * HLD;
#pragma unroll
        for (int j = 0; j < BR; ++j) *reinterpret_cast<f32x4*>(Bd + (lrow + 32 * j) * HLD + lq * 4) = rb[j];
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto compute = [&](int buf, int toff) {
        const float* Bb = Bs + buf * BN * HLD + (wn * (BN / 2) + lr) * HLD + lh * 4;
#pragma unroll
        for (int kg = 0; kg < HBK / 8; ++kg) {
            f32x4 fa[MI], fb[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) fa[i] = *reinterpret_cast<const f32x4*>(Hs + (hb[i] + toff) * HLD + lh * 4 + kg * 8);
#pragma unroll
            for (int j = 0; j < NI; ++j) fb[j] = *reinterpret_cast<const f32x4*>(Bb + j * 32 * HLD + kg * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][e], fb[j][e], acc[i][j], 0, 0, 0);
        }
    };

    halo_prefetch(0);
    b_prefetch(0);
    halo_store(0);
    b_store(0);
    __syncthreads();
    int s = 0;
    for (int chunk = 0; chunk < nchunk; ++chunk) {
        const bool next_chunk = chunk + 1 < nchunk;
        for (int tap = 0; tap < 9; ++tap, ++s) {
            const bool more = s + 1 < nsteps;
            if (more) b_prefetch(s + 1);
            if (tap == 2 && next_chunk) halo_prefetch(chunk + 1);
            const int kh = tap / 3;
            compute(s & 1, kh * g.HPW + (tap - kh * 3));
            if (more) b_store((s + 1) & 1);
            __syncthreads();
        }
        if (next_chunk) {
            halo_store(chunk + 1);
            __syncthreads();
        }
    }

    // ---- epilogue
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int co = n0 + wn * (BN / 2) + j * 32 + lr;
        if (co >= a.Cout) continue;
        const float bv = a.bias ? a.bias[co] : 0.f;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int x = m & (TW - 1), y = (m >> g.tw_log) & (TH - 1), f = m >> (g.tw_log + g.th_log);
                const int n = f0 + f;
                if (n < a.nfr) {
                    const size_t pix = ((size_t)n * Hl + ty0 + y) * Wl + tx0 + x;
                    float v = acc[i][j][r] + bv;
                    if (a.res) v += a.res[pix * a.res_ld + co];
                    if (a.fbias) v += a.fbias[(size_t)n * a.fbias_ld + co];
                    a.out[pix * a.ldo + co] = v;
                }
            }
        }
    }
}

static int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }
static bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

bool conv_halo_supported(const IgemmArgs& a) {
    const int Hl = a.Hs << a.ups, Wl = a.Ws << a.ups;
    return a.ksz == 3 && a.stride == 1 && a.pad == 1 && is_pow2(Hl) && is_pow2(Wl) && Hl >= 4 && Wl >= 4;
}

template <int BM, int BN>
static int launch_halo_t(const IgemmArgs& a, hipStream_t s) {
    const int Hl = a.Hs << a.ups, Wl = a.Ws << a.ups;
    HaloGeom g;
    const int TW = std::min(Wl, BM == 128 ? 16 : 8);
    const int TH = std::min(Hl, BM / TW >= 8 ? 8 : BM / TW);
    g.tw_log = ilog2(TW); g.th_log = ilog2(TH);
    g.TF = BM / (TW * TH);
    g.HPW = TW + 2; g.HPH = TH + 2; g.HP = g.TF * g.HPW * g.HPH;
    g.tiles_x = Wl / TW; g.tiles_y = Hl / TH;
    if (g.HP > 32 * (BM == 128 ? 7 : 4)) return 1;        // tiny images (4x4 tiles): caller falls back to the generic kernel
    const size_t lds = ((size_t)g.HP * HLD + 2 * BN * HLD) * sizeof(float);
    static size_t attr_lds = 0;
    if (lds > attr_lds) {
        VD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_halo_kernel<BM, BN>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_lds = lds;
    }
    const int fgroups = (a.nfr + g.TF - 1) / g.TF;
    dim3 grid(g.tiles_x * g.tiles_y * fgroups, (a.Cout + BN - 1) / BN);
    hipLaunchKernelGGL((conv3x3_halo_kernel<BM, BN>), grid, dim3(256), lds, s, a, g);
    VD_HIP(hipGetLastError());
    return 0;
}

int launch_conv_halo(const IgemmArgs& a, int tile_class, hipStream_t s) {
    switch (tile_class) {
        case 0: return launch_halo_t<128, 128>(a, s);
        case 1: return launch_halo_t<128, 64>(a, s);
        case 2: return launch_halo_t<64, 128>(a, s);
        default: return launch_halo_t<64, 64>(a, s);
    }
}

}  // namespace vd
