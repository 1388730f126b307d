// Implicit-GEMM convolution / linear layer on fp32 MFMA (v_mfma_f32_32x32x2_f32), gfx950.
//
// One kernel covers every dense contraction of the denoise step:
//   conv3x3 s1/s2 (+fused nearest-x2 upsample, + virtual channel concat), conv1x1, nn.Linear.
// out[m][co] = bias[co] + res[m][co] + sum_{tap,ci} f(in[pixel(m)+tap][ci]) * w[tap][co][ci]
// with f(v) = silu(v*A[n][ci] + B[n][ci]) -- GroupNorm (+FiLM scale/shift) and SiLU of the *producer*
// folded into this kernel's operand load, so normalised activations are never materialised
// (ResBlock: unet.py:185-198).
//
// Tiling (wave64, 4 waves = 2x2, each wave (BM/2)x(BN/2) as 32x32 MFMA tiles):
//   K-step = (tap, 32 input channels).  A tile [BM][32] gathered per tap from NHWC, B tile [BN][32]
//   from weights stored [tap][Cout][Cin].  Rows padded to 36 floats: ds_read_b128 of 16 lanes on
//   16 distinct 16-byte slots -> conflict-free (guide: LDS banking, ds_read_b128 lane groups).
//   MFMA 32x32x2 takes one f32 per lane per operand with k = lane>>5; a lane's float4 from LDS
//   feeds 4 consecutive MFMAs (k order is free as long as A and B agree), so operand traffic is
//   one ds_read_b128 per 4 MFMAs per fragment.
//   Pipeline: global loads of step s+1 are issued before the MFMAs of step s (register staging:
//   the operand transform needs VALU anyway), written to the other LDS buffer after them; one
//   barrier per K-step.
#include <algorithm>

#include <string>

#include "vd_common.h"

namespace vd {

constexpr int BK = 32;
constexpr int LDP = 36;   // padded LDS row (floats)

template <int BM, int BN>
__global__ __launch_bounds__(256) void igemm_kernel(IgemmArgs a) {
    constexpr int MI = BM / 64, NI = BN / 64;     // 32x32 tiles per wave in M / N
    constexpr int AR = BM / 32, BR = BN / 32;     // loader rows per thread
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                     // [2][BM][LDP]
    float* Bs = smem + 2 * BM * LDP;      // [2][BN][LDP]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 31, lh = lane >> 5;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;

    // ---- loader geometry: thread -> (row lrow + 32j, 4-float quad lq of the 32-channel chunk)
    const int lrow = tid >> 3, lq = tid & 7;
    const int HWo = a.Ho * a.Wo;
    const int Hl = a.Hs << a.ups, Wl = a.Ws << a.ups;
    int pn[AR], py[AR], px[AR];
#pragma unroll
    for (int j = 0; j < AR; ++j) {
        int m = m0 + lrow + 32 * j;
        if (m < a.M) {
            int n = m / HWo, r = m - n * HWo;
            int oy = r / a.Wo;
            pn[j] = n; py[j] = oy * a.stride - a.pad; px[j] = (r - oy * a.Wo) * a.stride - a.pad;
        } else {
            pn[j] = -1; py[j] = 0; px[j] = 0;
        }
    }
    const int ntaps = a.ksz * a.ksz;
    const int nchunk = a.Cin / BK;
    const int nsteps = ntaps * nchunk;
    const int C1 = a.Cin - a.C0;

    f32x4 ra[AR], rb[BR];
    unsigned avalid = 0;

    auto prefetch = [&](int s) {
        const int chunk = s / ntaps, tap = s - chunk * ntaps;
        const int kh = tap / a.ksz, kw = tap - kh * a.ksz;
        const int c = chunk * BK + lq * 4;
        const float* base; int cc, ld;
        if (c < a.C0) { base = a.src0; cc = c; ld = a.C0; } else { base = a.src1; cc = c - a.C0; ld = C1; }
        avalid = 0;
#pragma unroll
        for (int j = 0; j < AR; ++j) {
            int iy = py[j] + kh, ix = px[j] + kw;
            const bool ok = pn[j] >= 0 && iy >= 0 && iy < Hl && ix >= 0 && ix < Wl;
            // unconditional load (out-of-image taps read pixel 0 and are zeroed when staged): a branch around a
            // load makes hipcc drain the whole queue (vmcnt(0)) at the join
            const size_t pix = ok ? ((size_t)pn[j] * a.Hs + (iy >> a.ups)) * a.Ws + (ix >> a.ups) : 0;
            ra[j] = *reinterpret_cast<const f32x4*>(base + pix * ld + cc);
            avalid |= (ok ? 1u : 0u) << j;
        }
        const float* wt = a.w + ((size_t)tap * a.Cout) * a.Cin + c;
#pragma unroll
        for (int j = 0; j < BR; ++j) {
            const int co = min(n0 + lrow + 32 * j, a.Cout - 1);       // rows past Cout: duplicates, masked at the store
            rb[j] = *reinterpret_cast<const f32x4*>(wt + (size_t)co * a.Cin);
        }
    };

    auto stage = [&](int s, int buf) {
        const int chunk = s / ntaps;
        const int c = chunk * BK + lq * 4;
        float* Ad = As + buf * BM * LDP;
        float* Bd = Bs + buf * BN * LDP;
#pragma unroll
        for (int j = 0; j < AR; ++j) {
            f32x4 v = ra[j];
            if (a.affA) {
                const size_t fr = (size_t)max(pn[j], 0) * a.Cin + c;
                const f32x4 sa = *reinterpret_cast<const f32x4*>(a.affA + fr);
                const f32x4 sb = *reinterpret_cast<const f32x4*>(a.affB + fr);
                v = v * sa + sb;
            }
            if (a.act) { v.x = silu_f(v.x); v.y = silu_f(v.y); v.z = silu_f(v.z); v.w = silu_f(v.w); }
            if (!(avalid & (1u << j))) v = f32x4{0.f, 0.f, 0.f, 0.f};   // zero padding AFTER norm+activation (conv pads its input)
            *reinterpret_cast<f32x4*>(Ad + (lrow + 32 * j) * LDP + lq * 4) = v;
        }
#pragma unroll
        for (int j = 0; j < BR; ++j)
            *reinterpret_cast<f32x4*>(Bd + (lrow + 32 * j) * LDP + lq * 4) = rb[j];
    };

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto compute = [&](int buf) {
        const float* Ab = As + buf * BM * LDP + (wm * (BM / 2) + lr) * LDP + lh * 4;
        const float* Bb = Bs + buf * BN * LDP + (wn * (BN / 2) + lr) * LDP + lh * 4;
#pragma unroll
        for (int kg = 0; kg < BK / 8; ++kg) {
            f32x4 fa[MI], fb[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) fa[i] = *reinterpret_cast<const f32x4*>(Ab + i * 32 * LDP + kg * 8);
#pragma unroll
            for (int j = 0; j < NI; ++j) fb[j] = *reinterpret_cast<const f32x4*>(Bb + j * 32 * LDP + kg * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][e], fb[j][e], acc[i][j], 0, 0, 0);
        }
    };

    prefetch(0);
    stage(0, 0);
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
        const bool more = s + 1 < nsteps;
        if (more) prefetch(s + 1);
        compute(s & 1);
        if (more) stage(s + 1, (s + 1) & 1);
        __syncthreads();
    }

    // ---- epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int co = n0 + wn * (BN / 2) + j * 32 + lr;
        if (co >= a.Cout) continue;
        const float bv = a.bias ? a.bias[co] : 0.f;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            // residual loads batched ahead of the stores (unconditional, row clamped): one wait per tile
            // instead of one drained L2 round trip per element
            const int mb = m0 + wm * (BM / 2) + i * 32 + 4 * lh;
            f32x16 v = acc[i][j];
            if (a.res) {
                f32x16 rv;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    rv[r] = a.res[(size_t)min(mb + (r & 3) + 8 * (r >> 2), a.M - 1) * a.res_ld + co];
                v += rv;
            }
            if (a.fbias) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    v[r] += a.fbias[(size_t)(min(mb + (r & 3) + 8 * (r >> 2), a.M - 1) / HWo) * a.fbias_ld + co];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mb + (r & 3) + 8 * (r >> 2);
                if (m < a.M) a.out[(size_t)m * a.ldo + co] = v[r] + bv;
            }
        }
    }
}

template <int BM, int BN>
static int launch_t(const IgemmArgs& a, hipStream_t s) {
    const size_t lds = 2 * (BM + BN) * LDP * sizeof(float);
    VD_RAISE_LDS((&igemm_kernel<BM, BN>), lds);
    dim3 grid((a.M + BM - 1) / BM, (a.Cout + BN - 1) / BN);
    hipLaunchKernelGGL((igemm_kernel<BM, BN>), grid, dim3(256), lds, s, a);
    VD_HIP(hipGetLastError());
    return 0;
}

// One launch: every operand of `a` is small enough for the 32-bit byte offsets the kernels address with.
static int launch_igemm_one(const IgemmArgs& a, hipStream_t s) {
    if (a.ups_phase) {                                                  // sub-pixel weight image: only conv_wino_r64.hip reads it
        VD_REQUIRE(conv_wino_r64_supported(a), "sub-pixel Upsample conv: shape not covered by conv_wino_r64.hip");
        return launch_conv_wino_r64(a, s);
    }
    if (gemm_split_supported(a) || conv_split_supported(a)) return launch_gemm_split(a, igemm_tile_class(igemm_sel_M(a), a.Cout), s);
    if (conv_wino_z128_act_supported(a) || conv_wino_z128_supported(a)) return launch_conv_wino_z128(a, s);
    if (conv_wino_r64_supported(a)) return launch_conv_wino_r64(a, s);
    VD_REQUIRE(!a.wsplit, "split weight image given for a shape the split kernels do not cover");
    if (gemm_frag_supported(a)) return launch_gemm_frag(a, igemm_tile_class(igemm_sel_M(a), a.Cout), s);
    if (conv_wino_supported(a)) return launch_conv_wino(a, s);
    VD_REQUIRE(a.w != nullptr, "this shape runs on the generic kernel and needs [tap][Cout][Cin] weights");
    switch (igemm_tile_class(igemm_sel_M(a), a.Cout)) {
        case 0: return launch_t<128, 128>(a, s);
        case 1: return launch_t<128, 64>(a, s);
        case 2: return launch_t<64, 128>(a, s);
        default: return launch_t<64, 64>(a, s);
    }
}

// Frames (rows, for a linear layer) per launch.  The fast kernels reach their operands through buffer descriptors with
// 32-bit byte offsets and bound each operand to 2^28 elements (*_supported); a window that is merely BIG -- 160 frames of
// 128x128x128 channels, BASELINE configs[4] -- is cut along the frame dimension into equal launches with the base
// pointers advanced.  Frames are independent in every one of these kernels (per-frame affine / bias / statistics rows
// move with them), so the results are those of one launch, bit for bit.
int igemm_frames_per_launch(const IgemmArgs& a) {
    const size_t lim = ((size_t)1 << 28) - 1;
    const size_t in_pf = (size_t)a.Hs * a.Ws * std::max(a.C0, a.Cin - a.C0);
    const size_t out_pf = (size_t)a.Ho * a.Wo * std::max(a.ldo, a.res ? a.res_ld : 0);
    const size_t pf = std::max(in_pf, out_pf);
    if (pf == 0 || (size_t)a.nfr * pf <= lim || a.zcount > 1) return a.nfr;
    const size_t align = (a.Hs == 1 && a.Ws == 1) ? 256 : 4;       // whole 128-row tiles / whole 4-frame groups (conv_wino_r64 TF4)
    size_t maxfr = lim / pf / align * align;
    if (maxfr == 0) return 0;
    const size_t nl = ((size_t)a.nfr + maxfr - 1) / maxfr;
    size_t per = ((size_t)a.nfr + nl - 1) / nl;                       // equal launches
    per = (per + align - 1) / align * align;
    return (int)std::min(per, maxfr);
}

int launch_igemm(const IgemmArgs& a, hipStream_t s) {
    VD_REQUIRE(a.Cin % BK == 0, "Cin must be a multiple of 32 (pad the operand)");
    VD_REQUIRE(a.C0 % BK == 0 && a.C0 <= a.Cin, "concat split must be a multiple of 32");
    VD_REQUIRE(a.src1 != nullptr || a.C0 == a.Cin, "second source missing");
    VD_REQUIRE(a.ksz == 1 || a.ksz == 3, "kernel size 1 or 3");
    VD_REQUIRE(a.M > 0 && a.Cout > 0, "empty problem");
    VD_REQUIRE(a.M == a.nfr * a.Ho * a.Wo, "M != nfr*Ho*Wo");
    const int per = igemm_frames_per_launch(a);
    VD_REQUIRE(per > 0, "one frame of this layer exceeds 2^28 elements");
    if (per >= a.nfr) return launch_igemm_one(a, s);
    VD_REQUIRE(!(a.stats && a.stats_hw > 0), "GroupNorm partial sums from the split GEMM: one launch only (the tile choice, and with it the table, depends on M)");
    const size_t HWi = (size_t)a.Hs * a.Ws, HWo = (size_t)a.Ho * a.Wo;
    for (int f0 = 0; f0 < a.nfr; f0 += per) {
        IgemmArgs b = a;
        b.nfr = std::min(per, a.nfr - f0);
        b.nfr_sel = a.nfr_sel ? a.nfr_sel : a.nfr;
        b.M = b.nfr * a.Ho * a.Wo;
        b.src0 = a.src0 + (size_t)f0 * HWi * a.C0;
        if (a.src1) b.src1 = a.src1 + (size_t)f0 * HWi * (a.Cin - a.C0);
        if (a.res) b.res = a.res + (size_t)f0 * HWo * a.res_ld;
        b.out = a.out + (size_t)f0 * HWo * a.ldo;
        if (a.affA) { b.affA = a.affA + (size_t)f0 * a.Cin; b.affB = a.affB + (size_t)f0 * a.Cin; }
        if (a.fbias) b.fbias = a.fbias + (size_t)f0 * a.fbias_ld;
        if (a.stats) b.stats = a.stats + (size_t)f0 * a.stats_split * a.Cout * 2;
        if (a.side) { b.side = a.side + (size_t)f0 * HWo * a.Cin; b.sideA = a.sideA + (size_t)f0 * a.Cin; b.sideB = a.sideB + (size_t)f0 * a.Cin; }
        const int rc = launch_igemm_one(b, s);
        if (rc) return rc;
    }
    return 0;
}

// Tile choice: big tiles when the grid still fills 256 CUs x 2 blocks, smaller ones otherwise.
// 0: 128x128, 1: 128x64, 2: 64x128, 3: 64x64
int igemm_tile_class(int M, int Cout) {
    const long t128 = (long)((M + 127) / 128) * ((Cout + 127) / 128);
    if (Cout <= 64) return M >= 128 * 512 ? 1 : 3;
    if (t128 >= 512) return 0;
    const long t64n = (long)((M + 63) / 64) * ((Cout + 127) / 128);
    if (t64n >= 384) return 2;
    return 3;
}

}  // namespace vd
