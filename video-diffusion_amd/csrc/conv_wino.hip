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
    int tt_log;                  // tiles per dim per frame in the block (8 -> 3, 4 -> 2)
    int TF;                      // frames per block (1 or 4)
    int P;                       // patch width = 2*tiles + 2
    int PX;                      // patch pixels = TF * P * P
    int tiles_x, tiles_y;        // blocks per frame
};

// PRO: operand prologue (affine + SiLU) compiled in/out; TF4: four 8x8 frames per block (else one frame, 16x16 pixels).
template <bool PRO, bool TF4>
__global__ __launch_bounds__(256, 1) void conv3x3_wino_kernel(IgemmArgs a, WinoGeom g) {
    constexpr int NX = TF4 ? 7 : 6;            // patch float4 per thread: 400 / 324 pixels x 4 quads over 256 threads
    constexpr int NAF = TF4 ? 4 : 1;           // frames whose affine pairs a thread may need
    extern __shared__ __attribute__((aligned(16))) float smem[];      // [2][PX][WLD]; reused as Z exchange at the end
    const int tid = threadIdx.x, lane = tid & 63, wi = tid >> 6;      // wi = Winograd row owned by this wave
    const int lr = lane & 31, lh = lane >> 5;
    const int TT = 1 << g.tt_log;
    int bx = blockIdx.x;
    const int bxx = bx % g.tiles_x; bx /= g.tiles_x;
    const int byy = bx % g.tiles_y; bx /= g.tiles_y;
    const int f0 = bx * g.TF;
    const int ox0 = bxx * 2 * TT, oy0 = byy * 2 * TT;               // output-pixel origin of the block
    const int Hl = a.Hs << a.ups, Wl = a.Ws << a.ups;
    const int C1 = a.Cin - a.C0;
    const int nchunk = a.Cin / WKC, ncoblk = a.Cout >> 5;
    const int cob0 = blockIdx.y * 2;                                 // BN = 64 = 2 cout blocks

    // ---- patch elements owned by this thread (pixel = idx>>2, quad = tid&3): source pixel | frame slot<<28, or -1
    const int lq = tid & 3;
    int soff[NX];
#pragma unroll
    for (int e = 0; e < NX; ++e) {
        const int pix = (tid >> 2) + e * 64;
        soff[e] = -1;
        if (pix < g.PX) {
            const int per = g.P * g.P;
            const int f = pix / per, rem = pix - f * per;
            const int py = rem / g.P, px = rem - py * g.P;
            const int ly = oy0 + py - 1, lx = ox0 + px - 1, n = f0 + f;
            if (n < a.nfr && ly >= 0 && ly < Hl && lx >= 0 && lx < Wl)
                soff[e] = ((n * a.Hs + (ly >> a.ups)) * a.Ws + (lx >> a.ups)) | (f << 28);
        }
    }
    f32x4 rx[NX], aff[NAF][2];
    auto x_load = [&](int chunk) {
        const int c = chunk * WKC + lq * 4;
        const float* base; int cc, ld;
        if (c < a.C0) { base = a.src0; cc = c; ld = a.C0; } else { base = a.src1; cc = c - a.C0; ld = C1; }
#pragma unroll
        for (int e = 0; e < NX; ++e)
            rx[e] = *reinterpret_cast<const f32x4*>(base + (size_t)(soff[e] < 0 ? 0 : (soff[e] & 0x0fffffff)) * ld + cc);
        if constexpr (PRO) {
#pragma unroll
            for (int f = 0; f < NAF; ++f) {
                const int n = min(f0 + (f < g.TF ? f : 0), a.nfr - 1);
                aff[f][0] = *reinterpret_cast<const f32x4*>(a.affA + (size_t)n * a.Cin + c);
                aff[f][1] = *reinterpret_cast<const f32x4*>(a.affB + (size_t)n * a.Cin + c);
            }
        }
    };
    // elements [e0, e1) of this thread's share of the patch: the staging work is spread over the four MFMA groups
    auto x_store = [&](float* Xd, int e0, int e1) {
#pragma unroll
        for (int e = 0; e < NX; ++e) {
            if (e < e0 || e >= e1) continue;
            const int pix = (tid >> 2) + e * 64;
            if (pix < g.PX) {
                f32x4 v = rx[e];
                if constexpr (PRO) {
                    f32x4 sa = aff[0][0], sb = aff[0][1];
                    if constexpr (TF4) {
                        const int fs = (soff[e] >> 28) & 3;
                        sa = fs == 0 ? aff[0][0] : (fs == 1 ? aff[1][0] : (fs == 2 ? aff[2][0] : aff[3][0]));
                        sb = fs == 0 ? aff[0][1] : (fs == 1 ? aff[1][1] : (fs == 2 ? aff[2][1] : aff[3][1]));
                    }
                    v = v * sa + sb;
                    v.x = silu_f(v.x); v.y = silu_f(v.y); v.z = silu_f(v.z); v.w = silu_f(v.w);
                }
                if (soff[e] < 0) v = f32x4{0.f, 0.f, 0.f, 0.f};      // zero padding AFTER norm + activation
                *reinterpret_cast<f32x4*>(Xd + pix * WLD + lq * 4) = v;
            }
        }
    };

    // ---- A fragments: lane (tile m*32+lr, k-half lh); row wi of B^T = [[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]]
    const int r0 = wi == 0 ? 0 : 1, r1 = wi == 3 ? 3 : 2;
    const float s0 = wi == 2 ? -1.f : 1.f, s1 = (wi == 0 || wi == 3) ? -1.f : 1.f;
    int xb[2];                                                      // LDS float offset of the tile's patch origin
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int t = m * 32 + lr;
        const int tx = t & (TT - 1), ty = (t >> g.tt_log) & (TT - 1), f = t >> (2 * g.tt_log);
        xb[m] = ((f * g.P + 2 * ty) * g.P + 2 * tx) * WLD + lh * 4;
    }
    const int rowo0 = r0 * g.P * WLD, rowo1 = r1 * g.P * WLD;
    f32x4 raw[2][8], frag[2][4];
    auto a_read = [&](int slot, const float* Xc, int m, int kg) {
        const float* p = Xc + xb[m] + kg * 8;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            raw[slot][j] = *reinterpret_cast<const f32x4*>(p + rowo0 + j * WLD);
            raw[slot][4 + j] = *reinterpret_cast<const f32x4*>(p + rowo1 + j * WLD);
        }
    };
    auto a_transform = [&](int slot) {          // raw[slot] -> frag[slot]
        f32x4 t[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] = raw[slot][j] * s0 + raw[slot][4 + j] * s1;
        frag[slot][0] = t[0] - t[2]; frag[slot][1] = t[1] + t[2]; frag[slot][2] = t[2] - t[1]; frag[slot][3] = t[1] - t[3];
    };

    // ---- B fragments: U[chunk][xi = 4*wi + j][cob][kg][lane][4]
    const float* ul = a.wwino + lane * 4;
    f32x4 bfr[2][4][2];
    auto b_load = [&](int slot, int chunk, int kg) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int n = 0; n < 2; ++n)
                bfr[slot][j][n] = *reinterpret_cast<const f32x4*>(
                    ul + ((((size_t)chunk * 16 + wi * 4 + j) * ncoblk + min(cob0 + n, ncoblk - 1)) * 2 + kg) * 256);
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

    const int xbuf = g.PX * WLD;
    x_load(0);
    b_load(0, 0, 0);
    x_store(smem, 0, NX);
    __syncthreads();

    for (int chunk = 0; chunk < nchunk; ++chunk) {
        const int nxt = min(chunk + 1, nchunk - 1);
        const float* Xc = smem + (chunk & 1) * xbuf;
        float* Xn = smem + ((chunk + 1) & 1) * xbuf;
        x_load(nxt);
        a_read(0, Xc, 0, 0);
        a_read(1, Xc, 1, 0);
        a_transform(0);
#pragma unroll
        for (int gi = 0; gi < 4; ++gi) {                            // groups (kg, m) = (0,0) (0,1) (1,0) (1,1)
            const int kg = gi >> 1, m = gi & 1;
            if (gi == 0) b_load(1, chunk, 1);
            if (gi == 2) b_load(0, nxt, 0);
            if (gi + 2 < 4) a_read(gi & 1, Xc, gi & 1, 1);          // raw[gi&1] was consumed one region ago
            __builtin_amdgcn_sched_barrier(0);
            // one scheduling region: this group's 32 MFMAs, the NEXT group's fragment transform and a quarter of the
            // next chunk's patch staging (affine + SiLU + LDS write): with one wave per SIMD nothing else can fill
            // the matrix pipe, so the VALU work must sit in the MFMA shadows (each MFMA holds issue for 8 of 64 cycles)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        acc[m][j][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(frag[gi & 1][j][e], bfr[kg][j][n][e], acc[m][j][n], 0, 0, 0);
            if (gi + 1 < 4) a_transform((gi + 1) & 1);
            x_store(Xn, (NX * gi) / 4, (NX * (gi + 1)) / 4);
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // 1 MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);   // up to 5 VALU in its shadow
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);   // a DS write if one is ready
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }

    // ---- output transform.  A^T = [[1,1,1,0],[0,1,-1,-1]].  Wave-local: Z[q] = sum_j M[wi][j] A[j][q]
    // Z exchange layout in LDS: [i 4][q 2][m 2][n 2][reg16/4][lane 64][4]  (128 KB; the X buffers are dead now)
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
    const int p = wi >> 1, q = wi & 1;
    const int n0 = blockIdx.y * 64;
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
            const int co = n0 + n * 32 + lr;
            if (co >= a.Cout) continue;
            const float bv = a.bias ? a.bias[co] : 0.f;
            size_t pix[16];
            bool ok[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int t = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;        // C/D row = tile
                const int tx = t & (TT - 1), ty = (t >> g.tt_log) & (TT - 1), f = t >> (2 * g.tt_log);
                const int nf = f0 + f;
                ok[r] = nf < a.nfr;
                pix[r] = ((size_t)min(nf, a.nfr - 1) * Hl + oy0 + 2 * ty + p) * Wl + ox0 + 2 * tx + q;
            }
            if (a.res) {
                f32x16 rv;
#pragma unroll
                for (int r = 0; r < 16; ++r) rv[r] = a.res[pix[r] * a.res_ld + co];
                y += rv;
            }
            if (a.fbias) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int t = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    y[r] += a.fbias[(size_t)min(f0 + (t >> (2 * g.tt_log)), a.nfr - 1) * a.fbias_ld + co];
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (ok[r]) a.out[pix[r] * a.ldo + co] = y[r] + bv;
        }
}

static bool wino_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

bool conv_wino_supported(const IgemmArgs& a) {
    const int Hl = a.Hs << a.ups, Wl = a.Ws << a.ups;
    return a.wwino != nullptr && a.ksz == 3 && a.stride == 1 && a.pad == 1 && Hl == Wl && wino_pow2(Hl) && Hl >= 8 &&
           a.Cout % 64 == 0 && a.Cin % WKC == 0 && a.C0 % WKC == 0 && (a.affA != nullptr) == (a.act != 0);
}

int launch_conv_wino(const IgemmArgs& a, hipStream_t s) {
    const int Hl = a.Hs << a.ups;
    WinoGeom g;
    const int TT = Hl >= 16 ? 8 : 4;                   // tiles per dim per frame in a block
    g.tt_log = TT == 8 ? 3 : 2;
    g.TF = 64 / (TT * TT);                             // 1 or 4 frames
    g.P = 2 * TT + 2;
    g.PX = g.TF * g.P * g.P;                           // 324 or 400
    g.tiles_x = Hl / (2 * TT); g.tiles_y = Hl / (2 * TT);
    VD_REQUIRE(g.PX * 4 <= (g.TF == 4 ? 7 : 6) * 256, "patch does not fit the per-thread element table");
    const size_t lds = std::max((size_t)2 * g.PX * WLD * sizeof(float), (size_t)4 * 2 * 2 * 2 * 4 * 64 * 4 * sizeof(float));
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
                    out[((((((size_t)ch * 16 + i * 4 + j) * ncoblk + cb) * 2 + kg) * 64) + h * 32 + r) * 4 + e] = (float)U[i][j];
        }
}

}  // namespace vd
