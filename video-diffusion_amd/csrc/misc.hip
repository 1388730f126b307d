// Small HBM-bound kernels around the UNet torso and the per-step posterior update, gfx950.
#include <algorithm>

#include "vd_common.h"

namespace vd {

// ------------------------------------------------------------------ network input (CondMargVideoModel.forward,
// unet.py:951-983,991-1013): 5 channels [x*lat + obs_src*obs + x*(1-any), obs, kinda_marg]; per-frame timesteps;
// attention mask.  The 5-channel 3x3 stem (input_blocks.0.0) has K = 45: far too thin for the implicit-GEMM conv
// kernels (13 TFLOP/s with the channels padded to 32), so the input is written directly as its im2col matrix
// [B*T*H*W][Kpad = 64], k = tap*5 + channel (zeros for taps outside the image and for k >= 45), and the stem runs as a
// plain K = 64 GEMM on gemm_frag_kernel.
__global__ __launch_bounds__(256) void assemble_kernel(AssembleArgs a) {
    const int n = a.frame_list ? a.frame_list[blockIdx.y] : blockIdx.y;      // source frame; the im2col rows go to block blockIdx.y
    const int HW = a.H * a.W;
    const float om = a.obs_mask[n], lm = a.lat_mask[n], km = a.km_mask[n];
    const float any = fminf(om + lm + km, 1.0f);
    const int Cs = a.cond_mode == 0 ? 5 : (a.cond_mode == 1 ? 6 : 3);      // stem input channels (unet.py:932-940)
    if (blockIdx.x == 0 && threadIdx.x == 0 && !a.frame_list) {
        const float t = a.t_model[n / a.T];
        const float tobs = a.obs_t_mode == 0 ? 0.f : (a.obs_t_mode == 1 ? t : t - 1.f);
        // 'channel': observed frames carry the timestep of their source (unet.py:991-1013); 'duplicate' / 'all': every frame
        // keeps t; 't=0' (unet.py:1018-1019): `timesteps[obs_mask == 1] = -1` writes through a (B, 1) -> (B, T) expanded tensor,
        // i.e. ONE stored value per batch item: all frames of an item with any observed frame get -1 (mirrored as computed)
        float tf = t;
        if (a.cond_mode == 0) tf = tobs * om + t * (1.f - om);
        else if (a.cond_mode == 2) {
            bool any_obs = false;
            for (int k = 0; k < a.T; ++k) any_obs |= a.obs_mask[(n / a.T) * a.T + k] == 1.f;
            if (any_obs) tf = -1.f;
        }
        a.t_frames[n] = tf;
        a.amask[n] = any;
    }
    // Block = a strip of up to 64 pixels of one image row.  The three source rows of the strip (+ one column either side) are read ONCE, along
    // x (coalesced), mixed into the Cs stem channels and kept in LDS; an im2col row of Kpad = 64 floats is nine shifted copies of those and
    // leaves as sixteen 16-byte stores of neighbouring lanes.  (First version: thread = (pixel, tap), every tap re-reading x and x0 with
    // scalar loads and Cs scalar stores: 71 us for the 134 MB of the headline window; 16-byte stores alone, the gathers still per lane: 79.)
    if (a.scalars_only) return;
    constexpr int SW = 64;
    __shared__ float tile[3][SW + 2][6];
    __shared__ int lut[256];
    const int strips = (a.W + SW - 1) / SW;
    const int y = blockIdx.x / strips, x0 = (blockIdx.x - y * strips) * SW;
    for (int i = threadIdx.x; i < 9 * (SW + 2); i += 256) {
        const int xx = i % (SW + 2), rc = i / (SW + 2), r = rc % 3, c = rc / 3;
        const int yy = y + r - 1, gx = x0 + xx - 1;
        const bool in = yy >= 0 && yy < a.H && gx >= 0 && gx < a.W;
        float xv = 0.f, ov = 0.f;
        if (in) {
            const size_t q = ((size_t)n * 3 + c) * HW + yy * a.W + gx;
            xv = a.x[q];
            ov = a.obs_src[q];
        }
        if (a.cond_mode == 0) {
            tile[r][xx][c] = in ? xv * lm + ov * om + xv * (1.f - any) : 0.f;
            if (c == 0) { tile[r][xx][3] = in ? om : 0.f; tile[r][xx][4] = in ? km : 0.f; }
        } else if (a.cond_mode == 1) {                                      // obs_src = x0 (unet.py:1014-1017)
            tile[r][xx][c] = in ? xv * lm + xv * (1.f - any) : 0.f;
            tile[r][xx][3 + c] = in ? ov * om : 0.f;
        } else tile[r][xx][c] = xv;
    }
    for (int k = threadIdx.x; k < a.Kpad; k += 256) {                     // column k -> its float in the tile of pixel 0 (or none)
        const int tap = k / Cs, c = k - tap * Cs;
        lut[k] = k < 9 * Cs ? ((tap / 3) * (SW + 2) + tap % 3) * 6 + c : -1;
    }
    __syncthreads();
    const int qpr = a.Kpad >> 2, sh = 31 - __clz(qpr);                      // (Kpad / 4 is a power of two: launch_assemble)
    const float* tf = &tile[0][0][0];
    for (int i = threadIdx.x; i < SW * qpr; i += 256) {
        const int px = i >> sh, j = i & (qpr - 1);
        if (x0 + px >= a.W) continue;
        f32x4 out;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int o = lut[4 * j + e];
            out[e] = o >= 0 ? tf[o + px * 6] : 0.f;
        }
        *reinterpret_cast<f32x4*>(a.x_cols + ((size_t)blockIdx.y * HW + (size_t)y * a.W + x0 + px) * a.Kpad + 4 * j) = out;
    }
}

int launch_assemble(const AssembleArgs& a, hipStream_t s) {
    VD_REQUIRE(a.Kpad >= 64 && a.Kpad <= 256 && (a.Kpad & (a.Kpad - 1)) == 0, "padded im2col width: 64, 128 or 256");
    VD_REQUIRE(!(a.frame_list && a.scalars_only), "assemble: a frame list writes im2col rows only");
    if (a.frame_list && a.n_list == 0) return 0;
    const dim3 grid(a.scalars_only ? 1 : a.H * ((a.W + 63) / 64), a.frame_list ? a.n_list : a.B * a.T);
    hipLaunchKernelGGL(assemble_kernel, grid, dim3(256), 0, s, a);
    VD_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ frame-granular gather / scatter (window prefix cache)
template <bool SCATTER>
__global__ __launch_bounds__(256) void move_rows_kernel(const float* __restrict__ src, const int* __restrict__ list, size_t row4,
                                                        float* __restrict__ dst) {
    const int i = blockIdx.y, f = list[i];
    const f32x4* s4 = reinterpret_cast<const f32x4*>(src) + (size_t)(SCATTER ? i : f) * row4;
    f32x4* d4 = reinterpret_cast<f32x4*>(dst) + (size_t)(SCATTER ? f : i) * row4;
    for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < row4; k += (size_t)gridDim.x * 256) d4[k] = s4[k];
}

static int move_rows(bool scatter, const float* src, const int* list, int n, size_t row_floats, float* dst, hipStream_t s) {
    VD_REQUIRE(row_floats % 4 == 0, "frame rows in 16-byte units");
    if (n == 0) return 0;
    const size_t row4 = row_floats / 4;
    const int bx = (int)std::min<size_t>((row4 + 1023) / 1024, 64);
    if (scatter) hipLaunchKernelGGL(move_rows_kernel<true>, dim3(bx, n), dim3(256), 0, s, src, list, row4, dst);
    else hipLaunchKernelGGL(move_rows_kernel<false>, dim3(bx, n), dim3(256), 0, s, src, list, row4, dst);
    VD_HIP(hipGetLastError());
    return 0;
}
int launch_gather_rows(const float* src, const int* list, int n, size_t row_floats, float* dst, hipStream_t s) {
    return move_rows(false, src, list, n, row_floats, dst, s);
}
int launch_scatter_rows(const float* src, const int* list, int n, size_t row_floats, float* dst, hipStream_t s) {
    return move_rows(true, src, list, n, row_floats, dst, s);
}

// src [n][split][C][2] -> dst [frame][C][2] (one entry per frame and channel: the sum over the producer's pixel ranges, fp64)
__global__ void scatter_stats_kernel(const double* __restrict__ src, int split, int C, const int* __restrict__ list,
                                     double* __restrict__ dst) {
    const int i = blockIdx.x, f = list[i];
    for (int k = threadIdx.x; k < 2 * C; k += blockDim.x) {
        double v = 0;
        for (int sp = 0; sp < split; ++sp) v += src[((size_t)i * split + sp) * 2 * C + k];
        dst[(size_t)f * 2 * C + k] = v;
    }
}
int launch_scatter_stats(const double* src, int split, int C, const int* list, int n, double* dst, hipStream_t s) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(scatter_stats_kernel, dim3(n), dim3(256), 0, s, src, split, C, list, dst);
    VD_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ sinusoidal embedding (nn.py:89-107)
// out[n] = [cos(t*f_i) | sin(t*f_i)], f from a host-built table so the angles match the reference bit for bit.
__global__ void sinus_kernel(const float* __restrict__ t, const float* __restrict__ freqs, int half, int dim,
                             float* __restrict__ out) {
    const int n = blockIdx.x;
    const float tv = t[n];
    for (int i = threadIdx.x; i < half; i += blockDim.x) {
        const float arg = tv * freqs[i];
        out[(size_t)n * dim + i] = cosf(arg);
        out[(size_t)n * dim + half + i] = sinf(arg);
    }
    if ((dim & 1) && threadIdx.x == 0) out[(size_t)n * dim + dim - 1] = 0.f;
}

int launch_sinus_embed(const float* t, int n, int dim, const float* freqs, float* out, hipStream_t s) {
    hipLaunchKernelGGL(sinus_kernel, dim3(n), dim3(64), 0, s, t, freqs, dim / 2, dim, out);
    VD_HIP(hipGetLastError());
    return 0;
}

// frame embedding values (unet.py:914-926): t = fi (- mean over the window when centred)
__global__ void frame_t_kernel(const int64_t* __restrict__ fidx, int T, int center, float* __restrict__ tv) {
    const int b = blockIdx.x;
    float mean = 0.f;
    if (center) {
        float s = 0.f;
        for (int t = 0; t < T; ++t) s += (float)fidx[b * T + t];
        mean = s / (float)T;
    }
    for (int t = threadIdx.x; t < T; t += blockDim.x) tv[b * T + t] = (float)fidx[b * T + t] - mean;
}

int launch_frame_t(const int64_t* fidx, int B, int T, int center, float* tv, hipStream_t s) {
    hipLaunchKernelGGL(frame_t_kernel, dim3(B), dim3(64), 0, s, fidx, T, center, tv);
    VD_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ RPENet hidden layer (unet.py:283-296)
__global__ __launch_bounds__(256) void rpe_hidden_kernel(const float* __restrict__ te, int te_ld,
                                                         const float* __restrict__ Wd, const float* __restrict__ bd,
                                                         const int64_t* __restrict__ fidx, int T, int C,
                                                         float* __restrict__ E, int zs_te, int zs_w, int zs_b, size_t zs_e) {
    te += blockIdx.y * zs_te; Wd += blockIdx.y * zs_w; bd += blockIdx.y * zs_b; E += blockIdx.y * zs_e;
    const int row = blockIdx.x;                 // (b*T + t)*T + s
    const int s_ = row % T, bt = row / T, b = bt / T;
    const float d = (float)(fidx[bt] - fidx[b * T + s_]);
    const float f0 = logf(1.0f + fmaxf(d, 0.f)), f1 = logf(1.0f + fmaxf(-d, 0.f)), f2 = d == 0.f ? 1.f : 0.f;
    for (int c = threadIdx.x; c < C; c += 256) {
        const float lin = bd[c] + f0 * Wd[c * 3] + f1 * Wd[c * 3 + 1] + f2 * Wd[c * 3 + 2];
        E[(size_t)row * C + c] = silu_f(te[(size_t)bt * te_ld + c] + lin);
    }
}

// One block per (batch item, query frame, net): the T key frames of a query frame share the timestep-embedding row, and a thread keeps its
// channel's four constants in registers (one block per (b, t, s) row -- 30 k blocks of 1.5 KB each -- wrote its 47 / 75 MB at 1.6 TB/s: 2 x 38 us).
__global__ __launch_bounds__(256) void rpe_hidden_tab_kernel(const float* __restrict__ te, int te_ld, const float* __restrict__ wbase,
                                                             const long long* __restrict__ tab, const int64_t* __restrict__ fidx, int T, int C,
                                                             float* __restrict__ E, size_t zs_e) {
    __shared__ float feat[64][3];
    const long long* tz = tab + 3 * blockIdx.y;
    te += tz[0];
    const float* Wd = wbase + tz[1];
    const float* bd = wbase + tz[2];
    E += blockIdx.y * zs_e;
    const int bt = blockIdx.x, b = bt / T;      // rows (b*T + t)*T + s, s = 0 .. T-1
    for (int s0 = 0; s0 < T; s0 += 64) {
        const int ns = min(64, T - s0);
        __syncthreads();
        if ((int)threadIdx.x < ns) {
            const float d = (float)(fidx[bt] - fidx[b * T + s0 + threadIdx.x]);
            feat[threadIdx.x][0] = logf(1.0f + fmaxf(d, 0.f));
            feat[threadIdx.x][1] = logf(1.0f + fmaxf(-d, 0.f));
            feat[threadIdx.x][2] = d == 0.f ? 1.f : 0.f;
        }
        __syncthreads();
        for (int c = threadIdx.x; c < C; c += 256) {
            const float bc = bd[c], w0 = Wd[c * 3], w1 = Wd[c * 3 + 1], w2 = Wd[c * 3 + 2], tv = te[(size_t)bt * te_ld + c];
            for (int k = 0; k < ns; ++k) {
                const float lin = bc + feat[k][0] * w0 + feat[k][1] * w1 + feat[k][2] * w2;
                E[((size_t)bt * T + s0 + k) * C + c] = silu_f(tv + lin);
            }
        }
    }
}

int launch_rpe_hidden_tab(const float* te, int te_ld, const float* wbase, const long long* tab, const int64_t* fidx, int B, int T, int C,
                          float* E, int nz, size_t zs_e, hipStream_t s) {
    hipLaunchKernelGGL(rpe_hidden_tab_kernel, dim3(B * T, nz), dim3(256), 0, s, te, te_ld, wbase, tab, fidx, T, C, E, zs_e);
    VD_HIP(hipGetLastError());
    return 0;
}

int launch_rpe_hidden(const float* te, int te_ld, const float* Wd, const float* bd, const int64_t* fidx, int B, int T,
                      int C, float* E, int nz, int zs_te, int zs_w, int zs_b, size_t zs_e, hipStream_t s) {
    hipLaunchKernelGGL(rpe_hidden_kernel, dim3(B * T * T, nz), dim3(256), 0, s, te, te_ld, Wd, bd, fidx, T, C, E, zs_te, zs_w,
                       zs_b, zs_e);
    VD_HIP(hipGetLastError());
    return 0;
}

// bucket-table relative positions (RPE.get_bucket_ids, unet.py:330-347; iRPE eq. 18)
__global__ __launch_bounds__(256) void rpe_table_kernel(const float* __restrict__ table, const int64_t* __restrict__ fidx,
                                                        int T, int C, float alpha, float beta, float gamma, float lg,
                                                        float* __restrict__ R) {
    const int row = blockIdx.x;
    const int s_ = row % T, bt = row / T, b = bt / T;
    const long long d = fidx[bt] - fidx[b * T + s_];
    long long id = d;
    const float ad = fabsf((float)d);
    if (ad > alpha) {
        const float coef = logf(ad / alpha) / lg;
        const float v = fminf(beta, alpha + coef * (beta - alpha));
        id = (long long)(int)v * (d > 0 ? 1 : -1);
    }
    const int nb = 2 * (int)beta + 1;
    if (id < 0) id += nb;                       // negative indices wrap, as torch indexing does
    for (int c = threadIdx.x; c < C; c += 256) R[(size_t)row * C + c] = table[(size_t)id * C + c];
}

int launch_rpe_table(const float* table, const int64_t* fidx, int B, int T, int C, float alpha, float beta,
                     float gamma, float* R, hipStream_t s) {
    const float lg = (float)log((double)gamma / (double)alpha);
    hipLaunchKernelGGL(rpe_table_kernel, dim3(B * T * T), dim3(256), 0, s, table, fidx, T, C, alpha, beta, gamma, lg, R);
    VD_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ h + spatial_encoding (+ frame embedding)
__global__ __launch_bounds__(256) void posenc_kernel(const float* __restrict__ x, const float* __restrict__ P,
                                                     const float* __restrict__ femb, size_t per_frame4, int C4,
                                                     size_t total4, float* __restrict__ y) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (size_t)gridDim.x * 256) {
        f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
        if (P) v += reinterpret_cast<const f32x4*>(P)[i % per_frame4];
        if (femb) v += reinterpret_cast<const f32x4*>(femb)[(i / per_frame4) * C4 + (i % C4)];
        reinterpret_cast<f32x4*>(y)[i] = v;
    }
}

int launch_posenc_add(const float* x, const float* P, const float* femb, int nfr, int HW, int C, float* y,
                      hipStream_t s) {
    const size_t total4 = (size_t)nfr * HW * C / 4;
    const int grid = (int)std::min<size_t>((total4 + 255) / 256, 4096);
    hipLaunchKernelGGL(posenc_kernel, dim3(grid), dim3(256), 0, s, x, P, femb, (size_t)HW * C / 4, C / 4, total4, y);
    VD_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ output head: GN-affine + SiLU + conv3x3 C->Cout (<=4)
// NHWC in, NCHW out (the caller's layout).  16x16 output pixels per block, the activated halo tile
// staged through LDS 32 channels at a time; VALU (3 output channels cannot feed a 32x32 MFMA tile).
constexpr int OT = 16;
template <int NCO>      // accumulators per pixel: 4 (eps: 3 outputs) or 8 (learn_sigma: eps | variance values, 6 outputs)
__global__ __launch_bounds__(256) void out_conv_kernel(const float* __restrict__ x, const float* __restrict__ affA,
                                                       const float* __restrict__ affB, const float* __restrict__ w,
                                                       const float* __restrict__ bias, int H, int W, int C, int Cout,
                                                       float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) float tile[(OT + 2) * (OT + 2) * 36];
    __shared__ __attribute__((aligned(16))) float ws[9 * NCO * 32];
    const int n = blockIdx.z, y0 = blockIdx.y * OT, x0 = blockIdx.x * OT;
    const int tid = threadIdx.x, ty = tid / OT, tx = tid % OT;
    float acc[NCO];
#pragma unroll
    for (int co = 0; co < NCO; ++co) acc[co] = 0.f;
    for (int c0 = 0; c0 < C; c0 += 32) {
        __syncthreads();
        for (int i = tid; i < (OT + 2) * (OT + 2) * 8; i += 256) {
            const int pix = i >> 3, q = i & 7;
            const int iy = y0 + pix / (OT + 2) - 1, ix = x0 + pix % (OT + 2) - 1;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
                v = *reinterpret_cast<const f32x4*>(x + (((size_t)n * H + iy) * W + ix) * C + c0 + q * 4);
                const f32x4 A = *reinterpret_cast<const f32x4*>(affA + (size_t)n * C + c0 + q * 4);
                const f32x4 B = *reinterpret_cast<const f32x4*>(affB + (size_t)n * C + c0 + q * 4);
                v = v * A + B;
                v.x = silu_f(v.x); v.y = silu_f(v.y); v.z = silu_f(v.z); v.w = silu_f(v.w);
            }
            *reinterpret_cast<f32x4*>(tile + pix * 36 + q * 4) = v;
        }
        for (int i = tid; i < 9 * NCO * 32; i += 256) {
            const int tap = i / (NCO * 32), co = (i / 32) % NCO, c = i % 32;
            ws[i] = co < Cout ? w[((size_t)tap * Cout + co) * C + c0 + c] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const float* tp = tile + ((ty + tap / 3) * (OT + 2) + tx + tap % 3) * 36;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(tp + q * 4);
#pragma unroll
                for (int co = 0; co < NCO; ++co) {
                    const f32x4 wv = *reinterpret_cast<const f32x4*>(ws + (tap * NCO + co) * 32 + q * 4);
                    acc[co] += v.x * wv.x + v.y * wv.y + v.z * wv.z + v.w * wv.w;
                }
            }
        }
    }
    const int oy = y0 + ty, ox = x0 + tx;
    if (oy < H && ox < W)
        for (int co = 0; co < Cout; ++co) out[(((size_t)n * Cout + co) * H + oy) * W + ox] = acc[co] + bias[co];
}

int launch_out_conv(const float* x, const float* affA, const float* affB, const float* w, const float* bias, int nfr,
                    int H, int W, int C, int Cout, float* out_nchw, hipStream_t s) {
    VD_REQUIRE(Cout <= 8 && C % 32 == 0, "output head: Cout <= 8, C multiple of 32");
    const dim3 grid((W + OT - 1) / OT, (H + OT - 1) / OT, nfr);
    if (Cout <= 4) hipLaunchKernelGGL(out_conv_kernel<4>, grid, dim3(256), 0, s, x, affA, affB, w, bias, H, W, C, Cout, out_nchw);
    else hipLaunchKernelGGL(out_conv_kernel<8>, grid, dim3(256), 0, s, x, affA, affB, w, bias, H, W, C, Cout, out_nchw);
    VD_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ output head as GEMM + gather (engine.hip: forward)
// ------------------------------------------------------------------ the output head's GEMM (unet.py:744-749,838)
// T[pixel][co * 9 + tap] = silu(h[pixel][:] * A[frame][:] + B[frame][:]) . w[co][:][tap]: the head's 3x3 conv (C -> 3 | 6 channels behind
// GroupNorm + SiLU) as a 1x1 GEMM over its 27 | 54 (cout, tap) columns, followed by out_gather_kernel.  C is 128 and N is 32: no tile of
// the general kernels fits (the generic fp32 kernel ran it at 1.4 TB/s, 185 us of the headline step for 268 MB).  Here: fp32 MFMA
// 32x32x2 (exact fp32 products; the k order is free as long as A and B agree, so one float4 per lane feeds four MFMAs), the weights of a
// column tile resident in registers, a block = 1024 pixels of one frame (its (A, B) in LDS), a wave = 8 tiles of 32 pixels with the next
// tile's rows requested before the current one is multiplied.  Bound by the read of h.
template <int NCT, int CMAX>
__global__ __launch_bounds__(256, 2) void head_gemm_kernel(const float* __restrict__ h, const float* __restrict__ affA, const float* __restrict__ affB,
                                                           const float* __restrict__ Wt, int HW, int C, int TPW, float* __restrict__ T) {
    constexpr int NI = CMAX / 8, LDT = 32 * NCT;
    __shared__ __attribute__((aligned(16))) float sab[2 * CMAX];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, lr = lane & 31, kh = lane >> 5;
    const size_t pix_blk = (size_t)blockIdx.x * (128 * TPW);          // 128 TPW pixels of ONE frame (HW % 1024 == 0; TPW = 8 | 2: tiles per wave)
    const int n = (int)(pix_blk / HW);
    for (int c = tid; c < C; c += 256) { sab[c] = affA[(size_t)n * C + c]; sab[CMAX + c] = affB[(size_t)n * C + c]; }
    const int ni = C >> 3;
    f32x4 w[NCT][NI];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int i = 0; i < NI; ++i)
            w[ct][i] = i < ni ? *reinterpret_cast<const f32x4*>(Wt + (size_t)(ct * 32 + lr) * C + 8 * i + 4 * kh) : f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    const size_t pix0 = pix_blk + (size_t)wv * TPW * 32;
    const float* src = h + (pix0 + lr) * C + 4 * kh;
    f32x4 xa[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) xa[i] = i < ni ? *reinterpret_cast<const f32x4*>(src + 8 * i) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int it = 0; it < TPW; ++it) {
        f32x4 xn[NI];
        const float* nsrc = src + (size_t)(it + 1 < TPW ? it + 1 : it) * 32 * C;
#pragma unroll
        for (int i = 0; i < NI; ++i) xn[i] = i < ni ? *reinterpret_cast<const f32x4*>(nsrc + 8 * i) : f32x4{0.f, 0.f, 0.f, 0.f};
        f32x16 acc[NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            if (i < ni) {
                const f32x4 Aq = *reinterpret_cast<const f32x4*>(sab + 8 * i + 4 * kh), Bq = *reinterpret_cast<const f32x4*>(sab + CMAX + 8 * i + 4 * kh);
                f32x4 v = xa[i] * Aq + Bq;
                v.x = silu_f(v.x); v.y = silu_f(v.y); v.z = silu_f(v.z); v.w = silu_f(v.w);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[e], w[ct][i][e], acc[ct], 0, 0, 0);
            }
        }
        float* dst = T + (pix0 + (size_t)it * 32 + 4 * kh) * LDT + lr;
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) dst[(size_t)((r & 3) + 8 * (r >> 2)) * LDT + ct * 32] = acc[ct][r];
#pragma unroll
        for (int i = 0; i < NI; ++i) xa[i] = xn[i];
    }
}

// (32 columns: the 3-channel head; the 6-channel head of a learn_sigma network needs twice the weight registers and stays on the generic kernel)
bool head_gemm_supported(int HW, int C, int ldt) { return HW % 1024 == 0 && C % 8 == 0 && C <= 128 && ldt == 32; }

int launch_head_gemm(const float* h, const float* affA, const float* affB, const float* Wt, int nfr, int HW, int C, int ldt, float* T, hipStream_t s) {
    VD_REQUIRE(head_gemm_supported(HW, C, ldt), "output head GEMM: whole 1024-pixel blocks of a frame, C <= 128, 32 columns");
    // a small window (one rank's share of a strong-scaling split: 16 frames = 64 blocks of 1024 pixels) takes 256-pixel blocks: head 58 -> 29 us there
    const int tpw = (size_t)nfr * HW / 1024 < 512 ? 2 : 8;
    const dim3 grid((unsigned)((size_t)nfr * HW / (128 * tpw)));
    hipLaunchKernelGGL((head_gemm_kernel<1, 128>), grid, dim3(256), 0, s, h, affA, affB, Wt, HW, C, tpw, T);
    VD_HIP(hipGetLastError());
    return 0;
}

// T[pixel][co * 9 + tap] holds what tap `tap` of output channel co contributes FROM this pixel; output pixel (y, x) sums the entry of
// tap (dy, dx) at pixel (y + dy - 1, x + dx - 1).  Block = 16 x 16 output pixels: the 18 x 18 halo of T rows is staged through LDS
// with coalesced 16-byte loads (rows padded to an odd number of quads: lanes of a row walk different banks), then 9 * Cout LDS reads
// per pixel.  NCHW out.
template <int LDT>
__global__ __launch_bounds__(256) void out_gather_kernel(const float* __restrict__ T, const float* __restrict__ bias, int H, int W, int Cout,
                                                         float* __restrict__ out) {
    constexpr int ROW = LDT + 4;                                      // floats per staged pixel
    __shared__ __attribute__((aligned(16))) float tile[(OT + 2) * (OT + 2) * ROW];
    const int n = blockIdx.z, y0 = blockIdx.y * OT, x0 = blockIdx.x * OT;
    const int tid = threadIdx.x, ty = tid / OT, tx = tid % OT;
    for (int i = tid; i < (OT + 2) * (OT + 2) * (LDT / 4); i += 256) {
        const int pix = i / (LDT / 4), q = i - pix * (LDT / 4);
        const int iy = y0 + pix / (OT + 2) - 1, ix = x0 + pix % (OT + 2) - 1;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = *reinterpret_cast<const f32x4*>(T + (((size_t)n * H + iy) * W + ix) * LDT + q * 4);
        *reinterpret_cast<f32x4*>(tile + pix * ROW + q * 4) = v;
    }
    __syncthreads();
    const int oy = y0 + ty, ox = x0 + tx;
    if (oy >= H || ox >= W) return;
    for (int co = 0; co < Cout; ++co) {
        float acc = bias[co];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) acc += tile[((ty + tap / 3) * (OT + 2) + tx + tap % 3) * ROW + co * 9 + tap];
        out[(((size_t)n * Cout + co) * H + oy) * W + ox] = acc;
    }
}

int launch_out_gather(const float* T, const float* bias, int nfr, int H, int W, int ldt, int Cout, float* out_nchw, hipStream_t s) {
    VD_REQUIRE((ldt == 32 || ldt == 64) && 9 * Cout <= ldt, "output head: 3 or 6 output channels");
    const dim3 grid((W + OT - 1) / OT, (H + OT - 1) / OT, nfr);
    if (ldt == 32) hipLaunchKernelGGL(out_gather_kernel<32>, grid, dim3(256), 0, s, T, bias, H, W, Cout, out_nchw);
    else hipLaunchKernelGGL(out_gather_kernel<64>, grid, dim3(256), 0, s, T, bias, H, W, Cout, out_nchw);
    VD_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ Philox4x32-10 + Box-Muller
__device__ __forceinline__ void philox4x32_10(unsigned long long ctr, unsigned long long key, unsigned (&o)[4]) {
    unsigned c0 = (unsigned)ctr, c1 = (unsigned)(ctr >> 32), c2 = 0x5eed5eedu, c3 = 0;
    unsigned k0 = (unsigned)key, k1 = (unsigned)(key >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0;
        const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1;
        const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}

// element i -> standard normal; elements 4j..4j+3 share one Philox block
__device__ __forceinline__ float normal_at(unsigned long long seed, unsigned long long offset, unsigned long long i) {
    unsigned r[4];
    philox4x32_10(offset + (i >> 2), seed, r);
    const int pair = (int)(i & 2);
    const float u1 = ((float)r[pair] + 1.0f) * 2.3283064365386963e-10f;       // (0,1]
    const float u2 = (float)r[pair + 1] * 2.3283064365386963e-10f;
    const float rad = sqrtf(-2.0f * logf(u1));
    float sn, cs;
    sincosf(6.283185307179586f * u2, &sn, &cs);
    return (i & 1) ? rad * sn : rad * cs;
}

__global__ __launch_bounds__(256) void randn_kernel(float* out, size_t n, unsigned long long seed,
                                                    unsigned long long offset) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        out[i] = normal_at(seed, offset, i);
}

int launch_randn(float* out, long n, unsigned long long seed, unsigned long long offset, hipStream_t s) {
    const int grid = (int)std::min<long>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(randn_kernel, dim3(grid), dim3(256), 0, s, out, (size_t)n, seed, offset);
    VD_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ posterior update (one fused pass)
// p_sample (gaussian_diffusion.py:319-343,374-382,208-227,438-443) and ddim_sample (:597-634).
// Coefficients are float32 casts of the float64 tables, exactly what _extract_into_tensor yields.
__global__ __launch_bounds__(256) void posterior_kernel(PosteriorArgs a) {
    const size_t total = (size_t)a.B * a.per;
    if (a.dstate) { a.seed = a.dstate[0]; a.offset = a.dstate[1]; }
    bool nonfinite = false;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int b = (int)(i / a.per);
        const long long tl = a.t[b];
        const int NT = a.num_timesteps;
        if (tl < 0 || tl >= NT) {                // the reference raises IndexError (_extract_into_tensor); no table read here:
            if (a.sample) a.sample[i] = __builtin_nanf("");   // the element is poisoned and map_t_kernel has set the error flag
            if (a.xstart) a.xstart[i] = __builtin_nanf("");
            if (a.mean) a.mean[i] = __builtin_nanf("");
            continue;
        }
        const int t = (int)tl;
        const float* tb = a.tab + t;
        const float x = a.x[i];
        float x0 = a.x0_given ? a.x0_given[i] : tb[TAB_SQRT_RECIP * NT] * x - tb[TAB_SQRT_RECIPM1 * NT] * a.eps[i];
        // A network output that is not finite -- in the f16x3 arithmetic: an operand beyond fp16's range (|x| > 65504) anywhere in the
        // network -- must not leave through the clamp below as a plausible -1 (fmaxf(NaN, -1) = -1): it stays NaN and sets bit 1.
        const bool bad = !(fabsf(x0) <= 3.4028234e38f);
        nonfinite |= bad;
        if (a.clip && !bad) x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
        if (a.xstart) a.xstart[i] = x0;
        if (a.mean) a.mean[i] = tb[TAB_COEF1 * NT] * x0 + tb[TAB_COEF2 * NT] * x;
        if (!a.sample) continue;                 // p_mean_variance only
        const float z = a.noise ? a.noise[i] : normal_at(a.seed, a.offset, i);
        const float nz = t != 0 ? 1.0f : 0.0f;
        float smp;
        if (a.mode == 0) {
            const float mean = tb[TAB_COEF1 * NT] * x0 + tb[TAB_COEF2 * NT] * x;
            smp = mean + nz * expf(0.5f * tb[TAB_LOGVAR * NT]) * z;
        } else {
            const float e2 = (tb[TAB_SQRT_RECIP * NT] * x - x0) / tb[TAB_SQRT_RECIPM1 * NT];
            const float ab = tb[TAB_ACP * NT], abp = tb[TAB_ACP_PREV * NT];
            const float sigma = a.eta * sqrtf((1.f - abp) / (1.f - ab)) * sqrtf(1.f - ab / abp);
            const float mean = x0 * sqrtf(abp) + sqrtf(1.f - abp - sigma * sigma) * e2;
            smp = mean + nz * sigma * z;
        }
        a.sample[i] = smp;
    }
    if (nonfinite && a.err) atomicOr(a.err, VD_ERR_NONFINITE);     // (never taken on a healthy step: no cost beside the compare above)
}

int launch_posterior(const PosteriorArgs& a, hipStream_t s) {
    const size_t total = (size_t)a.B * a.per;
    const int grid = (int)std::min<size_t>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(posterior_kernel, dim3(grid), dim3(256), 0, s, a);
    VD_HIP(hipGetLastError());
    return 0;
}

// q_sample (gaussian_diffusion.py:190-206)
__global__ __launch_bounds__(256) void q_sample_kernel(const float* x0, const float* noise, const int64_t* t,
                                                       const float* tab, int NT, size_t per, size_t total, float* out) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        long long tt = t[i / per];
        if (tt < 0) tt += NT;                   // t-1 at t=0 wraps like numpy/torch indexing (gaussian_diffusion.py:565-568)
        if (tt < 0 || tt >= NT) { out[i] = __builtin_nanf(""); continue; }       // IndexError in the reference
        out[i] = fmaf(tab[TAB_SQRT_1M_ACP * NT + tt], noise[i], tab[TAB_SQRT_ACP * NT + tt] * x0[i]);   // (one rounding order for this kernel and q_sample_prev_kernel)
    }
}

int launch_q_sample(const float* x0, const float* noise, const int64_t* t, const float* tab, int num_timesteps, int B,
                    long per, float* out, hipStream_t s) {
    const size_t total = (size_t)B * per;
    const int grid = (int)std::min<size_t>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(q_sample_kernel, dim3(grid), dim3(256), 0, s, x0, noise, t, tab, num_timesteps, (size_t)per,
                       total, out);
    VD_HIP(hipGetLastError());
    return 0;
}

// The window executor's `observed_frames = 'x_t_minus_1'` (gaussian_diffusion.py:565-568: x_t_minus_1 = q_sample(x0, t - 1, fresh
// noise) before every step of p_sample_loop): t and the Philox {seed, offset} are read from device memory, the noise is
// element i of the stream at offset + draw_offset, index t - 1 = -1 wraps to the last entry like the reference's negative index.
__global__ __launch_bounds__(256) void q_sample_prev_kernel(const float* x0, const long long* t, const float* tab, int NT, size_t per, size_t total,
                                                            const unsigned long long* dstate, unsigned long long draw_offset, float* out) {
    const unsigned long long seed = dstate[0], offset = dstate[1] + draw_offset;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        long long tt = t[i / per] - 1;
        if (tt < 0) tt += NT;
        if (tt < 0 || tt >= NT) { out[i] = __builtin_nanf(""); continue; }
        out[i] = fmaf(tab[TAB_SQRT_1M_ACP * NT + tt], normal_at(seed, offset, i), tab[TAB_SQRT_ACP * NT + tt] * x0[i]);
    }
}

int launch_q_sample_prev(const float* x0, const int64_t* t, const float* tab, int num_timesteps, int B, long per, const unsigned long long* dstate,
                         unsigned long long draw_offset, float* out, hipStream_t s) {
    const size_t total = (size_t)B * per;
    const int grid = (int)std::min<size_t>((total + 255) / 256, 4096);
    hipLaunchKernelGGL(q_sample_prev_kernel, dim3(grid), dim3(256), 0, s, x0, reinterpret_cast<const long long*>(t), tab, num_timesteps, (size_t)per,
                       total, dstate, draw_offset, out);
    VD_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ NLL terms (gaussian_diffusion.py:750-790, losses.py)
__device__ __forceinline__ float approx_std_normal_cdf(float x) {          // losses.py:38-43
    return 0.5f * (1.0f + tanhf(0.7978845608028654f * (x + 0.044715f * x * x * x)));
}

// grid (nblk, B): block (k, b) sums its slice of batch element b; thread-local and cross-thread sums in fp64, one partial
// triple per block (deterministic: no atomics), folded by vb_final_kernel.
__global__ __launch_bounds__(256) void vb_terms_kernel(VbArgs a) {
    const int b = blockIdx.y;
    const long long tl = a.t[b];
    const int NT = a.num_timesteps;
    const bool ok = tl >= 0 && tl < NT;
    const int t = ok ? (int)tl : 0;
    const float* tb = a.tab + t;
    const float sr = tb[TAB_SQRT_RECIP * NT], srm1 = tb[TAB_SQRT_RECIPM1 * NT], c1 = tb[TAB_COEF1 * NT], c2 = tb[TAB_COEF2 * NT];
    const float lv = tb[TAB_LOGVAR * NT], tlv = tb[TAB_POST_LOGVAR * NT];
    const float inv_stdv = expf(-0.5f * lv), e_dlv = expf(tlv - lv), e_mlv = expf(-lv);
    const long fsz = a.per / a.T;
    double s0 = 0, s1 = 0, s2 = 0;
    for (long j = (long)blockIdx.x * 256 + threadIdx.x; j < a.per; j += (long)gridDim.x * 256) {
        const size_t i = (size_t)b * a.per + j;
        const float xs = a.x_start[i], xt = a.x_t[i];
        float x0 = a.start_x ? a.eps[i] : sr * xt - srm1 * a.eps[i];  // START_X: pred_xstart = process_xstart(model_output) (gaussian_diffusion.py:326-341)
        const bool bad = !(fabsf(x0) <= 3.4028234e38f);               // (posterior_kernel: a non-finite eps must not be clamped into range)
        if (bad && a.err) atomicOr(a.err, VD_ERR_NONFINITE);
        if (a.clip && !bad) x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
        if (a.pred_xstart) a.pred_xstart[i] = ok ? x0 : __builtin_nanf("");
        const float mean = c1 * x0 + c2 * xt, tmean = c1 * xs + c2 * xt;
        const float m = a.mask ? a.mask[(size_t)b * a.T + j / fsz] : 1.0f;
        float term;
        if (t == 0) {                                                      // decoder NLL (losses.py:46-76)
            const float cx = xs - mean;
            const float cdf_plus = approx_std_normal_cdf(inv_stdv * (cx + 1.0f / 255.0f));
            const float cdf_min = approx_std_normal_cdf(inv_stdv * (cx - 1.0f / 255.0f));
            const float lp = xs < -0.999f ? logf(fmaxf(cdf_plus, 1e-12f))
                           : xs > 0.999f ? logf(fmaxf(1.0f - cdf_min, 1e-12f)) : logf(fmaxf(cdf_plus - cdf_min, 1e-12f));
            term = -lp;
        } else {                                                           // KL(q(x_{t-1}|x_t,x_0) || p(x_{t-1}|x_t)) (losses.py:13-35)
            const float d = tmean - mean;
            term = 0.5f * (-1.0f + lv - tlv + e_dlv + d * d * e_mlv);
        }
        s0 += (double)(term * m);
        const float dx = x0 - xs;
        s1 += (double)(dx * dx * m);
        if (a.noise) {
            const float e2 = (sr * xt - x0) / srm1 - a.noise[i];           // _predict_eps_from_xstart (gaussian_diffusion.py:392-396)
            s2 += (double)(e2 * e2 * m);
        }
    }
    __shared__ double red[3][256];
    red[0][threadIdx.x] = s0; red[1][threadIdx.x] = s1; red[2][threadIdx.x] = s2;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o)
            for (int k = 0; k < 3; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x < 3) a.part[((size_t)b * gridDim.x + blockIdx.x) * 3 + threadIdx.x] = ok ? red[threadIdx.x][0] : __builtin_nan("");
}

__global__ void vb_final_kernel(const double* part, int nblk, double per, float* vb, float* xstart_mse, float* mse, float scale0) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= (int)gridDim.x * (int)blockDim.x) return;
    double s[3] = {0, 0, 0};
    for (int k = 0; k < nblk; ++k)
        for (int q = 0; q < 3; ++q) s[q] += part[((size_t)b * nblk + k) * 3 + q];
    if (vb) vb[b] = (float)(s[0] / per) * scale0;                           // mean_flat(...) / ln 2
    if (xstart_mse) xstart_mse[b] = (float)(s[1] / per);
    if (mse) mse[b] = (float)(s[2] / per);
}

int vb_terms_blocks(long per) { return (int)std::max<long>(1, std::min<long>(64, (per + 256 * 16 - 1) / (256 * 16))); }

int launch_vb_terms(const VbArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(vb_terms_kernel, dim3(a.nblk, a.B), dim3(256), 0, s, a);
    VD_HIP(hipGetLastError());
    hipLaunchKernelGGL(vb_final_kernel, dim3(a.B), dim3(1), 0, s, a.part, a.nblk, (double)a.per, a.vb, a.xstart_mse,
                       a.noise ? a.mse : nullptr, 1.4426950408889634f);
    VD_HIP(hipGetLastError());
    return 0;
}

__global__ __launch_bounds__(256) void prior_bpd_kernel(const float* x_start, const float* mask, const float* tab, int NT, int T,
                                                        long per, double* part) {
    const int b = blockIdx.y;
    const float sa = tab[TAB_SQRT_ACP * NT + NT - 1], lv = tab[TAB_LOG_1M_ACP * NT + NT - 1];
    const float ev = expf(lv);
    const long fsz = per / T;
    double s0 = 0;
    for (long j = (long)blockIdx.x * 256 + threadIdx.x; j < per; j += (long)gridDim.x * 256) {
        const float mu = sa * x_start[(size_t)b * per + j];
        const float m = mask ? mask[(size_t)b * T + j / fsz] : 1.0f;
        s0 += (double)(0.5f * (-1.0f - lv + ev + mu * mu) * m);              // normal_kl(mean, logvar, 0, 0)
    }
    __shared__ double red[256];
    red[threadIdx.x] = s0;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        double* p = part + ((size_t)b * gridDim.x + blockIdx.x) * 3;
        p[0] = red[0]; p[1] = 0; p[2] = 0;
    }
}

int launch_prior_bpd(const float* x_start, const float* mask, const float* tab, int num_timesteps, int B, int T, long per,
                     double* part, int nblk, float* out, hipStream_t s) {
    hipLaunchKernelGGL(prior_bpd_kernel, dim3(nblk, B), dim3(256), 0, s, x_start, mask, tab, num_timesteps, T, per, part);
    VD_HIP(hipGetLastError());
    hipLaunchKernelGGL(vb_final_kernel, dim3(B), dim3(1), 0, s, part, nblk, (double)per, out, nullptr, nullptr, 1.4426950408889634f);
    VD_HIP(hipGetLastError());
    return 0;
}

}  // namespace vd
