// GroupNorm(32 groups, eps 1e-5) for NHWC activations, gfx950.  HBM-bound kernels.
//
// The reference normalises, applies SiLU and (in ResBlocks) FiLM scale/shift as separate ATen ops
// (nn.py:15-17, unet.py:185-198): 67 GroupNorm + 101 SiLU launches and ~28 GB of traffic per step at
// B=8.  Here only the *statistics* pass reads the tensor; the normalise/scale/shift/
// activation is folded into a per-(frame, channel) affine pair consumed by the next convolution's
// operand load (igemm.hip).  Sums are accumulated in fp64, so E[x^2]-E[x]^2 is safe.
#include <cstdlib>

#include "vd_common.h"

namespace vd {

// ------------------------------------------------------------------ per-channel partial sums
// grid (split, nfr); thread -> fixed channel quad (C/4 lanes per pixel), PPI pixels per iteration.
__global__ __launch_bounds__(256) void gn_stats_partial(const float* __restrict__ src0, const float* __restrict__ src1,
                                                        int C0, int C, int HW, int split, double* __restrict__ part) {
    const int n = blockIdx.y, sp = blockIdx.x;
    const int tpp = C >> 2;                 // threads per pixel
    const int ppi = 256 / tpp;              // pixels per iteration (>=1, C <= 1024)
    const int tid = threadIdx.x;
    const int pl = tid / tpp, q = tid - pl * tpp;
    const int per = (HW + split - 1) / split;
    const int p_begin = sp * per, p_end = min(HW, p_begin + per);
    const int c = q * 4;
    const float* base; int cc, ld;
    if (c < C0) { base = src0; cc = c; ld = C0; } else { base = src1; cc = c - C0; ld = C - C0; }
    double s[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
    if (pl < ppi) {
        // four independent 16-byte loads in flight per thread (one per iteration left the kernel latency-bound at
        // ~3 TB/s); every term goes through fp64 (a common offset 1e4 times the spread must not destroy
        // E[x^2] - E[x]^2: tests/test_gpu_ops.py::test_groupnorm_fold_large_mean)
        const float* src = base + (size_t)n * HW * ld + cc;
        int p = p_begin + pl;
        for (; p + 3 * ppi < p_end; p += 4 * ppi) {
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f32x4*>(src + (size_t)(p + u * ppi) * ld);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int u = 0; u < 4; ++u) { const double d = (double)v[u][e]; s[e] += d; ss[e] += d * d; }
            }
        }
        for (; p < p_end; p += ppi) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(src + (size_t)p * ld);
#pragma unroll
            for (int e = 0; e < 4; ++e) { s[e] += v[e]; ss[e] += (double)v[e] * v[e]; }
        }
    }
    __shared__ double red[256 * 8];
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[e * 256 + tid] = s[e]; red[(4 + e) * 256 + tid] = ss[e]; }      // one plane per sum: conflict-free
    __syncthreads();
    if (tid < tpp) {
        double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int k = 0; k < ppi; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] += red[e * 256 + k * tpp + tid];
        double* o = part + (((size_t)n * split + sp) * C + tid * 4) * 2;
#pragma unroll
        for (int e = 0; e < 4; ++e) { o[e * 2] = a[e]; o[e * 2 + 1] = a[4 + e]; }
    }
}

int gn_stats_split(int nfr, int HW, int C) {
    const int ppi = 256 / (C / 4);
    int split = 1;
    // enough blocks to fill the chip, at least ~8 iterations of work per block
    while (nfr * split < 2048 && HW / (split * 2) >= ppi * 8) split *= 2;
    return split;
}

int launch_gn_stats(const float* src0, const float* src1, int C0, int C, int nfr, int HW, double* part, int split,
                    hipStream_t s) {
    VD_REQUIRE(C % 32 == 0 && C <= 1024 && C0 % 4 == 0, "GroupNorm32 channel constraints");
    hipLaunchKernelGGL(gn_stats_partial, dim3(split, nfr), dim3(256), 0, s, src0, src1, C0, C, HW, split, part);
    VD_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ group statistics -> per-(frame,channel) affine
//   y = ((x-mean)*rstd*gamma + beta) * (1+scale) + shift  =  x*A + B
// The per-channel partial sums come from up to two tables (the two halves of a virtual concat: channels [0, C0) from
// part0 with split0 entries per frame, [C0, C) from part1 with split1), each written either by gn_stats_partial or by
// the epilogue of the convolution that produced the tensor (conv_wino.hip).  One block per frame: 8 lanes per group
// reduce the fp64 partials with shuffles, then all 256 threads fold mean/rstd/gamma/beta and the FiLM pair into A, B.
__global__ __launch_bounds__(256) void gn_final_affine_kernel(const double* __restrict__ part0, int split0, int C0,
                                                              const double* __restrict__ part1, int split1, double count,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta,
                                                              const float* __restrict__ film, int film_ld, int C,
                                                              float* __restrict__ affA, float* __restrict__ affB, float* __restrict__ mr_out) {
    __shared__ float mr[64];
    const int n = blockIdx.x, tid = threadIdx.x;
    const int g = tid >> 3, l = tid & 7;
    const int cg = C / 32, C1 = C - C0;
    // the group's cg x split (sum, sum of squares) pairs dealt to its 8 lanes, FOUR requests in flight per lane: the kernel is nothing but
    // dependent round trips to L2 (one pair per trip and lane: 10 us per launch at 16 pixel ranges, 24 launches per step)
    typedef double f64x2 __attribute__((ext_vector_type(2)));
    const int smax = max(split0, split1), items = cg * smax;
    auto fetch = [&](int it) -> f64x2 {
        const int ci = it / smax, sp = it - ci * smax, c = g * cg + ci;
        const bool second = c >= C0;
        const int split = second ? split1 : split0, ld = second ? C1 : C0, cc = second ? c - C0 : c;
        if (it >= items || sp >= split) return f64x2{0.0, 0.0};
        return *reinterpret_cast<const f64x2*>((second ? part1 : part0) + (((size_t)n * split + sp) * ld + cc) * 2);
    };
    f64x2 acc[4] = {{0.0, 0.0}, {0.0, 0.0}, {0.0, 0.0}, {0.0, 0.0}};
    for (int it = l; it < items; it += 32) {
        f64x2 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = fetch(it + 8 * u);
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] += v[u];
    }
    double s = (acc[0].x + acc[1].x) + (acc[2].x + acc[3].x), ss = (acc[0].y + acc[1].y) + (acc[2].y + acc[3].y);
#pragma unroll
    for (int o = 4; o > 0; o >>= 1) { s += __shfl_xor(s, o, 8); ss += __shfl_xor(ss, o, 8); }
    if (l == 0) {
        const double mean = s / count;
        double var = ss / count - mean * mean;
        if (var < 0) var = 0;
        mr[g * 2] = (float)mean;
        mr[g * 2 + 1] = (float)(1.0 / sqrt(var + 1e-5));
        if (mr_out) { mr_out[((size_t)n * 32 + g) * 2] = mr[g * 2]; mr_out[((size_t)n * 32 + g) * 2 + 1] = mr[g * 2 + 1]; }
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        const int gg = c / cg;
        float A = mr[gg * 2 + 1] * gamma[c];
        float B = beta[c] - mr[gg * 2] * A;
        if (film) {
            const float sc = 1.0f + film[(size_t)n * film_ld + c];
            const float sh = film[(size_t)n * film_ld + C + c];
            A *= sc;
            B = B * sc + sh;
        }
        affA[(size_t)n * C + c] = A;
        affB[(size_t)n * C + c] = B;
    }
}

int launch_gn_affine(const double* part0, int split0, int C0, const double* part1, int split1, double count,
                     const float* gamma, const float* beta, const float* film, int film_ld, int nfr, int C, float* affA,
                     float* affB, hipStream_t s, float* mr_out) {
    VD_REQUIRE(part0 && (C0 == C || part1), "GroupNorm partial tables");
    hipLaunchKernelGGL(gn_final_affine_kernel, dim3(nfr), dim3(256), 0, s, part0, split0, C0, part1, split1, count, gamma,
                       beta, film, film_ld, C, affA, affB, mr_out);
    VD_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ y = silu(x*A[n][c] + B[n][c]) over a virtual concat
// GroupNorm(+FiLM)+SiLU materialised once for the 3x3 convs.  The conv kernels can apply the same affine + SiLU in their
// operand load, but every block that touches a pixel then repeats it (Cout/64 column blocks x the halo overlap: 2.5x to
// 12x), and on gfx950 VALU work does not hide behind fp32 MFMAs (tools/mfma_peak.hip): one HBM-bound pass is cheaper.
// grid (pixel ranges, frames); thread -> a fixed channel quad of the (virtually concatenated) tensor and every ppi-th pixel
// of its block's range, so the source pointer, the concat side and the (A, B) pair are per-thread constants: the loop is
// one 16-byte load, 4 fma + 4 SiLU, one 16-byte store.  (The first version walked a flat index with two 64-bit divisions
// per element: ~108 VALU per 16 bytes, which rocprofv3 showed to be the bound -- 5.3 TB/s -- not the memory.)
__global__ __launch_bounds__(256) void affine_act_kernel(const float* __restrict__ src0, const float* __restrict__ src1,
                                                         int C0, int C, const float* __restrict__ affA,
                                                         const float* __restrict__ affB, int HW, int per, int act,
                                                         float* __restrict__ y) {
    const int n = blockIdx.y;
    const int tpp = C >> 2, ppi = blockDim.x / tpp;       // blockDim.x is a multiple of tpp (launch_affine_act)
    const int tid = threadIdx.x;
    const int pl = tid / tpp, c = (tid - pl * tpp) * 4;
    const int p_begin = blockIdx.x * per, p_end = min(HW, p_begin + per);
    const float* src; int ld;
    if (c < C0) { src = src0 + (size_t)n * HW * C0 + c; ld = C0; } else { src = src1 + (size_t)n * HW * (C - C0) + (c - C0); ld = C - C0; }
    float* dst = y + (size_t)n * HW * C + c;
    const f32x4 A = *reinterpret_cast<const f32x4*>(affA + (size_t)n * C + c);
    const f32x4 B = *reinterpret_cast<const f32x4*>(affB + (size_t)n * C + c);
    auto one = [&](f32x4 v) {
        f32x4 r = v * A + B;
        if (act) { r.x = silu_f(r.x); r.y = silu_f(r.y); r.z = silu_f(r.z); r.w = silu_f(r.w); }
        return r;
    };
#ifndef VD_AA_NT
#define VD_AA_NT 1          // bit 0 nontemporal loads (the source is read once: 5.41 -> 5.79 TB/s), bit 1 nontemporal stores (4.93)
#endif
    auto ld4 = [&](const float* q) { return (VD_AA_NT & 1) ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(q)) : *reinterpret_cast<const f32x4*>(q); };
    auto st4 = [&](float* q, f32x4 v) { if (VD_AA_NT & 2) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(q)); else *reinterpret_cast<f32x4*>(q) = v; };
    int p = p_begin + pl;
    for (; p + 3 * ppi < p_end; p += 4 * ppi) {           // four loads in flight
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = ld4(src + (size_t)(p + u * ppi) * ld);
#pragma unroll
        for (int u = 0; u < 4; ++u) st4(dst + (size_t)(p + u * ppi) * C, one(v[u]));
    }
    for (; p < p_end; p += ppi) *reinterpret_cast<f32x4*>(dst + (size_t)p * C) = one(*reinterpret_cast<const f32x4*>(src + (size_t)p * ld));
}

// Blocks of the activation pass: pixel ranges of a frame, at least `aa_min_iters` trips of four 16-byte loads per thread, until the grid has
// `aa_target_blocks` blocks.  tools/probes/stream_probe.hip (r05f, 128 ch x 64 x 64 x 128 frames, read + write): 2048 blocks of 8 trips
// 5.24 TB/s, 8192 blocks of 2 trips 5.95 TB/s (a flat one-trip mapping 6.16; hipMemcpy 5.43) -- many short blocks keep more requests in
// flight across the tail of each wave than few long ones.  (VD_AA_BLOCKS / VD_AA_ITERS: A/B knobs, read once.)
static int aa_target_blocks() { static const int v = getenv("VD_AA_BLOCKS") ? atoi(getenv("VD_AA_BLOCKS")) : 8192; return v; }
static int aa_min_iters() { static const int v = getenv("VD_AA_ITERS") ? atoi(getenv("VD_AA_ITERS")) : 1; return v; }

int launch_affine_act(const float* src0, const float* src1, int C0, int C, const float* affA, const float* affB, int nfr,
                      int HW, int act, float* y, hipStream_t s) {
    VD_REQUIRE(C % 4 == 0 && C0 % 4 == 0 && C <= 1024 && (src1 != nullptr || C0 == C), "affine_act: channel counts");
    const int tpp = C / 4, ppi = 256 / tpp, threads = ppi * tpp;
    // pixel ranges: enough blocks to fill the chip (>= 2048), at least 4 iterations of 4 loads per block where the frame allows
    int split = 1;
    while (nfr * split < aa_target_blocks() && HW / (split * 2) >= ppi * aa_min_iters() * 4) split *= 2;
    const int per = (HW + split - 1) / split;
    hipLaunchKernelGGL(affine_act_kernel, dim3(split, nfr), dim3(threads), 0, s, src0, src1, C0, C, affA, affB, HW, per, act, y);
    VD_HIP(hipGetLastError());
    return 0;
}

static int fold_target_blocks() { static const int v = getenv("VD_FOLD_BLOCKS") ? atoi(getenv("VD_FOLD_BLOCKS")) : 2048; return v; }   // A/B knobs, read once
static int fold_min_rows() { static const int v = getenv("VD_FOLD_ROWS") ? atoi(getenv("VD_FOLD_ROWS")) : 16; return v; }

// ------------------------------------------------------------------ the fold inside the pass
// gn_final_affine_kernel is 7.5 us of dependent round trips per launch whatever the size (56 launches per step, 0.42 ms: the
// largest item of the step that is not work).  Here every block of the activation pass folds the statistics of ITS frame
// itself -- the per-channel sums of the producer's pixel ranges (thread (pixel lane, quad) walks the ranges pl, pl + ppi, ..),
// summed over the pixel lanes in a fixed order, then the group's cg channels -- and forms (A, B) of its channel quad with the
// formulas of gn_final_affine_kernel.  The table of a frame is 2 * split * C doubles from L2; the blocks of a CU overlap
// each other's fold.  The backward pass (tape) and the consumers that read (A, B) as arrays keep the two-launch form.
__global__ __launch_bounds__(256) void affine_act_fold_kernel(const float* __restrict__ src0, const float* __restrict__ src1,
                                                              int C0, int C, GnFold f, int HW, int per, int act,
                                                              float* __restrict__ y) {
    // LDS (dynamic, sized by C: the blocks of a CU fold side by side, and 32 KB of static arrays allowed five of them):
    //   red [8 sums][thread]   the threads' partial sums over the producer's pixel ranges -- one PLANE per sum, so that a wave's
    //                          64 stores / loads of a plane are 64 consecutive doubles (round 5's [thread][8] layout put every fourth lane
    //                          on the same bank: rocprofv3 counted bank-conflict cycles for 0.7 of this kernel's LDS cycles)
    //   gs  [2][C]             per-channel (sum | sum of squares) of the frame, two planes
    //   mr  [32][2] floats     per-group (mean, rstd): formed ONCE per block by one lane per group.  Round 5 had every thread fold the groups
    //                          of its four channels itself: 8 cg conflicting ds_read_b64 and four fp64 divisions + square roots per thread,
    //                          256 times the same 32 results -- the fold, not HBM, set the time of the small tensors (3.5 TB/s).
    // Same sums in the same order as round 5's kernel (pixel lanes k ascending, then the group's channels ascending), same formulas: its bits.
    // Against gn_final_affine_kernel's (A, B): same formulas, fp64 sums in another order (that kernel deals the pairs to eight lanes).
    extern __shared__ __attribute__((aligned(16))) double fold_lds[];
    const int NT = blockDim.x;
    double* const red = fold_lds;                // [8][NT]
    double* const gs = fold_lds + 8 * NT;        // [2][C]
    float* const mr = reinterpret_cast<float*>(gs + 2 * C);
    const int n = blockIdx.y;
    const int tpp = C >> 2, ppi = NT / tpp;
    const int tid = threadIdx.x;
    const int pl = tid / tpp, c = (tid - pl * tpp) * 4;
    const int C1 = C - C0, cg = C / 32;
    // the first four rows of this thread are requested BEFORE the fold: all blocks of a launch are resident at once and fold at the
    // same time -- without this HBM idles for the fold's dependent round trips at the start of every launch
    const int p_begin = blockIdx.x * per, p_end = min(HW, p_begin + per);
    const float* src; int ld;
    if (c < C0) { src = src0 + (size_t)n * HW * C0 + c; ld = C0; } else { src = src1 + (size_t)n * HW * C1 + (c - C0); ld = C1; }
    float* dst = y + (size_t)n * HW * C + c;
    auto ld4 = [&](const float* q) { return (VD_AA_NT & 1) ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(q)) : *reinterpret_cast<const f32x4*>(q); };
    int p = p_begin + pl;
    const bool head = p + 3 * ppi < p_end;
    f32x4 v0[4] = {};
    if (head) {
#pragma unroll
        for (int u = 0; u < 4; ++u) v0[u] = ld4(src + (size_t)(p + u * ppi) * ld);
    }
    {
        const bool second = c >= C0;
        const int split = second ? f.split1 : f.split0, ld = second ? C1 : C0, cc = second ? c - C0 : c;
        const double* p = (second ? f.part1 : f.part0) + ((size_t)n * split * ld + cc) * 2;
        double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int sp = pl; sp < split; sp += ppi) {
            const double* q = p + (size_t)sp * ld * 2;
#pragma unroll
            for (int e = 0; e < 4; ++e) { a[e] += q[2 * e]; a[4 + e] += q[2 * e + 1]; }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) red[e * NT + tid] = a[e];
    }
    __syncthreads();
    if (tid < tpp) {
        double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int k = 0; k < ppi; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] += red[e * NT + k * tpp + tid];
#pragma unroll
        for (int e = 0; e < 4; ++e) { gs[tid * 4 + e] = a[e]; gs[C + tid * 4 + e] = a[4 + e]; }
    }
    __syncthreads();
    if (tid < 32) {
        double s = 0, ss = 0;
        for (int k = 0; k < cg; ++k) { s += gs[tid * cg + k]; ss += gs[C + tid * cg + k]; }
        const double mean = s / f.count;
        double var = ss / f.count - mean * mean;
        if (var < 0) var = 0;
        mr[tid * 2] = (float)mean;
        mr[tid * 2 + 1] = (float)(1.0 / sqrt(var + 1e-5));
    }
    __syncthreads();
    f32x4 A, B;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int g = (c + e) / cg;
        const float mf = mr[g * 2], rf = mr[g * 2 + 1];
        float Ae = rf * f.gamma[c + e];
        float Be = f.beta[c + e] - mf * Ae;
        if (f.film) {
            const float sc = 1.0f + f.film[(size_t)n * f.film_ld + c + e];
            const float sh = f.film[(size_t)n * f.film_ld + C + c + e];
            Ae *= sc;
            Be = Be * sc + sh;
        }
        A[e] = Ae; B[e] = Be;
    }
    auto one = [&](f32x4 v) {
        f32x4 r = v * A + B;
        if (act) { r.x = silu_f(r.x); r.y = silu_f(r.y); r.z = silu_f(r.z); r.w = silu_f(r.w); }
        return r;
    };
    if (head) {
#pragma unroll
        for (int u = 0; u < 4; ++u) *reinterpret_cast<f32x4*>(dst + (size_t)(p + u * ppi) * C) = one(v0[u]);
        p += 4 * ppi;
    }
    for (; p + 3 * ppi < p_end; p += 4 * ppi) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = ld4(src + (size_t)(p + u * ppi) * ld);
#pragma unroll
        for (int u = 0; u < 4; ++u) *reinterpret_cast<f32x4*>(dst + (size_t)(p + u * ppi) * C) = one(v[u]);
    }
    for (; p < p_end; p += ppi) *reinterpret_cast<f32x4*>(dst + (size_t)p * C) = one(*reinterpret_cast<const f32x4*>(src + (size_t)p * ld));
}

int launch_affine_act_fold(const float* src0, const float* src1, int C0, int C, const GnFold& f, int nfr, int HW, int act, float* y,
                           hipStream_t s) {
    VD_REQUIRE(C % 32 == 0 && C0 % 4 == 0 && C <= 1024 && (src1 != nullptr || C0 == C), "affine_act_fold: channel counts");
    VD_REQUIRE(f.part0 && (C0 == C || f.part1), "affine_act_fold: GroupNorm partial tables");
    const int tpp = C / 4, ppi = 256 / tpp, threads = ppi * tpp;
    int split = 1;                                  // (coarse blocks: every block folds its frame's table first -- 8192 blocks: 1.74 -> 2.58 ms per step, r05h)
    while (nfr * split < fold_target_blocks() && HW / (split * 2) >= ppi * fold_min_rows()) split *= 2;
    const int per = (HW + split - 1) / split;
    const size_t lds = (size_t)(8 * threads + 2 * C) * sizeof(double) + 64 * sizeof(float);
    hipLaunchKernelGGL(affine_act_fold_kernel, dim3(split, nfr), dim3(threads), lds, s, src0, src1, C0, C, f, HW, per, act, y);
    VD_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------ y = x*A[n][c] + B[n][c]
// the same pass without the activation (attention blocks: the normalised tensor is also the residual, unet.py:474,538)
int launch_affine_apply(const float* x, const float* affA, const float* affB, int nfr, int HW, int C, float* y,
                        hipStream_t s) {
    return launch_affine_act(x, nullptr, C, C, affA, affB, nfr, HW, 0, y, s);
}

// ------------------------------------------------------------------ temporal GroupNorm (unet.py:472-475 on (B*HW, C, T))
// Statistics over (T x C/32) for each (b, pixel, group).  One pass: the T rows of a pixel stay in
// registers between the statistics and the normalisation.  Thread -> (pixel slot, channel quad).
template <int TMAX>
__global__ __launch_bounds__(256) void gn_temporal_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, int T, int HW, int C,
                                                          float* __restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) double sred[];   // [thread][2]: per-thread (4 channels x T) partial sums
    const int tpp = C >> 2, ppb = 256 / tpp;
    const int tid = threadIdx.x;
    const int pl = tid / tpp, q = tid - pl * tpp;
    const int b = blockIdx.y;
    const int p = blockIdx.x * ppb + pl;
    const bool active = pl < ppb && p < HW;
    f32x4 v[TMAX];
    double ps = 0, pss = 0;                       // this thread's share of its group: 4 channels x T frames
    const int cg = C / 32, tpg = cg >> 2;        // channels per group (a multiple of 4), threads per group
    if (active) {
#pragma unroll
        for (int t = 0; t < TMAX; ++t)
            if (t < T) v[t] = *reinterpret_cast<const f32x4*>(x + (((size_t)b * T + t) * HW + p) * C + q * 4);
#pragma unroll
        for (int t = 0; t < TMAX; ++t)
            if (t < T) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { const double d = (double)v[t][e]; ps += d; pss += d * d; }
            }
        sred[(size_t)tid * 2] = ps;
        sred[(size_t)tid * 2 + 1] = pss;
    }
    __syncthreads();
    if (!active) return;
    const double cnt = (double)cg * T;
    double gs = 0, gss = 0;
    const int g0 = tid - (q % tpg);              // first thread of this thread's group (same pixel slot)
    for (int k = 0; k < tpg; ++k) { gs += sred[(size_t)(g0 + k) * 2]; gss += sred[(size_t)(g0 + k) * 2 + 1]; }
    const double mean = gs / cnt;
    double var = gss / cnt - mean * mean;
    if (var < 0) var = 0;
    const float rstd = (float)(1.0 / sqrt(var + 1e-5));
    f32x4 A, Bv;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int c = q * 4 + e;
        A[e] = rstd * gamma[c];
        Bv[e] = beta[c] - (float)mean * A[e];
    }
#pragma unroll
    for (int t = 0; t < TMAX; ++t)
        if (t < T) *reinterpret_cast<f32x4*>(y + (((size_t)b * T + t) * HW + p) * C + q * 4) = v[t] * A + Bv;
}

// any channels-per-group count (the tiny test models have 2): every thread folds its channels' groups from a per-channel table
template <int TMAX>
__global__ __launch_bounds__(256) void gn_temporal_generic_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, int T, int HW, int C,
                                                          float* __restrict__ y) {
    extern __shared__ __attribute__((aligned(16))) double sred[];   // [ppb][C][2]
    const int tpp = C >> 2, ppb = 256 / tpp;
    const int tid = threadIdx.x;
    const int pl = tid / tpp, q = tid - pl * tpp;
    const int b = blockIdx.y;
    const int p = blockIdx.x * ppb + pl;
    const bool active = pl < ppb && p < HW;
    f32x4 v[TMAX];
    double s[4] = {0, 0, 0, 0}, ss[4] = {0, 0, 0, 0};
    if (active) {
#pragma unroll
        for (int t = 0; t < TMAX; ++t) {
            if (t < T) {
                v[t] = *reinterpret_cast<const f32x4*>(x + (((size_t)b * T + t) * HW + p) * C + q * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { s[e] += v[t][e]; ss[e] += (double)v[t][e] * v[t][e]; }
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            sred[((size_t)pl * C + q * 4 + e) * 2] = s[e];
            sred[((size_t)pl * C + q * 4 + e) * 2 + 1] = ss[e];
        }
    }
    __syncthreads();
    if (!active) return;
    const int cg = C / 32;
    const double cnt = (double)cg * T;
    f32x4 A, Bv;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int c = q * 4 + e, g = c / cg;
        double gs = 0, gss = 0;
        for (int k = 0; k < cg; ++k) {
            gs += sred[((size_t)pl * C + g * cg + k) * 2];
            gss += sred[((size_t)pl * C + g * cg + k) * 2 + 1];
        }
        const double mean = gs / cnt;
        double var = gss / cnt - mean * mean;
        if (var < 0) var = 0;
        const float rstd = (float)(1.0 / sqrt(var + 1e-5));
        A[e] = rstd * gamma[c];
        Bv[e] = beta[c] - (float)mean * A[e];
    }
#pragma unroll
    for (int t = 0; t < TMAX; ++t)
        if (t < T) *reinterpret_cast<f32x4*>(y + (((size_t)b * T + t) * HW + p) * C + q * 4) = v[t] * A + Bv;
}

int launch_gn_temporal(const float* x, const float* gamma, const float* beta, int B, int T, int HW, int C, float* y,
                       hipStream_t s) {
    VD_REQUIRE(C % 32 == 0 && C <= 1024, "GroupNorm32 channel constraints");
    VD_REQUIRE(T >= 1 && T <= 32, "temporal window of 1..32 frames");
    const int ppb = 256 / (C / 4);
    dim3 grid((HW + ppb - 1) / ppb, B);
    if ((C / 32) % 4 == 0) {                      // a thread's channel quad lies in one group: the fast kernel
        const size_t lds = (size_t)256 * 2 * sizeof(double);
        if (T <= 16)
            hipLaunchKernelGGL(gn_temporal_kernel<16>, grid, dim3(256), lds, s, x, gamma, beta, T, HW, C, y);
        else
            hipLaunchKernelGGL(gn_temporal_kernel<32>, grid, dim3(256), lds, s, x, gamma, beta, T, HW, C, y);
    } else {
        const size_t lds = (size_t)ppb * C * 2 * sizeof(double);
        if (T <= 16)
            hipLaunchKernelGGL(gn_temporal_generic_kernel<16>, grid, dim3(256), lds, s, x, gamma, beta, T, HW, C, y);
        else
            hipLaunchKernelGGL(gn_temporal_generic_kernel<32>, grid, dim3(256), lds, s, x, gamma, beta, T, HW, C, y);
    }
    VD_HIP(hipGetLastError());
    return 0;
}

}  // namespace vd
