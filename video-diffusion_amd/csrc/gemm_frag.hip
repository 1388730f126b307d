// Linear layers and 1x1 convolutions on fp32 MFMA with fragment-major weights, gfx950.
//
//   out[m][n] = bias[n] + res[m][n] + sum_k f(A[m][k]) * W[n][k]        f = identity | SiLU
//
// Same operand plan as conv_halo.hip, minus the halo: the A tile [BM][32] of a K-chunk goes through LDS
// (coalesced 128-byte row segments in, ds_read_b128 fragments out, double-buffered, ONE barrier per
// chunk), the B operand is fetched by each wave straight from L2 in MFMA-fragment order
//     [K/32][N/32][kgroup 4][lane 64][4 floats]
// one coalesced 1 KiB load per k-group, two k-groups ahead in a 4-deep register ring.
// A may be a virtual concat of two row-major sources (ResBlock skip connection over cat([h, skip])).
#include "vd_common.h"

namespace vd {

constexpr int GLD = 36;

#ifdef VD_GEMM_TIMING
// kernel-experiment builds only (tools/gemm_timing.py): shader-clock stamps of one mid-grid block, wave 0
__device__ unsigned long long g_gemm_stamp[6];      // [4], [5]: constant-rate (100 MHz) clock at stamps 0 and 3
#define GEMM_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x == gridDim.x / 2 && blockIdx.y == 0) {                  \
        g_gemm_stamp[i] = __builtin_readcyclecounter();                                                             \
        if (i == 0) g_gemm_stamp[4] = __builtin_amdgcn_s_memrealtime();                                             \
        if (i == 3) g_gemm_stamp[5] = __builtin_amdgcn_s_memrealtime(); } } while (0)
extern "C" int vd_debug_gemm_stamps(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_gemm_stamp), sizeof(g_gemm_stamp));
}
#else
#define GEMM_STAMP(i)
#endif

// On gfx950 VALU instructions do not overlap fp32 MFMAs of the same wave (tools/mfma_peak.hip), while memory and scalar
// issue is free: all addressing therefore goes through buffer descriptors (lane offsets computed once, per-chunk
// offsets scalar, out-of-range rows handled by the range check), and the SiLU prologue is compiled out when unused.
template <int BM, int BN, bool ACT>
__global__ __launch_bounds__(256, 3) void gemm_frag_kernel(IgemmArgs a) {
    constexpr int MI = BM / 64, NI = BN / 64, AR = BM / 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];          // [2][BM][GLD]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int lr = lane & 31, lh = lane >> 5;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int lrow = tid >> 3, lq = tid & 7;
    const int nchunk = a.Cin >> 5, ncoblk = a.Cout >> 5;
    const int C1 = a.Cin - a.C0;
    GEMM_STAMP(0);
    if (a.zcount > 1) {                          // batched problems of one shape
        const int z = blockIdx.z;
        a.src0 += (size_t)z * a.zs_a; a.wfrag += (size_t)z * a.zs_w; a.out += (size_t)z * a.zs_out;
        if (a.bias) a.bias += (size_t)z * a.zs_bias;
    }

    // A rows: byte offsets into either source (rows past M: duplicates of the last row, dropped at the store)
    const auto asrc0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.src0), 0, a.M * a.C0 * 4, 0x00020000);
    const auto asrc1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.src1 ? a.src1 : a.src0), 0,
                                                         a.src1 ? a.M * C1 * 4 : 0, 0x00020000);
    unsigned ao0[AR], ao1[AR];
#pragma unroll
    for (int j = 0; j < AR; ++j) {
        const unsigned row = (unsigned)min(m0 + lrow + 32 * j, a.M - 1);
        ao0[j] = row * (unsigned)(a.C0 * 4) + lq * 16u;
        ao1[j] = row * (unsigned)(C1 * 4) + lq * 16u;
    }
    // B fragments [K/32][N/32][kgroup 4][lane 64][4]: lane offset per column block, (chunk, k-group) scalar
    const auto bsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wfrag), 0, a.Cin * a.Cout * 4, 0x00020000);
    unsigned bo[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) bo[j] = (unsigned)min((int)blockIdx.y * (BN / 32) + wn * NI + j, ncoblk - 1) * 4096u + lane * 16u;

    f32x4 ra[AR], bfr[4][NI], afr[2][MI];      // weight ring of 4 = k-groups per chunk: slot == kg, compile-time
    auto a_prefetch = [&](int chunk) {
        const int c = chunk * 32;
        if (c < a.C0) {
#pragma unroll
            for (int j = 0; j < AR; ++j)
                ra[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(asrc0, ao0[j], c * 4, 0));
        } else {
#pragma unroll
            for (int j = 0; j < AR; ++j)
                ra[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(asrc1, ao1[j], (c - a.C0) * 4, 0));
        }
    };
    auto a_store = [&](float* Ad) {
#pragma unroll
        for (int j = 0; j < AR; ++j) {
            f32x4 v = ra[j];
            if constexpr (ACT) { v.x = silu_f(v.x); v.y = silu_f(v.y); v.z = silu_f(v.z); v.w = silu_f(v.w); }
            *reinterpret_cast<f32x4*>(Ad + (lrow + 32 * j) * GLD + lq * 4) = v;
        }
    };
    auto b_load = [&](int slot, int chunk, int kg) {
        const int so = chunk * ncoblk * 4096 + kg * 1024;
#pragma unroll
        for (int j = 0; j < NI; ++j) bfr[slot][j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(bsrc, bo[j], so, 0));
    };

    // output / residual element (row, column): lane part of the byte offset per (row group i, column block j) + a scalar
    // per accumulator row r -- ONE VALU add per access.  Rows past M fall off the end of the descriptor, columns past
    // Cout are parked beyond 2^31: the range check drops both.  (Epilogue VALU matters: beside the other blocks' MFMAs
    // every VALU instruction of this wave waits for a free ALU slot, tools/gemm_timing.py.)
    const auto osrc = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, a.M * a.ldo * 4, 0x00020000);
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.res ? a.res : a.out), 0,
                                                        a.res ? a.M * a.ldo * 4 : 0, 0x00020000);
    unsigned vb[MI][NI];
    float bv[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int co = n0 + wn * (BN / 2) + j * 32 + lr;
        bv[j] = a.bias && co < a.Cout ? a.bias[co] : 0.f;
#pragma unroll
        for (int i = 0; i < MI; ++i)
            vb[i][j] = co < a.Cout ? (unsigned)((m0 + wm * (BM / 2) + i * 32 + 4 * lh) * a.ldo + co) * 4u : 0x80000000u;
    }
    // accumulators start at bias + residual: the residual tile streams in under the first A-tile staging
    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int srow = ((r & 3) + 8 * (r >> 2)) * a.ldo * 4;                       // scalar
                acc[i][j][r] = bv[j] + __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, vb[i][j] + srow, 0, 0));
            }

    a_prefetch(0);
    b_load(0, 0, 0);
    b_load(1, 0, 1);
    a_store(smem);
    __syncthreads();

    const int aoff = (wm * (BM / 2) + lr) * GLD + lh * 4;
    GEMM_STAMP(1);
    for (int chunk = 0; chunk < nchunk; ++chunk) {
        const int nxt = min(chunk + 1, nchunk - 1);           // last chunk: redundant prefetch instead of a branch
        const float* Acur = smem + (chunk & 1) * BM * GLD + aoff;
        float* Anext = smem + ((chunk + 1) & 1) * BM * GLD;
        a_prefetch(nxt);
#pragma unroll
        for (int i = 0; i < MI; ++i) afr[0][i] = *reinterpret_cast<const f32x4*>(Acur + i * 32 * GLD);
#pragma unroll
        for (int kg = 0; kg < 4; ++kg) {
            if (kg + 2 < 4) b_load(kg + 2, chunk, kg + 2);      // two groups ahead; slot (kg+2)&3 was consumed two groups ago
            else b_load(kg - 2, nxt, kg - 2);
            if (kg + 1 < 4) {
#pragma unroll
                for (int i = 0; i < MI; ++i)
                    afr[(kg + 1) & 1][i] = *reinterpret_cast<const f32x4*>(Acur + i * 32 * GLD + (kg + 1) * 8);
            }
            __builtin_amdgcn_sched_barrier(0);               // keep the prefetches ahead of the MFMAs (see conv_halo.hip)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < MI; ++i)
#pragma unroll
                    for (int j = 0; j < NI; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(afr[kg & 1][i][e], bfr[kg][j][e], acc[i][j], 0, 0, 0);
            if (kg == 3) a_store(Anext);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }
    GEMM_STAMP(2);

#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            unsigned vbe = vb[i][j];
            asm volatile("" : "+v"(vbe));                // opaque: the 64 offsets are re-added here, not kept live
#pragma unroll                                           // across the K loop (64 VGPRs -> spills at 3 blocks per CU)
            for (int r = 0; r < 16; ++r) {
                const int srow = ((r & 3) + 8 * (r >> 2)) * a.ldo * 4;
                const float val = acc[i][j][r];          // (bit_cast straight from a vector-element lvalue reads element 0)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), osrc, vbe + srow, 0, 0);
            }
        }
    GEMM_STAMP(3);
}

#ifdef VD_GEMM_TIMING
extern "C" int vd_debug_gemm_occupancy(void) {
    int n = -1;
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, gemm_frag_kernel<128, 128, false>, 256, (size_t)2 * 128 * GLD * sizeof(float));
    return n;
}
#endif

bool gemm_frag_supported(const IgemmArgs& a) {
    return a.wfrag != nullptr && !a.wsplit && a.ksz == 1 && a.stride == 1 && a.pad == 0 && a.ups == 0 && a.Cout % 32 == 0 &&
           a.affA == nullptr && a.fbias == nullptr &&
           // 32-bit byte offsets (bit 31 marks out-of-range rows)
           (a.res == nullptr || a.res_ld == a.ldo) &&
           (size_t)a.M * std::max(std::max(a.C0, a.Cin - a.C0), a.ldo) < (1u << 28);
}

template <int BM, int BN>
static int launch_gf(const IgemmArgs& a, hipStream_t s) {
    const size_t lds = (size_t)2 * BM * GLD * sizeof(float);
    dim3 grid((a.M + BM - 1) / BM, (a.Cout + BN - 1) / BN, a.zcount > 1 ? a.zcount : 1);
    if (a.act) hipLaunchKernelGGL((gemm_frag_kernel<BM, BN, true>), grid, dim3(256), lds, s, a);
    else hipLaunchKernelGGL((gemm_frag_kernel<BM, BN, false>), grid, dim3(256), lds, s, a);
    VD_HIP(hipGetLastError());
    return 0;
}

int launch_gemm_frag(const IgemmArgs& a, int tile_class, hipStream_t s) {
    switch (tile_class) {
        case 0: return launch_gf<128, 128>(a, s);
        case 1: return launch_gf<128, 64>(a, s);
        case 2: return launch_gf<64, 128>(a, s);
        default: return launch_gf<64, 64>(a, s);
    }
}

// host-side repack of `rows` rows of a [N_total][K] row-major matrix (rows row0 .. row0+rows-1, all multiples of 32)
// into the fragment-major image of the WHOLE matrix: [K/32][N_total/32][4][64][4]
void pack_linear_frag(const float* w, float* out_base, int rows, int K, int n_total, int row0) {
    const int nchunk = K / 32, ncoblk = n_total / 32;
    for (int ch = 0; ch < nchunk; ++ch)
        for (int cb = 0; cb < rows / 32; ++cb)
            for (int kg = 0; kg < 4; ++kg)
                for (int h = 0; h < 2; ++h)
                    for (int r = 0; r < 32; ++r)
                        for (int e = 0; e < 4; ++e)
                            out_base[((((size_t)ch * ncoblk + row0 / 32 + cb) * 4 + kg) * 64 + h * 32 + r) * 4 + e] =
                                w[(size_t)(cb * 32 + r) * K + ch * 32 + kg * 8 + h * 4 + e];
}

}  // namespace vd
