// Large linear layers / 1x1 convolutions at fp32 accuracy on the bf16 matrix cores: one WAVE = one 128 x (32*NT) output
// tile, no LDS, no barrier.  gfx950.
//
//   out[m][n] = bias[n] + res[m][n] + sum_k f(A[m][k]) * W[n][k]        f = identity | SiLU        (as gemm_split.hip)
//
// Same arithmetic and the same weight image as gemm_split.hip (fp32 operands split exactly into three bf16 pieces, six
// piece products per element, weights [K/16][N/32][3][64][8] bf16).  What changes is the work split.  gemm_split.hip runs
// two 4-wave blocks per CU with 2x2 accumulator tiles per wave, one barrier per 32-k chunk and the A tile split through
// LDS; measured with everything but the MFMAs removed it still reaches only 0.62 of the pipe, and 0.40 in full
// (tools/gemm_split_timing.py) -- per-chunk overhead on 48 MFMAs.  Here a wave has the SIMD to itself (one wave per
// SIMD, 256 AGPRs of accumulators = 4 x 4 tiles of 32x32): a split A fragment feeds 24 MFMAs and a weight fragment 4, so
// per 32-k chunk it issues 192 MFMAs against 16 + 24 loads and 352 VALU (1.8 per MFMA slot), and nothing is shared:
//   * A fragments come straight from global memory in MFMA order (lane = row lr, k-half lh: 8 consecutive k = 2 x 16 B);
//     the four loads of a row tile and chunk cover whole 128-byte lines, issued together, one chunk ahead -- each row
//     tile's registers are refilled right after its last value has been split;
//   * the weights of the next k-step stream into the other half of a two-step ring while this one is multiplied;
//   * the split of row tile i+1 runs under the 6*NT MFMAs of row tile i.
// The launcher picks NT in {4, 3, 2} so that the wave tiles fill the 1024 SIMDs in whole rounds; small problems stay on
// gemm_split.hip (igemm.hip).  Result: correct (tests/test_gpu_ops.py runs the large shapes through both kernels) and
// within +-10 % of gemm_split.hip on every shape of the model, so it is opt-in (VD_GEMM_WAVE=1) and kept as the
// record of that experiment.
#include <algorithm>

#include "vd_common.h"

namespace vd {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int NT, bool ACT>
__global__ __launch_bounds__(256, 1) void gemm_wave_kernel(IgemmArgs a, int ntn) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    const int item = blockIdx.x * 4 + wave;                          // wave tile: column tile fastest (A rows shared in L2)
    const int tn = item % ntn, tm = item / ntn;
    const int m0 = tm * 128, cb0 = tn * NT;                           // first row, first 32-wide column block
    if (m0 >= a.M) return;                                            // (whole wave; there is no barrier in this kernel)
    const int nchunk = a.Cin >> 5, ncoblk = a.Cout >> 5;
    const int C1 = a.Cin - a.C0;

    // ---- A: lane (row lr of row tile i, k-half lh) reads 8 consecutive k of each 16-k step
    const auto asrc0 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.src0), 0, a.M * a.C0 * 4, 0x00020000);
    const auto asrc1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.src1 ? a.src1 : a.src0), 0,
                                                         a.src1 ? a.M * C1 * 4 : 0, 0x00020000);
    unsigned ao0[4], ao1[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned row = (unsigned)min(m0 + 32 * i + lr, a.M - 1);
        ao0[i] = row * (unsigned)(a.C0 * 4) + lh * 32u;
        ao1[i] = row * (unsigned)(C1 * 4) + lh * 32u;
    }
    f32x4 raw[4][4];                                                  // [row tile][k-step 2 x quad 2]
    auto a_load_tile = [&](int chunk, int i) {
        const int c = chunk * 32;
        if (c < a.C0) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                raw[i][e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(asrc0, ao0[i], c * 4 + (e >> 1) * 64 + (e & 1) * 16, 0));
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                raw[i][e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(asrc1, ao1[i], (c - a.C0) * 4 + (e >> 1) * 64 + (e & 1) * 16, 0));
        }
    };
    u32x4 apc[2][3];                                                  // [ring slot][piece] = 8 bf16
    float sr0, sr1;
    auto s_split = [&](int slot, int i, int ks, int pr, int stage) {  // pair pr = k 2pr, 2pr+1 of the lane's 8
        if (stage == 0) {
            const f32x4 v = raw[i][ks * 2 + (pr >> 1)];
            float x0 = (pr & 1) ? v.z : v.x, x1 = (pr & 1) ? v.w : v.y;
            if constexpr (ACT) { x0 = silu_f(x0); x1 = silu_f(x1); }
            unsigned p1;
            split_a(x0, x1, p1, sr0, sr1, 0x07060302u);
            apc[slot][0][pr] = p1;
        } else {
            unsigned p2, p3;
            split_b(sr0, sr1, p2, p3, 0x07060302u);
            apc[slot][1][pr] = p2; apc[slot][2][pr] = p3;
        }
    };

    // ---- B: [K/16][N/32][piece][lane][8 bf16] = 3072 bytes per (k-step, column block), two k-steps in registers
    const auto bsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wfrag), 0, a.Cin * a.Cout * 6, 0x00020000);
    const unsigned blane = lane * 16u;
    bf16x8 bfr[2][NT][3];
    auto b_load_one = [&](int slot, int kstep, int j, int p) {
        bfr[slot][j][p] = __builtin_bit_cast(
            bf16x8, __builtin_amdgcn_raw_buffer_load_b128(bsrc, blane, (kstep * ncoblk + min(cb0 + j, ncoblk - 1)) * 3072 + p * 1024, 0));
    };

    f32x16 acc[4][NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int co = (cb0 + j) * 32 + lr;
        const float bv = a.bias && co < a.Cout ? a.bias[co] : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = bv;
    }

    // ---- prologue: chunk 0 of A, k-step 0 of the weights, pieces of (row tile 0, k-step 0)
    const int nks = 2 * nchunk;
#pragma unroll
    for (int i = 0; i < 4; ++i) a_load_tile(0, i);
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int p = 0; p < 3; ++p) b_load_one(0, 0, j, p);
#pragma unroll
    for (int pr = 0; pr < 4; ++pr) { s_split(0, 0, 0, pr, 0); s_split(0, 0, 0, pr, 1); }

    // one position = (k-step ks of the chunk, row tile i): 6*NT MFMA slots
    constexpr int SL = 6 * NT;
    auto position = [&](int chunk, int ks, int i) {
        const int pi = i & 1;                                         // 4 positions per k-step: the ring slot is the row tile's parity
        const int ni = (i + 1) & 3, nks_ = i == 3 ? ks ^ 1 : ks;      // the position whose pieces are made now
        const int g = 2 * chunk + ks;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            f32x16 c = acc[i][j];
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                constexpr int PA[6] = {0, 1, 2, 0, 1, 0}, PB[6] = {2, 1, 0, 1, 0, 0};     // small terms first
                const int k = j * 6 + q;
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, apc[pi][PA[q]]), bfr[ks][j][PB[q]], c, 0, 0, 0);
                // split of the next position: pair t in slots 3t, 3t+1 (from slot 0; raw of the next chunk's row tile 0 has been
                // in flight for three positions when (ks = 1, i = 3) splits it)
                if (k < 12 && k % 3 != 2) s_split(pi ^ 1, ni, nks_, k / 3, k % 3);
                // weights of the next k-step into the other ring half: 3*NT requests over the four positions of this one
                {
                    const int w = i * SL + k;                         // slot within the k-step, 0 .. 4*SL-1
                    if (w % 8 == 5 && w / 8 < 3 * NT) b_load_one(ks ^ 1, min(g + 1, nks - 1), (w / 8) / 3, (w / 8) % 3);
                }
                // A of the next chunk: row tile i's registers are free once (ks = 1, i) has been split, i.e. from position
                // (ks = 1, i) on (its split ran during (1, i-1) resp. (0, 3) for i = 0)
                if (ks == 1 && k >= SL - 4) {
                    const int e = k - (SL - 4);
                    const int cn = min(chunk + 1, nchunk - 1) * 32;
                    if (cn < a.C0)
                        raw[i][e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(asrc0, ao0[i], cn * 4 + (e >> 1) * 64 + (e & 1) * 16, 0));
                    else
                        raw[i][e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(asrc1, ao1[i], (cn - a.C0) * 4 + (e >> 1) * 64 + (e & 1) * 16, 0));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            acc[i][j] = c;
        }
    };
#pragma nounroll
    for (int chunk = 0; chunk < nchunk; ++chunk) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i) position(chunk, ks, i);
    }

    // ---- epilogue: residual, store (rows past M and columns past Cout fall to the descriptor's range check)
    const auto osrc = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, a.M * a.ldo * 4, 0x00020000);
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.res ? a.res : a.out), 0, a.res ? a.M * a.ldo * 4 : 0, 0x00020000);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int co = (cb0 + j) * 32 + lr;
            unsigned vb = co < a.Cout ? (unsigned)((m0 + i * 32 + 4 * lh) * a.ldo + co) * 4u : 0x80000000u;
            asm volatile("" : "+v"(vb));
            f32x16 v = acc[i][j];
            if (a.res) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    v[r] += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc, vb + ((r & 3) + 8 * (r >> 2)) * a.ldo * 4, 0, 0));
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float val = v[r];
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), osrc, vb + ((r & 3) + 8 * (r >> 2)) * a.ldo * 4, 0, 0);
            }
        }
}

// Wave tiles of 128 x 32*NT: the NT (dividing N/32) whose tile count fills the SIMDs of the device in the fullest rounds;
// 0 if the problem is not this kernel's (small, ragged, batched, or a convolution)
static int gemm_wave_nt(const IgemmArgs& a) {
    static int nsimd = 0;
    if (!nsimd) {
        int dev = 0, ncu = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
        nsimd = 4 * ncu;
    }
    const long mt = (a.M + 127) / 128;
    int best = 0;
    double beste = 0.0;
    for (int nt = 4; nt >= 2; --nt) {
        if ((a.Cout / 32) % nt) continue;
        const long items = mt * (a.Cout / 32 / nt);
        const long rounds = (items + nsimd - 1) / nsimd;
        const double e = (double)items / (double)(rounds * nsimd) * (nt == 4 ? 1.0 : nt == 3 ? 0.97 : 0.9);   // narrower tiles re-read A more
        if (e > beste) { beste = e; best = nt; }
    }
    return beste >= 0.7 ? best : 0;
}

bool gemm_wave_supported(const IgemmArgs& a) {
    // Opt-in (VD_GEMM_WAVE=1): parity-tested, but measured level with gemm_split.hip on the model's shapes (0.36-0.43 of the
    // nominal bf16 peak either way, tools/gemm_split_timing.py) -- both sit at the clock the chip sustains under this load
    static const bool on = [] { const char* e = getenv("VD_GEMM_WAVE"); return e && e[0] == '1'; }();
    if (!on || !gemm_split_supported(a) || a.ksz != 1 || a.zcount > 1 || a.M < 4096 || a.Cin % 32) return false;
    return gemm_wave_nt(a) != 0;
}

int launch_gemm_wave(const IgemmArgs& a, hipStream_t s) {
    const int nt = gemm_wave_nt(a);
    VD_REQUIRE(nt != 0, "gemm_wave: shape not covered");
    const int ntn = a.Cout / 32 / nt;
    const long items = (long)((a.M + 127) / 128) * ntn;
    dim3 grid((unsigned)((items + 3) / 4));
#define VD_GW_LAUNCH(NTV)                                                                                              \
    if (a.act) hipLaunchKernelGGL((gemm_wave_kernel<NTV, true>), grid, dim3(256), 0, s, a, ntn);                       \
    else hipLaunchKernelGGL((gemm_wave_kernel<NTV, false>), grid, dim3(256), 0, s, a, ntn)
    if (nt == 4) { VD_GW_LAUNCH(4); }
    else if (nt == 3) { VD_GW_LAUNCH(3); }
    else { VD_GW_LAUNCH(2); }
#undef VD_GW_LAUNCH
    VD_HIP(hipGetLastError());
    return 0;
}

}  // namespace vd
