// Would TWO waves per SIMD, each with 8 accumulator tiles and the WHOLE transform + split of its Winograd row (9 vector
// instructions per MFMA instead of 4.5), run the matrix pipe better than one wave with 16 tiles?  No memory traffic:
// cycles per MFMA per SIMD with 1 wave x (MFMA + NV vector instructions) and with 2 waves x (MFMA + 2 NV).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_w8_probe.hip -o /tmp/p8 && /tmp/p8
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int NV, int THREADS>
__global__ __launch_bounds__(THREADS, 1) void loop(float* out, unsigned long long* ticks, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    u32x4 af[2][3], bf[2][3];
    for (int i = 0; i < 2; ++i) for (int p = 0; p < 3; ++p) for (int e = 0; e < 4; ++e) { af[i][p][e] = 0x3f803f80u + threadIdx.x + i + p + e; bf[i][p][e] = 0x3f003f80u + threadIdx.x * 3 + i + p + e; }
    float tv[8];
    for (int i = 0; i < 8; ++i) tv[i] = a0 * i + threadIdx.x * 0.37f;
    unsigned sel = 0x07060302u + (unsigned)iters * 0;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int pos = 0; pos < 8; ++pos) {
            const int cur = pos & 1, nxt = cur ^ 1;
            constexpr int PA[6] = {2, 1, 1, 0, 0, 0}, PB[6] = {0, 1, 0, 2, 1, 0};
            constexpr int NS = NACC == 16 ? 12 : 6;                     // slots per position
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                const int q = NACC == 16 ? k >> 1 : k, n = NACC == 16 ? k & 1 : 0;
                const int ai = (pos * (NACC == 16 ? 2 : 1) + n) & (NACC - 1);
                acc[ai] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[cur][PA[q]]), __builtin_bit_cast(bf16x8, bf[n][PB[q]]), acc[ai], 0, 0, 0);
                // NV vector instructions: split steps of channel pairs, round-robin
#pragma unroll
                for (int v = 0; v < NV; v += 5) {
                    const int pr = (k + v / 5) & 3;
                    unsigned p1; float r0, r1, h0, h1;
                    asm volatile("v_and_b32 %3, 0xffff0000, %5\n\tv_and_b32 %4, 0xffff0000, %6\n\tv_perm_b32 %0, %6, %5, %7\n\t"
                                 "v_sub_f32 %1, %5, %3\n\tv_sub_f32 %2, %6, %4"
                                 : "=&v"(p1), "=&v"(r0), "=&v"(r1), "=&v"(h0), "=&v"(h1) : "v"(tv[2 * pr]), "v"(tv[2 * pr + 1]), "s"(sel));
                    af[nxt][v / 5 % 3][pr] = p1; tv[2 * pr] = r0 + a0; tv[2 * pr + 1] = r1;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 8; ++i) s += tv[i];
    out[blockIdx.x * THREADS + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}
template <int NACC, int NV, int THREADS>
void run(float* out, unsigned long long* ticks, unsigned long long* rt, const char* what) {
    const int iters = 200;
    hipLaunchKernelGGL((loop<NACC, NV, THREADS>), dim3(256), dim3(THREADS), 0, 0, out, ticks, 10, 1.f, 2.f);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((loop<NACC, NV, THREADS>), dim3(256), dim3(THREADS), 0, 0, out, ticks, iters, 1.f, 2.f);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_simd = (double)iters * 8 * (NACC == 16 ? 12 : 6) * (THREADS / 256);
    printf("%-64s %7.1f ns per MFMA per SIMD (wall)  = %5.1f cycles at 2.0 GHz\n", what, ms * 1e6 / mfma_per_simd, ms * 1e6 / mfma_per_simd * 2.0);
}
int main() {
    float* out; unsigned long long* ticks;
    (void)hipMalloc(&out, 4096 * 512 * 4); (void)hipMalloc(&ticks, 8);
    run<16, 0, 256>(out, ticks, nullptr, "1 wave/SIMD, 16 tiles, MFMA only");
    run<16, 5, 256>(out, ticks, nullptr, "1 wave/SIMD, 16 tiles, 5 VALU per MFMA");
    run<8, 0, 512>(out, ticks, nullptr, "2 waves/SIMD, 8 tiles each, MFMA only");
    run<8, 5, 512>(out, ticks, nullptr, "2 waves/SIMD, 8 tiles each, 5 VALU per MFMA");
    run<8, 10, 512>(out, ticks, nullptr, "2 waves/SIMD, 8 tiles each, 10 VALU per MFMA (whole transform per wave)");
    return 0;
}
