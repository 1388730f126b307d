// Issue rate of v_mfma_f32_32x32x16_bf16 on gfx950 by waves per SIMD and accumulator rotation (no memory traffic).
// NACC = accumulator tiles a wave rotates over (1 = every MFMA waits for its predecessor), WPS = waves per SIMD.
// Printed: shader cycles per MFMA per SIMD, and the chip's rate from the wall clock (hipEvents).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_rate.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int NACC, int WPS>
__global__ __launch_bounds__(256 * WPS, 1) void loop(float* out, unsigned long long* ticks, int iters) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(1.f + (threadIdx.x & 3) + i); b[i] = (__bf16)(0.001f * i); }
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 32; ++k) acc[k % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[k % NACC], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int q = 0; q < 16; ++q) s += acc[i][q];
    out[blockIdx.x * 256 * WPS + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}
// the conv kernel's pattern: 8 accumulator tiles, six MFMAs in a row on each, 3 A and 12 B fragments
template <int WPS, int LB, bool RANDOM>
__global__ __launch_bounds__(256 * WPS, LB) void loop6(float* out, unsigned long long* ticks, int iters) {
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a[3], b[12];
    // RANDOM: operands with random bits (like real activations / weights): the switching activity sets the power, and the power
    // the clock the chip sustains
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    auto rnd = [&]() { h ^= h << 13; h ^= h >> 17; h ^= h << 5; return RANDOM ? (float)(int)(h & 0xffff) * (1.f / 65536.f) - 0.5f : 1.f; };
    for (int f = 0; f < 3; ++f) for (int i = 0; i < 8; ++i) a[f][i] = (__bf16)(RANDOM ? rnd() : 1.f + (threadIdx.x & 3) + i + f);
    for (int f = 0; f < 12; ++f) for (int i = 0; i < 8; ++i) b[f][i] = (__bf16)(RANDOM ? rnd() * 0.01f : 0.001f * i * f);
    constexpr int PA[6] = {2, 1, 1, 0, 0, 0}, PB[6] = {0, 1, 0, 2, 1, 0};
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 48; ++k) {
            const int t = k / 6, q = k % 6;
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA[q]], b[(t & 3) * 3 + PB[q]], acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 8; ++i) for (int q = 0; q < 16; ++q) s += acc[i][q];
    out[blockIdx.x * 256 * WPS + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}
template <int WPS, int LB, bool RANDOM>
void run6(float* out, unsigned long long* ticks) {
    const int blocks = 256 * LB, iters = 40000;
    hipLaunchKernelGGL((loop6<WPS, LB, RANDOM>), dim3(blocks), dim3(256 * WPS), 0, 0, out, ticks, 100);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((loop6<WPS, LB, RANDOM>), dim3(blocks), dim3(256 * WPS), 0, 0, out, ticks, iters);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double mf = (double)blocks * 4 * WPS * iters * 48.0;
    printf("conv pattern, %d waves per block, %d block(s) per CU, %s operands: %7.1f TFLOP/s\n", 4 * WPS, LB, RANDOM ? "random" : "constant",
           mf * 32768.0 / (ms * 1e-3) * 1e-12);
}
template <int NACC, int WPS>
void run(float* out, unsigned long long* ticks) {
    const int blocks = 256, iters = 20000;
    hipLaunchKernelGGL((loop<NACC, WPS>), dim3(blocks), dim3(256 * WPS), 0, 0, out, ticks, 100);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((loop<NACC, WPS>), dim3(blocks), dim3(256 * WPS), 0, 0, out, ticks, iters);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long t;
    (void)hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
    const double mf = (double)blocks * 4 * WPS * iters * 32.0;
    printf("accumulators %d, waves/SIMD %d: %6.2f cycles per MFMA per SIMD, %7.1f TFLOP/s, clock %.2f GHz\n", NACC, WPS,
           (double)t / (iters * 32.0 * WPS), mf * 32768.0 / (ms * 1e-3) * 1e-12, (double)t / (ms * 1e-3) * 1e-9);
}
int main() {
    float* out; unsigned long long* ticks;
    (void)hipMalloc(&out, 4096 * 1024 * 4); (void)hipMalloc(&ticks, 8);
    run<1, 1>(out, ticks); run<2, 1>(out, ticks); run<4, 1>(out, ticks); run<8, 1>(out, ticks);
    run<1, 2>(out, ticks); run<2, 2>(out, ticks); run<4, 2>(out, ticks);
    run6<1, 1, false>(out, ticks); run6<2, 1, false>(out, ticks); run6<1, 2, false>(out, ticks);
    run6<1, 1, true>(out, ticks); run6<2, 1, true>(out, ticks); run6<1, 1, true>(out, ticks);
    run<1, 4>(out, ticks); run<2, 4>(out, ticks); run<4, 4>(out, ticks);
    return 0;
}
