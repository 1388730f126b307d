// How many vector-ALU instructions hide behind a v_mfma_f32_32x32x16_f16, one wave per SIMD (gfx950)?  Cycles per MFMA with NF
// instructions of one kind issued by the same wave behind each MFMA: independent v_fma_f32, a DEPENDENT chain of v_fma_f32, packed
// v_pk_fma_f32 / v_pk_add_f32 (two fp32 results per instruction), v_pk_mul_f16, v_xor_b32.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_valu_curve.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
static const char* NAMES[] = {"v_fma_f32 independent", "v_fma_f32 dependent chain", "v_pk_fma_f32", "v_pk_add_f32", "v_pk_mul_f16", "v_xor_b32"};
template <int NF, int MODE>
__global__ __launch_bounds__(256, 1) void loop(float* out, unsigned long long* ticks, int iters, float a0, float b0) {
    f32x16 acc[16];
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(a0 + (threadIdx.x & 7) + i); b[i] = (_Float16)(b0 * i); }
    float x[16]; f32x2 y[8];
    for (int i = 0; i < 16; ++i) x[i] = a0 * i + threadIdx.x;
    for (int i = 0; i < 8; ++i) y[i] = f32x2{a0 * i, b0 + threadIdx.x};
    float fa = a0, fb = b0; f32x2 pa = {a0, a0}, pb = {b0, b0};
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 128; ++k) {
            acc[k & 15] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[k & 15], 0, 0, 0);
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[f & 15]) : "v"(fa), "v"(fb));
                if (MODE == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[0]) : "v"(fa), "v"(fb));
                if (MODE == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(y[f & 7]) : "v"(pa), "v"(pb));
                if (MODE == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(y[f & 7]) : "v"(pa));
                if (MODE == 4) asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(x[f & 15]) : "v"(fa));
                if (MODE == 5) asm volatile("v_xor_b32 %0, 0x80008000, %0" : "+v"(x[f & 15]));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 16; ++i) s += x[i];
    for (int i = 0; i < 8; ++i) s += y[i].x + y[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}
template <int NF, int MODE>
double run(float* out, unsigned long long* ticks) {
    const int blocks = 256, iters = 100;
    hipLaunchKernelGGL((loop<NF, MODE>), dim3(blocks), dim3(256), 0, 0, out, ticks, 5, 1.f, 2.f);
    (void)hipDeviceSynchronize();
    hipLaunchKernelGGL((loop<NF, MODE>), dim3(blocks), dim3(256), 0, 0, out, ticks, iters, 1.f, 2.f);
    (void)hipDeviceSynchronize();
    unsigned long long t;
    (void)hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
    return (double)t / (iters * 128.0);
}
template <int MODE>
void sweep(float* out, unsigned long long* ticks) {
    printf("%-28s NF = 0 1 2 3 4 5 6 8 10 12:", NAMES[MODE]);
    printf(" %5.1f", run<0, MODE>(out, ticks)); printf(" %5.1f", run<1, MODE>(out, ticks)); printf(" %5.1f", run<2, MODE>(out, ticks));
    printf(" %5.1f", run<3, MODE>(out, ticks)); printf(" %5.1f", run<4, MODE>(out, ticks)); printf(" %5.1f", run<5, MODE>(out, ticks));
    printf(" %5.1f", run<6, MODE>(out, ticks)); printf(" %5.1f", run<8, MODE>(out, ticks)); printf(" %5.1f", run<10, MODE>(out, ticks));
    printf(" %5.1f\n", run<12, MODE>(out, ticks));
}
int main() {
    float* out; unsigned long long* ticks;
    (void)hipMalloc(&out, 4096 * 256 * 4); (void)hipMalloc(&ticks, 8);
    sweep<0>(out, ticks); sweep<1>(out, ticks); sweep<2>(out, ticks); sweep<3>(out, ticks); sweep<4>(out, ticks); sweep<5>(out, ticks);
    return 0;
}
