// fp32-MFMA issue-rate probe (no memory traffic): what v_mfma_f32_32x32x2_f32 sustains on gfx950 with one wave per SIMD
// and 16 independent accumulator tiles, and how much VALU work hides in the gap behind each MFMA.
//   NF independent v_fma_f32 per gap (MODE 0), NF v_exp_f32 (MODE 1), one dependent chain of NF v_fma_f32 (MODE 2).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NF, int MODE>
__global__ __launch_bounds__(256, 1) void mfma_loop(float* out, unsigned long long* ticks, int iters, float a0, float b0) {
    f32x16 acc[16];
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x, b = b0;
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = a0 * i + threadIdx.x;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x2 px[8], pa = {a0, b0};
    for (int i = 0; i < 8; ++i) px[i] = f32x2{a0 * i, b0 + i};
    unsigned sc = (unsigned)iters;
    __shared__ float ldsbuf[4096];
    ldsbuf[threadIdx.x] = a0;
    f32x4 lv[4] = {};
    unsigned ldsa = (threadIdx.x & 63) * 16;
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 128; ++k) {
            acc[k & 15] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[k & 15], 0, 0, 0);
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[f & 15]) : "v"(a), "v"(b));
                if (MODE == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(x[f & 15]));
                if (MODE == 2) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[0]) : "v"(a), "v"(b));
                if (MODE == 3) asm volatile("v_exp_f32 %0, %0\n s_nop 0\n v_add_f32 %0, 1.0, %0" : "+v"(x[0]));
                if (MODE == 4) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(px[f & 7]) : "v"(pa));
                if (MODE == 5) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sc));
                if (MODE == 6) asm volatile("s_nop 0");
                if (MODE == 7) asm volatile("v_mov_b32 %0, %1" : "=v"(x[f & 15]) : "v"(a));
                if (MODE == 8) asm volatile("ds_read_b128 %0, %1" : "=v"(lv[f & 3]) : "v"(ldsa));
                if (MODE == 9) asm volatile("v_cndmask_b32 %0, 0, %0, vcc" : "+v"(x[f & 15]));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 16; ++i) s += x[i];
    for (int i = 0; i < 8; ++i) s += px[i].x + px[i].y;
    if (MODE == 8) { asm volatile("s_waitcnt lgkmcnt(0)"); for (int i = 0; i < 4; ++i) s += lv[i].x; }
    s += (float)sc;
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}
template <int NF, int MODE>
void run(float* out, unsigned long long* ticks) {
    const int blocks = 256, iters = 400;
    hipLaunchKernelGGL((mfma_loop<NF, MODE>), dim3(blocks), dim3(256), 0, 0, out, ticks, 10, 1.f, 2.f);
    (void)hipDeviceSynchronize();
    hipLaunchKernelGGL((mfma_loop<NF, MODE>), dim3(blocks), dim3(256), 0, 0, out, ticks, iters, 1.f, 2.f);
    (void)hipDeviceSynchronize();
    unsigned long long t;
    (void)hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
    printf("mode %d fillers %2d: %.2f ticks per MFMA\n", MODE, NF, (double)t / (iters * 128.0));
}
int main() {
    float* out; unsigned long long* ticks;
    (void)hipMalloc(&out, 4096 * 256 * 4); (void)hipMalloc(&ticks, 8);
    run<0, 0>(out, ticks); run<4, 0>(out, ticks); run<8, 0>(out, ticks); run<10, 0>(out, ticks); run<12, 0>(out, ticks);
    run<14, 0>(out, ticks); run<16, 0>(out, ticks); run<20, 0>(out, ticks);
    run<2, 1>(out, ticks); run<4, 1>(out, ticks); run<6, 1>(out, ticks); run<8, 1>(out, ticks);
    run<4, 2>(out, ticks); run<8, 2>(out, ticks); run<12, 2>(out, ticks);
    run<1, 3>(out, ticks); run<2, 3>(out, ticks); run<3, 3>(out, ticks); run<4, 3>(out, ticks);
    run<4, 4>(out, ticks); run<8, 4>(out, ticks); run<4, 5>(out, ticks); run<8, 5>(out, ticks); run<4, 6>(out, ticks);
    run<8, 6>(out, ticks); run<4, 7>(out, ticks); run<8, 7>(out, ticks); run<2, 8>(out, ticks); run<4, 8>(out, ticks);
    run<4, 9>(out, ticks); run<8, 9>(out, ticks);
    return 0;
}
