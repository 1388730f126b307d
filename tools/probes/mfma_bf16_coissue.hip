// bf16-MFMA co-issue probe (no memory traffic), gfx950: cycles per v_mfma_f32_32x32x16_bf16 (32 cycles alone) with NF
// vector-ALU instructions of one kind issued by the SAME wave behind each MFMA.  One wave per SIMD, 16 accumulator tiles.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_bf16_coissue.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
static const char* NAMES[] = {"v_fma_f32", "v_and_b32", "v_perm_b32", "v_sub_f32", "v_dot2c_f32_bf16", "v_cvt_pk_bf16_f32", "v_pk_add_f32",
                              "v_pk_fma_f32", "v_accvgpr_read", "v_accvgpr_write"};
template <int NF, int MODE>
__global__ __launch_bounds__(256, 1) void loop(float* out, unsigned long long* ticks, int iters, float a0, float b0) {
    f32x16 acc[16];
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(a0 + threadIdx.x + i); b[i] = (__bf16)(b0 * i); }
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = a0 * i + threadIdx.x;
    f32x2 px[8], pa = {a0, b0};
    for (int i = 0; i < 8; ++i) px[i] = f32x2{a0 * i, b0 + i};
    float fa = a0, fb = b0;
    unsigned sel = 0x07060302u + (unsigned)iters * 0;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 128; ++k) {
            acc[k & 15] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[k & 15], 0, 0, 0);
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[f & 15]) : "v"(fa), "v"(fb));
                if (MODE == 1) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(x[f & 15]));
                if (MODE == 2) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x[f & 15]) : "v"(fa), "s"(sel));
                if (MODE == 3) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(x[f & 15]) : "v"(fa));
                if (MODE == 4) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(x[f & 15]) : "v"(fa), "v"(fb));
                if (MODE == 5) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x[f & 15]) : "v"(fa));
                if (MODE == 6) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(px[f & 7]) : "v"(pa));
                if (MODE == 7) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(px[f & 7]) : "v"(pa));
                if (MODE == 8) asm volatile("v_accvgpr_read_b32 %0, a255" : "=v"(x[f & 15]));
                if (MODE == 9) asm volatile("v_accvgpr_write_b32 a255, %0" : : "v"(x[f & 15]));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 16; ++i) s += x[i];
    for (int i = 0; i < 8; ++i) s += px[i].x + px[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}
template <int NF, int MODE>
void run(float* out, unsigned long long* ticks) {
    const int blocks = 256, iters = 200;
    hipLaunchKernelGGL((loop<NF, MODE>), dim3(blocks), dim3(256), 0, 0, out, ticks, 10, 1.f, 2.f);
    (void)hipDeviceSynchronize();
    hipLaunchKernelGGL((loop<NF, MODE>), dim3(blocks), dim3(256), 0, 0, out, ticks, iters, 1.f, 2.f);
    (void)hipDeviceSynchronize();
    unsigned long long t;
    (void)hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
    printf("%-20s x%2d: %6.2f cycles per MFMA\n", NAMES[MODE], NF, (double)t / (iters * 128.0));
}
template <int MODE>
void sweep(float* out, unsigned long long* ticks) {
    run<0, MODE>(out, ticks); run<2, MODE>(out, ticks); run<4, MODE>(out, ticks); run<6, MODE>(out, ticks); run<8, MODE>(out, ticks);
    run<12, MODE>(out, ticks);
}
int main() {
    float* out; unsigned long long* ticks;
    (void)hipMalloc(&out, 4096 * 256 * 4); (void)hipMalloc(&ticks, 8);
    sweep<0>(out, ticks); sweep<1>(out, ticks); sweep<2>(out, ticks); sweep<3>(out, ticks); sweep<4>(out, ticks); sweep<5>(out, ticks);
    sweep<6>(out, ticks); sweep<7>(out, ticks); sweep<8>(out, ticks); sweep<9>(out, ticks);
    return 0;
}
