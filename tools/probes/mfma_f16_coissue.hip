// f16-MFMA co-issue probe (no memory traffic), gfx950: cycles per v_mfma_f32_32x32x16_f16 (32 alone) with NF vector-ALU
// instructions of one kind issued by the SAME wave behind each MFMA -- the candidates for the two-piece fp16 split of an
// fp32 operand (x = a0 + 2^-12 a1', a0 = f16(x), a1' = f16((x - a0) * 2^12)) -- and the split's exactness on random values.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_f16_coissue.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
static const char* NAMES[] = {"v_fma_f32", "v_cvt_pk_f16_f32", "v_fma_mix_f32", "v_cvt_f32_f16", "v_cvt_f32_f16_sdwa", "v_mul_f32", "v_ldexp_f32",
                              "split pair (6 instr)", "v_and_b32"};
template <int NF, int MODE>
__global__ __launch_bounds__(256, 1) void loop(float* out, unsigned long long* ticks, int iters, float a0, float b0) {
    f32x16 acc[16];
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(a0 + (threadIdx.x & 7) + i); b[i] = (_Float16)(b0 * i); }
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = a0 * i + threadIdx.x;
    float fa = a0, fb = b0;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 128; ++k) {
            acc[k & 15] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[k & 15], 0, 0, 0);
            if (MODE == 7) {
#pragma unroll
                for (int f = 0; f < NF; ++f) {    // NF pairs: pk, 2 mix, 2 scale folded (mix with -4096 and pre-scaled x), pk
                    unsigned p0, p1; float r0, r1, s0, s1;
                    asm volatile("v_cvt_pk_f16_f32 %0, %6, %7\n\t"
                                 "v_mul_f32 %2, 0x45800000, %6\n\t"
                                 "v_mul_f32 %3, 0x45800000, %7\n\t"
                                 "v_fma_mix_f32 %4, %0, %8, %2 op_sel_hi:[1,0,0]\n\t"
                                 "v_fma_mix_f32 %5, %0, %8, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
                                 "v_cvt_pk_f16_f32 %1, %4, %5"
                                 : "=&v"(p0), "=&v"(p1), "=&v"(s0), "=&v"(s1), "=&v"(r0), "=&v"(r1)
                                 : "v"(x[(2 * f) & 15]), "v"(x[(2 * f + 1) & 15]), "v"(fb));
                    x[(2 * f) & 15] = __builtin_bit_cast(float, p0 ^ p1);
                }
            } else {
#pragma unroll
                for (int f = 0; f < NF; ++f) {
                    if (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[f & 15]) : "v"(fa), "v"(fb));
                    if (MODE == 1) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(x[f & 15]) : "v"(fa));
                    if (MODE == 2) asm volatile("v_fma_mix_f32 %0, %0, %1, %2 op_sel_hi:[1,0,0]" : "+v"(x[f & 15]) : "v"(fa), "v"(fb));
                    if (MODE == 3) asm volatile("v_cvt_f32_f16 %0, %0" : "+v"(x[f & 15]));
                    if (MODE == 4) asm volatile("v_cvt_f32_f16_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "+v"(x[f & 15]));
                    if (MODE == 5) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[f & 15]) : "v"(fa));
                    if (MODE == 6) asm volatile("v_ldexp_f32 %0, %0, 12" : "+v"(x[f & 15]));
                    if (MODE == 8) asm volatile("v_and_b32 %0, 0xffffe000, %0" : "+v"(x[f & 15]));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 16; ++i) s += x[i];
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
    printf("%-22s x%2d: %6.2f cycles per MFMA\n", NAMES[MODE], NF, (double)t / (iters * 128.0));
}
template <int MODE>
void sweep(float* out, unsigned long long* ticks) {
    run<0, MODE>(out, ticks); run<2, MODE>(out, ticks); run<4, MODE>(out, ticks); run<6, MODE>(out, ticks); run<8, MODE>(out, ticks);
}

// exactness of the split: a0 = f16(x) (RNE), a1 = f16((x - a0) * 4096) via v_fma_mix_f32; back = a0 + a1 / 4096 in double
__global__ void split_check(const float* x, float* a0o, float* a1o, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    const float xa = x[2 * i], xb = x[2 * i + 1];
    unsigned p0, p1; float r0, r1;
    const float m4096 = -4096.f;
    asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(p0) : "v"(xa), "v"(xb));
    const float sa = xa * 4096.f, sb = xb * 4096.f;
    asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(p0), "v"(m4096), "v"(sa));
    asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(p0), "v"(m4096), "v"(sb));
    asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(p1) : "v"(r0), "v"(r1));
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const h2 h0 = __builtin_bit_cast(h2, p0), h1 = __builtin_bit_cast(h2, p1);
    a0o[2 * i] = (float)h0[0]; a0o[2 * i + 1] = (float)h0[1];
    a1o[2 * i] = (float)h1[0]; a1o[2 * i + 1] = (float)h1[1];
}

int main() {
    float* out; unsigned long long* ticks;
    (void)hipMalloc(&out, 4096 * 256 * 4); (void)hipMalloc(&ticks, 8);
    sweep<0>(out, ticks); sweep<1>(out, ticks); sweep<2>(out, ticks); sweep<3>(out, ticks); sweep<4>(out, ticks); sweep<5>(out, ticks);
    sweep<6>(out, ticks); sweep<8>(out, ticks);
    run<1, 7>(out, ticks); run<2, 7>(out, ticks);
    // split exactness over magnitudes 2^-30 .. 2^15
    const int n = 1 << 20;
    std::vector<float> hx(n), h0(n), h1(n);
    srand(1);
    for (int i = 0; i < n; ++i) {
        const double m = 1.0 + (rand() / (double)RAND_MAX), e = -30 + (rand() % 46);
        hx[i] = (float)((rand() & 1 ? -1 : 1) * m * std::pow(2.0, e));
    }
    float *dx, *d0, *d1;
    (void)hipMalloc(&dx, n * 4); (void)hipMalloc(&d0, n * 4); (void)hipMalloc(&d1, n * 4);
    (void)hipMemcpy(dx, hx.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(split_check, dim3(n / 2 / 256), dim3(256), 0, 0, dx, d0, d1, n);
    (void)hipMemcpy(h0.data(), d0, n * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(h1.data(), d1, n * 4, hipMemcpyDeviceToHost);
    double worst[46] = {};
    for (int i = 0; i < n; ++i) {
        const double back = (double)h0[i] + (double)h1[i] / 4096.0, rel = std::fabs(back - (double)hx[i]) / std::fabs((double)hx[i]);
        int e; std::frexp(hx[i], &e); e = e - 1 + 30;
        if (e >= 0 && e < 46 && rel > worst[e]) worst[e] = rel;
    }
    for (int e = 0; e < 46; e += 3) printf("|x| in 2^%d: worst relative error of a0 + a1/4096: %.3e (2^%.1f)\n", e - 30, worst[e], worst[e] > 0 ? std::log2(worst[e]) : -99.0);
    return 0;
}
