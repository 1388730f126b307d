// Why do the vector instructions of conv_wino_r64's slots not hide behind its MFMAs?  (PMC r02l: MFMA-busy cycles + VALU issue
// cycles = wave cycles; tools/mfma_bf16_coissue.hip: six independent VALU per MFMA are free.)  One wave per SIMD, 16
// accumulator tiles, the kernel's slot shape rebuilt step by step:
//   MODE 0  MFMA on 16 accumulators round-robin, constant operands, NV independent v_and per slot          (the old probe)
//   MODE 1  as the kernel: 12 MFMAs on TWO alternating accumulators per position, then the next pair
//   MODE 2  + the A operand of the next position is written by the slot's VALU (split asm: and/and/perm/sub/sub)
//   MODE 3  + the B operand changes per position (register copies standing in for the weight loads)
//   MODE 4  MODE 2 with the VALU reading/writing registers produced by v_fma (the transform) two slots earlier
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_slot_probe.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int NV>
__global__ __launch_bounds__(256, 1) void loop(float* out, unsigned long long* ticks, int iters, float a0, float b0) {
    f32x16 acc[16];
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    u32x4 af[2][3], bf[4][3];
    for (int i = 0; i < 2; ++i) for (int p = 0; p < 3; ++p) for (int e = 0; e < 4; ++e) af[i][p][e] = 0x3f803f80u + threadIdx.x + i + p + e;
    for (int i = 0; i < 4; ++i) for (int p = 0; p < 3; ++p) for (int e = 0; e < 4; ++e) bf[i][p][e] = 0x3f003f80u + threadIdx.x * 3 + i + p + e;
    float tv[8];
    for (int i = 0; i < 8; ++i) tv[i] = a0 * i + threadIdx.x * 0.37f;
    unsigned sel = 0x07060302u + (unsigned)iters * 0;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int pos = 0; pos < 8; ++pos) {                      // 8 positions x 12 slots
            const int cur = pos & 1, nxt = cur ^ 1;
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                const int q = k >> 1, n = k & 1;
                constexpr int PA[6] = {2, 1, 1, 0, 0, 0}, PB[6] = {0, 1, 0, 2, 1, 0};
                const int ai = MODE == 0 ? (pos * 12 + k) & 15 : (pos * 2 + n) & 15;
                const int bi = MODE >= 3 ? (pos & 1) * 2 + n : n;
                acc[ai] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[MODE >= 2 ? cur : 0][PA[q]]),
                                                                   __builtin_bit_cast(bf16x8, bf[bi][PB[q]]), acc[ai], 0, 0, 0);
                if (MODE <= 1) {
#pragma unroll
                    for (int f = 0; f < NV; ++f) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(tv[f & 7]));
                } else {
                    // the kernel's split of one channel pair: 5 (first step) or 6 (second step) instructions, results into the
                    // NEXT position's A fragment
                    const int pr = k & 3;
                    if (k < 4 && NV >= 5) {
                        unsigned p1; float r0, r1, h0, h1;
                        asm volatile("v_and_b32 %3, 0xffff0000, %5\n\tv_and_b32 %4, 0xffff0000, %6\n\tv_perm_b32 %0, %6, %5, %7\n\t"
                                     "v_sub_f32 %1, %5, %3\n\tv_sub_f32 %2, %6, %4"
                                     : "=&v"(p1), "=&v"(r0), "=&v"(r1), "=&v"(h0), "=&v"(h1) : "v"(tv[2 * pr]), "v"(tv[2 * pr + 1]), "s"(sel));
                        af[nxt][0][pr] = p1; tv[2 * pr] = r0 + (MODE == 4 ? a0 : 0.f); tv[2 * pr + 1] = r1;
                    } else if (k < 8 && NV >= 5) {
                        unsigned p2, p3; float h0, h1;
                        asm volatile("v_and_b32 %2, 0xffff0000, %4\n\tv_and_b32 %3, 0xffff0000, %5\n\tv_perm_b32 %0, %5, %4, %6\n\t"
                                     "v_sub_f32 %2, %4, %2\n\tv_sub_f32 %3, %5, %3\n\tv_perm_b32 %1, %3, %2, %6"
                                     : "=&v"(p2), "=&v"(p3), "=&v"(h0), "=&v"(h1) : "v"(tv[2 * pr]), "v"(tv[2 * pr + 1]), "s"(sel));
                        af[nxt][1][pr] = p2; af[nxt][2][pr] = p3;
                    } else if (NV >= 5) {
#pragma unroll
                        for (int f = 0; f < 4; ++f) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(tv[(k * 4 + f) & 7]) : "v"(a0), "v"(b0));
                    }
                    if (MODE >= 3 && k == 11) {
#pragma unroll
                        for (int p = 0; p < 3; ++p) bf[(pos & 1) * 2 + 0][p] += 1u;       // "new weights" for this slot pair
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 8; ++i) s += tv[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}
template <int MODE, int NV>
void run(float* out, unsigned long long* ticks, const char* what) {
    const int blocks = 256, iters = 200;
    hipLaunchKernelGGL((loop<MODE, NV>), dim3(blocks), dim3(256), 0, 0, out, ticks, 10, 1.f, 2.f);
    (void)hipDeviceSynchronize();
    hipLaunchKernelGGL((loop<MODE, NV>), dim3(blocks), dim3(256), 0, 0, out, ticks, iters, 1.f, 2.f);
    (void)hipDeviceSynchronize();
    unsigned long long t;
    (void)hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
    printf("mode %d NV %d  %-60s %6.2f cycles per MFMA\n", MODE, NV, what, (double)t / (iters * 96.0));
}
int main() {
    float* out; unsigned long long* ticks;
    (void)hipMalloc(&out, 4096 * 256 * 4); (void)hipMalloc(&ticks, 8);
    run<0, 0>(out, ticks, "16 accumulators round-robin, no VALU");
    run<0, 5>(out, ticks, "16 accumulators round-robin, 5 v_and");
    run<1, 0>(out, ticks, "2 alternating accumulators per position, no VALU");
    run<1, 5>(out, ticks, "2 alternating accumulators, 5 v_and");
    run<1, 6>(out, ticks, "2 alternating accumulators, 6 v_and");
    run<2, 0>(out, ticks, "A fragment double-buffered, no VALU");
    run<2, 5>(out, ticks, "A fragment written by the split asm (5-6 per slot)");
    run<3, 5>(out, ticks, "+ B operand changing per position");
    run<4, 5>(out, ticks, "+ fma-produced split inputs");
    return 0;
}
