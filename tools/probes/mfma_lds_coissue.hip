// What an un-waited LDS / memory instruction costs the bf16 matrix pipe (gfx950).  Every wave loops
//   { v_mfma_f32_32x32x16_bf16 ; NF instructions of one kind }  over 4 accumulator tiles,
// WPS waves per SIMD (block = 256 * WPS threads, one block per CU).  Printed: cycles per MFMA *per SIMD*
// (32 = the pipe is never idle).  Nothing waits on the results inside the loop except the counter limits.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_lds_coissue.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
static const char* NAMES[] = {"ds_read_b128", "ds_write_b64", "ds_write_b128", "ds_read_b64", "buffer_load_b128 (L2)", "buffer_load_lds b128", "v_and_b32 (dependent)",
                              "ds_read_b128 + 4 VALU dep", "ds_write_b64 + 4 VALU dep"};
template <int NF, int MODE, int WPS>
__global__ __launch_bounds__(256 * WPS, 1) void loop(float* out, const float* src, unsigned long long* ticks, int iters) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(1.f + threadIdx.x + i); b[i] = (__bf16)(2.f * i); }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned la = wv * 4096 + lane * 16;                  // conflict-free for every width used here
    f32x4 r[8];
    for (int i = 0; i < 8; ++i) r[i] = f32x4{1.f * i, 2.f, 3.f, 4.f};
    float x = threadIdx.x;
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, 1 << 20, 0x00020000);
    typedef __attribute__((address_space(3))) void* lds_ptr;
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            acc[k & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[k & 3], 0, 0, 0);
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const int slot = (k * NF + f) & 7;
                if (MODE == 0 || MODE == 7) asm volatile("ds_read_b128 %0, %1" : "=v"(r[slot]) : "v"(la + slot * 1024) : "memory");
                if (MODE == 1 || MODE == 8) asm volatile("ds_write_b64 %0, %1" : : "v"(la / 2 + slot * 512), "v"(__builtin_shufflevector(r[0], r[0], 0, 1)) : "memory");
                if (MODE == 2) asm volatile("ds_write_b128 %0, %1" : : "v"(la + slot * 1024), "v"(r[0]) : "memory");
                if (MODE == 3) asm volatile("ds_read_b64 %0, %1" : "=v"(*reinterpret_cast<double*>(&r[slot])) : "v"(la / 2 + slot * 512) : "memory");
                if (MODE == 4) r[slot] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, threadIdx.x * 16u, slot * 8192 + (blockIdx.x & 7) * 65536, 0));
#if defined(__HIP_DEVICE_COMPILE__)
                if (MODE == 5) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr)(lds + 65536 + wv * 1024 + slot * 8192), 16, threadIdx.x * 16u, slot * 8192, 0, 0);
#endif
                if (MODE == 6) asm volatile("v_and_b32 %0, 0xffff0000, %0" : "+v"(x));
                if (MODE == 7 || MODE == 8)
                    asm volatile("v_and_b32 %0, 0xffff0000, %0\n\tv_and_b32 %0, 0xffff0000, %0\n\tv_and_b32 %0, 0xffff0000, %0\n\tv_and_b32 %0, 0xffff0000, %0" : "+v"(x));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE == 4 || MODE == 5) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = x;
    for (int i = 0; i < 4; ++i) for (int q = 0; q < 16; ++q) s += acc[i][q];
    for (int i = 0; i < 8; ++i) s += r[i].x + r[i].y + r[i].z + r[i].w;
    out[blockIdx.x * 256 * WPS + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}
template <int NF, int MODE, int WPS>
double run(float* out, float* src, unsigned long long* ticks) {
    const int blocks = 256, iters = 100;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&loop<NF, MODE, WPS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((loop<NF, MODE, WPS>), dim3(blocks), dim3(256 * WPS), 144 * 1024, 0, out, src, ticks, 5);
    (void)hipDeviceSynchronize();
    hipLaunchKernelGGL((loop<NF, MODE, WPS>), dim3(blocks), dim3(256 * WPS), 144 * 1024, 0, out, src, ticks, iters);
    (void)hipDeviceSynchronize();
    unsigned long long t;
    (void)hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
    return (double)t / (iters * 32.0 * WPS);
}
template <int MODE>
void sweep(float* out, float* src, unsigned long long* ticks) {
    printf("%-28s per MFMA 0/1/2/3:  1 wave/SIMD %6.1f %6.1f %6.1f %6.1f   2 waves/SIMD %6.1f %6.1f %6.1f %6.1f\n", NAMES[MODE],
           run<0, MODE, 1>(out, src, ticks), run<1, MODE, 1>(out, src, ticks), run<2, MODE, 1>(out, src, ticks), run<3, MODE, 1>(out, src, ticks),
           run<0, MODE, 2>(out, src, ticks), run<1, MODE, 2>(out, src, ticks), run<2, MODE, 2>(out, src, ticks), run<3, MODE, 2>(out, src, ticks));
}
int main() {
    float *out, *src; unsigned long long* ticks;
    (void)hipMalloc(&out, 4096 * 512 * 4); (void)hipMalloc(&src, 1 << 21); (void)hipMalloc(&ticks, 8);
    (void)hipMemset(src, 0, 1 << 21);
    sweep<0>(out, src, ticks); sweep<1>(out, src, ticks); sweep<2>(out, src, ticks); sweep<3>(out, src, ticks); sweep<4>(out, src, ticks);
    sweep<5>(out, src, ticks); sweep<6>(out, src, ticks); sweep<7>(out, src, ticks); sweep<8>(out, src, ticks);
    return 0;
}
