// Probe of v_dot2c_f32_bf16 / v_cvt_pk_bf16_f32 on gfx950: semantics of the constant operand and issue rate.
//   hipcc --offload-arch=gfx950 -O3 tools/dot2_probe.hip -o /tmp/dot2_probe && /tmp/dot2_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__global__ void sem(const float* x, float* o, unsigned mlo, unsigned mhi) {
    f32x2 v = {x[0], x[1]};
    bf16x2 p = __builtin_convertvector(v, bf16x2);
    const bf16x2 m0 = {(__bf16)-1.0f, (__bf16)0.0f}, m1 = {(__bf16)0.0f, (__bf16)-1.0f};
    o[0] = __builtin_amdgcn_fdot2_f32_bf16(p, m0, v.x, false);              // compiler's constant form
    o[1] = __builtin_amdgcn_fdot2_f32_bf16(p, m1, v.y, false);
    o[2] = __builtin_amdgcn_fdot2_f32_bf16(p, __builtin_bit_cast(bf16x2, mlo), v.x, false);   // register form
    o[3] = __builtin_amdgcn_fdot2_f32_bf16(p, __builtin_bit_cast(bf16x2, mhi), v.y, false);
    unsigned pb = __builtin_bit_cast(unsigned, p);
    o[4] = __builtin_bit_cast(float, pb << 16);
    o[5] = __builtin_bit_cast(float, pb & 0xffff0000u);
}

template <int MODE>
__global__ void rate(float* o, unsigned mlo, long long* cyc) {
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i;
    unsigned q = mlo + threadIdx.x;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < 1024; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(a[i]) : "v"(q), "v"(mlo));
            if (MODE == 1) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(q), "v"(mlo));
            if (MODE == 2) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "+v"(a[i]) : "v"(q), "v"(mlo));
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    o[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[MODE] = t1 - t0;
}

int main() {
    float hx[2] = {1.2345678f, -3.1415927e-3f}, *dx, *dout;
    long long* dc;
    hipMalloc(&dx, 8); hipMalloc(&dout, 4096); hipMalloc(&dc, 64);
    hipMemcpy(dx, hx, 8, hipMemcpyHostToDevice);
    sem<<<1, 1>>>(dx, dout, 0x0000bf80u, 0xbf800000u);
    float ho[6];
    hipMemcpy(ho, dout, 24, hipMemcpyDeviceToHost);
    printf("x = %.9g %.9g   p(as f32) = %.9g %.9g   exact r = %.9g %.9g\n", hx[0], hx[1], ho[4], ho[5], hx[0] - ho[4], hx[1] - ho[5]);
    printf("const form: %.9g %.9g    register form: %.9g %.9g\n", ho[0], ho[1], ho[2], ho[3]);
    rate<0><<<1, 64>>>(dout, 0x0000bf80u, dc);
    rate<1><<<1, 64>>>(dout, 0x0000bf80u, dc);
    rate<2><<<1, 64>>>(dout, 0x0000bf80u, dc);
    long long hc[3];
    hipMemcpy(hc, dc, 24, hipMemcpyDeviceToHost);
    printf("cycles per instruction (one wave, 8 independent chains): dot2c %.2f  fmac %.2f  cvt_pk_bf16 %.2f\n", hc[0] / 8192.0, hc[1] / 8192.0,
           hc[2] / 8192.0);
    return 0;
}
