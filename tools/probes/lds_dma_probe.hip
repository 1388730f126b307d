// buffer_load_dwordx4 ... lds (LDS-DMA) semantics on gfx950 through __builtin_amdgcn_raw_ptr_buffer_load_lds:
// lane l of the wave writes 16 bytes at  M0 base + l*16  whatever its (gathered) global offset; lanes whose offset fails
// the descriptor's range check write zeros.   hipcc --offload-arch=gfx950 -O3 tools/lds_dma_probe.hip -o /tmp/p && /tmp/p
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* x, float* o, int bytes) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 4096; i += 256) sm[i] = -1.f;
    __syncthreads();
    auto r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, bytes, 0x00020000);
    // wave w: reversed gather; lanes 60..63 out of range; second instruction with an scalar offset of 1024 bytes
    const unsigned vo = lane < 60 ? (unsigned)(63 - lane) * 16u + w * 1024u : 0x80000000u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(sm + w * 512), 16, vo, 0, 0, 0);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(sm + 2048 + w * 512), 16, vo, 4096, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = tid; i < 4096; i += 256) o[i] = sm[i];
}
int main() {
    std::vector<float> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = (float)i;
    float *dx, *dout;
    (void)hipMalloc(&dx, 16384); (void)hipMalloc(&dout, 16384);
    (void)hipMemcpy(dx, h.data(), 16384, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(256), 16384, 0, dx, dout, 16384);
    std::vector<float> o(4096);
    (void)hipMemcpy(o.data(), dout, 16384, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int w = 0; w < 4; ++w)
        for (int l = 0; l < 64; ++l)
            for (int e = 0; e < 4; ++e) {
                const float exp0 = l < 60 ? (float)(w * 256 + (63 - l) * 4 + e) : 0.f;
                const float exp1 = l < 60 ? (float)(1024 + w * 256 + (63 - l) * 4 + e) : 0.f;
                const float g0 = o[w * 512 + l * 4 + e], g1 = o[2048 + w * 512 + l * 4 + e];
                if (g0 != exp0 || g1 != exp1) { if (bad < 8) printf("wave %d lane %d e %d: got %g %g expected %g %g\n", w, l, e, g0, g1, exp0, exp1); ++bad; }
            }
    for (int w = 0; w < 4; ++w) if (o[w * 512 + 256] != -1.f) { printf("wave %d wrote past its 1 KiB\n", w); ++bad; }
    printf(bad ? "MISMATCH (%d)\n" : "LDS-DMA: lane l -> base + 16*l, gathered, out-of-range lanes write zeros: OK\n", bad);
    return bad != 0;
}
