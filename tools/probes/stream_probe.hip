// HBM streaming probe for the elementwise passes (norm.hip: affine_act_fold): y = silu(x * A[c] + B[c]) over [frames][pixels][C] fp32,
// read + write once.  Which mapping / depth reaches the board's streaming rate (torch.mul out of place: 6.1 TB/s on this board)?
//   hipcc --offload-arch=gfx950 -O3 tools/probes/stream_probe.hip -o gpurun_out/stream_probe && gpurun_out/stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float silu_f(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }
template <bool ACT> __device__ __forceinline__ f32x4 one(f32x4 v, f32x4 A, f32x4 B) {
    f32x4 r = v * A + B;
    if (ACT) { r.x = silu_f(r.x); r.y = silu_f(r.y); r.z = silu_f(r.z); r.w = silu_f(r.w); }
    return r;
}
// V0: the product kernel's mapping: block = (pixel range, frame), thread = (pixel lane, channel quad), DEPTH loads in flight
template <bool ACT, int DEPTH, bool NT>
__global__ __launch_bounds__(256) void v0(const float* __restrict__ x, const float* __restrict__ Aa, const float* __restrict__ Ba, int C, int HW, int per,
                                          float* __restrict__ y) {
    const int n = blockIdx.y, tpp = C >> 2, ppi = 256 / tpp, tid = threadIdx.x, pl = tid / tpp, c = (tid - pl * tpp) * 4;
    const f32x4 A = *reinterpret_cast<const f32x4*>(Aa + n * C + c), B = *reinterpret_cast<const f32x4*>(Ba + n * C + c);
    const float* src = x + (size_t)n * HW * C + c;
    float* dst = y + (size_t)n * HW * C + c;
    const int p_begin = blockIdx.x * per, p_end = min(HW, p_begin + per);
    for (int p = p_begin + pl; p + (DEPTH - 1) * ppi < p_end; p += DEPTH * ppi) {
        f32x4 v[DEPTH];
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) v[u] = NT ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + (size_t)(p + u * ppi) * C))
                                                  : *reinterpret_cast<const f32x4*>(src + (size_t)(p + u * ppi) * C);
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) *reinterpret_cast<f32x4*>(dst + (size_t)(p + u * ppi) * C) = one<ACT>(v[u], A, B);
    }
}
// V1: flat mapping, no loop: block = 256 threads x UN float4 = UN KiB-rows of 4 KiB; (256 % (C/4) == 0: a thread's quad is fixed)
template <bool ACT, int UN, bool NT>
__global__ __launch_bounds__(256) void v1(const float* __restrict__ x, const float* __restrict__ Aa, const float* __restrict__ Ba, int C, size_t per_frame4,
                                          float* __restrict__ y) {
    const size_t base = (size_t)blockIdx.x * 256 * UN + threadIdx.x;
    const int n = (int)(base / per_frame4), c = (int)(threadIdx.x % (C >> 2)) * 4;
    const f32x4 A = *reinterpret_cast<const f32x4*>(Aa + n * C + c), B = *reinterpret_cast<const f32x4*>(Ba + n * C + c);
    f32x4 v[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) v[u] = NT ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(x) + base + u * 256) : reinterpret_cast<const f32x4*>(x)[base + u * 256];
#pragma unroll
    for (int u = 0; u < UN; ++u) reinterpret_cast<f32x4*>(y)[base + u * 256] = one<ACT>(v[u], A, B);
}
// V2: persistent grid-stride over 4 KiB rows, 2 rows in flight
template <bool ACT, bool NT>
__global__ __launch_bounds__(256) void v2(const float* __restrict__ x, const float* __restrict__ Aa, const float* __restrict__ Ba, int C, size_t per_frame4, size_t total4,
                                          float* __restrict__ y) {
    const int c = (int)(threadIdx.x % (C >> 2)) * 4;
    for (size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; i < total4; i += (size_t)gridDim.x * 1024) {
        const int n = (int)(i / per_frame4);
        const f32x4 A = *reinterpret_cast<const f32x4*>(Aa + n * C + c), B = *reinterpret_cast<const f32x4*>(Ba + n * C + c);
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = NT ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(x) + i + u * 256) : reinterpret_cast<const f32x4*>(x)[i + u * 256];
#pragma unroll
        for (int u = 0; u < 4; ++u) reinterpret_cast<f32x4*>(y)[i + u * 256] = one<ACT>(v[u], A, B);
    }
}
int main() {
    const int nfr = 128, HW = 4096, NB = 3;
    for (int C : {128, 256, 384}) {
        const int HWc = C == 128 ? 4096 : C == 256 ? 1024 : 4096;
        const size_t n = (size_t)nfr * HWc * C;
        std::vector<float*> xs(NB), ys(NB);
        for (int i = 0; i < NB; ++i) { (void)hipMalloc(&xs[i], n * 4); (void)hipMalloc(&ys[i], n * 4); (void)hipMemset(xs[i], 0x3c, n * 4); }
        float *A, *B; (void)hipMalloc(&A, nfr * C * 4); (void)hipMalloc(&B, nfr * C * 4); (void)hipMemset(A, 0x3c, nfr * C * 4); (void)hipMemset(B, 0, nfr * C * 4);
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        auto time = [&](const char* name, auto launch) {
            for (int i = 0; i < 3; ++i) launch(xs[i % NB], ys[i % NB]);
            (void)hipEventRecord(e0);
            const int reps = 30;
            for (int i = 0; i < reps; ++i) launch(xs[i % NB], ys[i % NB]);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            printf("C=%3d HW=%4d %-44s %7.1f us  %6.2f TB/s\n", C, HWc, name, ms / reps * 1e3, 2.0 * n * 4 / (ms / reps * 1e-3) / 1e12);
        };
        const int tpp = C / 4, ppi = 256 / tpp, threads = ppi * tpp;
        int split = 1;
        while (nfr * split < 2048 && HWc / (split * 2) >= ppi * 16) split *= 2;
        const int per = (HWc + split - 1) / split;
        const size_t pf4 = (size_t)HWc * C / 4, tot4 = n / 4;
        if (256 % tpp == 0) {
            time("v0 product mapping, 4 deep, silu, nt", [&](float* x, float* y) { hipLaunchKernelGGL((v0<true, 4, true>), dim3(split, nfr), dim3(threads), 0, 0, x, A, B, C, HWc, per, y); });
            time("v0 4 deep, silu, plain loads", [&](float* x, float* y) { hipLaunchKernelGGL((v0<true, 4, false>), dim3(split, nfr), dim3(threads), 0, 0, x, A, B, C, HWc, per, y); });
            time("v0 4 deep, NO silu, nt", [&](float* x, float* y) { hipLaunchKernelGGL((v0<false, 4, true>), dim3(split, nfr), dim3(threads), 0, 0, x, A, B, C, HWc, per, y); });
            time("v0 8 deep, silu, nt", [&](float* x, float* y) { hipLaunchKernelGGL((v0<true, 8, true>), dim3(split, nfr), dim3(threads), 0, 0, x, A, B, C, HWc, per, y); });
            time("v0 2 deep, silu, nt", [&](float* x, float* y) { hipLaunchKernelGGL((v0<true, 2, true>), dim3(split, nfr), dim3(threads), 0, 0, x, A, B, C, HWc, per, y); });
            time("v0 4 deep, silu, nt, split x4", [&](float* x, float* y) { hipLaunchKernelGGL((v0<true, 4, true>), dim3(split * 4, nfr), dim3(threads), 0, 0, x, A, B, C, HWc, per / 4, y); });
            time("v1 flat, 4 per thread, silu", [&](float* x, float* y) { hipLaunchKernelGGL((v1<true, 4, false>), dim3(tot4 / 1024), dim3(256), 0, 0, x, A, B, C, pf4, y); });
            time("v1 flat, 4 per thread, silu, nt", [&](float* x, float* y) { hipLaunchKernelGGL((v1<true, 4, true>), dim3(tot4 / 1024), dim3(256), 0, 0, x, A, B, C, pf4, y); });
            time("v1 flat, 4 per thread, NO silu", [&](float* x, float* y) { hipLaunchKernelGGL((v1<false, 4, false>), dim3(tot4 / 1024), dim3(256), 0, 0, x, A, B, C, pf4, y); });
            time("v1 flat, 8 per thread, silu", [&](float* x, float* y) { hipLaunchKernelGGL((v1<true, 8, false>), dim3(tot4 / 2048), dim3(256), 0, 0, x, A, B, C, pf4, y); });
            time("v1 flat, 2 per thread, silu", [&](float* x, float* y) { hipLaunchKernelGGL((v1<true, 2, false>), dim3(tot4 / 512), dim3(256), 0, 0, x, A, B, C, pf4, y); });
            time("v2 grid-stride 2048 blocks, silu", [&](float* x, float* y) { hipLaunchKernelGGL((v2<true, false>), dim3(2048), dim3(256), 0, 0, x, A, B, C, pf4, tot4, y); });
            time("v2 grid-stride 1024 blocks, silu, nt", [&](float* x, float* y) { hipLaunchKernelGGL((v2<true, true>), dim3(1024), dim3(256), 0, 0, x, A, B, C, pf4, tot4, y); });
        }
        time("hipMemcpyDtoD", [&](float* x, float* y) { (void)hipMemcpyAsync(y, x, n * 4, hipMemcpyDeviceToDevice, 0); });
        for (int i = 0; i < NB; ++i) { (void)hipFree(xs[i]); (void)hipFree(ys[i]); }
        (void)hipFree(A); (void)hipFree(B);
    }
    return 0;
}
