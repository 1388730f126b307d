// Timing-only prototype of the GEMM phase of a Winograd F(2x2,3x3) convolution on fp32 MFMA (gfx950).
// Question it answers: with 16 Winograd positions x (M tiles) x (N couts) per block, V tile in LDS and the
// transformed weights streamed fragment-major from L2, does the loop stay MFMA-bound?  (No transforms, fake data.)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/wino_proto.hip -o /tmp/wino_proto && /tmp/wino_proto
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int KC = 16;          // channels per chunk
constexpr int VLD = KC + 4;     // V row stride (floats)

// block: 512 threads = 8 waves; M = 64 tiles (16x16 output pixels), BN = 64 couts.
// wave w: xi in {2w, 2w+1} x m-tile {0,1} x n-tile {0,1} = 8 accumulators.
__global__ __launch_bounds__(512, 2) void wino_gemm_proto(const float* __restrict__ U, const float* __restrict__ src,
                                                          float* __restrict__ out, int nchunk, int ncoblk) {
    extern __shared__ __attribute__((aligned(16))) float V[];     // [2][16][64][VLD]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 31, lh = lane >> 5;
    f32x16 acc[2][2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[x][m][n][r] = 0.f;
    const int cob0 = blockIdx.y * 2;
    const float* ul = U + lane * 4;
    // stand-in for the input transform: every thread writes 8 float4 of the next V tile per chunk
    auto fill = [&](float* Vd, int chunk) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(src + ((size_t)blockIdx.x * nchunk + chunk) * 2048 + tid * 4);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int idx = tid + e * 512;                    // (xi*64 + tile)*4 + quad
            *reinterpret_cast<f32x4*>(Vd + (idx >> 2) * VLD + (idx & 3) * 4) = v * (float)(e + 1);
        }
    };
    f32x4 bfr[2][2][2];                                      // [kg parity][xi][n]
    auto b_load = [&](int slot, int chunk, int kg) {
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int n = 0; n < 2; ++n)
                bfr[slot][x][n] = *reinterpret_cast<const f32x4*>(
                    ul + ((((size_t)chunk * 16 + wave * 2 + x) * ncoblk + cob0 + n) * 2 + kg) * 256);
    };
    fill(V, 0);
    b_load(0, 0, 0);
    __syncthreads();
    for (int chunk = 0; chunk < nchunk; ++chunk) {
        const int nxt = min(chunk + 1, nchunk - 1);
        const float* Vc = V + (chunk & 1) * 16 * 64 * VLD;
        float* Vn = V + ((chunk + 1) & 1) * 16 * 64 * VLD;
#pragma unroll
        for (int kg = 0; kg < 2; ++kg) {
            if (kg == 0) b_load(1, chunk, 1); else b_load(0, nxt, 0);
            f32x4 afr[2][2];
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int m = 0; m < 2; ++m)
                    afr[x][m] = *reinterpret_cast<const f32x4*>(Vc + ((wave * 2 + x) * 64 + m * 32 + lr) * VLD + kg * 8 + lh * 4);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int x = 0; x < 2; ++x)
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int n = 0; n < 2; ++n)
                            acc[x][m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(afr[x][m][e], bfr[kg][x][n][e], acc[x][m][n], 0, 0, 0);
            if (kg == 1) fill(Vn, nxt);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }
    float s = 0.f;
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) s += acc[x][m][n][r];
    out[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 512 + tid] = s;
}

int main() {
    struct Shape { int nfr, H, Cin, Cout; } shapes[] = {{128, 64, 128, 128}, {128, 32, 256, 256}, {128, 16, 384, 384}, {128, 64, 256, 128}};
    for (auto sh : shapes) {
        const int tiles = sh.nfr * (sh.H / 2) * (sh.H / 2);
        const int mblocks = tiles / 64, nblocks = sh.Cout / 64, nchunk = sh.Cin / KC, ncoblk = sh.Cout / 32;
        float *U, *src, *out;
        const size_t usz = (size_t)nchunk * 16 * ncoblk * 2 * 256;
        hipMalloc(&U, usz * 4); hipMalloc(&src, (size_t)mblocks * nchunk * 2048 * 4); hipMalloc(&out, (size_t)mblocks * nblocks * 512 * 4);
        hipMemset(U, 0, usz * 4); hipMemset(src, 0, (size_t)mblocks * nchunk * 2048 * 4);
        const size_t lds = 2 * 16 * 64 * VLD * 4;
        hipFuncSetAttribute((const void*)wino_gemm_proto, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(a);
            for (int i = 0; i < 10; ++i)
                hipLaunchKernelGGL(wino_gemm_proto, dim3(mblocks, nblocks), dim3(512), lds, 0, U, src, out, nchunk, ncoblk);
            hipEventRecord(b); hipEventSynchronize(b);
        }
        float ms; hipEventElapsedTime(&ms, a, b); ms /= 10;
        const double direct = 2.0 * sh.nfr * sh.H * sh.H * (double)sh.Cout * sh.Cin * 9;
        printf("N=%d %dx%d %d->%d: %.1f us  winograd-MFMA %.1f TFLOP/s  direct-equivalent %.1f TFLOP/s  (LDS %zu B, err=%s)\n", sh.nfr, sh.H,
               sh.H, sh.Cin, sh.Cout, ms * 1e3, direct / 2.25 / ms / 1e9, direct / ms / 1e9, lds, hipGetErrorString(hipGetLastError()));
        hipFree(U); hipFree(src); hipFree(out);
    }
    return 0;
}
