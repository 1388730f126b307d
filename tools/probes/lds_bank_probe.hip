// ds_read_b128 bank-conflict probe: cycles per instruction for a given per-lane byte-address pattern (one wave per SIMD,
// reads kept in flight, no dependent use).   hipcc --offload-arch=gfx950 -O3 tools/lds_bank_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256, 1) void probe(const int* __restrict__ addr, unsigned long long* ticks, float* sink, int iters) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 16384; i += 256) lds[i] = (float)i;
    __syncthreads();
    const unsigned a = (unsigned)addr[threadIdx.x & 63];
    f32x4 acc = {0, 0, 0, 0};
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        f32x4 v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) asm volatile("ds_read_b128 %0, %1" : "=v"(v[k]) : "v"(a));
        asm volatile("s_waitcnt lgkmcnt(0)");
#pragma unroll
        for (int k = 0; k < 8; ++k) acc += v[k];
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    sink[blockIdx.x * 256 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
    if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
}
int main() {
    int* d_addr; unsigned long long* d_t; float* d_s;
    (void)hipMalloc(&d_addr, 64 * 4); (void)hipMalloc(&d_t, 8); (void)hipMalloc(&d_s, 256 * 256 * 4);
    auto run = [&](const char* name, std::vector<int> a) {
        (void)hipMemcpy(d_addr, a.data(), 64 * 4, hipMemcpyHostToDevice);
        const int iters = 2000;
        hipLaunchKernelGGL(probe, dim3(256), dim3(256), 65536, 0, d_addr, d_t, d_s, iters);
        (void)hipDeviceSynchronize();
        unsigned long long t; (void)hipMemcpy(&t, d_t, 8, hipMemcpyDeviceToHost);
        printf("%-44s %6.1f cycles per ds_read_b128 (4 waves per CU issuing)\n", name, (double)t / (iters * 8.0));
    };
    std::vector<int> a(64);
    for (int l = 0; l < 64; ++l) a[l] = l * 16; run("linear lane*16", a);
    for (int l = 0; l < 64; ++l) a[l] = l * 80; run("stride 80 B", a);
    for (int l = 0; l < 64; ++l) a[l] = l * 160; run("stride 160 B", a);
    for (int l = 0; l < 64; ++l) a[l] = l * 32; run("stride 32 B", a);
    for (int l = 0; l < 64; ++l) a[l] = l * 64; run("stride 64 B", a);
    for (int l = 0; l < 64; ++l) a[l] = l * 128; run("stride 128 B", a);
    for (int l = 0; l < 64; ++l) a[l] = l * 256; run("stride 256 B (all one slot)", a);
    // old patch image: tile (tx, ty) of a 16x16-pixel block, pixel (2ty, 2tx), 80 B per pixel, 18 pixels per row
    for (int l = 0; l < 64; ++l) { int r = l & 31, h = l >> 5, tx = r & 7, ty = r >> 3; a[l] = ((2 * ty) * 18 + 2 * tx) * 80 + h * 32; }
    run("old image (linear pixels, 80 B)", a);
    // new image: even/odd planes, row stride 1472 B
    for (int l = 0; l < 64; ++l) { int r = l & 31, h = l >> 5, tx = r & 7, ty = r >> 3; a[l] = (2 * ty) * 1472 + tx * 80 + h * 32; }
    run("new image (planes, rows 1472 B)", a);
    for (int l = 0; l < 64; ++l) { int r = l & 31, h = l >> 5, tx = r & 7, ty = r >> 3; a[l] = (2 * ty) * 1472 + tx * 80 + h * 16; }
    run("new image, halves 16 B apart", a);
    for (int l = 0; l < 64; ++l) { int r = l & 31, h = l >> 5, tx = r & 7, ty = r >> 3; a[l] = (2 * ty) * 1472 + tx * 80 + h * 40; }
    run("new image, halves 40 B apart (unaligned!)", a);
    for (int l = 0; l < 64; ++l) { int r = l & 31, h = l >> 5, tx = r & 7, ty = r >> 3; a[l] = (2 * ty) * 1408 + tx * 64 + h * 32 ; }
    run("planes, 64 B pixels, rows 1408 B", a);
    for (int l = 0; l < 64; ++l) { int r = l & 31, h = l >> 5; a[l] = r * 80 + h * 32; }
    run("32 consecutive pixels of 80 B, halves +32", a);
    for (int l = 0; l < 64; ++l) { int r = l & 31, h = l >> 5; a[l] = r * 80 + h * 2560; }
    run("32 consecutive pixels of 80 B, halves +2560", a);
    // which 16 lanes share a pass?  slot = 16-byte slot within the 256 B of the 64 banks
    for (int l = 0; l < 64; ++l) a[l] = ((l & 7) + 8 * ((l >> 5) & 1)) * 16 + (l >> 3 & 3) * 4096; run("slots (l&7)+8*(l>>5): free iff pass = {8 lanes, +32}", a);
    for (int l = 0; l < 64; ++l) a[l] = ((l & 7) + 8 * ((l >> 4) & 1)) * 16 + (l >> 3 & 1) * 4096 + (l >> 5) * 8192; run("slots (l&7)+8*(l>>4&1): free iff pass = {8 lanes, +16}", a);
    for (int l = 0; l < 64; ++l) a[l] = (l & 15) * 16 + (l >> 4) * 4096; run("slots l&15: free iff pass = 16 contiguous lanes", a);
    for (int l = 0; l < 64; ++l) a[l] = ((l & 3) + 4 * (l >> 4)) * 16 + (l >> 2 & 3) * 4096; run("slots (l&3)+4*(l>>4): free iff pass = {4 lanes of each 16}", a);
    // V fragment image of conv_wino_split64: lane (tile lr, k-half lh) reads slot lr*4 + ((2*lh+e) ^ ((lr>>2)&3))
    for (int l = 0; l < 64; ++l) { int lr = l & 31, lh = l >> 5; a[l] = lr * 64 + (((2 * lh) ^ ((lr >> 2) & 3)) * 16); }
    run("V fragment read, swizzle (lr>>2)&3", a);
    for (int l = 0; l < 64; ++l) { int lr = l & 31, lh = l >> 5; a[l] = lr * 64 + (((2 * lh) ^ ((lr >> 2) & 3) ^ ((lr >> 4) & 1)) * 16); }
    run("V fragment read, swizzle (lr>>2)&3 ^ lr>>4", a);
    for (int l = 0; l < 64; ++l) { int lr = l & 31, lh = l >> 5; a[l] = lr * 80 + lh * 32; }
    run("V fragment read, 80 B rows", a);
    // conv_wino_s64: block transform reads of the raw image, V writes, staging stores
    for (int l = 0; l < 64; ++l) { int t = l >> 2, kq = l & 3, tx = t & 7, ty = t >> 3; a[l] = 2 * ty * 1280 + tx * 64 + kq * 16; }
    run("s64 raw read, 8x8 tiles (rows 1280 B)", a);
    for (int l = 0; l < 64; ++l) { int t = l >> 2, kq = l & 3, tx = t & 3, ty = t >> 2; a[l] = 2 * ty * 768 + tx * 64 + kq * 16; }
    run("s64 raw read, 4x4 tiles (rows 768 B)", a);
    for (int l = 0; l < 64; ++l) { int t = l >> 2, kq = l & 3; a[l] = t * 64 + ((kq ^ ((t >> 2) & 3)) * 16); }
    run("s64 V write (swizzled slots)", a);
    for (int l = 0; l < 64; ++l) { int pl = l >> 2, lq = l & 3, py = pl / 18, px = pl % 18; a[l] = py * 1280 + (px & 1) * 640 + (px >> 1) * 64 + lq * 16; }
    run("s64 staging store, 18-wide patch", a);
    for (int l = 0; l < 64; ++l) { int pl = l >> 2, lq = l & 3, py = pl / 10, px = pl % 10; a[l] = py * 768 + (px & 1) * 384 + (px >> 1) * 64 + lq * 16; }
    run("s64 staging store, 10-wide patch", a);
    return 0;
}
