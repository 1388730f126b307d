// Common definitions for the gfx950 kernels and the step engine.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <string>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace vd {

// Error plumbing: every C-ABI entry returns 0 or a negative code; text via vd_last_error().
void set_error(const std::string& msg);
extern thread_local std::string g_last_error;

#define VD_HIP(call)                                                                   \
    do {                                                                               \
        hipError_t e_ = (call);                                                        \
        if (e_ != hipSuccess) {                                                        \
            vd::set_error(std::string(#call) + ": " + hipGetErrorString(e_));          \
            return -2;                                                                 \
        }                                                                              \
    } while (0)

#define VD_REQUIRE(cond, msg)                                                          \
    do {                                                                               \
        if (!(cond)) {                                                                 \
            vd::set_error(std::string("requirement failed: ") + #cond + " -- " + msg); \
            return -1;                                                                 \
        }                                                                              \
    } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize belongs to a DEVICE's copy of a function: raise it once per (launch site, device) -- a process
// that drives several devices must not find the flag of device 0 set when it launches on device 1.  `st` is the site's static.
struct LdsAttr { size_t have[64] = {}; };
inline int raise_dynamic_lds(LdsAttr& st, const void* fn, size_t bytes) {
    int dev = 0;
    VD_HIP(hipGetDevice(&dev));
    VD_REQUIRE(dev >= 0 && dev < 64, "device ordinal beyond the per-device attribute table");
    if (bytes > st.have[dev]) {
        VD_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        st.have[dev] = bytes;
    }
    return 0;
}
#define VD_RAISE_LDS(fn, bytes)                                                                        \
    do {                                                                                               \
        static vd::LdsAttr lds_attr_;                                                                  \
        if (int rc_ = vd::raise_dynamic_lds(lds_attr_, reinterpret_cast<const void*>(fn), (bytes))) return rc_; \
    } while (0)

// ---------------------------------------------------------------- kernel argument blocks
// Implicit-GEMM convolution / linear layer on NHWC activations (see igemm.hip).
struct IgemmArgs {
    const float* src0;   // [nfr][Hs][Ws][C0]
    const float* src1;   // [nfr][Hs][Ws][Cin-C0] or null  (virtual channel concat, unet.py:826-828)
    int C0, Cin;
    int nfr, Hs, Ws;     // stored source dims
    int ups;             // 1: source is read through a nearest x2 upsample (unet.py:69)
    int nfr_sel = 0;     // frames of the WHOLE layer call when this launch is a part of one (a frame range of a big window, the compact batch
                         // of a window suffix): kernel variants that are chosen by grid size (split-K, conv_wino_z128.hip) are chosen for it,
                         // so a frame's arithmetic does not depend on which other frames share its launch.  0: nfr
    int ups_phase = 0;   // with ups = 1: wwino holds the sub-pixel image (pack_conv3_wino_ups); conv_wino_r64.hip
    // gemm_split.hip, 1x1 over a (virtually concatenated) input, 128x128 tile: the column blocks 0 also write
    // side[m][k] = SiLU(A[m][k] * sideA[frame][k] + sideB[frame][k]) -- the GroupNorm(+FiLM)+SiLU image the ResBlock's first
    // conv reads (norm.hip: affine_act_kernel) -- from the A tile they stage anyway; side_hw = rows per frame (% 128 == 0)
    float* side = nullptr; const float* sideA = nullptr; const float* sideB = nullptr; int side_hw = 0;
    int stride, pad, ksz;
    int Ho, Wo;
    const float* w;      // [ksz*ksz][Cout][Cin]           (generic kernel)
    const float* wfrag;  // [tap][Cin/32][Cout/32][4][64][4] fragment-major (3x3 halo kernel / linear), or null
    const float* wwino;  // Winograd-transformed 3x3 weights [Cin/16][16][Cout/32][2][64][4] (conv_wino.hip), or null
    const float* bias;   // [Cout] or null
    const float* affA;   // [nfr][Cin] per-frame per-channel scale  (GroupNorm/FiLM folded), or null
    const float* affB;   // [nfr][Cin] shift
    int act;             // 1: SiLU on the (affine-transformed) input
    const float* res;    // [M][res_ld] residual added in the epilogue, or null
    int res_ld;
    const float* fbias;  // [nfr][fbias_ld] per-frame bias (ResBlock without scale-shift: h + emb_out, unet.py:196), or null
    int fbias_ld;
    float* out;          // [M][ldo]
    int ldo;
    int Cout, M;
    // GroupNorm statistics of the OUTPUT, produced by the epilogue (conv_wino.hip only): per (frame, block-of-frame,
    // channel) [sum, sum of squares] as doubles, the layout gn_stats_partial writes -> the consumer's GroupNorm needs no
    // pass over the tensor.  stats_split = blocks per frame (conv_wino_stats_split).  Null: not produced.
    double* stats;
    int stats_split;
    // gemm_split.hip (linear layers whose rows are the pixels of consecutive frames): rows per frame; the table is then
    // [frame][stats_split = stats_hw / rows of a wave tile][Cout][2] (gemm_split_stats_rows)
    int stats_hw;
    // gemm_frag.hip only: zcount > 1 runs zcount independent problems of identical shape in one launch (blockIdx.z);
    // problem z reads src0 + z*zs_a, wfrag + z*zs_w, bias + z*zs_bias and writes out + z*zs_out (element strides).
    // The three RPE-net output layers of an attention block (unet.py:283-298) go out this way.
    int zcount, zs_a, zs_w, zs_bias, zs_out;
    // gemm_split.hip: problems whose weights do not sit at a constant stride (the RPE nets of ALL attention blocks of one width in one
    // launch, engine.hip: rpe_all): problem z takes wfrag = zbase + ztab[2z], bias = zbase + ztab[2z + 1] (float offsets, device table)
    const long long* ztab = nullptr; const float* zbase = nullptr;
    // wsplit == 2: wwino is the image of pack_conv3_wino_split (conv_wino_r64.hip).  Otherwise:
    // wfrag is the bf16-split image of the weights (gemm_split.hip: fp32 accuracy from six bf16 piece products);
    // zs_w then counts floats of that image as well
    int wsplit;
    // conv_wino_r64.hip, small grids only: scratch for split-K partial outputs (conv_wino_r64_ksplit_floats), or null
    float* ksplit_ws;
    size_t ksplit_ws_floats;
};

struct AttnSpatialArgs {
    const float* qkv;    // [nfr*L][3C]: q | k | v, each [heads][F]
    float* out;          // [nfr*L][C]
    int nfr, L, C, heads;
    float scale;
};

struct AttnTemporalArgs {
    const float* qkv;    // [B*T*HW][3C]
    const float* Rk;     // [B][T][T][C] or null
    const float* Rq;
    const float* Rv;
    const float* mask;   // [B][T] (1 = real frame) or null
    float* out;          // [B*T*HW][C]
    int B, T, HW, C, heads;
    int allow_pad;       // allow_interactions_between_padding
    float scale;
};

int launch_igemm(const IgemmArgs& a, hipStream_t s);
int igemm_frames_per_launch(const IgemmArgs& a);   // frames (rows) per launch: big windows are cut along the frame dimension
int igemm_tile_class(int M, int Cout);   // 0: 128x128, 1: 128x64, 2: 64x128, 3: 64x64
// linear / 1x1 path with fragment-major weights (gemm_frag.hip); wfrag = [K/32][N/32][4][64][4]
bool gemm_frag_supported(const IgemmArgs& a);
int launch_gemm_frag(const IgemmArgs& a, int tile_class, hipStream_t s);
void pack_linear_frag(const float* w, float* out_base, int rows, int K, int n_total, int row0);
// Winograd F(2x2,3x3) path (conv_wino.hip): 2.25x fewer MFMAs than the direct 3x3 kernels
bool conv_wino_supported(const IgemmArgs& a);
// rows of the whole layer call a launch belongs to (IgemmArgs::nfr_sel): what the tile classes are chosen for
inline int igemm_sel_M(const IgemmArgs& a) { return a.nfr_sel > a.nfr ? a.nfr_sel * a.Ho * a.Wo : a.M; }
int conv_wino_stats_split(int Hout);
// ---- exact three-way bf16 split of a pair of fp32 values (conv_wino_r64.hip, VD_MATH=bf16x6), plain VALU only: v_pk_*_f32
// and v_dot2c_f32_bf16 do not overlap the bf16 MFMA (tools/mfma_bf16_coissue.hip).  Every piece is rounded to NEAREST
// (v_cvt_pk_bf16_f32): rounds 1-3 truncated (v_and_b32 0xffff0000), which gave the three dropped piece products the sign of
// a*b -- a systematic bias of ~10 % of the mean error (tools/mode_check.py signed_mean_err) -- at the same instruction count.
// first half: p1 = bf16 pair of (x0, x1), r = x - p1 (exact in fp32)
__device__ __forceinline__ void split_a(float x0, float x1, unsigned& p1, float& r0, float& r1, unsigned) {
    float h0, h1;
    asm("v_cvt_pk_bf16_f32 %0, %5, %6\n\t"
        "v_lshlrev_b32 %3, 16, %0\n\t"
        "v_and_b32 %4, 0xffff0000, %0\n\t"
        "v_sub_f32 %1, %5, %3\n\t"
        "v_sub_f32 %2, %6, %4"
        : "=&v"(p1), "=&v"(r0), "=&v"(r1), "=&v"(h0), "=&v"(h1)
        : "v"(x0), "v"(x1));
}
// second half: p2 = bf16 pair of r, p3 = r - p2 (at most 8 significant bits are left: exactly a bf16 value)
__device__ __forceinline__ void split_b(float r0, float r1, unsigned& p2, unsigned& p3, unsigned) {
    float h0, h1;
    asm("v_cvt_pk_bf16_f32 %0, %4, %5\n\t"
        "v_lshlrev_b32 %2, 16, %0\n\t"
        "v_and_b32 %3, 0xffff0000, %0\n\t"
        "v_sub_f32 %2, %4, %2\n\t"
        "v_sub_f32 %3, %5, %3\n\t"
        "v_cvt_pk_bf16_f32 %1, %2, %3"
        : "=&v"(p2), "=&v"(p3), "=&v"(h0), "=&v"(h1)
        : "v"(r0), "v"(r1));
}

// ---- arithmetic of the matrix products (VD_MATH, read once per process; every rank of a job must agree: the packed weight
// image depends on it -- vd_weights_layout_id):
//   f16x3  (default)  an fp32 operand x is carried as TWO fp16 pieces, x ~ a0 + 2^-12 a1 with a0 = f16(x) (round to nearest
//                     even) and a1 = f16((x - a0) * 2^12) (x - a0 is exact in fp32): 22 significand bits, relative error
//                     <= 2^-22 for 2^-14 <= |x| <= 65504, absolute error <= 2^-37 below (tools/mfma_f16_coissue.hip measures
//                     2^-23 worst over 2^20 values per binade); |x| > 65504 becomes inf and the product NaN -- loudly, never
//                     silently.  A product a*b is THREE piece products on v_mfma_f32_32x32x16_f16 with fp32 accumulation:
//                     a0*b0 (exact in fp32: 11 x 11 bits) + a0*b1 + a1*(2^-12 b0); the dropped a1*b1 is <= 2^-22 |ab|.
//                     Weights are scaled per output channel by a power of two on the host (max |w s| in [2^13, 2^14): every
//                     piece, 2^-12 b0 included, sits in fp16's normal range down to 2^-15 of the row's largest weight) and the
//                     epilogue multiplies by 1/s -- exact.  Measured against fp64 the result is at least as close as the fp32
//                     MFMA's (max <= 1.06x, mean <= 1.02x: profiles/r04_split_accuracy.json; tests/test_gpu_ops.py asserts
//                     <= 1.5x / 1.25x): the rounding of the fp32 accumulator dominates, and every mode shares it.
//   bf16x6            the EXACT split: x = x1 + x2 + x3, three bf16 pieces, six piece products (the default of rounds 1-3).
//   fp32              every product on v_mfma_f32_32x32x2_f32 (gemm_frag.hip, conv_wino.hip).
enum { MATH_F16X3 = 0, MATH_BF16X6 = 1, MATH_FP32 = 2 };
int math_mode();
inline bool f16_math() { return math_mode() == MATH_F16X3; }
// Split weight images (gemm_split.hip, conv_wino_r64.hip) are followed by a trailer of 2 * N floats, N = output channels of
// the image: [N] the power-of-two scale s the row was multiplied with (1 in bf16x6), [N] its reciprocal.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
// device: the two fp16 pieces of a pair of fp32 values.  p0 = {f16(x0), f16(x1)}; p1 = {f16((x0 - p0.lo) * 4096), f16(...)}:
// v_fma_mix_f32 reads the fp16 half directly (x - a0 in one instruction).  Plain VALU issue cost each
// (tools/mfma_f16_coissue.hip: they co-issue with the f16 MFMA like v_fma_f32): 2.5 instructions per value in all.
__device__ __forceinline__ unsigned f16_pack(float x0, float x1) {
    unsigned p;
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(p) : "v"(x0), "v"(x1));
    return p;
}
__device__ __forceinline__ float f16_rem_lo(unsigned p0, float x0) {
    float r;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(p0), "v"(x0));
    return r;
}
__device__ __forceinline__ float f16_rem_hi(unsigned p0, float x1) {
    float r;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(p0), "v"(x1));
    return r;
}
// p1 = {f16(r0 * 2^12), f16(r1 * 2^12)}: v_ldexp_f32 + v_cvt_pk_f16_f32.  (v_fma_mixlo/hi_f16 would scale and round in one
// instruction each, but they write HALF a register: gfx950 needs a wait state behind such a write before the register is read
// again, hipcc pads every one with an s_nop, and with one wave per SIMD an s_nop is an issue slot like any other: the ISA of
// round 4's first f16x3 loop carried nine of them per Winograd position.)
__device__ __forceinline__ unsigned f16_pack_scaled(float r0, float r1) {
    float s0, s1;
    unsigned p;
    asm("v_ldexp_f32 %0, %1, 12" : "=v"(s0) : "v"(r0));
    asm("v_ldexp_f32 %0, %1, 12" : "=v"(s1) : "v"(r1));
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(p) : "v"(s0), "v"(s1));
    return p;
}
// host packers of the split images (split_pack.hip); sizes in uint16 INCLUDING the trailer
size_t split_image_u16(size_t n_out, size_t k_total);            // n_out * k_total * 3 + 4 * n_out
// fp32-accurate Winograd conv on the 16-bit matrix cores (conv_wino_r64.hip); weights: [Cin/16][16][Cout/32][3][64][8] + trailer
void pack_conv3_wino_split(const float* oihw, unsigned short* out, int O, int I);
bool conv_wino_r64_supported(const IgemmArgs& a);        // wsplit == 2
int launch_conv_wino_r64(const IgemmArgs& a, hipStream_t s);
// the same convolution with the column half of the output transform on the matrix pipe, 64 tiles x 128 couts per block
// (conv_wino_z128.hip): f16x3, maps >= 16 x 16, Cout % 128 == 0; same weight image
bool conv_wino_z128_supported(const IgemmArgs& a);
bool conv_wino_z128_shape(int nfr, int H, int Cin, int Cout);     // the shape half of that decision (grid fill; conv_wino_z128.hip)
int launch_conv_wino_z128(const IgemmArgs& a, hipStream_t s);
// ... and with the GroupNorm(+FiLM) affine + SiLU of the input applied in the kernel's patch staging (a.affA / a.affB / a.act = 1): no activation image
bool conv_wino_z128_act_shape(int nfr, int H, int Cin, int Cout);
bool conv_wino_z128_act_supported(const IgemmArgs& a);
// Upsample (nearest x2) + conv3x3 in its sub-pixel form (conv_wino_r64.hip): four phase kernels per real cout over the
// LOW-resolution map, one of the four Winograd columns structurally zero and skipped.  IgemmArgs::ups_phase selects it;
// the GroupNorm table then has conv_wino_ups_stats_split(Hs) entries per frame.
void pack_conv3_wino_ups(const float* oihw, unsigned short* out, int O, int I);     // image of 4*O phase kernels
int conv_wino_ups_stats_split(int Hs);
int conv_wino_r64_ksplit(int nfr, int Hl, int Cin, int Cout);            // slices of the channel loop a small grid is cut into (1: none)
size_t conv_wino_r64_ksplit_floats(int nfr, int Hl, int Cin, int Cout, int nfr_sel = 0);  // floats of scratch the caller then provides in ksplit_ws (slices chosen for nfr_sel frames)
bool gemm_split_supported(const IgemmArgs& a);
bool gemm_split_side_supported(const IgemmArgs& a);   // IgemmArgs::side (the ResBlock's activation image from the skip conv's A tiles)            // fp32-accurate GEMM on the bf16 matrix cores (gemm_split.hip)
bool conv_split_supported(const IgemmArgs& a);          // 3x3, stride 1|2: the same kernel over an implicit im2col A
void pack_conv3_split(const float* w_oihw, unsigned short* out, int Cout, int Cin);
int launch_gemm_split(const IgemmArgs& a, int tile_class, hipStream_t s);
int gemm_split_tile_class(int M, int Cout);               // igemm_tile_class, or 4 = the 128x192 tile
int gemm_split_stats_rows(int M, int Cout);               // rows of a wave tile = rows behind one GroupNorm partial sum (64 | 32)
void pack_linear_split(const float* w, unsigned short* out_base, int rows, int K, int n_total, int row0);          // blocks per frame = partial sums per (frame, channel)
int launch_conv_wino(const IgemmArgs& a, hipStream_t s);
void pack_conv3_wino(const float* oihw, float* out, int O, int I);      // out: 16*O*I floats
int launch_attn_spatial(const AttnSpatialArgs& a, hipStream_t s);
int launch_attn_temporal(const AttnTemporalArgs& a, hipStream_t s);

// GroupNorm statistics over (pixels x channels-of-group) of one frame, 32 groups, virtual concat.
// part: workspace of nfr*split*C*2 doubles (per-channel fp64 sum / sum of squares per pixel range).
int launch_gn_stats(const float* src0, const float* src1, int C0, int C, int nfr, int HW, double* part, int split,
                    hipStream_t s);
int gn_stats_split(int nfr, int HW, int C);
// affA/affB[n][c] = fold(mean, rstd, gamma, beta, FiLM scale/shift).  film: [nfr][film_ld], scale at +0, shift at +C.
// count = elements per group (HW * C/32)
int launch_gn_affine(const double* part0, int split0, int C0, const double* part1, int split1, double count,
                     const float* gamma, const float* beta, const float* film, int film_ld, int nfr, int C, float* affA,
                     float* affB, hipStream_t s, float* mr_out = nullptr);      // mr_out: [nfr][32][2] mean, rstd (backward pass)
// launch_gn_affine + launch_affine_act in one launch: every block folds its frame's statistics itself (norm.hip).
struct GnFold { const double* part0; int split0; const double* part1; int split1; double count; const float* gamma; const float* beta;
                const float* film; int film_ld; };
int launch_affine_act_fold(const float* src0, const float* src1, int C0, int C, const GnFold& f, int nfr, int HW, int act, float* y,
                           hipStream_t s);
// y = x*A[n][c] + B[n][c]  (materialised normalisation for the attention residual, unet.py:474,538)
// y[n][p][0..C) = silu?(concat(src0, src1)[n][p][c] * A[n][c] + B[n][c])
int launch_affine_act(const float* src0, const float* src1, int C0, int C, const float* affA, const float* affB, int nfr,
                      int HW, int act, float* y, hipStream_t s);
int launch_affine_apply(const float* x, const float* affA, const float* affB, int nfr, int HW, int C, float* y,
                        hipStream_t s);
// Temporal GroupNorm: stats over (T x C/32) for every (b, pixel); writes the normalised tensor.
int launch_gn_temporal(const float* x, const float* gamma, const float* beta, int B, int T, int HW, int C, float* y,
                       hipStream_t s);

struct AssembleArgs {
    const float* x;      // [B][T][3][H][W]
    const float* obs_src;// frames substituted where obs_mask=1 (x0 / x / x_t_minus_1)
    const float* obs_mask, *lat_mask, *km_mask;   // [B*T]
    const float* t_model;// [B] value handed to the network (already mapped / rescaled)
    int obs_t_mode;      // 0: 'x_0' (obs frames see t=0), 1: 'x_t', 2: 'x_t_minus_1'
    int B, T, H, W, Kpad;
    int cond_mode;       // vd_config::cond_emb_type: 0 channel (5 stem channels), 1 duplicate|all (6), 2 t=0 (3)
    float* x_cols;       // [B*T*H*W][Kpad]: im2col of the stem's input for the 3x3 stem, k = tap*Cs + channel
    float* t_frames;     // [B*T]
    float* amask;        // [B*T] anything mask
    // window prefix cache (engine.hip): with a frame list the im2col rows of frames list[0..n_list) are written COMPACTLY
    // (row block i <- frame list[i]) and the per-frame scalars are left alone; scalars_only writes t_frames / amask of every
    // frame and no im2col
    const int* frame_list = nullptr;
    int n_list = 0;
    int scalars_only = 0;
};
int launch_assemble(const AssembleArgs& a, hipStream_t s);
// frame-granular moves of the window prefix cache: dst[i] = src[list[i]] (rows of row_floats), dst[list[i]] = src[i], and the
// GroupNorm partial table of a compact tensor folded to ONE entry per (frame, channel) at its frame's place
int launch_gather_rows(const float* src, const int* list, int n, size_t row_floats, float* dst, hipStream_t s);
int launch_scatter_rows(const float* src, const int* list, int n, size_t row_floats, float* dst, hipStream_t s);
int launch_scatter_stats(const double* src, int split, int C, const int* list, int n, double* dst, hipStream_t s);
// out[n] = [cos(t*f) | sin(t*f)] with the frequency table built on the host (nn.py:89-107)
int launch_sinus_embed(const float* t, int n, int dim, const float* freqs, float* out, hipStream_t s);
// RPENet hidden: E[b,t,s,c] = silu(te[b*T+t][c] + Wd[c][:]*feat(d) + bd[c]), d = fi[b,t]-fi[b,s]  (unet.py:283-296)
// nz nets at once (blockIdx.y): net z reads te + z*zs_te, Wd + z*zs_w, bd + z*zs_b and writes E + z*zs_e
int launch_rpe_hidden(const float* te, int te_ld, const float* Wd, const float* bd, const int64_t* fidx, int B, int T,
                      int C, float* E, int nz, int zs_te, int zs_w, int zs_b, size_t zs_e, hipStream_t s);
// the same for nets at arbitrary places: net z reads te + tab[3z], wbase + tab[3z + 1] (Wd), wbase + tab[3z + 2] (bd)
int launch_rpe_hidden_tab(const float* te, int te_ld, const float* wbase, const long long* tab, const int64_t* fidx, int B, int T, int C,
                          float* E, int nz, size_t zs_e, hipStream_t s);
// bucket-table path (unet.py:330-347): R[b,t,s,:] = table[bucket(d)]
int launch_rpe_table(const float* table, const int64_t* fidx, int B, int T, int C, float alpha, float beta,
                     float gamma, float* R, hipStream_t s);
// h = x + P[pixel][c] (+ frame embedding [n][c])
int launch_posenc_add(const float* x, const float* P, const float* femb, int nfr, int HW, int C, float* y,
                      hipStream_t s);
// tv[b*T+t] = fidx - (center ? mean_t fidx : 0)   (unet.py:914-926)
int launch_frame_t(const int64_t* fidx, int B, int T, int center, float* tv, hipStream_t s);
// out conv: GN-affine + SiLU + conv3x3 (C -> Cout<=8), NHWC in, NCHW out  (unet.py:744-749,838)
int launch_out_conv(const float* x, const float* affA, const float* affB, const float* w, const float* bias, int nfr,
                    int H, int W, int C, int Cout, float* out_nchw, hipStream_t s);

// the head's first half as its own kernel: T[pixel][ldt] = silu(h * A + B) . Wt^T, Wt [ldt][C] fp32 (misc.hip)
bool head_gemm_supported(int HW, int C, int ldt);
int launch_head_gemm(const float* h, const float* affA, const float* affB, const float* Wt, int nfr, int HW, int C, int ldt, float* T, hipStream_t s);
// the head's second half: out_nchw[n][co][y][x] = bias[co] + sum_tap T[n][y + dy][x + dx][co * 9 + tap] (zero outside the image); T [nfr][H][W][ldt]
int launch_out_gather(const float* T, const float* bias, int nfr, int H, int W, int ldt, int Cout, float* out_nchw, hipStream_t s);

struct PosteriorArgs {
    const float* x;      // x_t   [B][per]
    const float* eps;    // model output
    const float* noise;  // explicit noise or null (then Philox(seed, offset))
    const int64_t* t;    // [B] respaced index
    const float* tab;    // [NTAB][num_timesteps] float32 tables (gathered in f64 on host, cast like the reference)
    int num_timesteps;
    int B; long per;
    int clip;
    int mode;            // 0: p_sample, 1: ddim
    float eta;
    unsigned long long seed, offset;
    float* sample; float* xstart;   // outputs (either may be null; sample may alias x: the update is elementwise)
    float* mean;                    // p_mean_variance's 'mean' (posterior mean of the clipped x0 prediction), or null
    const unsigned long long* dstate;   // window executor: {seed, offset} read from device memory instead of the arguments
    const float* x0_given = nullptr;    // denoised_fn path: the caller's x_0 prediction replaces the one derived from eps
    int* err = nullptr;                 // the engine's sticky error word: bit 1 is set when the network output (eps, or the x_0 handed in)
                                        // is not finite -- the clamp of clip_denoised would otherwise turn a NaN into a plausible -1 silently
};
enum { VD_ERR_TIMESTEP = 1, VD_ERR_NONFINITE = 2 };
int launch_posterior(const PosteriorArgs& a, hipStream_t s);
int launch_q_sample(const float* x0, const float* noise, const int64_t* t, const float* tab, int num_timesteps, int B,
                    long per, float* out, hipStream_t s);
int launch_randn(float* out, long n, unsigned long long seed, unsigned long long offset, hipStream_t s);
int launch_q_sample_prev(const float* x0, const int64_t* t, const float* tab, int num_timesteps, int B, long per, const unsigned long long* dstate,
                         unsigned long long draw_offset, float* out, hipStream_t s);

enum { TAB_SQRT_RECIP = 0, TAB_SQRT_RECIPM1, TAB_COEF1, TAB_COEF2, TAB_LOGVAR, TAB_ACP, TAB_ACP_PREV,
       TAB_SQRT_ACP, TAB_SQRT_1M_ACP, TAB_POST_LOGVAR, TAB_LOG_1M_ACP, TAB_ALPHA, NTAB };

// ---- backward-data pieces of `use_gradient_method` (backward.hip)
struct GnBwdArgs {
    const float* x0; const float* x1;   // the GroupNorm's input: one tensor or a virtual concat [N][HW][C0] | [N][HW][C-C0]
    int C0, C;
    const float* A; const float* B;     // the forward's folded affine [N][C]
    const float* mr;                    // [N][32][2] group mean, rstd (gn_final_affine's optional output)
    const float* dy;                    // [N][HW][C] gradient w.r.t. act(x*A + B)
    int act;                            // 1: SiLU behind the affine
    int N, HW;
    float* dx0; float* dx1;             // gradients of the two sources, each assigned (acc = 0) or accumulated (acc = 1)
    int acc0, acc1;
    const float* extra;                 // [N][HW][C] added to the result (the ResBlock's skip path), or null
    double* part;                       // [N][gn_bwd_split][C][2] workspace
    float* K;                           // [N][32][2] workspace
};
int gn_bwd_split(int nfr, int HW, int C);
int launch_gn_bwd(const GnBwdArgs& a, hipStream_t s);
int launch_gn_temporal_bwd(const float* x, const float* gamma, const float* dy, int B, int T, int HW, int C, int accumulate,
                           float* dx, hipStream_t s);
int launch_attn_temporal_bwd(const AttnTemporalArgs& a, const float* dout, float* dqkv, hipStream_t s);
size_t attn_spatial_bwd_ws_floats(const AttnSpatialArgs& a);
int launch_attn_spatial_bwd(const AttnSpatialArgs& a, const float* dout, float* dqkv, float* ws, hipStream_t s);
int launch_zero_stuff2(const float* x, int nfr, int Ho, int Wo, int C, float* y, hipStream_t s);
int launch_sumpool2(const float* x, int nfr, int Ho, int Wo, int C, int accumulate, float* y, hipStream_t s);
int launch_add(const float* x, size_t n, int accumulate, float* y, hipStream_t s);
int launch_out_conv_bwd(const float* deps, const float* w, int nfr, int H, int W, int C, int Cout, float* da, hipStream_t s);
int launch_stem_col2im(const float* dcols, const float* obs, const float* lat, const float* km, int nfr, int H, int W, int Kpad, int cond_mode,
                       float* dx, hipStream_t s);
// return_attn_weights (unet.py:457-466): softmax weights averaged over the heads, |.|; temporal [B*HW][T][T], spatial [nfr][L][L]
int launch_attn_temporal_weights(const AttnTemporalArgs& a, float* out, hipStream_t s);
int launch_attn_spatial_weights(const AttnSpatialArgs& a, float* out, hipStream_t s);
struct GuidedArgs {
    const float* x; const float* eps; const float* noise; const float* xtm1;   // [B][per]
    const float* obs;                   // [B*T] the ORIGINAL observation mask (the network saw every frame as latent)
    const int64_t* t; const float* tab; int num_timesteps;
    int B, T; long per; int clip;
    float* deps; float* dxd;            // d loss / d eps, and the direct part of d loss / d x
    float* mean; float* xstart;         // the unguided posterior mean and the x_0 prediction
    int* err = nullptr;                 // sticky error word (bit 1: eps not finite)
    const float* gscale = nullptr;      // {s, 1 / s}: the backward pass ran on s * d loss / d eps (backward.hip: launch_grad_rescale); null: 1
};
int launch_grad_rescale(float* deps, size_t n, float* gs, hipStream_t st);
int launch_guided_grad(const GuidedArgs& a, hipStream_t s);
int launch_guided_final(const GuidedArgs& a, const float* dx_net, const float* noise2, float* grad, float* mean_out, float* sample,
                        hipStream_t s);

// _vb_terms_bpd + the two MSEs of calc_bpd_loop_subsampled's loop body (gaussian_diffusion.py:750-790, 975-990), given eps
struct VbArgs {
    const float* x_start; const float* x_t; const float* eps;
    const float* noise;          // for mse = mean((eps_from_xstart - noise)^2); null: not computed
    const int64_t* t;            // [B]
    const float* tab; int num_timesteps;
    const float* mask;           // [B*T] latent_mask or null (mean_flat(..., mask): multiply, then mean over ALL elements)
    int B, T; long per;          // per = T*3*H*W
    int clip;
    float* pred_xstart;          // optional output
    double* part;                // [B][nblk][3] partial sums
    int nblk;
    float* vb; float* xstart_mse; float* mse;      // [B] outputs (the last two may be null)
    int* err = nullptr;          // sticky error word (bit 1: eps not finite)
    int start_x = 0;             // ModelMeanType.START_X (predict_xstart=True): `eps` holds the network's x_0 prediction itself
};
int launch_vb_terms(const VbArgs& a, hipStream_t s);
int vb_terms_blocks(long per);
// _prior_bpd (gaussian_diffusion.py:909-926): KL(q(x_T | x_0) || N(0, I)) in bits per dim, masked mean
int launch_prior_bpd(const float* x_start, const float* mask, const float* tab, int num_timesteps, int B, int T, long per,
                     double* part, int nblk, float* out, hipStream_t s);

// x * sigmoid(x) with v_exp_f32 + v_rcp_f32 (each <= 1 ulp): 5 VALU instructions instead of the ~15 of an IEEE
// division.  The operand transform of the conv kernels runs this on every staged input element.
__device__ __forceinline__ float silu_f(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

}  // namespace vd
