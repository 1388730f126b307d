// Host side of the split matrix kernels (gemm_split.hip, conv_wino_r64.hip): the arithmetic mode and the weight images.
//
// Every image is [k-step][output block of 32][piece 3][lane 64][8 x 16 bit] -- lane 32h + r of a (k-step, block) holds output
// 32*blk + r, k = 16*step + 8h + e -- followed by a trailer of 2*N floats: the per-output power-of-two scale and its
// reciprocal (vd_common.h).  Pieces of a weight w (s = the row's scale):
//   f16x3:  [0] b0 = f16(w s)   [1] b1 = f16(w s - b0)   [2] 2^-12 b0      (the partner of the activation's scaled remainder)
//   bf16x6: [0] bf16(w)  [1] bf16(w - p0)  [2] bf16(w - p0 - p1)           (exact: 3 x 8 significand bits; s = 1)
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "vd_common.h"

namespace vd {

int math_mode() {
    static const int v = [] {
        const char* e = getenv("VD_MATH");
        if (!e || !*e || std::string(e) == "f16x3") return (int)MATH_F16X3;
        if (std::string(e) == "bf16x6") return (int)MATH_BF16X6;
        if (std::string(e) == "fp32") return (int)MATH_FP32;
        fprintf(stderr, "libvdamd: VD_MATH=%s is not one of f16x3 | bf16x6 | fp32\n", e);
        abort();
    }();
    return v;
}

size_t split_image_u16(size_t n_out, size_t k_total) { return n_out * k_total * 3 + 4 * n_out; }

// fp32 -> bf16, round to nearest even on the top 16 bits
static inline unsigned short bf16_rne(float f) {
    unsigned u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);          // NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
static inline float bf16_to_f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; std::memcpy(&f, &u, 4); return f; }
static inline unsigned short f16_bits(float f) { const _Float16 h = (_Float16)f; unsigned short b; std::memcpy(&b, &h, 2); return b; }
static inline float f16_round(float f) { return (float)(_Float16)f; }

// power of two s with max |w s| in [2^13, 2^14)
float split_row_scale(double maxabs) {
    if (math_mode() != MATH_F16X3 || !(maxabs > 0.0) || !std::isfinite(maxabs)) return 1.f;
    int ex;
    (void)std::frexp(maxabs, &ex);                    // maxabs = m 2^ex, m in [0.5, 1)
    const int e = std::max(-100, std::min(100, 14 - ex));
    return std::ldexp(1.f, e);
}

// the three 16-bit pieces of weight v under row scale s
void split_weight(float v, float s, unsigned short out[3]) {
    if (math_mode() == MATH_F16X3) {
        const float vs = v * s;                       // exact: s is a power of two
        const float b0 = f16_round(vs);
        out[0] = f16_bits(b0);
        out[1] = f16_bits(vs - b0);
        out[2] = f16_bits(b0 * (1.f / 4096.f));
        return;
    }
    out[0] = bf16_rne(v);
    const float r1 = v - bf16_to_f(out[0]);
    out[1] = bf16_rne(r1);
    out[2] = bf16_rne(r1 - bf16_to_f(out[1]));
}

// rows [row0, row0 + rows) of a [n_total][K] row-major matrix into the image of the WHOLE matrix at out_base (trailer
// entries of those rows included)
void pack_linear_split(const float* w, unsigned short* out_base, int rows, int K, int n_total, int row0) {
    const int ncoblk = n_total / 32;
    float* trailer = reinterpret_cast<float*>(out_base + (size_t)n_total * K * 3);
    for (int n = 0; n < rows; ++n) {
        double mx = 0.0;
        for (int k = 0; k < K; ++k) mx = std::max(mx, (double)std::fabs(w[(size_t)n * K + k]));
        const float s = split_row_scale(mx);
        trailer[row0 + n] = s;
        trailer[n_total + row0 + n] = 1.f / s;
        const int cb = (row0 + n) / 32, r = (row0 + n) % 32;
        for (int k = 0; k < K; ++k) {
            unsigned short p[3];
            split_weight(w[(size_t)n * K + k], s, p);
            const int ks = k / 16, h = (k % 16) / 8, j = k % 8;
            for (int q = 0; q < 3; ++q) out_base[((((size_t)ks * ncoblk + cb) * 3 + q) * 64 + h * 32 + r) * 8 + j] = p[q];
        }
    }
}

// OIHW 3x3 weights -> the image of the [Cout][9*Cin] matrix with k = tap*Cin + c (the order gemm_split's CONV mode walks K)
void pack_conv3_split(const float* w, unsigned short* out, int Cout, int Cin) {
    std::vector<float> lin((size_t)Cout * 9 * Cin);
    for (int o = 0; o < Cout; ++o)
        for (int i = 0; i < Cin; ++i)
            for (int t = 0; t < 9; ++t) lin[(size_t)o * 9 * Cin + (size_t)t * Cin + i] = w[((size_t)o * Cin + i) * 9 + t];
    pack_linear_split(lin.data(), out, Cout, 9 * Cin, Cout, 0);
}

// ---- Winograd F(2x2,3x3) image: U = G g G^T (fp64, rounded once to fp32, row 3 negated), [Cin/16][xi 16][O/32][piece 3][lane 64][8]:
// lane 32h + r holds U[xi][co = 32*blk + r][ci = 16*chunk + 8h + e]; trailer [2][O]
static void wino_U(const double* gk, double U[16]) {
    static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    double tmp[4][3];
    for (int i = 0; i < 4; ++i)
        for (int c = 0; c < 3; ++c) tmp[i][c] = G[i][0] * gk[0 * 3 + c] + G[i][1] * gk[1 * 3 + c] + G[i][2] * gk[2 * 3 + c];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            const double u = tmp[i][0] * G[j][0] + tmp[i][1] * G[j][1] + tmp[i][2] * G[j][2];
            U[i * 4 + j] = i == 3 ? -u : u;
        }
}
// one output channel `co` of an image with O outputs: gk[ci][9] fp64 kernels
static void pack_wino_cout(const std::vector<double>& gk, unsigned short* out, int O, int I, int co) {
    const int ncoblk = O / 32, cb = co >> 5, r = co & 31;
    std::vector<float> U((size_t)I * 16);
    double mx = 0.0;
    for (int ci = 0; ci < I; ++ci) {
        double u[16];
        wino_U(gk.data() + (size_t)ci * 9, u);
        for (int x = 0; x < 16; ++x) { U[(size_t)ci * 16 + x] = (float)u[x]; mx = std::max(mx, std::fabs((double)(float)u[x])); }
    }
    const float s = split_row_scale(mx);
    float* trailer = reinterpret_cast<float*>(out + (size_t)16 * O * I * 3);
    trailer[co] = s;
    trailer[O + co] = 1.f / s;
    for (int ci = 0; ci < I; ++ci) {
        const int ch = ci / 16, k = ci % 16, h = k >> 3, e = k & 7;
        for (int x = 0; x < 16; ++x) {
            unsigned short pc[3];
            split_weight(U[(size_t)ci * 16 + x], s, pc);
            for (int q3 = 0; q3 < 3; ++q3) out[(((((size_t)ch * 16 + x) * ncoblk + cb) * 3 + q3) * 64 + h * 32 + r) * 8 + e] = pc[q3];
        }
    }
}

void pack_conv3_wino_split(const float* oihw, unsigned short* out, int O, int I) {
    std::vector<double> gk((size_t)I * 9);
    for (int co = 0; co < O; ++co) {
        for (int ci = 0; ci < I; ++ci)
            for (int t = 0; t < 9; ++t) gk[(size_t)ci * 9 + t] = oihw[((size_t)co * I + ci) * 9 + t];
        pack_wino_cout(gk, out, O, I, co);
    }
}

// Upsample (nearest x2) + conv3x3 as four 3x3 kernels over the SOURCE map (conv_wino_r64.hip): output pixel (2y + a, 2x + b)
// reads source rows (y-1, y, y+1) with (w0, w1 + w2, 0) for a = 0 and (0, w0 + w1, w2) for a = 1, columns alike with b; the
// sums are formed in fp64.  Image of 4*O couts: cout ((b * O/32 + cb) * 2 + a) * 32 + r is phase (a, b) of real cout 32*cb + r.
void pack_conv3_wino_ups(const float* oihw, unsigned short* out, int O, int I) {
    const int ncb = O / 32;
    std::vector<double> gk((size_t)I * 9);
    for (int co = 0; co < O; ++co)
        for (int a = 0; a < 2; ++a)
            for (int b = 0; b < 2; ++b) {
                for (int ci = 0; ci < I; ++ci) {
                    const float* w = oihw + ((size_t)co * I + ci) * 9;
                    double rows[3][3];
                    double* g = gk.data() + (size_t)ci * 9;
                    for (int c = 0; c < 3; ++c) {                     // vertical combination, per kernel column
                        const double w0 = w[0 * 3 + c], w1 = w[1 * 3 + c], w2 = w[2 * 3 + c];
                        rows[0][c] = a == 0 ? w0 : 0.0; rows[1][c] = a == 0 ? w1 + w2 : w0 + w1; rows[2][c] = a == 0 ? 0.0 : w2;
                    }
                    for (int r = 0; r < 3; ++r) {
                        g[r * 3 + 0] = b == 0 ? rows[r][0] : 0.0;
                        g[r * 3 + 1] = b == 0 ? rows[r][1] + rows[r][2] : rows[r][0] + rows[r][1];
                        g[r * 3 + 2] = b == 0 ? 0.0 : rows[r][2];
                    }
                }
                const int cb = co >> 5, r32 = co & 31;
                pack_wino_cout(gk, out, 4 * O, I, ((b * ncb + cb) * 2 + a) * 32 + r32);
            }
}

}  // namespace vd
