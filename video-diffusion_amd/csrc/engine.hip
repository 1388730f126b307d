// Step engine: UNet topology, packed weights, the per-step launch sequence and the C ABI.
//
// One process per GPU owns one engine.  A denoise step is a fixed sequence of ~600 kernel launches
// on the caller's stream; activations live in one bump-allocated arena sized per (B, T) window (288 GB
// of HBM: no buffer reuse games, the same addresses every step, so the sequence is graph-capturable).
// Weights live in ONE packed device buffer in kernel-ready layouts (see Param::kind).
//
// Reference call path being replaced (paths relative to /root/reference/improved_diffusion):
//   _WrappedModel.__call__ respace.py:111-119 -> CondMargVideoModel.forward unet.py:949-1026 ->
//   UNetVideoModel.forward :898-912 -> UNetModel.forward :768-839 ; p_sample gaussian_diffusion.py:403-448.
#include <cmath>
#include <cstring>
#include <unordered_map>
#include <vector>

#include "../../include/vd_amd.h"
#include "vd_common.h"

namespace vd {

thread_local std::string g_last_error;
void set_error(const std::string& msg) { g_last_error = msg; }

int launch_frame_t(const int64_t* fidx, int B, int T, int center, float* tv, hipStream_t s);

// PK_CONV3: [tap][Cout][Cin] for the generic kernel (igemm.hip).  PK_LINF / PK_STEM: every nn.Linear, 1x1 conv and the stem as a
// matrix image -- the split image of gemm_split.hip (split_pack.hip; VD_MATH=fp32: the fragment-major fp32 image of gemm_frag.hip).
// PK_CONV3W: Winograd F(2x2,3x3) image (conv_wino_r64.hip; fp32: conv_wino.hip).  PK_CONV3S: stride-2 conv on the split GEMM.
enum ParamKind { PK_RAW = 0, PK_CONV3, PK_STEM, PK_POSENC, PK_OUTCONV, PK_LINF, PK_CONV3W, PK_CONV3S,
                 PK_CONV3WU };      // PK_CONV3WU: Upsample + conv3x3 in its sub-pixel form (conv_wino_r64.hip): image of 4 x Cout phase kernels

struct Param {
    std::string name;
    int nd = 0;
    long long shape[4] = {1, 1, 1, 1};
    size_t numel = 0;        // as stored in the checkpoint
    size_t packed = 0;       // floats in the packed buffer
    size_t off = 0;          // float offset in the packed buffer
    int kind = PK_RAW;
    bool loaded = false;
    // PK_LINF members of a batched matrix (all FiLM projections / all RPE time projections are ONE GEMM):
    // `off` is the base of the whole fragment-major image, rows [frag_row0, +shape[0]) of frag_rows in total
    int frag_rows = 0, frag_row0 = 0;
    // backward-data image (use_gradient_method): the transposed matrix of a linear layer / 1x1 conv, the 180-degree-rotated
    // transposed kernel of a 3x3 conv, in the layout of kind_bwd; -1: the backward pass does not need this parameter
    int kind_bwd = -1;
    size_t off_bwd = 0, packed_bwd = 0;
};

struct ResP {
    int cin, cout;
    int gn1w, gn1b, c1w, c1b, embw, embb, gn2w, gn2b, c2w, c2b, skw = -1, skb = -1;
    int film_off = 0;        // column offset in the batched emb projection
};
struct RpeP { int dw = -1, db = -1, tw = -1, tb = -1, ow = -1, ob = -1, table = -1; int te_off = 0; };
struct AttP { int normw, normb, qkvw, qkvb, projw, projb; };
struct AttnP { int C; AttP sp, tp; RpeP rq, rk, rv; };
struct ConvP { int w, b, c; };
struct Layer { int type; int idx; };      // 0 stem, 1 res, 2 attn, 3 down, 4 up
// An activation tensor [N][H][H][C]; part/split: its GroupNorm partial sums [N][split][C][2] when the producing
// convolution's epilogue wrote them (conv_wino.hip), else null and the consumer's GroupNorm runs the statistics pass.
struct Tens { float* p; int C, H; double* part = nullptr; int split = 0; };

// ---- tape of one forward pass, kept when the step is guided (use_gradient_method): every op with the tensors its
// backward reads.  Pointers are arena addresses; nothing of a taped forward is released before the backward has run.
struct TapeRes { int idx; Tens x0, x1; bool has_x1; int N; float *A1, *B1, *mr1, *h, *A2, *B2, *mr2, *o; };
struct TapeAttn { int idx; Tens x; int B, T; float *xn, *qkv, *Rk, *Rq, *Rv, *o, *xt, *A, *Bf, *mr, *xn2, *qkv2, *o2, *xs; const float* amask; };
struct TapeConv { int type, idx; Tens in, out; };                     // 0 stem, 3 down, 4 up
struct TapePos { Tens in, out; };
struct TapeOp { int kind, i; };                                       // kind: 0 conv, 1 res, 2 attn, 5 posenc
struct Tape {
    std::vector<TapeOp> ops;
    std::vector<TapeRes> res; std::vector<TapeAttn> attn; std::vector<TapeConv> conv; std::vector<TapePos> pos;
    Tens head; float *headA = nullptr, *headB = nullptr, *head_mr = nullptr;
    void clear() { ops.clear(); res.clear(); attn.clear(); conv.clear(); pos.clear(); }
};

static const int CH_MULT_256[] = {1, 1, 2, 2, 4, 4};
static const int CH_MULT_128[] = {1, 1, 2, 3, 4};
static const int CH_MULT_64[] = {1, 2, 3, 4};
static const int CH_MULT_32[] = {1, 2, 2, 2};
// VD_MATH (vd_common.h): f16x3 (default) and bf16x6 run every matrix product on the split kernels (gemm_split.hip,
// conv_wino_r64.hip) over 16-bit piece images; fp32 keeps them on the fp32 MFMA (gemm_frag.hip, conv_wino.hip).
static bool split_math() { return math_mode() != MATH_FP32; }
static bool split_conv() { return split_math(); }

constexpr int STEM_KPAD = 64;          // im2col width of the 5-channel 3x3 stem (45 real columns)

struct Arena {
    char* base = nullptr;
    size_t cap = 0, used = 0, peak = 0;     // cap: bytes this arena may hand out (the workspace up to its tail; 0 = unchecked)
    bool dry = false, overflow = false;
    template <class Tp> Tp* get(size_t n) {
        size_t bytes = (n * sizeof(Tp) + 255) & ~(size_t)255;
        if (!dry && cap && used + bytes > cap) {
            // the dry run that sized the workspace did not see this request: never write past the arena (the tail holds t_model
            // and the eps scratch).  The request is served from the base -- results are garbage -- and the forward fails loudly.
            overflow = true;
            return reinterpret_cast<Tp*>(base);
        }
        char* p = (dry ? reinterpret_cast<char*>(0x1000) : base) + used;     // dry run: distinct, never dereferenced (the backward keys gradients by tensor address)
        used += bytes;
        peak = std::max(peak, used);
        return reinterpret_cast<Tp*>(p);
    }
    // stack discipline for transients: everything is stream-ordered on one stream, so a released range may be handed
    // out again by the next get()
    size_t mark() const { return used; }
    void release(size_t m) { used = m; }
};

// ---- per-class kernel timing with HIP events on the launch stream (bench.py's live roofline figure)
enum ProfClass { PC_IGEMM_128x128 = 0, PC_IGEMM_128x64, PC_IGEMM_64x128, PC_IGEMM_64x64, PC_GN_STATS, PC_GN_TEMPORAL,
                 PC_ATTN_SPATIAL, PC_ATTN_TEMPORAL, PC_OUT_CONV, PC_ELEMENTWISE, PC_POSTERIOR,
                 PC_CONV_128x128, PC_CONV_128x64, PC_CONV_64x128, PC_CONV_64x64, PC_CONV_WINO, PC_CONV_WINO_R64, PC_IGEMM_128x192,
                 PC_CONV_WINO_R64_UPS, PC_CONV_WINO_Z128, PC_COUNT };
// names of the kernels a class runs on: [0] split arithmetic (f16x3 | bf16x6), [1] VD_MATH=fp32
static const char* kProfNames[PC_COUNT][2] = {
    {"gemm_split_kernel<128,128>", "gemm_frag_kernel<128,128>"}, {"gemm_split_kernel<128,64>", "gemm_frag_kernel<128,64>"},
    {"gemm_split_kernel<64,128>", "gemm_frag_kernel<64,128>"}, {"gemm_split_kernel<64,64>", "gemm_frag_kernel<64,64>"},
    {"gn_stats_partial", "gn_stats_partial"}, {"gn_temporal_kernel", "gn_temporal_kernel"},
    {"attn_spatial_kernel", "attn_spatial_kernel"}, {"attn_temporal_kernel", "attn_temporal_kernel"},
    {"out_conv_kernel", "out_conv_kernel"}, {"affine_act_kernel", "affine_act_kernel"}, {"posterior_kernel", "posterior_kernel"},
    {"igemm_kernel<128,128> 3x3", "igemm_kernel<128,128> 3x3"}, {"igemm_kernel<128,64> 3x3", "igemm_kernel<128,64> 3x3"},
    {"igemm_kernel<64,128> 3x3", "igemm_kernel<64,128> 3x3"}, {"igemm_kernel<64,64> 3x3", "igemm_kernel<64,64> 3x3"},
    {"conv3x3_wino_kernel", "conv3x3_wino_kernel"}, {"conv3x3_wino_r64_kernel", "conv3x3_wino_r64_kernel"},
    {"gemm_split_kernel<128,192>", "gemm_split_kernel<128,192>"},
    {"conv3x3_wino_r64_ups_kernel", "conv3x3_wino_r64_ups_kernel"}, {"conv3x3_wino_z128_kernel", "conv3x3_wino_z128_kernel"}};
struct ProfRec { int cls; double flops, bytes; hipEvent_t a, b; char tag[56]; };
struct Profiler {
    bool on = false;
    std::vector<ProfRec> recs;
};
static Profiler g_prof;
static const char* prof_name(int i) {
    if (i < 0 || i >= PC_COUNT) return "";
    return kProfNames[i][i == PC_CONV_WINO || i == PC_CONV_WINO_R64 ? !split_conv() : !split_math()];
}

struct ProfScope {
    hipStream_t st; bool live;
    ProfScope(int cls, double flops, double bytes, hipStream_t s, const char* tag = "") : st(s), live(g_prof.on) {
        if (!live) return;
        ProfRec r{cls, flops, bytes, nullptr, nullptr, {0}};
        snprintf(r.tag, sizeof(r.tag), "%s", tag);
        (void)hipEventCreate(&r.a); (void)hipEventCreate(&r.b);
        (void)hipEventRecord(r.a, st);
        g_prof.recs.push_back(r);
    }
    ~ProfScope() { if (live) (void)hipEventRecord(g_prof.recs.back().b, st); }
};

// res_block decides from the SHAPE alone (conv_wino_z128_act_shape) that a conv activates in its own patch staging and then never
// materialises the activation image; the dispatcher takes the activating kernel only when conv_wino_z128_act_supported holds for the
// call as built (weight image, piece count, strides, size limits).  If the two ever disagree the conv would get a raw tensor plus
// (A, B) that no other kernel applies: say so here, by name, instead of failing later with the generic "no kernel covers" error.
static bool act_conv_will_dispatch(const IgemmArgs& g) {
    IgemmArgs one = g;
    one.nfr = std::max(1, std::min(g.nfr, igemm_frames_per_launch(g)));
    one.M = one.nfr * g.Ho * g.Wo;
    return conv_wino_z128_act_supported(one);
}

static int igemm_p(const IgemmArgs& g, hipStream_t st, int cin_alg = 0) {
    const double taps = (double)g.ksz * g.ksz, cin = cin_alg ? cin_alg : g.Cin;
    const double in_pix = (double)g.nfr * g.Hs * g.Ws;
    const double nz = g.zcount > 1 ? g.zcount : 1;
    // algorithmic bytes: input once, output once (+ residual), weights once -- and, for a skip convolution that also writes the ResBlock's
    // GroupNorm + SiLU image from the A rows it stages (IgemmArgs::side: one more tensor of the input's size written by this launch, the
    // pass affine_act would otherwise be), that image: counted since r05 (round 4's figure omitted it and read as 2x traffic waste)
    const double bytes = nz * 4.0 * (in_pix * cin + (double)g.M * g.Cout * (g.res ? 2 : 1) + taps * cin * g.Cout + (g.side ? in_pix * cin : 0.0));
    IgemmArgs one = g;                         // a big window goes out as several launches over frame ranges (igemm.hip)
    one.nfr = std::max(1, std::min(g.nfr, igemm_frames_per_launch(g)));
    one.M = one.nfr * g.Ho * g.Wo;
    const bool zact = conv_wino_z128_act_supported(one);              // conv_wino_z128.hip with the activation in its patch staging
    const bool wino = zact || conv_wino_supported(one) || conv_wino_r64_supported(one);
    const bool split_gemm = gemm_split_supported(one) || conv_split_supported(one);
    const int cls = wino ? (int)(g.ups_phase ? PC_CONV_WINO_R64_UPS : zact || conv_wino_z128_supported(one) ? PC_CONV_WINO_Z128
                                                               : conv_wino_r64_supported(one) ? PC_CONV_WINO_R64 : PC_CONV_WINO)
                    : split_gemm && gemm_split_tile_class(igemm_sel_M(one), g.Cout) == 4 ? (int)PC_IGEMM_128x192
                    : igemm_tile_class(igemm_sel_M(one), g.Cout) + (g.ksz == 3 && !split_gemm ? (int)PC_CONV_128x128 : 0);   // 3x3 on the generic kernel
    char tag[56];
    snprintf(tag, sizeof(tag), "M=%d N=%d K=%d k%d s%d%s%s%s", g.M, g.Cout, g.Cin, g.ksz, g.stride, g.ups ? " ups" : "",
             g.affA ? " pro" : (g.act ? " act" : ""), g.res ? " res" : "");
    VD_REQUIRE(g.stats == nullptr || wino || (split_gemm && g.stats_hw > 0),
               "GroupNorm partial sums requested from a kernel that has no such epilogue");
    ProfScope ps(cls, nz * 2.0 * g.M * g.Cout * cin * taps, bytes, st, tag);
    return launch_igemm(g, st);
}

static int affine_act(const float* src0, const float* src1, int C0, int C, const float* A, const float* B, int N, int HW,
                      float* y, hipStream_t st) {
    ProfScope ps(PC_ELEMENTWISE, 0.0, 8.0 * N * HW * C, st);
    return launch_affine_act(src0, src1, C0, C, A, B, N, HW, 1, y, st);
}

struct FwdIn {
    int B, T;
    const float *x, *obs_src, *obs, *lat, *km, *t_model;
    const int64_t* fidx;
    int obs_mode;
    float* eps;
};

// Window prefix cache (vd_window_begin, opt-in).  The input blocks before the first attention layer treat frames as
// independent batch entries, and in 'x_0' mode an observed frame's network input (its x0 pixels, the indicator channels, the
// timestep 0 embedding) does not change over the steps of a window: its activations there are computed once per window.
// A forward with a plan runs those blocks on the `n` listed frames only (a compact batch), scatters every block output --
// and its GroupNorm partial sums -- into a persistent full-size tensor that already holds the other frames' rows, and
// carries on from there with the full batch.
struct PrefixStore { std::vector<float*> tens; std::vector<double*> parts; std::vector<size_t> floats; };
struct PrefixPlan {
    int n = 0;                       // frames in the list
    const int* list = nullptr;       // device: compact index -> frame
    bool build_only = false;         // stop behind the prefix (the once-per-window pass over the observed frames)
    PrefixStore* store = nullptr;
};

// Window suffix skip (vd_window_begin, opt-in).  Behind the LAST attention layer the network treats frames as independent batch
// entries again (ResBlocks, Upsample, the output head: unet.py:820-839), and the caller keeps only the latent frames of a window
// (scripts/video_sample.py:170-186) while an observed frame's sample is replaced by the observation before the network sees it
// again (unet.py:958-983, modes 'x_0' and 'x_t_minus_1').  A forward with this plan gathers the `n` listed frames (every frame
// that is not a pure observation) out of the last attention layer's output and out of each skip tensor the remaining decoder
// blocks read -- rows and GroupNorm partial sums -- runs those blocks and the head on the compact batch, and scatters eps
// into a zeroed full-size tensor.  Kernel variants are chosen for the full frame count (IgemmArgs::nfr_sel): the listed frames'
// eps are those of the full forward, bit for bit.
struct SuffixPlan { int n = 0; const int* list = nullptr; };

static inline int out_t_cols(int Cout) { return (9 * Cout + 31) / 32 * 32; }       // columns of the head's (cout, tap) GEMM

// frames a layer call stands for when it runs on a part of them (conv_args -> IgemmArgs::nfr_sel); set around the compact suffix
static thread_local int g_sel_nfr = 0;

}  // namespace vd

using namespace vd;

struct vd_engine {
    vd_config cfg;
    int E = 0;                       // time_embed_dim
    std::vector<Param> params;
    std::unordered_map<std::string, int> pidx;
    std::vector<ResP> res;
    std::vector<AttnP> attn;
    std::vector<ConvP> convs;        // stem / down / up
    std::vector<std::vector<Layer>> input_blocks, output_blocks;
    std::vector<Layer> middle;
    int n_before_attn = 0, pos_res = 0, pos_ch = 0, final_ch = 0;
    int suf_blk = -2, suf_layer = -1;   // last attention layer: output block (-1: none in the decoder, the suffix starts at block 0) and layer in it
    int p_posenc = -1, p_te0w, p_te0b, p_te2w, p_te2b, p_outgw, p_outgb, p_outw, p_outb;
    size_t film_w_off = 0, film_b_off = 0, te_w_off = 0, te_b_off = 0;
    int film_total = 0, te_total = 0;
    size_t packed_total = 0;
    float* wbuf = nullptr;           // caller-owned packed weights
    bool wbuf_on_host = false;       // vd_set_weight_storage_host: the packed image is assembled in host memory
    int put(void* dst, const void* src, size_t bytes) {
        if (wbuf_on_host) { std::memcpy(dst, src, bytes); return 0; }
        VD_HIP(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
        return 0;
    }
    // rpe_all: the relative-position nets of ALL attention blocks at the start of a forward, two launches per channel width instead of two
    // per block (they depend on the timestep embedding and the frame indices only, unet.py:283-298).  Per width: the blocks of that width,
    // a device table of offsets (rpe_hidden: te column, Wd, bd; output layer: weight image, bias) and where the blocks' R tensors land.
    struct RpeGroup { int C; std::vector<int> blocks; long long* d_hid = nullptr; long long* d_out = nullptr; };
    std::vector<RpeGroup> rpe_groups;
    std::vector<float*> rpe_R;        // [attention block][3]: R of the running forward (k, q, v order of attn_block), or empty
    int rpe_all(const float* te, const int64_t* fidx, int B, int T, hipStream_t st, Arena& ar);
    int rpe_tables();
    // small device tables
    float* d_freq_time = nullptr; int n_freq_time = 0;
    float* d_freq_frame = nullptr; int n_freq_frame = 0;
    float* d_tab = nullptr; int num_timesteps = 0;
    float* d_tmap = nullptr; float rescale = 1.f;
    // workspace: activations of one (B, T) window [0, ws_tail), then t_model [B] and the eps scratch
    char* ws = nullptr; size_t ws_cap = 0;
    bool ws_suf = false;
    int ws_B = 0, ws_T = 0; size_t ws_tail = 0;      // the window shape ws_tail was computed for (dry run of the topology)
    std::unordered_map<long long, size_t> ws_peaks;  // (B << 32 | T) -> arena peak
    // ---- use_gradient_method: second packed image (backward-data weights) + the tape of the guided step's forward
    float* wbuf_bwd = nullptr; bool wbuf_bwd_on_host = false; size_t packed_bwd_total = 0;
    Tape* tape = nullptr;
    // return_attn_weights: device buffers for the next forward, one pair per attention block in execution order
    std::vector<float*> attn_cap_t, attn_cap_s;
    int attn_seq = 0;
    int mean_type = 0;                               // what the network's output IS: 0 eps (ModelMeanType.EPSILON), 1 x_0 (START_X)
    int* d_err = nullptr;                            // sticky device flags: bit 0 = timestep index out of range, bit 1 = network output not finite
    int device = -1;
    double* d_part = nullptr; size_t part_cap = 0;   // NLL partial sums
    // ---- window executor (vd_window_*): device-resident step state + one captured graph per window signature
    struct WinKey {
        int B, T, obs_mode, sampler, clip, flags; float eta;      // flags: 1 prefix cache, 2 suffix skip
        const void *x, *obs_src, *obs, *lat, *km, *fidx;
        bool operator==(const WinKey& o) const { return std::memcmp(this, &o, sizeof(WinKey)) == 0; }
    };
    struct WinGraph {
        WinKey key; hipGraph_t graph; hipGraphExec_t exec;
        // window prefix cache: which frames are step-invariant (part of the signature: the compact batch size is baked into the
        // captured launches), the two device lists and the persistent tensors the captured step reads and writes
        std::vector<unsigned char> inv; int n_inv = 0, n_suf = 0;   // n_suf: frames the suffix runs on (0: all of them)
        int* d_lists = nullptr;                  // [n_act active frames | n_inv invariant frames]
        PrefixStore* store = nullptr;
    };
    bool prefix_cache_on = false;                // vd_set_window_prefix_cache
    bool suffix_skip_on = false;                 // vd_set_window_suffix_skip
    static void free_graph(WinGraph& g) {
        (void)hipGraphExecDestroy(g.exec); (void)hipGraphDestroy(g.graph);
        if (g.d_lists) (void)hipFree(g.d_lists);
        if (g.store) { for (float* p : g.store->tens) if (p) (void)hipFree(p); for (double* p : g.store->parts) if (p) (void)hipFree(p); delete g.store; }
    }
    std::vector<WinGraph> win_graphs;
    int win_cur = -1;
    bool win_lost = false;                               // the armed window's graph was dropped (workspace growth / new schedule)
    unsigned long long win_gen = 0;                      // bumped by every vd_window_begin: a run must name the window it continues
    long long win_left = 0;                              // steps the current window still has (t + 1)
    long long* d_win_t = nullptr; int win_t_cap = 0;     // [B] current respaced index of the window
    float* d_win_xtm1 = nullptr; size_t win_xtm1_cap = 0; // 'x_t_minus_1' windows: the observed frames re-noised to t - 1, redrawn every step
    unsigned long long* d_win_rng = nullptr;             // {seed, Philox offset}

    ~vd_engine() {
        if (d_freq_time) (void)hipFree(d_freq_time);
        if (d_freq_frame) (void)hipFree(d_freq_frame);
        for (auto& g : rpe_groups) { if (g.d_hid) (void)hipFree(g.d_hid); if (g.d_out) (void)hipFree(g.d_out); }
        if (d_tab) (void)hipFree(d_tab);
        if (d_tmap) (void)hipFree(d_tmap);
        if (ws) (void)hipFree(ws);
        if (d_err) (void)hipFree(d_err);
        if (d_part) (void)hipFree(d_part);
        if (d_win_t) (void)hipFree(d_win_t);
        if (d_win_rng) (void)hipFree(d_win_rng);
        if (d_win_xtm1) (void)hipFree(d_win_xtm1);
        for (auto& g : win_graphs) free_graph(g);
    }

    // Captured window graphs bake addresses (workspace, schedule tables, step counters) and values (num_timesteps, rescale)
    // into their kernel arguments: whatever replaces one of those drops every graph first.  A graph may still be queued on
    // the executor's stream, so the device is drained before an exec is destroyed.
    void drop_window_graphs() {
        if (win_graphs.empty()) { win_cur = -1; return; }
        (void)hipDeviceSynchronize();
        for (auto& g : win_graphs) free_graph(g);
        win_graphs.clear();
        win_cur = -1;
    }

    const float* W(int p) const { return wbuf + params[p].off; }
    void set_w(vd::IgemmArgs& g, int p) const {
        const int k = params[p].kind;
        g.w = g.wfrag = g.wwino = nullptr;
        if (k == PK_CONV3W) { g.wwino = W(p); g.wsplit = split_conv() ? 2 : 0; }
        else if (k == PK_CONV3WU) { g.wwino = W(p); g.wsplit = 2; g.ups_phase = 1; }
        else if (k == PK_CONV3S) { g.wfrag = W(p); g.wsplit = 1; }
        else if (k == PK_LINF) { g.wfrag = W(p); g.wsplit = split_math(); }
        else g.w = W(p);
    }

    int add(const std::string& name, std::initializer_list<long long> shape, int kind = PK_RAW) {
        Param p;
        p.name = name;
        p.nd = (int)shape.size();
        int i = 0;
        p.numel = 1;
        for (long long s : shape) { p.shape[i++] = s; p.numel *= (size_t)s; }
        p.kind = kind;
        p.packed = p.numel;
        // split images: three 16-bit pieces per weight + the trailer of per-output scales (2 floats per output; split_pack.hip)
        const size_t O = (size_t)p.shape[0];
        if (kind == PK_STEM) p.packed = O * STEM_KPAD;
        if ((kind == PK_STEM || kind == PK_LINF) && split_math()) p.packed = p.packed * 3 / 2 + 2 * O;
        if (kind == PK_CONV3S) p.packed = p.numel * 3 / 2 + 2 * O;
        if (kind == PK_CONV3W) p.packed = split_conv() ? (size_t)16 * O * p.shape[1] * 3 / 2 + 2 * O : (size_t)16 * O * p.shape[1];
        if (kind == PK_CONV3WU) p.packed = (size_t)4 * 16 * O * p.shape[1] * 3 / 2 + 8 * O;
        // output head: [tap][O][I] (out_conv_bwd, the op entry point) + [out_t_cols(O)][I], rows co * 9 + tap: the head as a 1x1 GEMM over
        // (cout, tap) columns followed by a 9-tap gather (misc.hip: out_gather_kernel)
        if (kind == PK_OUTCONV) p.packed = p.numel + (size_t)out_t_cols((int)O) * p.shape[1];
        params.push_back(p);
        pidx[name] = (int)params.size() - 1;
        return (int)params.size() - 1;
    }

    int build();
    int forward(const FwdIn& in, hipStream_t st, Arena& ar, const PrefixPlan* pp = nullptr, const SuffixPlan* sp = nullptr);
    int ensure_ws(int B, int T);
    int res_block(const ResP& r, Tens x0, const Tens* x1, int N, const float* film_all, const float* emb_unused,
                  hipStream_t st, Arena& ar, Tens* out);
    int attn_block(const AttnP& a, Tens x, int B, int T, const float* te_all, const int64_t* fidx, const float* amask,
                   hipStream_t st, Arena& ar, Tens* out);
    int gn_fold(const Tens& x0, const Tens* x1, int N, int gw, int gb, const float* film,
                int film_ld, hipStream_t st, Arena& ar, float** A, float** B, float** mr = nullptr);
    int gn_act(const Tens& x0, const Tens* x1, int N, int gw, int gb, const float* film, int film_ld, int act, float* y,
               hipStream_t st, Arena& ar);
    int backward(const FwdIn& in, const float* deps, float* dx, hipStream_t st, Arena& ar);
    int bwd_linear(const float* dy, int M, int pw, const float* res, float* out, hipStream_t st);
    int bwd_conv3(Tens dy, int N, int pw, int cout_bwd, const float* res, float* out, hipStream_t st, Arena& ar);
    int ensure_ws_guided(int B, int T);
    const float* WB(int p) const { return wbuf_bwd + params[p].off_bwd; }
    int linear(const float* a, int M, int K, int pw, int pb, int Nout, const float* wptr, const float* bptr, int act,
               const float* res, float* out, hipStream_t st, const Tens* stats_of = nullptr);
    void gemm_stats_table(Arena& ar, int M, int K, int Nout, Tens* t);
    void conv_split_stats_table(Arena& ar, const IgemmArgs& conv, int Cout, Tens* t);
};

// ------------------------------------------------------------------------------------------ topology
int vd_engine::build() {
    const int mc = cfg.num_channels, nrb = cfg.num_res_blocks;
    const int* mult; int nlev;
    switch (cfg.image_size) {
        case 256: mult = CH_MULT_256; nlev = 6; break;
        case 128: mult = CH_MULT_128; nlev = 5; break;
        case 64: mult = CH_MULT_64; nlev = 4; break;
        case 32: mult = CH_MULT_32; nlev = 4; break;
        default: set_error("unsupported image size: " + std::to_string(cfg.image_size)); return -1;
    }
    VD_REQUIRE(mc % 32 == 0, "num_channels must be a multiple of 32 (GroupNorm32)");
    VD_REQUIRE(cfg.num_heads > 0 && cfg.n_attention_ds >= 0 && cfg.n_attention_ds <= 8, "heads / attention_ds");
    E = mc * (cfg.time_embed_mult > 0 ? cfg.time_embed_mult : 4);
    auto in_att = [&](int ds) { for (int i = 0; i < cfg.n_attention_ds; ++i) if (cfg.attention_ds[i] == ds) return true; return false; };

    // The reference registers spatial_encoding first (a Parameter of the root module), but its shape is
    // known only after the input blocks are laid out; reserve the slot now.
    if (cfg.use_spatial_encoding) p_posenc = add("spatial_encoding", {1, 1, 1, 1}, PK_POSENC);
    p_te0w = add("time_embed.0.weight", {E, mc}, PK_LINF); p_te0b = add("time_embed.0.bias", {E});
    p_te2w = add("time_embed.2.weight", {E, E}, PK_LINF); p_te2b = add("time_embed.2.bias", {E});

    // 3x3 stride-1 convs at >= 8x8 with a multiple of 64 couts run as Winograd F(2x2,3x3); the rest on the generic kernel
    auto k3 = [&](int res_out, int cout) { return res_out >= 8 && cout % 64 == 0 ? (int)PK_CONV3W : (int)PK_CONV3; };
    // backward-data image of a 3x3 stride-1 conv with `co` outputs (= the forward's inputs) and `ci` inputs at resolution rs
    auto k3b = [&](int rs, int co, int ci) {
        return rs >= 8 && (rs & (rs - 1)) == 0 && co % 64 == 0 && ci % 32 == 0 && split_conv() ? (int)PK_CONV3W : (int)PK_CONV3;
    };
    auto bwd3 = [&](int p, int rs) { params[p].kind_bwd = k3b(rs, (int)params[p].shape[1], (int)params[p].shape[0]); };
    auto bwdl = [&](int p) { params[p].kind_bwd = PK_LINF; };
    auto add_res = [&](const std::string& pre, int cin, int cout, int rs) {
        ResP r; r.cin = cin; r.cout = cout;
        r.gn1w = add(pre + ".in_layers.0.weight", {cin}); r.gn1b = add(pre + ".in_layers.0.bias", {cin});
        r.c1w = add(pre + ".in_layers.2.weight", {cout, cin, 3, 3}, k3(rs, cout)); r.c1b = add(pre + ".in_layers.2.bias", {cout});
        const int eo = cfg.use_scale_shift_norm ? 2 * cout : cout;
        r.embw = add(pre + ".emb_layers.1.weight", {eo, E}, PK_LINF); r.embb = add(pre + ".emb_layers.1.bias", {eo});
        r.gn2w = add(pre + ".out_layers.0.weight", {cout}); r.gn2b = add(pre + ".out_layers.0.bias", {cout});
        r.c2w = add(pre + ".out_layers.3.weight", {cout, cout, 3, 3}, k3(rs, cout)); r.c2b = add(pre + ".out_layers.3.bias", {cout});
        if (cin != cout) {
            r.skw = add(pre + ".skip_connection.weight", {cout, cin, 1, 1}, PK_LINF); r.skb = add(pre + ".skip_connection.bias", {cout});
            bwdl(r.skw);
        }
        bwd3(r.c1w, rs); bwd3(r.c2w, rs);
        res.push_back(r);
        return (int)res.size() - 1;
    };
    auto add_att = [&](const std::string& pre, int C) {
        AttP a;
        a.qkvw = add(pre + ".qkv.weight", {3 * C, C}, PK_LINF); a.qkvb = add(pre + ".qkv.bias", {3 * C});
        a.projw = add(pre + ".proj_out.weight", {C, C}, PK_LINF); a.projb = add(pre + ".proj_out.bias", {C});
        a.normw = add(pre + ".norm.weight", {C}); a.normb = add(pre + ".norm.bias", {C});
        bwdl(a.qkvw); bwdl(a.projw);
        return a;
    };
    auto add_rpe = [&](const std::string& pre, int C) {
        RpeP r;
        if (cfg.use_rpe_net) {
            r.dw = add(pre + ".rpe_net.embed_distances.weight", {C, 3}); r.db = add(pre + ".rpe_net.embed_distances.bias", {C});
            r.tw = add(pre + ".rpe_net.embed_diffusion_time.weight", {C, E}, PK_LINF); r.tb = add(pre + ".rpe_net.embed_diffusion_time.bias", {C});
            r.ow = add(pre + ".rpe_net.out.weight", {C, C}, PK_LINF); r.ob = add(pre + ".rpe_net.out.bias", {C});
        } else {
            r.table = add(pre + ".lookup_table_weight", {2 * (long long)cfg.rp_beta + 1, cfg.num_heads, C / cfg.num_heads});
        }
        return r;
    };
    auto add_attn = [&](const std::string& pre, int C) {
        VD_REQUIRE(C % cfg.num_heads == 0 && (C / cfg.num_heads) % 8 == 0, "head dim must be a multiple of 8");
        AttnP a; a.C = C;
        a.sp = add_att(pre + ".spatial_attention", C);
        a.tp = add_att(pre + ".temporal_attention", C);
        a.rq = add_rpe(pre + ".temporal_attention.rpe_q", C);
        a.rk = add_rpe(pre + ".temporal_attention.rpe_k", C);
        a.rv = add_rpe(pre + ".temporal_attention.rpe_v", C);
        attn.push_back(a);
        return (int)attn.size() - 1;
    };
    auto add_conv = [&](const std::string& pre, int cin, int cout, int kind, int rs_bwd) {
        ConvP c; c.c = cout;
        c.w = add(pre + ".weight", {cout, cin, 3, 3}, kind); c.b = add(pre + ".bias", {cout});
        if (kind == PK_STEM) params[c.w].kind_bwd = PK_LINF;          // [STEM_KPAD][cout]: dcols = dy * W
        else bwd3(c.w, rs_bwd);                                       // Down: a stride-1 conv of the zero-stuffed gradient at the INPUT resolution
        convs.push_back(c);
        return (int)convs.size() - 1;
    };

    std::vector<int> chans;
    VD_REQUIRE(cfg.cond_emb_type >= 0 && cfg.cond_emb_type <= 2, "cond_emb_type: 0 channel, 1 duplicate|all, 2 t=0");
    const int stem_in = cfg.cond_emb_type == 0 ? 5 : (cfg.cond_emb_type == 1 ? 6 : 3);         // unet.py:932-940
    input_blocks.push_back({Layer{0, add_conv("input_blocks.0.0", stem_in, mc, PK_STEM, cfg.image_size)}});
    chans.push_back(mc);
    int ch = mc, ds = 1, first_ds = -1, first_ch = -1;
    n_before_attn = -1;
    for (int lvl = 0; lvl < nlev; ++lvl) {
        for (int k = 0; k < nrb; ++k) {
            if (in_att(ds) && n_before_attn < 0) { n_before_attn = (int)input_blocks.size(); first_ds = ds; first_ch = ch; }
            const std::string pre = "input_blocks." + std::to_string(input_blocks.size());
            std::vector<Layer> blk;
            int ri = add_res(pre + ".0", ch, mult[lvl] * mc, cfg.image_size / ds);
            blk.push_back(Layer{1, ri});
            ch = mult[lvl] * mc;
            if (in_att(ds)) { int ai = add_attn(pre + ".1", ch); if (ai < 0) return ai; blk.push_back(Layer{2, ai}); }
            input_blocks.push_back(blk);
            chans.push_back(ch);
        }
        if (lvl != nlev - 1) {
            const std::string pre = "input_blocks." + std::to_string(input_blocks.size());
            // Downsample (unet.py:98): stride-2 3x3 -> the split GEMM over an implicit im2col operand
            input_blocks.push_back({Layer{3, add_conv(pre + ".0.op", ch, ch, split_math() ? PK_CONV3S : PK_CONV3, cfg.image_size / ds)}});
            chans.push_back(ch);
            ds *= 2;
        }
    }
    if (n_before_attn < 0) { n_before_attn = (int)input_blocks.size(); first_ds = ds; first_ch = ch; }
    pos_res = cfg.image_size / first_ds; pos_ch = first_ch;
    if (p_posenc >= 0) {
        Param& p = params[p_posenc];
        p.nd = 4; p.shape[0] = 1; p.shape[1] = pos_ch; p.shape[2] = pos_res; p.shape[3] = pos_res;
        p.numel = p.packed = (size_t)pos_ch * pos_res * pos_res;
    }
    {
        int r0 = add_res("middle_block.0", ch, ch, cfg.image_size / ds);
        int a0 = add_attn("middle_block.1", ch); if (a0 < 0) return a0;
        int r1 = add_res("middle_block.2", ch, ch, cfg.image_size / ds);
        middle = {Layer{1, r0}, Layer{2, a0}, Layer{1, r1}};
    }
    for (int lvl = nlev - 1; lvl >= 0; --lvl) {
        for (int i = 0; i <= nrb; ++i) {
            const std::string pre = "output_blocks." + std::to_string(output_blocks.size());
            std::vector<Layer> blk;
            const int skip = chans.back(); chans.pop_back();
            blk.push_back(Layer{1, add_res(pre + ".0", ch + skip, mc * mult[lvl], cfg.image_size / ds)});
            ch = mc * mult[lvl];
            int li = 1;
            if (in_att(ds)) { int ai = add_attn(pre + "." + std::to_string(li++), ch); if (ai < 0) return ai; blk.push_back(Layer{2, ai}); }
            if (lvl && i == nrb) {
                // Upsample + conv: the sub-pixel form where the split Winograd kernel serves the SOURCE map
                const int rs_src = cfg.image_size / ds;
                int kup = k3(2 * rs_src, ch);
                if (kup == PK_CONV3W && split_conv() && rs_src >= 8 && (rs_src & (rs_src - 1)) == 0 && ch % 64 == 0) kup = PK_CONV3WU;
                blk.push_back(Layer{4, add_conv(pre + "." + std::to_string(li++) + ".conv", ch, ch, kup, 2 * rs_src)});
                ds /= 2;
            }
            output_blocks.push_back(blk);
        }
    }
    suf_blk = -1; suf_layer = -1;
    for (size_t i = 0; i < output_blocks.size(); ++i)
        for (size_t l = 0; l < output_blocks[i].size(); ++l)
            if (output_blocks[i][l].type == 2) { suf_blk = (int)i; suf_layer = (int)l; }
    final_ch = ch;
    p_outgw = add("out.0.weight", {ch}); p_outgb = add("out.0.bias", {ch});
    const int oc = cfg.learn_sigma ? 6 : 3;                                                    // script_util.py:129-131
    p_outw = add("out.2.weight", {oc, mc, 3, 3}, PK_OUTCONV); p_outb = add("out.2.bias", {oc});
    VD_REQUIRE(final_ch == mc, "channel_mult[0] must be 1");

    // ---- packed layout: [FiLM weights | FiLM biases | rpe-time weights | rpe-time biases | everything else]
    size_t off = 0;
    auto place = [&](int p) { params[p].off = off; off += (params[p].packed + 3) & ~(size_t)3; };
    film_w_off = off; film_total = 0;
    for (auto& r : res) { r.film_off = film_total; place(r.embw); film_total += (int)params[r.embw].shape[0]; }
    film_b_off = off;
    for (auto& r : res) place(r.embb);
    te_total = 0;
    if (cfg.use_rpe_net) {
        te_w_off = off;
        for (auto& a : attn) for (RpeP* r : {&a.rq, &a.rk, &a.rv}) { r->te_off = te_total; place(r->tw); te_total += a.C; }
        te_b_off = off;
        for (auto& a : attn) for (RpeP* r : {&a.rq, &a.rk, &a.rv}) place(r->tb);
    }
    std::vector<char> placed(params.size(), 0);
    for (auto& r : res) { placed[r.embw] = placed[r.embb] = 1; }
    if (cfg.use_rpe_net) for (auto& a : attn) for (RpeP* r : {&a.rq, &a.rk, &a.rv}) { placed[r->tw] = placed[r->tb] = 1; }
    for (size_t i = 0; i < params.size(); ++i) if (!placed[i]) place((int)i);
    packed_total = off;
    // The batched projections are ONE fragment-major matrix each: members share the region base and own a row range.
    for (auto& r : res) { Param& p = params[r.embw]; p.off = film_w_off; p.frag_rows = film_total; p.frag_row0 = r.film_off; }
    if (cfg.use_rpe_net)
        for (auto& a : attn)
            for (RpeP* r : {&a.rq, &a.rk, &a.rv}) { Param& p = params[r->tw]; p.off = te_w_off; p.frag_rows = te_total; p.frag_row0 = r->te_off; }
    // ---- backward-data image (its own buffer, set only when a guided step is wanted: vd_set_bwd_weight_storage)
    size_t offb = 0;
    for (auto& p : params) {
        if (p.kind_bwd < 0) continue;
        const size_t O = (size_t)p.shape[1], I = (size_t)p.shape[0];          // outputs / inputs of the BACKWARD operator
        if (p.kind == PK_STEM) p.packed_bwd = (size_t)STEM_KPAD * I * 3 / 2 + 2 * STEM_KPAD;
        else if (p.kind_bwd == PK_LINF) p.packed_bwd = O * I * 3 / 2 + 2 * O;
        else if (p.kind_bwd == PK_CONV3W) p.packed_bwd = 16 * O * I * 3 / 2 + 2 * O;
        else p.packed_bwd = 9 * O * I;
        p.off_bwd = offb;
        offb += (p.packed_bwd + 3) & ~(size_t)3;
    }
    packed_bwd_total = offb;
    for (auto& p : params)
        if (p.kind == PK_LINF) {
            if (!p.frag_rows) p.frag_rows = (int)p.shape[0];
            VD_REQUIRE(p.shape[0] % 32 == 0 && p.shape[1] % 32 == 0 && p.frag_row0 % 32 == 0, "linear layer dims must be multiples of 32");
        }
    return 0;
}

// ------------------------------------------------------------------------------------------ helpers
static IgemmArgs linear_args(int M, int K, int Nout) {
    IgemmArgs g{};
    g.C0 = K; g.Cin = K; g.nfr = M; g.Hs = 1; g.Ws = 1; g.stride = 1; g.ksz = 1; g.Ho = 1; g.Wo = 1;
    g.wsplit = split_math(); g.ldo = Nout; g.Cout = Nout; g.M = M; g.res_ld = Nout;
    return g;
}

// The output of a linear layer that is a GroupNorm's input (t: H x H pixels per frame, rows in (frame, pixel) order): where
// the split GEMM runs it in one launch, its epilogue writes the per-channel partial sums -- allocate the table (also in the
// arena's dry run) and let the tensor carry it; otherwise the tensor stays without and gn_fold runs a statistics pass
void vd_engine::gemm_stats_table(Arena& ar, int M, int K, int Nout, Tens* t) {
    IgemmArgs g = linear_args(M, K, Nout);
    g.wfrag = reinterpret_cast<const float*>(this);                                  // any non-null pointer: the shape test only
    const int HW = t->H * t->H, rows = gemm_split_stats_rows(M, Nout);
    if (!gemm_split_supported(g) || igemm_frames_per_launch(g) < M || HW % rows || M % HW) return;
    t->split = HW / rows;
    t->part = ar.get<double>((size_t)(M / HW) * t->split * Nout * 2);
}

// the same for a 3x3 conv that runs on the split GEMM's implicit-im2col mode (the stride-2 Downsample convs): the rows of
// its output are the Ho x Wo pixels of consecutive frames
void vd_engine::conv_split_stats_table(Arena& ar, const IgemmArgs& conv, int Cout, Tens* t) {
    IgemmArgs g = conv;
    g.wfrag = reinterpret_cast<const float*>(this); g.wsplit = split_math(); g.Cout = Cout; g.ldo = Cout; g.res_ld = Cout;
    const int HW = g.Ho * g.Wo, rows = gemm_split_stats_rows(g.M, Cout);
    if (!conv_split_supported(g) || igemm_frames_per_launch(g) < g.nfr || HW % rows) return;
    t->split = HW / rows;
    t->part = ar.get<double>((size_t)g.nfr * t->split * Cout * 2);
}

int vd_engine::linear(const float* a, int M, int K, int, int, int Nout, const float* wptr, const float* bptr, int act,
                      const float* resid, float* out, hipStream_t st, const Tens* stats_of) {
    IgemmArgs g{};
    g.src0 = a; g.src1 = nullptr; g.C0 = K; g.Cin = K;
    g.nfr = M; g.Hs = 1; g.Ws = 1; g.ups = 0; g.stride = 1; g.pad = 0; g.ksz = 1; g.Ho = 1; g.Wo = 1;
    g.w = nullptr; g.wfrag = wptr;           // every nn.Linear weight is stored fragment-major (PK_LINF)
    g.wsplit = split_math();
    g.bias = bptr; g.affA = nullptr; g.affB = nullptr; g.act = act;
    g.res = resid; g.res_ld = Nout; g.fbias = nullptr; g.fbias_ld = 0;
    g.out = out; g.ldo = Nout; g.Cout = Nout; g.M = M;
    if (stats_of && stats_of->part) { g.stats = stats_of->part; g.stats_split = stats_of->split; g.stats_hw = stats_of->H * stats_of->H; }
    return igemm_p(g, st);
}

int vd_engine::gn_fold(const Tens& x0, const Tens* x1, int N, int gw, int gb,
                       const float* film, int film_ld, hipStream_t st, Arena& ar, float** A, float** Bp, float** mr) {
    const int HW = x0.H * x0.H, C = x0.C + (x1 ? x1->C : 0);
    // per-channel partial sums of each source: the table its producer wrote, or one statistics pass over it
    const Tens* src[2] = {&x0, x1};
    const double* part[2] = {nullptr, nullptr};
    int split[2] = {0, 0};
    for (int i = 0; i < 2; ++i) {
        if (!src[i]) continue;
        if (src[i]->part) { part[i] = src[i]->part; split[i] = src[i]->split; continue; }
        split[i] = gn_stats_split(N, HW, src[i]->C);
        double* pt = ar.get<double>((size_t)N * split[i] * src[i]->C * 2);
        part[i] = pt;
        if (ar.dry) continue;
        ProfScope ps(PC_GN_STATS, 0.0, 4.0 * N * HW * src[i]->C, st);
        int rc = launch_gn_stats(src[i]->p, nullptr, src[i]->C, src[i]->C, N, HW, pt, split[i], st);
        if (rc) return rc;
    }
    *A = ar.get<float>((size_t)N * C);
    *Bp = ar.get<float>((size_t)N * C);
    float* mrp = tape && mr ? ar.get<float>((size_t)N * 64) : nullptr;      // group mean / rstd for the backward pass
    if (mr) *mr = mrp;
    if (ar.dry) return 0;
    return launch_gn_affine(part[0], split[0], x0.C, part[1], split[1], (double)HW * (C / 32), W(gw), W(gb), film, film_ld, N,
                            C, *A, *Bp, st, mrp);
}

// GroupNorm(+FiLM) + SiLU of a (virtually concatenated) tensor into y in ONE launch: the activation pass folds the statistics
// itself (norm.hip: affine_act_fold_kernel).  For consumers that need nothing but y; not with a tape (the backward pass reads
// A, B and mean / rstd).
static bool gn_fold_fused() { return true; }
int vd_engine::gn_act(const Tens& x0, const Tens* x1, int N, int gw, int gb, const float* film, int film_ld, int act, float* y,
                      hipStream_t st, Arena& ar) {
    const int HW = x0.H * x0.H, C = x0.C + (x1 ? x1->C : 0);
    const Tens* src[2] = {&x0, x1};
    const double* part[2] = {nullptr, nullptr};
    int split[2] = {0, 0};
    for (int i = 0; i < 2; ++i) {
        if (!src[i]) continue;
        if (src[i]->part) { part[i] = src[i]->part; split[i] = src[i]->split; continue; }
        split[i] = gn_stats_split(N, HW, src[i]->C);
        double* pt = ar.get<double>((size_t)N * split[i] * src[i]->C * 2);
        part[i] = pt;
        if (ar.dry) continue;
        ProfScope ps(PC_GN_STATS, 0.0, 4.0 * N * HW * src[i]->C, st);
        int rc = launch_gn_stats(src[i]->p, nullptr, src[i]->C, src[i]->C, N, HW, pt, split[i], st);
        if (rc) return rc;
    }
    // Big tensors (the 64 x 64 and 32 x 32 levels of a full window) take the two-launch form: (A, B) per (frame, channel) by
    // gn_final_affine, then the pass on many short blocks -- 5.95 TB/s against the 5.2 of the long blocks the in-kernel fold needs
    // (tools/probes/stream_probe.hip); the 7 us of the extra launch are repaid from ~64 MB on.  Same formulas and fp64 sums; the ORDER of the
    // sums differs (gn_final_affine deals a group's pairs to eight lanes, the folding pass walks them in sequence), so (A, B) agree to fp64
    // rounding -- after the cast to fp32 in practice to the bit (tools/switch_check.py), by construction to 1 ulp.  The 64 MB switch is by the
    // size of the whole layer call (g_sel_nfr): a full window and the compact suffix batch of the same layer take the same form.
    static const size_t big = getenv("VD_AA_BIG_MB") ? (size_t)atol(getenv("VD_AA_BIG_MB")) << 20 : (size_t)64 << 20;
    const bool two = (size_t)(g_sel_nfr > N ? g_sel_nfr : N) * HW * C * 4 >= big;      // by the size of the whole layer call: the compact suffix batch takes the full window's form
    float* Aab = two ? ar.get<float>((size_t)N * C) : nullptr;
    float* Bab = two ? ar.get<float>((size_t)N * C) : nullptr;
    if (ar.dry) return 0;
    if (two) {
        int rc = launch_gn_affine(part[0], split[0], x0.C, part[1], split[1], (double)HW * (C / 32), W(gw), W(gb), film, film_ld, N, C, Aab, Bab, st);
        if (rc) return rc;
        ProfScope ps(PC_ELEMENTWISE, 0.0, 8.0 * N * HW * C, st);
        return launch_affine_act(x0.p, x1 ? x1->p : nullptr, x0.C, C, Aab, Bab, N, HW, act, y, st);
    }
    GnFold f{part[0], split[0], part[1], split[1], (double)HW * (C / 32), W(gw), W(gb), film, film_ld};
    ProfScope ps(PC_ELEMENTWISE, 0.0, 8.0 * N * HW * C, st);
    return launch_affine_act_fold(x0.p, x1 ? x1->p : nullptr, x0.C, C, f, N, HW, act, y, st);
}

// the epilogue of the Winograd conv (the kernel every PK_CONV3W weight runs on) writes the GroupNorm partial sums of
// its output: allocate the table and hand it to the launch
static double* stats_table(Arena& ar, int N, int Hout, int Cout, int* split) {
    *split = conv_wino_stats_split(Hout);
    return ar.get<double>((size_t)N * *split * Cout * 2);
}

// split-K scratch of a small-grid Winograd conv (conv_wino_r64.hip).  The dry run that sizes the workspace takes the largest
// request over every batch 1..N: the prefix-cache forward runs the same layer on a compact batch (fewer frames = a smaller
// grid = possibly MORE slices), and the arena must hold that too
static size_t ksplit_scratch(const Arena& ar, int N, int H, int cin, int cout) {
    if (!ar.dry) return conv_wino_r64_ksplit_floats(N, H, cin, cout, g_sel_nfr);
    size_t m = 0;
    for (int n = 1; n <= N; ++n) m = std::max(m, conv_wino_r64_ksplit_floats(n, H, cin, cout));
    for (int n = 1; n <= N; ++n) m = std::max(m, conv_wino_r64_ksplit_floats(n, H, cin, cout, N));     // a compact suffix: slices chosen for N frames
    return m;
}

static IgemmArgs conv_args(Tens x0, const Tens* x1, int N, int ksz, int stride, int ups) {
    IgemmArgs g{};
    g.src0 = x0.p; g.C0 = x0.C; g.Cin = x0.C;
    if (x1) { g.src1 = x1->p; g.Cin += x1->C; }
    g.nfr = N; g.Hs = x0.H; g.Ws = x0.H; g.ups = ups; g.stride = stride; g.pad = ksz == 3 ? 1 : 0; g.ksz = ksz;
    g.nfr_sel = g_sel_nfr;
    const int Hl = x0.H << ups;
    g.Ho = g.Wo = (Hl + 2 * g.pad - ksz) / stride + 1;
    g.M = N * g.Ho * g.Wo;
    return g;
}

int vd_engine::res_block(const ResP& r, Tens x0, const Tens* x1, int N, const float* film_all, const float*,
                         hipStream_t st, Arena& ar, Tens* out) {
    const int H = x0.H, HW = H * H;
    const int cin = x0.C + (x1 ? x1->C : 0);
    VD_REQUIRE(cin == r.cin, "ResBlock input channels");
    const float* s1 = x1 ? x1->p : nullptr;
    float *A1 = nullptr, *B1 = nullptr, *A2 = nullptr, *B2 = nullptr, *mr1 = nullptr, *mr2 = nullptr;
    int rc;
    const float* film = film_all + r.film_off;
    // The skip convolution (1x1 over the block input, unet.py:159-166) reads the tensor the first GroupNorm+SiLU reads: where
    // it runs on the 128x128 tile of gemm_split.hip its column blocks 0 write that activation image from the rows they stage
    // (IgemmArgs::side) -- the block input, the largest tensor of a decoder block, is read once instead of twice.  Decided by
    // shape alone (the dry run lays the arena out the same way).
    IgemmArgs gsk{};
    bool fuse_skip = false;
    if (r.skw >= 0) {
        gsk = conv_args(x0, x1, N, 1, 1, 0);
        set_w(gsk, r.skw); gsk.Cout = r.cout; gsk.ldo = r.cout; gsk.side_hw = HW;
        IgemmArgs one = gsk; one.nfr = std::max(1, std::min(N, igemm_frames_per_launch(gsk)));
        one.M = (g_sel_nfr ? std::max(g_sel_nfr, one.nfr) : one.nfr) * HW;                     // (the tile class is chosen for the whole layer call)
        one.wfrag = one.wfrag ? one.wfrag : reinterpret_cast<const float*>(0x1000);      // dry run: no weight image yet, the shape decides
        fuse_skip = params[r.skw].kind == PK_LINF && split_math() && gemm_split_side_supported(one);
    }
    // the statistics folded by the activation pass itself (gn_act) wherever nothing else reads (A, B): not for the skip
    // convolution that writes the image (it takes the pair as arrays), not with a tape
    const bool fold = !tape && gn_fold_fused();
    // Where conv_wino_z128.hip takes the convolution and one or two cout blocks share a patch, the GroupNorm(+FiLM) affine + SiLU runs in
    // the kernel's patch staging and the activation image is never written (conv_wino_z128_act_shape): the pair goes in as arrays.  The
    // dry run lays the arena out for both forms (a compact batch may decide differently).
    const int nsel = g_sel_nfr > N ? g_sel_nfr : N;
    const bool za1 = !tape && params[r.c1w].kind == PK_CONV3W && f16_math() && x0.C % 16 == 0 && conv_wino_z128_act_shape(nsel, H, cin, r.cout);
    // ... and then the skip convolution has no image to write: it runs as a plain 1x1 (the dry run keeps the larger layout: sk_early under the transient)
    if (za1 && !ar.dry) fuse_skip = false;
    const bool fold1 = fold && !fuse_skip;
    const bool za2 = !tape && params[r.c2w].kind == PK_CONV3W && f16_math() && conv_wino_z128_act_shape(nsel, H, r.cout, r.cout);
    if ((!fold1 || za1 || ar.dry) && (rc = gn_fold(x0, x1, N, r.gn1w, r.gn1b, nullptr, 0, st, ar, &A1, &B1, &mr1))) return rc;
    float* h = ar.get<float>((size_t)N * HW * r.cout);
    Tens ht{h, r.cout, H};
    if (params[r.c1w].kind == PK_CONV3W) ht.part = stats_table(ar, N, H, r.cout, &ht.split);
    Tens ot{nullptr, r.cout, H};
    if (params[r.c2w].kind == PK_CONV3W) ot.part = stats_table(ar, N, H, r.cout, &ot.split);
    float* sk_early = fuse_skip ? ar.get<float>((size_t)N * HW * r.cout) : nullptr;
    {   // SiLU(GroupNorm(concat(x0, x1))) once, into a transient; the conv then reads plain activations (norm.hip)
        const size_t mk = ar.mark();
        float* a1 = ar.get<float>((size_t)N * HW * cin);
        const size_t ksf = ksplit_scratch(ar, N, H, cin, r.cout);      // small grids: split-K scratch (conv_wino_r64.hip)
        float* ksw = ksf ? ar.get<float>(ksf) : nullptr;
        if (fold1 && !za1 && (rc = gn_act(x0, x1, N, r.gn1w, r.gn1b, nullptr, 0, 1, a1, st, ar))) return rc;
        if (!ar.dry) {
            if (fuse_skip) {
                gsk.bias = W(r.skb); gsk.out = sk_early; gsk.side = a1; gsk.sideA = A1; gsk.sideB = B1;
                if ((rc = igemm_p(gsk, st))) return rc;
            } else if (!fold1 && !za1 && (rc = affine_act(x0.p, s1, x0.C, cin, A1, B1, N, HW, a1, st))) return rc;
            Tens at{a1, cin, H};
            IgemmArgs g = conv_args(za1 ? x0 : at, za1 ? x1 : nullptr, N, 3, 1, 0);
            if (za1) { g.affA = A1; g.affB = B1; g.act = 1; }
            set_w(g, r.c1w); g.bias = W(r.c1b);
            g.out = h; g.ldo = r.cout; g.Cout = r.cout; g.stats = ht.part; g.stats_split = ht.split;
            g.ksplit_ws = ksw; g.ksplit_ws_floats = ksf;
            if (!cfg.use_scale_shift_norm) { g.fbias = film; g.fbias_ld = film_total; }    // h + emb_out (unet.py:196)
            VD_REQUIRE(!za1 || act_conv_will_dispatch(g), "res_block: in_layers conv was planned on conv_wino_z128's activating form (no activation image "
                                                          "written) but the call as built is not one it takes");
            if ((rc = igemm_p(g, st))) return rc;
        }
        ar.release(mk);
    }
    if ((!fold || za2 || ar.dry) && (rc = gn_fold(ht, nullptr, N, r.gn2w, r.gn2b, cfg.use_scale_shift_norm ? film : nullptr, film_total, st, ar, &A2, &B2, &mr2)))
        return rc;
    const float* skip = x0.p;
    if (fuse_skip) skip = sk_early;
    else if (r.skw >= 0) {
        float* sk = ar.get<float>((size_t)N * HW * r.cout);
        if (!ar.dry) {
            IgemmArgs g = conv_args(x0, x1, N, 1, 1, 0);
            set_w(g, r.skw); g.bias = W(r.skb); g.out = sk; g.ldo = r.cout; g.Cout = r.cout;
            if ((rc = igemm_p(g, st))) return rc;
        }
        skip = sk;
    } else {
        VD_REQUIRE(x1 == nullptr, "identity skip over a concatenated input");
    }
    float* o = ar.get<float>((size_t)N * HW * r.cout);
    {
        const size_t mk = ar.mark();
        float* a2 = ar.get<float>((size_t)N * HW * r.cout);
        const size_t ksf = ksplit_scratch(ar, N, H, r.cout, r.cout);
        float* ksw = ksf ? ar.get<float>(ksf) : nullptr;
        if (fold && !za2 && (rc = gn_act(ht, nullptr, N, r.gn2w, r.gn2b, cfg.use_scale_shift_norm ? film : nullptr, film_total, 1, a2, st, ar))) return rc;
        if (!ar.dry) {
            if (!fold && !za2 && (rc = affine_act(h, nullptr, r.cout, r.cout, A2, B2, N, HW, a2, st))) return rc;
            Tens at{a2, r.cout, H};
            IgemmArgs g = conv_args(za2 ? ht : at, nullptr, N, 3, 1, 0);
            if (za2) { g.affA = A2; g.affB = B2; g.act = 1; }
            set_w(g, r.c2w); g.bias = W(r.c2b);
            g.res = skip; g.res_ld = r.cout; g.out = o; g.ldo = r.cout; g.Cout = r.cout;
            g.stats = ot.part; g.stats_split = ot.split;
            g.ksplit_ws = ksw; g.ksplit_ws_floats = ksf;
            VD_REQUIRE(!za2 || act_conv_will_dispatch(g), "res_block: out_layers conv was planned on conv_wino_z128's activating form (no activation image "
                                                          "written) but the call as built is not one it takes");
            if ((rc = igemm_p(g, st))) return rc;
        }
        ar.release(mk);
    }
    ot.p = o;
    *out = ot;
    if (tape) {
        tape->ops.push_back(TapeOp{1, (int)tape->res.size()});
        tape->res.push_back(TapeRes{(int)(&r - res.data()), x0, x1 ? *x1 : Tens{}, x1 != nullptr, N, A1, B1, mr1, h, A2, B2, mr2, o});
    }
    return 0;
}

int vd_engine::attn_block(const AttnP& a, Tens x, int B, int T, const float* te_all, const int64_t* fidx,
                          const float* amask, hipStream_t st, Arena& ar, Tens* out) {
    const int C = a.C, H = x.H, HW = H * H, N = B * T;
    const size_t tok = (size_t)N * HW;
    const float scale = 1.0f / sqrtf((float)(C / cfg.num_heads));
    int rc;
    // ---- temporal attention over the T frames of each (batch, pixel)      (unet.py:246-255)
    float* xn = ar.get<float>(tok * C);
    float* qkv = ar.get<float>(tok * 3 * C);
    // relative-position terms in memory order q, k, v: the three nets of a block have identical shapes and sit at a
    // constant stride in the packed weights, so each of their two layers is ONE launch (blockIdx.y / .z = net)
    const RpeP* rp[3] = {&a.rq, &a.rk, &a.rv};
    const size_t rrows = (size_t)B * T * T;
    const int blk = (int)(&a - attn.data());
    const bool pre = !rpe_R.empty();                                        // rpe_all has produced this forward's R tensors already
    float* Ehid = cfg.use_rpe_net && !pre ? ar.get<float>(3 * rrows * C) : nullptr;
    float* Rall = pre ? rpe_R[3 * blk] : ar.get<float>(3 * rrows * C);     // (memory order q, k, v)
    float* R[3] = {Rall + rrows * C, Rall, Rall + 2 * rrows * C};          // k, q, v as the attention kernel takes them
    float* o = ar.get<float>(tok * C);
    float* xt = ar.get<float>(tok * C);
    Tens xt_t{xt, C, H};                               // the spatial attention's GroupNorm reads it: statistics from the proj_out GEMM
    gemm_stats_table(ar, (int)tok, C, C, &xt_t);
    if (!ar.dry) {
        { ProfScope ps(PC_GN_TEMPORAL, 0.0, 8.0 * tok * C, st);
          rc = launch_gn_temporal(x.p, W(a.tp.normw), W(a.tp.normb), B, T, HW, C, xn, st); }
        if (rc) return rc;
        if ((rc = linear(xn, (int)tok, C, 0, 0, 3 * C, W(a.tp.qkvw), W(a.tp.qkvb), 0, nullptr, qkv, st))) return rc;
        if (pre) {
        } else if (cfg.use_rpe_net) {
            const int zs_dw = (int)(W(rp[1]->dw) - W(rp[0]->dw)), zs_db = (int)(W(rp[1]->db) - W(rp[0]->db));
            const int zs_ow = (int)(W(rp[1]->ow) - W(rp[0]->ow)), zs_ob = (int)(W(rp[1]->ob) - W(rp[0]->ob));
            VD_REQUIRE(W(rp[2]->dw) - W(rp[1]->dw) == zs_dw && W(rp[2]->db) - W(rp[1]->db) == zs_db &&
                       W(rp[2]->ow) - W(rp[1]->ow) == zs_ow && W(rp[2]->ob) - W(rp[1]->ob) == zs_ob &&
                       rp[1]->te_off - rp[0]->te_off == C && rp[2]->te_off - rp[1]->te_off == C, "RPE nets: constant stride");
            if ((rc = launch_rpe_hidden(te_all + rp[0]->te_off, te_total, W(rp[0]->dw), W(rp[0]->db), fidx, B, T, C, Ehid, 3, C,
                                        zs_dw, zs_db, rrows * C, st))) return rc;
            IgemmArgs g{};
            g.src0 = Ehid; g.C0 = C; g.Cin = C; g.nfr = (int)rrows; g.Hs = g.Ws = g.Ho = g.Wo = 1; g.stride = 1; g.ksz = 1;
            g.wfrag = W(rp[0]->ow); g.wsplit = split_math(); g.bias = W(rp[0]->ob); g.out = Rall; g.ldo = C; g.Cout = C; g.M = (int)rrows;
            g.zcount = 3; g.zs_a = g.zs_out = (int)(rrows * C); g.zs_w = zs_ow; g.zs_bias = zs_ob;
            if ((rc = igemm_p(g, st))) return rc;
        } else {
            for (int i = 0; i < 3; ++i)
                if ((rc = launch_rpe_table(W(rp[i]->table), fidx, B, T, C, cfg.rp_alpha, cfg.rp_beta, cfg.rp_gamma,
                                           Rall + i * rrows * C, st))) return rc;
        }
        AttnTemporalArgs ta{qkv, R[0], R[1], R[2], amask, o, B, T, HW, C, cfg.num_heads,
                            cfg.allow_interactions_between_padding, scale};
        { const double Fd = C / cfg.num_heads;
          ProfScope ps(PC_ATTN_TEMPORAL, 10.0 * B * HW * cfg.num_heads * T * T * Fd, 16.0 * tok * C + 12.0 * rrows * C, st);
          rc = launch_attn_temporal(ta, st); }
        if (rc) return rc;
        if (!attn_cap_t.empty() && attn_seq < (int)attn_cap_t.size() && attn_cap_t[attn_seq] &&
            (rc = launch_attn_temporal_weights(ta, attn_cap_t[attn_seq], st))) return rc;
        // proj_out + residual on the NORMALISED activations (unet.py:537-538)
        if ((rc = linear(o, (int)tok, C, 0, 0, C, W(a.tp.projw), W(a.tp.projb), 0, xn, xt, st, &xt_t))) return rc;
    }
    // ---- spatial attention over the HW pixels of each frame               (unet.py:258-267)
    float *A = nullptr, *Bf = nullptr, *mrs = nullptr;
    const bool fold = !tape && gn_fold_fused();
    if (!fold && (rc = gn_fold(xt_t, nullptr, N, a.sp.normw, a.sp.normb, nullptr, 0, st, ar, &A, &Bf, &mrs))) return rc;
    float* xn2 = ar.get<float>(tok * C);
    float* qkv2 = ar.get<float>(tok * 3 * C);
    float* o2 = ar.get<float>(tok * C);
    float* xs = ar.get<float>(tok * C);
    Tens xs_t{xs, C, H};                               // the block's output: the next ResBlock's GroupNorm (or a skip) reads it
    gemm_stats_table(ar, (int)tok, C, C, &xs_t);
    if (fold && (rc = gn_act(xt_t, nullptr, N, a.sp.normw, a.sp.normb, nullptr, 0, 0, xn2, st, ar))) return rc;
    if (!ar.dry) {
        if (!fold) { ProfScope ps(PC_ELEMENTWISE, 0.0, 8.0 * tok * C, st); rc = launch_affine_apply(xt, A, Bf, N, HW, C, xn2, st); }
        if (rc) return rc;
        if ((rc = linear(xn2, (int)tok, C, 0, 0, 3 * C, W(a.sp.qkvw), W(a.sp.qkvb), 0, nullptr, qkv2, st))) return rc;
        AttnSpatialArgs sa{qkv2, o2, N, HW, C, cfg.num_heads, scale};
        { ProfScope ps(PC_ATTN_SPATIAL, 4.0 * N * (double)HW * HW * C, 16.0 * tok * C, st); rc = launch_attn_spatial(sa, st); }
        if (rc) return rc;
        if (!attn_cap_s.empty() && attn_seq < (int)attn_cap_s.size() && attn_cap_s[attn_seq] &&
            (rc = launch_attn_spatial_weights(sa, attn_cap_s[attn_seq], st))) return rc;
        if ((rc = linear(o2, (int)tok, C, 0, 0, C, W(a.sp.projw), W(a.sp.projb), 0, xn2, xs, st, &xs_t))) return rc;
    }
    *out = xs_t;
    if (!ar.dry) ++attn_seq;
    if (tape) {
        tape->ops.push_back(TapeOp{2, (int)tape->attn.size()});
        tape->attn.push_back(TapeAttn{(int)(&a - attn.data()), x, B, T, xn, qkv, R[0], R[1], R[2], o, xt, A, Bf, mrs, xn2, qkv2, o2, xs, amask});
    }
    return 0;
}

// The relative-position nets of every attention block (RPENet, unet.py:283-298: silu(Linear(feat(d)) + Linear(emb)) -> Linear) read the
// timestep embedding and the frame indices only -- nothing a block computes.  Per block they were two small dependent launches (the three
// nets of a block batched): 22 of the ~210 launches of a step, each paying the ~5 us of a dependent dispatch on top of 5 - 15 us of work
// that leaves most CUs idle.  Here: per channel width ONE hidden-layer launch and ONE output-layer GEMM over all nets of that width (the
// 3 x 5 nets of 384 channels, the 3 x 6 of 512), at the start of the forward.  Same kernels, same arithmetic per net: bit-identical R.
// the attention blocks by channel width, and per width the device tables of offsets the two launches read (never inside a stream capture:
// ensure_ws calls this)
static bool rpe_all_on() { static const bool on = getenv("VD_NO_RPE_ALL") == nullptr; return on; }     // (A/B switch: the per-block launches of rounds 1-4)

int vd_engine::rpe_tables() {
    if (!cfg.use_rpe_net || !split_math() || attn.empty() || !te_total || !rpe_all_on()) return 0;
    if (rpe_groups.empty()) {
        for (size_t b = 0; b < attn.size(); ++b) {
            size_t gi = 0;
            while (gi < rpe_groups.size() && rpe_groups[gi].C != attn[b].C) ++gi;
            if (gi == rpe_groups.size()) rpe_groups.push_back(RpeGroup{attn[b].C, {}});
            rpe_groups[gi].blocks.push_back((int)b);
        }
    }
    for (auto& g : rpe_groups) {
        if (g.d_hid) continue;
        const int nn = 3 * (int)g.blocks.size();
        std::vector<long long> th(3 * nn), to(2 * nn);
        for (size_t i = 0; i < g.blocks.size(); ++i) {
            const AttnP& a = attn[g.blocks[i]];
            const RpeP* rp[3] = {&a.rq, &a.rk, &a.rv};
            for (int k = 0; k < 3; ++k) {
                const int z = 3 * (int)i + k;
                th[3 * z] = rp[k]->te_off; th[3 * z + 1] = (long long)params[rp[k]->dw].off; th[3 * z + 2] = (long long)params[rp[k]->db].off;
                to[2 * z] = (long long)params[rp[k]->ow].off; to[2 * z + 1] = (long long)params[rp[k]->ob].off;
            }
        }
        VD_HIP(hipMalloc(reinterpret_cast<void**>(&g.d_hid), th.size() * sizeof(long long)));
        VD_HIP(hipMalloc(reinterpret_cast<void**>(&g.d_out), to.size() * sizeof(long long)));
        VD_HIP(hipMemcpy(g.d_hid, th.data(), th.size() * sizeof(long long), hipMemcpyHostToDevice));
        VD_HIP(hipMemcpy(g.d_out, to.data(), to.size() * sizeof(long long), hipMemcpyHostToDevice));
    }
    return 0;
}

int vd_engine::rpe_all(const float* te, const int64_t* fidx, int B, int T, hipStream_t st, Arena& ar) {
    rpe_R.clear();
    if (!cfg.use_rpe_net || !split_math() || attn.empty() || !te_total || !rpe_all_on()) return 0;
    if (ar.dry && rpe_groups.empty()) {                                // (the dry run sizes the arena before any table exists: widths only)
        for (size_t b = 0; b < attn.size(); ++b) {
            size_t gi = 0;
            while (gi < rpe_groups.size() && rpe_groups[gi].C != attn[b].C) ++gi;
            if (gi == rpe_groups.size()) rpe_groups.push_back(RpeGroup{attn[b].C, {}});
            rpe_groups[gi].blocks.push_back((int)b);
        }
    }
    VD_REQUIRE(!rpe_groups.empty(), "rpe_all: ensure_ws has not run");
    const size_t rrows = (size_t)B * T * T;
    rpe_R.assign(3 * attn.size(), nullptr);
    int rc = 0;
    for (auto& g : rpe_groups) {
        const int nn = 3 * (int)g.blocks.size(), C = g.C;
        float* Rg = ar.get<float>((size_t)nn * rrows * C);
        for (size_t i = 0; i < g.blocks.size(); ++i)
            for (int k = 0; k < 3; ++k) rpe_R[3 * g.blocks[i] + k] = Rg + (3 * i + k) * rrows * C;
        const size_t mk = ar.mark();
        float* Eh = ar.get<float>((size_t)nn * rrows * C);
        ar.release(mk);                                                // a transient: the two launches are stream-ordered
        if (ar.dry) continue;
        VD_REQUIRE(g.d_hid && g.d_out, "rpe_all: offset tables missing (ensure_ws builds them)");
        if ((rc = launch_rpe_hidden_tab(te, te_total, wbuf, g.d_hid, fidx, B, T, C, Eh, nn, rrows * C, st))) return rc;
        IgemmArgs q{};
        q.src0 = Eh; q.C0 = C; q.Cin = C; q.nfr = (int)rrows; q.Hs = q.Ws = q.Ho = q.Wo = 1; q.stride = 1; q.ksz = 1;
        q.wfrag = wbuf + params[attn[g.blocks[0]].rq.ow].off; q.bias = wbuf + params[attn[g.blocks[0]].rq.ob].off;   // (problem 0's: the shape checks read them)
        q.wsplit = split_math(); q.out = Rg; q.ldo = C; q.Cout = C; q.M = (int)rrows;
        q.zcount = nn; q.zs_a = q.zs_out = (int)(rrows * C); q.ztab = g.d_out; q.zbase = wbuf;
        if ((rc = igemm_p(q, st))) return rc;
    }
    return 0;
}

// ------------------------------------------------------------------------------------------ forward
int vd_engine::forward(const FwdIn& in, hipStream_t st, Arena& ar, const PrefixPlan* pp, const SuffixPlan* sp) {
    const int B = in.B, T = in.T, N = B * T, S = cfg.image_size, mc = cfg.num_channels;
    int rc;
    VD_REQUIRE(!pp || (!tape && !ar.dry && pp->store && n_before_attn > 0), "prefix plan: executor steps only");
    VD_REQUIRE(!sp || (!tape && sp->n > 0 && sp->n <= N && (ar.dry || sp->list)), "suffix plan: executor steps only, at least one frame");
    struct SelGuard { ~SelGuard() { g_sel_nfr = 0; } } sel_guard;      // (every return path leaves the selection hint cleared)
    const int Npre = pp ? pp->n : N;                                 // frames the blocks before the first attention layer run on
    float* x8 = ar.get<float>((size_t)Npre * S * S * STEM_KPAD);    // im2col of the network input
    float* tfr = ar.get<float>(N);
    float* amask = ar.get<float>(N);
    float* tsin = ar.get<float>((size_t)N * mc);
    float* e1 = ar.get<float>((size_t)N * E);
    float* emb = ar.get<float>((size_t)N * E);
    float* film = ar.get<float>((size_t)N * film_total);
    float* te = te_total ? ar.get<float>((size_t)N * te_total) : nullptr;
    float* ftv = cfg.use_frame_encoding ? ar.get<float>(N) : nullptr;
    float* femb = cfg.use_frame_encoding ? ar.get<float>((size_t)N * pos_ch) : nullptr;
    if (!ar.dry) {
        VD_REQUIRE(d_freq_time && n_freq_time == mc / 2, "vd_set_freqs not called (time frequencies)");
        AssembleArgs aa{in.x, in.obs_src, in.obs, in.lat, in.km, in.t_model, in.obs_mode, B, T, S, S, STEM_KPAD, cfg.cond_emb_type, x8, tfr, amask};
        if (pp) {                                                    // per-frame scalars of every frame, im2col of the listed ones
            AssembleArgs sc = aa; sc.scalars_only = 1;
            if ((rc = launch_assemble(sc, st))) return rc;
            aa.frame_list = pp->list; aa.n_list = pp->n;
        }
        if ((rc = launch_assemble(aa, st))) return rc;
        if ((rc = launch_sinus_embed(tfr, N, mc, d_freq_time, tsin, st))) return rc;
        if ((rc = linear(tsin, N, mc, 0, 0, E, W(p_te0w), W(p_te0b), 0, nullptr, e1, st))) return rc;
        if ((rc = linear(e1, N, E, 0, 0, E, W(p_te2w), W(p_te2b), 1, nullptr, emb, st))) return rc;
        // every ResBlock's emb_layers (SiLU -> Linear, unet.py:143-150) in ONE GEMM; same for RPENet's
        // embed_diffusion_time (no SiLU, unet.py:294)
        if ((rc = linear(emb, N, E, 0, 0, film_total, wbuf + film_w_off, wbuf + film_b_off, 1, nullptr, film, st))) return rc;
        if (te_total && (rc = linear(emb, N, E, 0, 0, te_total, wbuf + te_w_off, wbuf + te_b_off, 0, nullptr, te, st))) return rc;
        if (cfg.use_frame_encoding) {
            VD_REQUIRE(d_freq_frame && n_freq_frame == pos_ch / 2, "vd_set_freqs not called (frame frequencies)");
            if ((rc = launch_frame_t(in.fidx, B, T, cfg.enforce_position_invariance, ftv, st))) return rc;
            if ((rc = launch_sinus_embed(ftv, N, pos_ch, d_freq_frame, femb, st))) return rc;
        }
    }
    rpe_R.clear();
    if (!(pp && pp->build_only) && (rc = rpe_all(te, in.fidx, B, T, st, ar))) return rc;      // every block's relative-position tensors, up front
    attn_seq = 0;
    std::vector<Tens> hs;
    Tens h{x8, STEM_KPAD, S};
    int Nrun = Npre;                                                 // batch of the block being run: Npre in the prefix, N behind it
    const float* film_full = film;
    if (pp || ar.dry) {                                              // FiLM rows of the listed frames, compactly (the dry run sizes it for any list)
        float* fc = ar.get<float>((size_t)std::max(pp ? Npre : N, 1) * film_total);
        if (pp) {
            if ((rc = launch_gather_rows(film_full, pp->list, Npre, film_total, fc, st))) return rc;
            film = fc;
        }
    }
    auto run = [&](const std::vector<Layer>& blk, Tens in0, const Tens* in1, Tens* outp, size_t l0 = 0, size_t l1 = ~(size_t)0) -> int {
        const int N = Nrun;
        Tens cur = in0;
        const Tens* second = in1;
        for (size_t li = l0; li < std::min(l1, blk.size()); ++li) {
            const Layer& L = blk[li];
            Tens nxt{};
            if (L.type == 1) {
                if ((rc = res_block(res[L.idx], cur, second, N, film, nullptr, st, ar, &nxt))) return rc;
            } else if (L.type == 2) {
                if ((rc = attn_block(attn[L.idx], cur, B, T, te, in.fidx, amask, st, ar, &nxt))) return rc;
            } else if (L.type == 0) {                                 // stem: im2col (assemble_kernel) x [64][mc] GEMM
                const ConvP& c = convs[L.idx];
                float* o = ar.get<float>((size_t)N * S * S * c.c);
                nxt = Tens{o, c.c, S};
                gemm_stats_table(ar, N * S * S, STEM_KPAD, c.c, &nxt);
                if (!ar.dry && (rc = linear(cur.p, N * S * S, STEM_KPAD, 0, 0, c.c, W(c.w), W(c.b), 0, nullptr, o, st, &nxt))) return rc;
                if (tape) { tape->ops.push_back(TapeOp{0, (int)tape->conv.size()}); tape->conv.push_back(TapeConv{0, L.idx, cur, nxt}); }
            } else {
                const ConvP& c = convs[L.idx];
                const int stride = L.type == 3 ? 2 : 1, ups = L.type == 4 ? 1 : 0;
                IgemmArgs g = conv_args(cur, nullptr, N, 3, stride, ups);
                float* o = ar.get<float>((size_t)g.M * c.c);
                nxt = Tens{o, c.c, g.Ho};
                const bool phase = params[c.w].kind == PK_CONV3WU;                    // sub-pixel Upsample conv: four table entries per tile group
                if (params[c.w].kind == PK_CONV3W) nxt.part = stats_table(ar, N, g.Ho, c.c, &nxt.split);
                else if (phase) { nxt.split = conv_wino_ups_stats_split(cur.H); nxt.part = ar.get<double>((size_t)N * nxt.split * c.c * 2); }
                else if (stride == 2) conv_split_stats_table(ar, g, c.c, &nxt);       // Downsample on the split GEMM
                const size_t mk = ar.mark();
                const size_t ksf = stride == 1 && !phase ? ksplit_scratch(ar, N, g.Ho, cur.C, c.c) : 0;
                float* ksw = ksf ? ar.get<float>(ksf) : nullptr;
                ar.release(mk);                                                       // a transient: the launches are stream-ordered
                if (!ar.dry) {
                    g.ksplit_ws = ksw; g.ksplit_ws_floats = ksf;
                    set_w(g, c.w); g.bias = W(c.b); g.out = o; g.ldo = c.c; g.Cout = c.c;
                    g.stats = nxt.part; g.stats_split = nxt.split;
                    if (nxt.part && params[c.w].kind != PK_CONV3W && !phase) g.stats_hw = g.Ho * g.Wo;
                    if ((rc = igemm_p(g, st))) return rc;
                }
                if (tape) { tape->ops.push_back(TapeOp{0, (int)tape->conv.size()}); tape->conv.push_back(TapeConv{L.type, L.idx, cur, nxt}); }
            }
            cur = nxt;
            second = nullptr;
            // a request the dry run did not see was served from the arena's base (Arena::get): stop before anything else is enqueued on
            // top of the aliased tensors -- inside a capture this also keeps the broken step out of the graph
            VD_REQUIRE(!ar.overflow, "workspace overflow: the forward asked for more than the dry run measured (this step's output is invalid)");
        }
        *outp = cur;
        return 0;
    };
    for (size_t i = 0; i < input_blocks.size(); ++i) {
        if (pp && (int)i < n_before_attn) {
            // compact batch -> this block's persistent full-size tensor (allocated by the window's first pass, outside the arena)
            Tens hc{};
            if (Npre > 0 && (rc = run(input_blocks[i], h, nullptr, &hc))) return rc;
            PrefixStore& ps = *pp->store;
            if (ps.tens.size() <= i) { ps.tens.resize(i + 1, nullptr); ps.parts.resize(i + 1, nullptr); ps.floats.resize(i + 1, 0); }
            if (Npre == 0) { VD_REQUIRE(ps.tens[i], "prefix cache: no tensor to reuse"); }
            else {
                const size_t per = (size_t)hc.H * hc.H * hc.C;
                if (!ps.tens[i]) {
                    VD_REQUIRE(pp->build_only, "prefix cache: the window's first pass allocates");
                    VD_HIP(hipMalloc(reinterpret_cast<void**>(&ps.tens[i]), (size_t)N * per * sizeof(float)));
                    ps.floats[i] = per | ((size_t)hc.C << 40) | ((size_t)hc.H << 52);
                    if (hc.part) VD_HIP(hipMalloc(reinterpret_cast<void**>(&ps.parts[i]), (size_t)N * hc.C * 2 * sizeof(double)));
                }
                VD_REQUIRE((ps.parts[i] != nullptr) == (hc.part != nullptr), "prefix cache: statistics table");
                if ((rc = launch_scatter_rows(hc.p, pp->list, Npre, per, ps.tens[i], st))) return rc;
                if (hc.part && (rc = launch_scatter_stats(hc.part, hc.split, hc.C, pp->list, Npre, ps.parts[i], st))) return rc;
            }
            const int Cc = (int)((ps.floats[i] >> 40) & 0xfff), Hc = (int)(ps.floats[i] >> 52);
            Tens full{ps.tens[i], Cc, Hc, ps.parts[i], ps.parts[i] ? 1 : 0};
            if ((int)i + 1 == n_before_attn) {
                if (pp->build_only) return 0;
                h = full; Nrun = N; film = const_cast<float*>(film_full);
            } else {
                h = hc;                                              // the next prefix block continues on the compact batch
            }
            hs.push_back(full);
        } else {
            if ((rc = run(input_blocks[i], h, nullptr, &h))) return rc;
            hs.push_back(h);
        }
        if ((int)i + 1 == n_before_attn && (cfg.use_spatial_encoding || cfg.use_frame_encoding)) {
            // added AFTER the skip push (unet.py:815-818): the skip keeps the un-encoded tensor
            const size_t n = (size_t)N * h.H * h.H * h.C;
            float* y = ar.get<float>(n);
            if (!ar.dry) {
                VD_REQUIRE(h.H == pos_res && h.C == pos_ch, "positional encoding shape");
                if ((rc = launch_posenc_add(h.p, cfg.use_spatial_encoding ? W(p_posenc) : nullptr, femb, N, h.H * h.H, h.C, y, st)))
                    return rc;
            }
            if (tape) { tape->ops.push_back(TapeOp{5, (int)tape->pos.size()}); tape->pos.push_back(TapePos{h, Tens{y, h.C, h.H}}); }
            h = Tens{y, h.C, h.H};
        }
    }
    if ((rc = run(middle, h, nullptr, &h))) return rc;
    // window suffix skip: from behind the last attention layer on, the listed frames only
    const int Nsuf = sp ? sp->n : N;
    bool compact = false;
    auto gather_t = [&](const Tens& t) -> Tens {                      // rows and GroupNorm partial sums of the listed frames
        const size_t per = (size_t)t.H * t.H * t.C;
        Tens c{ar.get<float>((size_t)Nsuf * per), t.C, t.H, nullptr, t.split};
        if (t.part) c.part = ar.get<double>((size_t)Nsuf * t.split * t.C * 2);
        if (!ar.dry) {
            if (launch_gather_rows(t.p, sp->list, Nsuf, per, c.p, st)) return Tens{};
            if (t.part && launch_gather_rows(reinterpret_cast<const float*>(t.part), sp->list, Nsuf, (size_t)t.split * t.C * 4,
                                             reinterpret_cast<float*>(c.part), st)) return Tens{};
        }
        return c;
    };
    auto go_compact = [&]() -> int {
        float* fc = ar.get<float>((size_t)Nsuf * film_total);
        if (!ar.dry && (rc = launch_gather_rows(film_full, sp->list, Nsuf, film_total, fc, st))) return rc;
        h = gather_t(h);
        VD_REQUIRE(ar.dry || h.p, "suffix skip: gather");
        film = fc; Nrun = Nsuf; g_sel_nfr = N; compact = true;
        return 0;
    };
    if (sp && suf_blk < 0 && (rc = go_compact())) return rc;
    for (size_t i = 0; i < output_blocks.size(); ++i) {
        Tens skip = hs.back(); hs.pop_back();
        if (sp && !compact && (int)i == suf_blk) {
            if ((rc = run(output_blocks[i], h, &skip, &h, 0, suf_layer + 1))) return rc;        // cat([h, hs.pop()]) read in place
            if ((rc = go_compact())) return rc;
            if ((rc = run(output_blocks[i], h, nullptr, &h, suf_layer + 1))) return rc;
            continue;
        }
        if (compact) { skip = gather_t(skip); VD_REQUIRE(ar.dry || skip.p, "suffix skip: gather"); }
        if ((rc = run(output_blocks[i], h, &skip, &h))) return rc;
    }
    float *A, *Bf, *mrh = nullptr;
    if ((rc = gn_fold(h, nullptr, Nrun, p_outgw, p_outgb, nullptr, 0, st, ar, &A, &Bf, &mrh))) return rc;
    if (tape) { tape->head = h; tape->headA = A; tape->headB = Bf; tape->head_mr = mrh; }
    const int oc = cfg.learn_sigma ? 6 : 3;
    float* eps_c = compact ? ar.get<float>((size_t)Nsuf * oc * S * S) : nullptr;
    float* head_t = ar.get<float>((size_t)Nrun * S * S * out_t_cols(oc));
    if (!ar.dry) {
        VD_REQUIRE(h.H == S && h.C == final_ch, "output head shape");
        { // out = conv3x3(silu(gn(h))) with 3 | 6 outputs (unet.py:744-749,838): T[pixel][cout * 9 + tap] = silu(A h + B) . w[cout][:][tap] as ONE
          // 1x1 GEMM over the 27 | 54 (cout, tap) columns (the generic kernel: GroupNorm affine + SiLU in its operand load), then
          // eps[cout][y][x] = bias + sum over the 9 taps of T at the neighbour the tap points to (out_gather_kernel)
          ProfScope ps(PC_OUT_CONV, 2.0 * Nrun * S * S * h.C * 27.0, 4.0 * Nrun * S * S * (h.C + 3.0), st);
          static const bool old_head = getenv("VD_HEAD_GENERIC") != nullptr;      // A/B switch: the generic fp32 kernel (rounds 4)
          if (!old_head && head_gemm_supported(S * S, h.C, out_t_cols(oc))) {
              rc = launch_head_gemm(h.p, A, Bf, W(p_outw) + (size_t)9 * oc * h.C, Nrun, S * S, h.C, out_t_cols(oc), head_t, st);
          } else {
              IgemmArgs g = conv_args(h, nullptr, Nrun, 1, 1, 0);
              g.w = W(p_outw) + (size_t)9 * oc * h.C; g.wsplit = 0; g.affA = A; g.affB = Bf; g.act = 1;
              g.out = head_t; g.ldo = out_t_cols(oc); g.Cout = out_t_cols(oc);
              rc = launch_igemm(g, st);
          }
          if (!rc) rc = launch_out_gather(head_t, W(p_outb), Nrun, S, S, out_t_cols(oc), oc, compact ? eps_c : in.eps, st); }
        if (rc) return rc;
        if (compact) {                                               // the other frames' eps: zeros (their samples stay finite; nobody reads them)
            VD_HIP(hipMemsetAsync(in.eps, 0, (size_t)N * oc * S * S * sizeof(float), st));
            if ((rc = launch_scatter_rows(eps_c, sp->list, Nsuf, (size_t)oc * S * S, in.eps, st))) return rc;
        }
    }
    VD_REQUIRE(!ar.overflow, "workspace overflow: the forward asked for more than the dry run measured (this step's output is invalid)");
    return 0;
}

int vd_engine::ensure_ws(int B, int T) {
    if (B == ws_B && T == ws_T && ws && ws_suf == suffix_skip_on) return 0;      // the common case: every step of a window
    { const int trc = rpe_tables(); if (trc) return trc; }
    // with the window suffix skip enabled the arena also holds the gathered skip tensors: sized for the worst list (every frame)
    const long long key = ((long long)B << 32) | (unsigned)T | (suffix_skip_on ? 1ll << 61 : 0);
    auto it = ws_peaks.find(key);
    if (it == ws_peaks.end()) {
        Arena dry; dry.dry = true;
        FwdIn fi{}; fi.B = B; fi.T = T;
        int rc = forward(fi, nullptr, dry);
        if (!rc && suffix_skip_on) {
            Arena dry2; dry2.dry = true;
            SuffixPlan wp; wp.n = B * T;
            rc = forward(fi, nullptr, dry2, nullptr, &wp);
            dry.peak = std::max(dry.peak, dry2.peak);
        }
        if (rc) return rc;
        it = ws_peaks.emplace(key, dry.peak).first;
    }
    const size_t tail = (it->second + 255) & ~(size_t)255;
    const size_t tm_bytes = ((size_t)B * sizeof(float) + 255) & ~(size_t)255;
    const size_t need = tail + tm_bytes + (size_t)B * T * (cfg.learn_sigma ? 6 : 3) * cfg.image_size * cfg.image_size * sizeof(float);
    if (need > ws_cap) {
        // captured window graphs hold addresses inside the old workspace: they die with it (a window in flight is lost:
        // vd_window_run then reports "window graphs invalidated")
        if (win_cur >= 0) win_lost = true;
        drop_window_graphs();
        if (ws) VD_HIP(hipFree(ws));
        ws = nullptr; ws_cap = 0; ws_B = ws_T = 0;
        VD_HIP(hipMalloc(reinterpret_cast<void**>(&ws), need));
        ws_cap = need;
    }
    ws_B = B; ws_T = T; ws_tail = tail; ws_suf = suffix_skip_on;
    return 0;
}


// ------------------------------------------------------------------------------------------ backward (use_gradient_method)
// d(out)/d(in) of every taped op, walked in reverse.  Gradients live in the same arena behind the forward's tensors; a
// tensor's gradient buffer is created by its first consumer (assigned) and accumulated into by the others (the skip
// connections: an encoder output feeds the next encoder block AND a decoder block).
namespace {
struct GradMap {
    std::unordered_map<const float*, std::pair<float*, bool>> m;     // tensor -> (gradient, written)
    Arena* ar;
    float* get(const Tens& t, int N, bool* fresh) {
        auto it = m.find(t.p);
        if (it == m.end()) it = m.emplace(t.p, std::make_pair(ar->get<float>((size_t)N * t.H * t.H * t.C), false)).first;
        *fresh = !it->second.second;
        it->second.second = true;
        return it->second.first;
    }
    float* have(const float* p) const { auto it = m.find(p); return it == m.end() ? nullptr : it->second.first; }
};
}  // namespace

// out[M][K_fwd] = dy[M][N_fwd] * W  (+ res): the forward's split GEMM over the transposed image of parameter pw
int vd_engine::bwd_linear(const float* dy, int M, int pw, const float* resid, float* out, hipStream_t st) {
    const Param& p = params[pw];
    const int Kb = p.kind == PK_STEM ? (int)p.shape[0] : (int)p.shape[0], Nb = p.kind == PK_STEM ? STEM_KPAD : (int)p.shape[1];
    IgemmArgs g = linear_args(M, Kb, Nb);
    g.src0 = dy; g.wfrag = WB(pw); g.wsplit = 1; g.res = resid; g.out = out;
    VD_REQUIRE(gemm_split_supported(g), "backward linear layer: shape not covered by gemm_split.hip");
    return launch_igemm(g, st);
}

// out[N][H][H][cout_bwd] = conv3x3_s1(dy, rotated transposed kernel of parameter pw) (+ res, in place allowed)
int vd_engine::bwd_conv3(Tens dy, int N, int pw, int cout_bwd, const float* resid, float* out, hipStream_t st, Arena& ar) {
    const Param& p = params[pw];
    IgemmArgs g = conv_args(dy, nullptr, N, 3, 1, 0);
    const size_t mk = ar.mark();
    const size_t ksf = p.kind_bwd == PK_CONV3W ? ksplit_scratch(ar, N, dy.H, dy.C, cout_bwd) : 0;
    g.ksplit_ws = ksf ? ar.get<float>(ksf) : nullptr; g.ksplit_ws_floats = ksf;
    ar.release(mk);
    if (ar.dry) return 0;
    if (p.kind_bwd == PK_CONV3W) { g.wwino = WB(pw); g.wsplit = 2; } else { g.w = WB(pw); g.wsplit = 0; }
    g.res = resid; g.res_ld = cout_bwd; g.out = out; g.ldo = cout_bwd; g.Cout = cout_bwd;
    return launch_igemm(g, st);
}

int vd_engine::backward(const FwdIn& in, const float* deps, float* dx, hipStream_t st, Arena& ar) {
    const int B = in.B, T = in.T, N = B * T, S = cfg.image_size;
    const bool dry = ar.dry;
    int rc;
    GradMap gm; gm.ar = &ar;
    bool fresh;
    // GroupNorm(+act) backward with its workspaces
    auto gn_bwd = [&](const Tens& x0, const Tens* x1, const float* A, const float* Bv, const float* mr, const float* dy, int act,
                      float* dx0, int acc0, float* dx1, int acc1, const float* extra) -> int {
        const int C = x0.C + (x1 ? x1->C : 0), HW = x0.H * x0.H;
        const size_t mk = ar.mark();
        double* part = ar.get<double>((size_t)N * gn_bwd_split(N, HW, C) * C * 2);
        float* K = ar.get<float>((size_t)N * 64);
        ar.release(mk);
        if (dry) return 0;
        GnBwdArgs a{x0.p, x1 ? x1->p : nullptr, x0.C, C, A, Bv, mr, dy, act, N, HW, dx0, dx1, acc0, acc1, extra, part, K};
        return launch_gn_bwd(a, st);
    };
    // ---- output head: eps = conv(silu(gn(h)))   (unet.py:744-749,838)
    {
        const Tens& h = tape->head;
        float* dh = gm.get(h, N, &fresh);
        const size_t mk = ar.mark();
        float* da = ar.get<float>((size_t)N * S * S * h.C);
        if (!dry && (rc = launch_out_conv_bwd(deps, W(p_outw), N, S, S, h.C, 3, da, st))) return rc;
        if ((rc = gn_bwd(h, nullptr, tape->headA, tape->headB, tape->head_mr, da, 1, dh, 0, nullptr, 0, nullptr))) return rc;
        ar.release(mk);
    }
    for (int oi = (int)tape->ops.size() - 1; oi >= 0; --oi) {
        const TapeOp op = tape->ops[oi];
        if (op.kind == 1) {                                              // ---- ResBlock (unet.py:185-198)
            const TapeRes& t = tape->res[op.i];
            const ResP& r = res[t.idx];
            const int H = t.x0.H, HW = H * H, cin = r.cin;
            float* d_o = gm.have(t.o);
            VD_REQUIRE(dry || d_o, "backward: ResBlock output without a gradient");
            if (!d_o) d_o = gm.get(Tens{t.o, r.cout, H}, N, &fresh);
            bool f0, f1 = true;
            float* dx0 = gm.get(t.x0, N, &f0);
            float* dx1 = t.has_x1 ? gm.get(t.x1, N, &f1) : nullptr;
            const size_t mk = ar.mark();
            float* d_a2 = ar.get<float>((size_t)N * HW * r.cout);
            float* d_h = ar.get<float>((size_t)N * HW * r.cout);
            float* d_a1 = ar.get<float>((size_t)N * HW * cin);
            float* d_sk = r.skw >= 0 ? ar.get<float>((size_t)N * HW * cin) : nullptr;
            if ((rc = bwd_conv3(Tens{d_o, r.cout, H}, N, r.c2w, r.cout, nullptr, d_a2, st, ar))) return rc;
            if ((rc = gn_bwd(Tens{t.h, r.cout, H}, nullptr, t.A2, t.B2, t.mr2, d_a2, 1, d_h, 0, nullptr, 0, nullptr))) return rc;
            if ((rc = bwd_conv3(Tens{d_h, r.cout, H}, N, r.c1w, cin, nullptr, d_a1, st, ar))) return rc;
            if (r.skw >= 0 && !dry && (rc = bwd_linear(d_o, N * HW, r.skw, nullptr, d_sk, st))) return rc;
            // the skip path's gradient (identity: d_o itself) rides on the GroupNorm backward's write
            if ((rc = gn_bwd(t.x0, t.has_x1 ? &t.x1 : nullptr, t.A1, t.B1, t.mr1, d_a1, 1, dx0, !f0, dx1, !f1, r.skw >= 0 ? d_sk : d_o))) return rc;
            ar.release(mk);
        } else if (op.kind == 2) {                                       // ---- FactorizedAttentionBlock (unet.py:236-267)
            const TapeAttn& t = tape->attn[op.i];
            const AttnP& a = attn[t.idx];
            const int C = a.C, H = t.x.H, HW = H * H;
            const size_t tok = (size_t)N * HW;
            const float scale = 1.0f / sqrtf((float)(C / cfg.num_heads));
            float* d_xs = gm.have(t.xs);
            VD_REQUIRE(dry || d_xs, "backward: attention output without a gradient");
            if (!d_xs) d_xs = gm.get(Tens{t.xs, C, H}, N, &fresh);
            float* dxin = gm.get(t.x, N, &fresh);
            const size_t mk = ar.mark();
            float* d_o2 = ar.get<float>(tok * C);
            float* d_qkv = ar.get<float>(tok * 3 * C);
            float* d_xn2 = ar.get<float>(tok * C);
            float* d_xt = ar.get<float>(tok * C);
            float* d_o = ar.get<float>(tok * C);
            float* d_xn = ar.get<float>(tok * C);
            AttnSpatialArgs sa{t.qkv2, nullptr, N, HW, C, cfg.num_heads, scale};
            float* sws = ar.get<float>(attn_spatial_bwd_ws_floats(sa));
            if (!dry) {
                // spatial half: xs = xn2 + proj(attn(qkv(xn2))), xn2 = gn(xt)
                if ((rc = bwd_linear(d_xs, (int)tok, a.sp.projw, nullptr, d_o2, st))) return rc;
                if ((rc = launch_attn_spatial_bwd(sa, d_o2, d_qkv, sws, st))) return rc;
                if ((rc = bwd_linear(d_qkv, (int)tok, a.sp.qkvw, d_xs, d_xn2, st))) return rc;
            }
            if ((rc = gn_bwd(Tens{t.xt, C, H}, nullptr, t.A, t.Bf, t.mr, d_xn2, 0, d_xt, 0, nullptr, 0, nullptr))) return rc;
            if (!dry) {
                // temporal half: xt = xn + proj(attn(qkv(xn), R)), xn = temporal gn(x)
                if ((rc = bwd_linear(d_xt, (int)tok, a.tp.projw, nullptr, d_o, st))) return rc;
                AttnTemporalArgs ta{t.qkv, t.Rk, t.Rq, t.Rv, t.amask, nullptr, t.B, t.T, HW, C, cfg.num_heads, cfg.allow_interactions_between_padding, scale};
                if ((rc = launch_attn_temporal_bwd(ta, d_o, d_qkv, st))) return rc;
                if ((rc = bwd_linear(d_qkv, (int)tok, a.tp.qkvw, d_xt, d_xn, st))) return rc;
                if ((rc = launch_gn_temporal_bwd(t.x.p, W(a.tp.normw), d_xn, t.B, t.T, HW, C, !fresh, dxin, st))) return rc;
            }
            ar.release(mk);
        } else if (op.kind == 5) {                                       // ---- + positional encodings: the gradient passes
            const TapePos& t = tape->pos[op.i];
            float* d_y = gm.have(t.out.p);
            VD_REQUIRE(dry || d_y, "backward: positional-encoding output without a gradient");
            if (!d_y) d_y = gm.get(t.out, N, &fresh);
            float* d_h = gm.get(t.in, N, &fresh);
            if (!dry && (rc = launch_add(d_y, (size_t)N * t.in.H * t.in.H * t.in.C, !fresh, d_h, st))) return rc;
        } else {
            const TapeConv& t = tape->conv[op.i];
            const ConvP& c = convs[t.idx];
            float* d_out = gm.have(t.out.p);
            VD_REQUIRE(dry || d_out, "backward: conv output without a gradient");
            if (!d_out) d_out = gm.get(t.out, N, &fresh);
            if (t.type == 0) {                                           // ---- stem: dcols = dy * W, then col2im onto x
                const size_t mk = ar.mark();
                float* dcols = ar.get<float>((size_t)N * S * S * STEM_KPAD);
                if (!dry) {
                    if ((rc = bwd_linear(d_out, N * S * S, c.w, nullptr, dcols, st))) return rc;
                    if ((rc = launch_stem_col2im(dcols, in.obs, in.lat, in.km, N, S, S, STEM_KPAD, cfg.cond_emb_type, dx, st))) return rc;
                }
                ar.release(mk);
            } else if (t.type == 3) {                                    // ---- Downsample (stride 2): stride-1 conv of the zero-stuffed gradient
                float* d_in = gm.get(t.in, N, &fresh);
                const size_t mk = ar.mark();
                float* dz = ar.get<float>((size_t)N * t.in.H * t.in.H * c.c);
                if (!dry && (rc = launch_zero_stuff2(d_out, N, t.out.H, t.out.H, c.c, dz, st))) return rc;
                if ((rc = bwd_conv3(Tens{dz, c.c, t.in.H}, N, c.w, t.in.C, fresh ? nullptr : d_in, d_in, st, ar))) return rc;
                ar.release(mk);
            } else {                                                     // ---- Upsample: conv backward at 2H, then the 2x2 sums
                float* d_in = gm.get(t.in, N, &fresh);
                const size_t mk = ar.mark();
                float* dup = ar.get<float>((size_t)N * t.out.H * t.out.H * t.in.C);
                if ((rc = bwd_conv3(Tens{d_out, c.c, t.out.H}, N, c.w, t.in.C, nullptr, dup, st, ar))) return rc;
                if (!dry && (rc = launch_sumpool2(dup, N, t.in.H, t.in.H, t.in.C, !fresh, d_in, st))) return rc;
                ar.release(mk);
            }
        }
    }
    return 0;
}

// workspace of a guided step: forward (taped, nothing released that the backward reads) + backward + the step's own buffers
int vd_engine::ensure_ws_guided(int B, int T) {
    { const int trc = rpe_tables(); if (trc) return trc; }
    const long long key = ((long long)B << 32) | (unsigned)T | (1ll << 62);
    auto it = ws_peaks.find(key);
    if (it == ws_peaks.end()) {
        Arena dry; dry.dry = true;
        Tape tp; tape = &tp;
        FwdIn fi{}; fi.B = B; fi.T = T;
        int rc = forward(fi, nullptr, dry);
        if (!rc) rc = backward(fi, nullptr, nullptr, nullptr, dry);
        tape = nullptr;
        if (rc) return rc;
        it = ws_peaks.emplace(key, dry.peak).first;
    }
    const size_t per = (size_t)B * T * 3 * cfg.image_size * cfg.image_size * sizeof(float);
    const size_t need = ((it->second + 255) & ~(size_t)255) + 4096 + 8 * per;
    if (need > ws_cap) {
        if (win_cur >= 0) win_lost = true;
        drop_window_graphs();
        if (ws) VD_HIP(hipFree(ws));
        ws = nullptr; ws_cap = 0; ws_B = ws_T = 0;
        VD_HIP(hipMalloc(reinterpret_cast<void**>(&ws), need));
        ws_cap = need;
    }
    ws_B = ws_T = 0;                       // the plain step recomputes its tail offsets on its next call
    return 0;
}

// ------------------------------------------------------------------------------------------ C ABI
extern "C" {

const char* vd_last_error(void) { return g_last_error.c_str(); }
const char* vd_version(void) {
    return math_mode() == MATH_F16X3 ? "vdamd 0.4 (gfx950; VD_MATH=f16x3: fp32 operands as two fp16 pieces (22 significand bits), three piece products, fp32 accumulation)"
           : math_mode() == MATH_BF16X6 ? "vdamd 0.4 (gfx950; VD_MATH=bf16x6: fp32 operands split exactly into three bf16 pieces, six piece products, fp32 accumulation)"
                                        : "vdamd 0.4 (gfx950; VD_MATH=fp32: every matrix product on the fp32 MFMA)";
}

int vd_create(const vd_config* cfg, vd_engine** out) {
    VD_REQUIRE(cfg && out, "null argument");
    vd_engine* e = new vd_engine();
    e->cfg = *cfg;
    int rc = e->build();
    if (rc) { delete e; return rc; }
    *out = e;
    return 0;
}

void vd_destroy(vd_engine* e) { delete e; }

int vd_param_count(vd_engine* e) { return e ? (int)e->params.size() : -1; }

int vd_param_info(vd_engine* e, int i, char* name, int cap, int* ndim, long long shape[4]) {
    VD_REQUIRE(e && i >= 0 && i < (int)e->params.size(), "parameter index");
    const Param& p = e->params[i];
    if (name && cap > 0) { std::strncpy(name, p.name.c_str(), cap - 1); name[cap - 1] = 0; }
    if (ndim) *ndim = p.nd;
    if (shape) for (int k = 0; k < 4; ++k) shape[k] = k < p.nd ? p.shape[k] : 1;
    return 0;
}

long long vd_weights_bytes(vd_engine* e) { return e ? (long long)(e->packed_total * sizeof(float)) : -1; }

// FNV-1a over everything that decides where a parameter's bytes sit in the packed buffer and how they are encoded
unsigned long long vd_weights_layout_id(vd_engine* e) {
    if (!e) return 0;
    unsigned long long h = 1469598103934665603ull;
    auto mix = [&](unsigned long long v) { for (int i = 0; i < 8; ++i) { h ^= (v >> (8 * i)) & 0xff; h *= 1099511628211ull; } };
    mix(math_mode()); mix(e->packed_total); mix(e->params.size());
    for (const Param& p : e->params) {
        for (char c : p.name) mix((unsigned char)c);
        mix(p.kind); mix(p.off); mix(p.packed); mix(p.frag_rows); mix(p.frag_row0);
    }
    return h;
}

// One process drives one GPU (the reference's launcher does the same, command_launchers.py:32-62): kernel attributes and
// CU counts are cached process-wide, an engine owns one workspace and runs its steps on one stream at a time.
static int g_bound_device = -1;

int vd_set_weight_storage(vd_engine* e, void* buf, long long bytes) {
    VD_REQUIRE(e && buf, "null argument");
    VD_REQUIRE(bytes >= (long long)(e->packed_total * sizeof(float)), "weight buffer too small");
    int dev = -1;
    VD_HIP(hipGetDevice(&dev));
    if (g_bound_device < 0) g_bound_device = dev;
    VD_REQUIRE(dev == g_bound_device, "libvdamd drives ONE device per process (one process per GPU); this process is already bound to another device");
    e->device = dev;
    e->wbuf_on_host = false;
    e->wbuf = static_cast<float*>(buf);
    return 0;
}

int vd_set_weight_storage_host(vd_engine* e, void* host_buf, long long bytes) {
    VD_REQUIRE(e && host_buf, "null argument");
    VD_REQUIRE(bytes >= (long long)(e->packed_total * sizeof(float)), "weight buffer too small");
    e->wbuf = static_cast<float*>(host_buf);
    e->wbuf_on_host = true;
    return 0;
}

int vd_load_weight(vd_engine* e, const char* name, const float* host, long long numel) {
    VD_REQUIRE(e && name && host, "null argument");
    VD_REQUIRE(e->wbuf, "vd_set_weight_storage first");
    auto it = e->pidx.find(name);
    if (it == e->pidx.end()) { set_error(std::string("unexpected key in state_dict: ") + name); return -1; }
    Param& p = e->params[it->second];
    if ((long long)p.numel != numel) {
        set_error("size mismatch for " + p.name + ": expected " + std::to_string(p.numel) + " got " + std::to_string(numel));
        return -1;
    }
    std::vector<float> tmp;
    const float* src = host;
    int rc_put = 0;
    if (p.kind == PK_LINF && split_math()) {
        // rows [row0, row0+rows) of a [frag_rows][K] matrix -> K/16 contiguous pieces of its bf16-split fragment image
        const int rows = (int)p.shape[0], K = (int)p.shape[1];
        std::vector<unsigned short> sp(split_image_u16(rows, K));
        pack_linear_split(host, sp.data(), rows, K, rows, 0);
        const size_t piece = (size_t)(rows / 32) * 1536;                   // ushorts per k-step of this member
        for (int ks = 0; ks < K / 16; ++ks) {
            float* dst = e->wbuf + p.off + ((size_t)ks * (p.frag_rows / 32) + p.frag_row0 / 32) * 768;
            if ((rc_put = e->put(dst, sp.data() + ks * piece, piece * sizeof(unsigned short)))) return rc_put;
        }
        // the rows' scales and reciprocals into the trailer of the whole (batched) image
        const float* tr = reinterpret_cast<const float*>(sp.data() + (size_t)rows * K * 3);
        float* trd = e->wbuf + p.off + (size_t)p.frag_rows * K * 3 / 2;
        if ((rc_put = e->put(trd + p.frag_row0, tr, rows * sizeof(float)))) return rc_put;
        if ((rc_put = e->put(trd + p.frag_rows + p.frag_row0, tr + rows, rows * sizeof(float)))) return rc_put;
        p.loaded = true;
        return 0;
    }
    if (p.kind == PK_LINF) {
        // rows [row0, row0+rows) of a [frag_rows][K] matrix -> K/32 contiguous pieces of its fragment-major image
        const int rows = (int)p.shape[0], K = (int)p.shape[1];
        tmp.resize(p.numel);
        pack_linear_frag(host, tmp.data(), rows, K, rows, 0);
        const size_t piece = (size_t)(rows / 32) * 1024;
        for (int ch = 0; ch < K / 32; ++ch) {
            float* dst = e->wbuf + p.off + ((size_t)ch * (p.frag_rows / 32) + p.frag_row0 / 32) * 1024;
            if ((rc_put = e->put(dst, tmp.data() + ch * piece, piece * sizeof(float)))) return rc_put;
        }
        p.loaded = true;
        return 0;
    }
    if (p.kind == PK_CONV3W && split_conv()) {
        tmp.resize(p.packed);
        pack_conv3_wino_split(host, reinterpret_cast<unsigned short*>(tmp.data()), (int)p.shape[0], (int)p.shape[1]);
        src = tmp.data();
    } else if (p.kind == PK_CONV3WU) {
        tmp.resize(p.packed);
        pack_conv3_wino_ups(host, reinterpret_cast<unsigned short*>(tmp.data()), (int)p.shape[0], (int)p.shape[1]);
        src = tmp.data();
    } else if (p.kind == PK_CONV3W) {
        tmp.resize(p.packed);
        pack_conv3_wino(host, tmp.data(), (int)p.shape[0], (int)p.shape[1]);
        src = tmp.data();
    } else if (p.kind == PK_CONV3S) {
        tmp.resize(p.packed);
        pack_conv3_split(host, reinterpret_cast<unsigned short*>(tmp.data()), (int)p.shape[0], (int)p.shape[1]);
        src = tmp.data();
    } else if (p.kind == PK_STEM) {                    // OIHW -> [O][k = tap*I + i] (zero padded) -> fragment-major linear
        const int O = (int)p.shape[0], I = (int)p.shape[1];
        std::vector<float> lin((size_t)O * STEM_KPAD, 0.f);
        for (int o = 0; o < O; ++o)
            for (int i = 0; i < I; ++i)
                for (int t = 0; t < 9; ++t) lin[(size_t)o * STEM_KPAD + t * I + i] = host[((size_t)o * I + i) * 9 + t];
        if (split_math()) {
            tmp.resize(p.packed);
            pack_linear_split(lin.data(), reinterpret_cast<unsigned short*>(tmp.data()), O, STEM_KPAD, O, 0);
        } else {
            tmp.resize((size_t)O * STEM_KPAD);
            pack_linear_frag(lin.data(), tmp.data(), O, STEM_KPAD, O, 0);
        }
        src = tmp.data();
    } else if (p.kind == PK_CONV3 || p.kind == PK_OUTCONV) {
        const int O = (int)p.shape[0], I = (int)p.shape[1];
        tmp.assign(p.packed, 0.f);
        for (int o = 0; o < O; ++o)
            for (int i = 0; i < I; ++i)
                for (int t = 0; t < 9; ++t) tmp[((size_t)t * O + o) * I + i] = host[((size_t)o * I + i) * 9 + t];
        if (p.kind == PK_OUTCONV)
            for (int o = 0; o < O; ++o)
                for (int i = 0; i < I; ++i)
                    for (int t = 0; t < 9; ++t) tmp[(size_t)9 * O * I + ((size_t)o * 9 + t) * I + i] = host[((size_t)o * I + i) * 9 + t];
        src = tmp.data();
    } else if (p.kind == PK_POSENC) {
        const int C = (int)p.shape[1], HW = (int)(p.shape[2] * p.shape[3]);
        tmp.resize((size_t)C * HW);
        for (int c = 0; c < C; ++c)
            for (int q = 0; q < HW; ++q) tmp[(size_t)q * C + c] = host[(size_t)c * HW + q];
        src = tmp.data();
    }
    if ((rc_put = e->put(e->wbuf + p.off, src, p.packed * sizeof(float)))) return rc_put;
    p.loaded = true;
    return 0;
}

int vd_weights_missing(vd_engine* e) {
    if (!e) return -1;
    int n = 0;
    for (auto& p : e->params) n += p.loaded ? 0 : 1;
    return n;
}

int vd_mark_weights_loaded(vd_engine* e) {
    VD_REQUIRE(e, "null argument");
    for (auto& p : e->params) p.loaded = true;
    return 0;
}

int vd_pos_channels(vd_engine* e) { return e ? e->pos_ch : -1; }
int vd_pos_resolution(vd_engine* e) { return e ? e->pos_res : -1; }

int vd_set_freqs(vd_engine* e, const float* tf, int nt, const float* ff, int nf) {
    VD_REQUIRE(e && tf && nt > 0, "time frequencies");
    if (e->d_freq_time) VD_HIP(hipFree(e->d_freq_time));
    VD_HIP(hipMalloc(reinterpret_cast<void**>(&e->d_freq_time), nt * sizeof(float)));
    VD_HIP(hipMemcpy(e->d_freq_time, tf, nt * sizeof(float), hipMemcpyHostToDevice));
    e->n_freq_time = nt;
    if (ff && nf > 0) {
        if (e->d_freq_frame) VD_HIP(hipFree(e->d_freq_frame));
        VD_HIP(hipMalloc(reinterpret_cast<void**>(&e->d_freq_frame), nf * sizeof(float)));
        VD_HIP(hipMemcpy(e->d_freq_frame, ff, nf * sizeof(float), hipMemcpyHostToDevice));
        e->n_freq_frame = nf;
    }
    return 0;
}

int vd_set_schedule(vd_engine* e, int nts, const float* tab, const int* tmap, float rescale) {
    VD_REQUIRE(e && tab && tmap && nts > 0, "schedule tables");
    // the tables' addresses, num_timesteps and rescale are kernel arguments of every captured window graph
    if (e->win_cur >= 0) e->win_lost = true;
    e->drop_window_graphs();
    if (e->d_tab) VD_HIP(hipFree(e->d_tab));
    if (e->d_tmap) VD_HIP(hipFree(e->d_tmap));
    VD_HIP(hipMalloc(reinterpret_cast<void**>(&e->d_tab), (size_t)NTAB * nts * sizeof(float)));
    VD_HIP(hipMemcpy(e->d_tab, tab, (size_t)NTAB * nts * sizeof(float), hipMemcpyHostToDevice));
    std::vector<float> tm(nts);
    for (int i = 0; i < nts; ++i) tm[i] = (float)tmap[i];
    VD_HIP(hipMalloc(reinterpret_cast<void**>(&e->d_tmap), nts * sizeof(float)));
    VD_HIP(hipMemcpy(e->d_tmap, tm.data(), nts * sizeof(float), hipMemcpyHostToDevice));
    e->num_timesteps = nts;
    e->rescale = rescale;
    if (!e->d_err) {
        VD_HIP(hipMalloc(reinterpret_cast<void**>(&e->d_err), sizeof(int)));
        VD_HIP(hipMemset(e->d_err, 0, sizeof(int)));
    }
    return 0;
}

// ModelMeanType of the bound diffusion (gaussian_diffusion.py:29-36): 0 EPSILON (default), 1 START_X (predict_xstart=True,
// script_util.py:429-431).  Captured window graphs bake it in: they are dropped on a change.
int vd_set_model_mean_type(vd_engine* e, int type) {
    VD_REQUIRE(e && (type == 0 || type == 1), "model mean type: 0 EPSILON, 1 START_X");
    if (type != e->mean_type) { if (e->win_cur >= 0) e->win_lost = true; e->drop_window_graphs(); }
    e->mean_type = type;
    return 0;
}

int vd_device_errors(vd_engine* e, int* flags) {
    VD_REQUIRE(e && flags, "null argument");
    *flags = 0;
    if (!e->d_err) return 0;
    // The steps may have run on any stream -- torch's side streams are non-blocking, and a NULL-stream copy is not ordered behind
    // those: wait for the whole device first, so that the word read is the word every step issued so far has left, and the clear
    // cannot race a step still in flight (read + clear are then two operations on an idle device).
    VD_HIP(hipDeviceSynchronize());
    VD_HIP(hipMemcpy(flags, e->d_err, sizeof(int), hipMemcpyDeviceToHost));
    if (*flags) { VD_HIP(hipMemset(e->d_err, 0, sizeof(int))); VD_HIP(hipDeviceSynchronize()); }
    return 0;
}

int vd_workspace_bytes(vd_engine* e, int B, int T, long long* bytes) {
    VD_REQUIRE(e && bytes && B > 0 && T > 0, "arguments");
    Arena dry; dry.dry = true;
    FwdIn fi{}; fi.B = B; fi.T = T;
    int rc = e->forward(fi, nullptr, dry);
    if (rc) return rc;
    *bytes = (long long)dry.peak;
    return 0;
}

static int check_ready(vd_engine* e, int B, int T) {
    VD_REQUIRE(e, "null engine");
    VD_REQUIRE(B > 0 && T > 0 && T <= 32, "window of 1..32 frames");
    VD_REQUIRE(e->wbuf && !e->wbuf_on_host, "weights not set (the packed image must live in device memory: vd_set_weight_storage)");
    int miss = vd_weights_missing(e);
    if (miss) {
        for (auto& p : e->params) if (!p.loaded) { set_error("missing key in state_dict: " + p.name + " (+" + std::to_string(miss - 1) + " more)"); break; }
        return -1;
    }
    return 0;
}

int vd_unet_forward(vd_engine* e, int B, int T, const float* x, const float* obs_src, const float* obs,
                    const float* lat, const float* km, const long long* fidx, const float* t_model, int obs_mode,
                    float* eps, void* stream) {
    int rc = check_ready(e, B, T);
    if (rc) return rc;
    VD_REQUIRE(x && obs_src && obs && lat && km && fidx && t_model && eps, "null tensor");
    VD_REQUIRE(obs_mode >= 0 && obs_mode <= 2, "observed_frames must be x_0 / x_t / x_t_minus_1");
    if ((rc = e->ensure_ws(B, T))) return rc;
    Arena ar; ar.base = e->ws; ar.cap = e->ws_tail;                   // the activations end where t_model and the eps scratch begin
    FwdIn fi{B, T, x, obs_src, obs, lat, km, t_model, reinterpret_cast<const int64_t*>(fidx), obs_mode, eps};
    return e->forward(fi, static_cast<hipStream_t>(stream), ar);
}

// _WrappedModel.__call__ (respace.py:111-119): t_model = map[t] * rescale.  The reference raises IndexError for an index
// outside the table; here the step stays asynchronous: the network is fed NaN, the posterior kernel writes NaN for that
// batch element (misc.hip) and bit 0 of the engine's sticky error word is set for vd_device_errors().
__global__ void map_t_kernel(const int64_t* t, const float* tmap, float rescale, int B, int nts, float* out, int* err) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) {
        const long long i = t[b];
        if (i < 0 || i >= nts) {
            out[b] = __builtin_nanf("");
            atomicOr(err, 1);
        } else {
            out[b] = tmap[i] * rescale;
        }
    }
}

// One denoise step on `st`: t -> t_model, UNet forward, posterior update.  `t` and (when rng != null) the Philox
// {seed, offset} are read from device memory, so the same launch sequence serves every step of a window (executor).
static int step_launches(vd_engine* e, int mode, int B, int T, const float* x, const float* obs_src, const float* obs,
                         const float* lat, const float* km, const long long* fidx, const long long* t, int obs_mode,
                         int clip, float eta, const float* noise, unsigned long long seed, unsigned long long offset,
                         const unsigned long long* rng, float* sample, float* xstart, float* mean, float* eps_out,
                         hipStream_t st, const PrefixPlan* pp = nullptr, const SuffixPlan* sp = nullptr) {
    int rc;
    // every sampler entry point comes through here (vd_p_sample / vd_ddim_sample / vd_p_mean_variance / the window executor): a
    // 6-channel network output must never reach the 3-channel posterior kernel
    VD_REQUIRE(!e->cfg.learn_sigma, "learn_sigma: the reference's sampler asserts on video tensors (gaussian_diffusion.py:283: model_output.shape == (B, 2*T, ...)); only the network forward is served");
    const size_t per = (size_t)T * 3 * e->cfg.image_size * e->cfg.image_size;
    // tail of the workspace: t_model [B] + eps scratch (sized by ensure_ws for this B)
    float* tm = reinterpret_cast<float*>(e->ws + e->ws_tail);
    float* eps = eps_out ? eps_out : reinterpret_cast<float*>(e->ws + e->ws_tail + (((size_t)B * sizeof(float) + 255) & ~(size_t)255));
    hipLaunchKernelGGL(map_t_kernel, dim3((B + 63) / 64), dim3(64), 0, st, reinterpret_cast<const int64_t*>(t), e->d_tmap,
                       e->rescale, B, e->num_timesteps, tm, e->d_err);
    Arena ar; ar.base = e->ws; ar.cap = e->ws_tail;                   // the activations end where t_model and the eps scratch begin
    FwdIn fi{B, T, x, obs_src, obs, lat, km, tm, reinterpret_cast<const int64_t*>(fidx), obs_mode, eps};
    if ((rc = e->forward(fi, st, ar, pp, sp))) return rc;
    PosteriorArgs pa{x, eps, noise, reinterpret_cast<const int64_t*>(t), e->d_tab, e->num_timesteps, B, (long)per, clip,
                     mode, eta, seed, offset, sample, xstart, mean, rng};
    pa.err = e->d_err;
    if (e->mean_type == 1) pa.x0_given = eps;        // START_X: pred_xstart = process_xstart(model_output) (gaussian_diffusion.py:326-341)
    ProfScope ps(PC_POSTERIOR, 0.0, 4.0 * B * per * 5.0, st);
    return launch_posterior(pa, st);
}

static int sample_impl(vd_engine* e, int mode, int B, int T, const float* x, const float* obs_src, const float* obs,
                       const float* lat, const float* km, const long long* fidx, const long long* t, int obs_mode,
                       int clip, float eta, const float* noise, unsigned long long seed, unsigned long long offset,
                       float* sample, float* xstart, float* eps_out, void* stream) {
    int rc = check_ready(e, B, T);
    if (rc) return rc;
    VD_REQUIRE(e->d_tab, "vd_set_schedule not called");
    VD_REQUIRE(x && obs_src && obs && lat && km && fidx && t && sample, "null tensor");
    VD_REQUIRE(obs_mode >= 0 && obs_mode <= 2, "observed_frames must be x_0 / x_t / x_t_minus_1");
    VD_REQUIRE(mode == 0 || eta >= 0.f, "eta");
    VD_REQUIRE(!e->cfg.learn_sigma, "learn_sigma: the reference's sampler asserts on video tensors (gaussian_diffusion.py:283: model_output.shape == (B, 2*T, ...)); only the network forward is served");
    if ((rc = e->ensure_ws(B, T))) return rc;
    return step_launches(e, mode, B, T, x, obs_src, obs, lat, km, fidx, t, obs_mode, clip, eta, noise, seed, offset, nullptr,
                         sample, xstart, nullptr, eps_out, static_cast<hipStream_t>(stream));
}

int vd_p_mean_variance(vd_engine* e, int B, int T, const float* x, const float* obs_src, const float* obs, const float* lat,
                       const float* km, const long long* fidx, const long long* t, int obs_mode, int clip, float* mean,
                       float* xstart, float* eps, void* stream) {
    int rc = check_ready(e, B, T);
    if (rc) return rc;
    VD_REQUIRE(e->d_tab, "vd_set_schedule not called");
    VD_REQUIRE(x && obs_src && obs && lat && km && fidx && t && (mean || xstart || eps), "null tensor");
    VD_REQUIRE(obs_mode >= 0 && obs_mode <= 2, "observed_frames must be x_0 / x_t / x_t_minus_1");
    if ((rc = e->ensure_ws(B, T))) return rc;
    return step_launches(e, 0, B, T, x, obs_src, obs, lat, km, fidx, t, obs_mode, clip, 0.f, nullptr, 0, 0, nullptr, nullptr,
                         xstart, mean, eps, static_cast<hipStream_t>(stream));
}

static int ensure_part(vd_engine* e, size_t doubles) {
    if (doubles <= e->part_cap) return 0;
    if (e->d_part) VD_HIP(hipFree(e->d_part));
    e->d_part = nullptr; e->part_cap = 0;
    VD_HIP(hipMalloc(reinterpret_cast<void**>(&e->d_part), doubles * sizeof(double)));
    e->part_cap = doubles;
    return 0;
}

int vd_vb_terms(vd_engine* e, int B, int T, const float* x_start, const float* x_t, const float* eps, const float* noise,
                const long long* t, int clip, const float* latent_mask, float* vb, float* xstart_mse, float* mse,
                float* pred_xstart, void* stream) {
    VD_REQUIRE(e && e->d_tab, "vd_set_schedule not called");
    VD_REQUIRE(B > 0 && T > 0 && x_start && x_t && eps && t && vb, "arguments");
    VD_REQUIRE(mse == nullptr || noise != nullptr, "mse needs the noise x_t was drawn with");
    const long per = (long)T * 3 * e->cfg.image_size * e->cfg.image_size;
    const int nblk = vb_terms_blocks(per);
    int rc = ensure_part(e, (size_t)B * nblk * 3);
    if (rc) return rc;
    VbArgs a{x_start, x_t, eps, noise, reinterpret_cast<const int64_t*>(t), e->d_tab, e->num_timesteps, latent_mask, B, T, per,
             clip, pred_xstart, e->d_part, nblk, vb, xstart_mse, mse};
    a.err = e->d_err;
    a.start_x = e->mean_type == 1;
    return launch_vb_terms(a, static_cast<hipStream_t>(stream));
}

int vd_prior_bpd(vd_engine* e, int B, int T, const float* x_start, const float* latent_mask, float* out, void* stream) {
    VD_REQUIRE(e && e->d_tab, "vd_set_schedule not called");
    VD_REQUIRE(B > 0 && T > 0 && x_start && out, "arguments");
    const long per = (long)T * 3 * e->cfg.image_size * e->cfg.image_size;
    const int nblk = vb_terms_blocks(per);
    int rc = ensure_part(e, (size_t)B * nblk * 3);
    if (rc) return rc;
    return launch_prior_bpd(x_start, latent_mask, e->d_tab, e->num_timesteps, B, T, per, e->d_part, nblk, out,
                            static_cast<hipStream_t>(stream));
}

// ------------------------------------------------------------------------------------------ window executor
// scripts/video_sample.py:149-168 runs `for timestep in reversed(range(num_timesteps)): local = p_sample(...)` from the
// host, ~330 launches per step.  Here the window's step index and Philox counter live in device memory, a step is ONE
// captured hipGraph (map_t -> UNet forward -> in-place posterior update -> advance the counters), cached per window
// signature, and a window is num_timesteps replays of it (BASELINE configs[4]).
__global__ void win_set_kernel(long long* t, unsigned long long* rng, int B, long long t_start, unsigned long long seed,
                               unsigned long long offset) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) t[b] = t_start;
    if (b == 0) { rng[0] = seed; rng[1] = offset; }
}

__global__ void win_advance_kernel(long long* t, unsigned long long* rng, int B, unsigned long long draws) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) t[b] -= 1;
    if (b == 0) rng[1] += draws;
}

int vd_window_begin(vd_engine* e, int B, int T, float* x, const float* obs_src, const float* obs, const float* lat,
                    const float* km, const long long* fidx, int obs_mode, int sampler, int clip, float eta,
                    unsigned long long seed, unsigned long long offset, long long t_start, void* stream) {
    int rc = check_ready(e, B, T);
    if (rc) return rc;
    VD_REQUIRE(e->d_tab, "vd_set_schedule not called");
    VD_REQUIRE(x && obs_src && obs && lat && km && fidx, "null tensor");
    VD_REQUIRE(obs_mode >= 0 && obs_mode <= 3, "observed_frames must be x_0 / x_t / x_t_minus_1 (2: re-noised per step, 3: the caller's tensor as it is)");
    // 3 = 'x_t_minus_1' with the caller's tensor read as it is at every step -- what scripts/video_sample.py:149-166 does (it hands x0 as a
    // clean placeholder); 2 = p_sample_loop's form, which re-noises the clean frames to t - 1 before each step (gaussian_diffusion.py:565-568)
    const int net_mode = obs_mode == 3 ? 2 : obs_mode;
    VD_REQUIRE(sampler == 0 || sampler == 1, "sampler: 0 p_sample, 1 ddim_sample");
    VD_REQUIRE(t_start >= 0 && t_start < e->num_timesteps, "t_start outside the schedule");
    hipStream_t st = static_cast<hipStream_t>(stream);
    VD_REQUIRE(st != nullptr, "the window executor captures a hipGraph: it needs a non-default stream");
    if ((rc = e->ensure_ws(B, T))) return rc;
    if (B > e->win_t_cap) {
        e->drop_window_graphs();                                // captured graphs hold the old counter address
        if (e->d_win_t) VD_HIP(hipFree(e->d_win_t));
        e->d_win_t = nullptr; e->win_t_cap = 0;
        VD_HIP(hipMalloc(reinterpret_cast<void**>(&e->d_win_t), (size_t)B * sizeof(long long)));
        e->win_t_cap = B;
    }
    if (!e->d_win_rng) VD_HIP(hipMalloc(reinterpret_cast<void**>(&e->d_win_rng), 2 * sizeof(unsigned long long)));
    const size_t per_w = (size_t)T * 3 * e->cfg.image_size * e->cfg.image_size;
    if (obs_mode == 2 && (size_t)B * per_w > e->win_xtm1_cap) {
        if (e->d_win_xtm1) {                                    // captured graphs hold the old buffer's address
            e->drop_window_graphs();
            VD_HIP(hipFree(e->d_win_xtm1));
        }
        e->d_win_xtm1 = nullptr; e->win_xtm1_cap = 0;
        VD_HIP(hipMalloc(reinterpret_cast<void**>(&e->d_win_xtm1), (size_t)B * per_w * sizeof(float)));
        e->win_xtm1_cap = (size_t)B * per_w;
    }
    // 'x_t_minus_1' (gaussian_diffusion.py:565-568): obs_src holds the CLEAN observed frames x0; every step first draws
    // x_t_minus_1 = q_sample(x0, t - 1) into the engine's buffer (the second half of the step's Philox range), the network reads that
    const float* net_obs_src = obs_mode == 2 ? e->d_win_xtm1 : obs_src;
    hipLaunchKernelGGL(win_set_kernel, dim3((B + 63) / 64), dim3(64), 0, st, e->d_win_t, e->d_win_rng, B, t_start, seed, offset);
    VD_HIP(hipGetLastError());
    vd_engine::WinKey key;
    std::memset(&key, 0, sizeof(key));
    key.B = B; key.T = T; key.obs_mode = obs_mode; key.sampler = sampler; key.clip = clip; key.eta = eta;
    key.x = x; key.obs_src = obs_src; key.obs = obs; key.lat = lat; key.km = km; key.fidx = fidx;
    e->win_cur = -1;
    e->win_lost = false;
    ++e->win_gen;
    e->win_left = t_start + 1;
    // window prefix cache (opt-in): the frames whose network input cannot change during the window -- observed (obs = 1,
    // lat = 0) in 'x_0' mode with the default 'channel' conditioning (assemble_kernel: v = obs_src, indicator channels,
    // timestep 0) -- run the blocks before the first attention layer ONCE, now; the captured step runs them on the others
    const int N = B * T;
    std::vector<unsigned char> inv;
    int n_inv = 0;
    // window suffix skip (opt-in): the same frames -- pure observations -- are not read back by the caller and re-enter the
    // network as the observation ('x_0', 'x_t_minus_1'; NOT 'x_t', where an observed frame's input is its own running sample):
    // everything behind the last attention layer runs without them
    const bool use_pre = e->prefix_cache_on && obs_mode == 0 && e->cfg.cond_emb_type == 0 && e->n_before_attn > 0;
    const bool want_suf = e->suffix_skip_on && obs_mode != 1 && e->cfg.cond_emb_type == 0 && e->suf_blk >= -1 && !e->attn.empty();
    if (use_pre || want_suf) {
        std::vector<float> hm(2 * (size_t)N);
        VD_HIP(hipMemcpyAsync(hm.data(), obs, N * sizeof(float), hipMemcpyDeviceToHost, st));
        VD_HIP(hipMemcpyAsync(hm.data() + N, lat, N * sizeof(float), hipMemcpyDeviceToHost, st));
        VD_HIP(hipStreamSynchronize(st));
        inv.resize(N);
        for (int n = 0; n < N; ++n) { inv[n] = hm[n] == 1.f && hm[N + n] == 0.f; n_inv += inv[n]; }
        if (n_inv == 0) inv.clear();
    }
    const bool pre_on = use_pre && n_inv > 0, suf_on = want_suf && n_inv > 0 && n_inv < N;
    key.flags = (pre_on ? 1 : 0) | (suf_on ? 2 : 0);
    auto build_pass = [&](vd_engine::WinGraph& g) -> int {               // the invariant frames' prefix, eagerly, into the store
        float* tm = reinterpret_cast<float*>(e->ws + e->ws_tail);
        float* eps = reinterpret_cast<float*>(e->ws + e->ws_tail + (((size_t)B * sizeof(float) + 255) & ~(size_t)255));
        hipLaunchKernelGGL(map_t_kernel, dim3((B + 63) / 64), dim3(64), 0, st, reinterpret_cast<const int64_t*>(e->d_win_t),
                           e->d_tmap, e->rescale, B, e->num_timesteps, tm, e->d_err);
        Arena ar; ar.base = e->ws; ar.cap = e->ws_tail;                   // the activations end where t_model and the eps scratch begin
        FwdIn fi{B, T, x, obs_src, obs, lat, km, tm, reinterpret_cast<const int64_t*>(fidx), obs_mode, eps};
        PrefixPlan bp; bp.n = g.n_inv; bp.list = g.d_lists + (N - g.n_inv); bp.build_only = true; bp.store = g.store;
        const bool prof = g_prof.on;
        g_prof.on = false;
        const int brc = e->forward(fi, st, ar, &bp);
        g_prof.on = prof;
        return brc;
    };
    for (size_t i = 0; i < e->win_graphs.size(); ++i)
        if (e->win_graphs[i].key == key && e->win_graphs[i].inv == inv) {
            e->win_cur = (int)i;
            return pre_on ? build_pass(e->win_graphs[i]) : 0;          // same buffers, possibly new contents: the cache is per window
        }
    // A new signature.  Everything a launch needs lazily (kernel attributes, CU count, the workspace) is set up by one
    // eager forward into the eps scratch before the capture; x is not touched by it.
    {
        float* tm = reinterpret_cast<float*>(e->ws + e->ws_tail);
        float* eps = reinterpret_cast<float*>(e->ws + e->ws_tail + (((size_t)B * sizeof(float) + 255) & ~(size_t)255));
        hipLaunchKernelGGL(map_t_kernel, dim3((B + 63) / 64), dim3(64), 0, st, reinterpret_cast<const int64_t*>(e->d_win_t),
                           e->d_tmap, e->rescale, B, e->num_timesteps, tm, e->d_err);
        Arena ar; ar.base = e->ws; ar.cap = e->ws_tail;                   // the activations end where t_model and the eps scratch begin
        if (obs_mode == 2 && (rc = launch_q_sample_prev(obs_src, reinterpret_cast<const int64_t*>(e->d_win_t), e->d_tab, e->num_timesteps, B, (long)per_w,
                                                        e->d_win_rng, (unsigned long long)B * per_w / 2, e->d_win_xtm1, st))) return rc;
        FwdIn fi{B, T, x, net_obs_src, obs, lat, km, tm, reinterpret_cast<const int64_t*>(fidx), net_mode, eps};
        if ((rc = e->forward(fi, st, ar))) return rc;
        VD_HIP(hipStreamSynchronize(st));
    }
    const bool prof = g_prof.on;
    g_prof.on = false;                                          // no event records inside a capture
    const size_t per = (size_t)T * 3 * e->cfg.image_size * e->cfg.image_size;
    vd_engine::WinGraph wg{};
    wg.key = key; wg.inv = inv; wg.n_inv = n_inv;
    PrefixPlan plan;
    SuffixPlan splan;
    if (pre_on || suf_on) {
        std::vector<int> lists;
        for (int n = 0; n < N; ++n) if (!inv[n]) lists.push_back(n);
        for (int n = 0; n < N; ++n) if (inv[n]) lists.push_back(n);
        VD_HIP(hipMalloc(reinterpret_cast<void**>(&wg.d_lists), N * sizeof(int)));
        VD_HIP(hipMemcpy(wg.d_lists, lists.data(), N * sizeof(int), hipMemcpyHostToDevice));
        wg.n_suf = suf_on ? N - n_inv : 0;
        splan.n = N - n_inv; splan.list = wg.d_lists;
    }
    if (pre_on) {
        wg.store = new PrefixStore();
        rc = build_pass(wg);
        if (!rc && hipStreamSynchronize(st) != hipSuccess) { set_error("prefix cache: first pass"); rc = -2; }
        if (rc) { g_prof.on = prof; wg.graph = nullptr; wg.exec = nullptr; if (wg.d_lists) (void)hipFree(wg.d_lists);
                  for (float* p : wg.store->tens) if (p) (void)hipFree(p); for (double* p : wg.store->parts) if (p) (void)hipFree(p); delete wg.store; return rc; }
        plan.n = N - n_inv; plan.list = wg.d_lists; plan.store = wg.store;
    }
    VD_HIP(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    rc = obs_mode == 2 ? launch_q_sample_prev(obs_src, reinterpret_cast<const int64_t*>(e->d_win_t), e->d_tab, e->num_timesteps, B, (long)per,
                                              e->d_win_rng, (unsigned long long)B * per / 2, e->d_win_xtm1, st) : 0;
    if (!rc) rc = step_launches(e, sampler, B, T, x, net_obs_src, obs, lat, km, fidx, e->d_win_t, net_mode, clip, eta, nullptr, 0, 0,
                                e->d_win_rng, x, nullptr, nullptr, nullptr, st, pre_on ? &plan : nullptr, suf_on ? &splan : nullptr);
    if (!rc) {
        hipLaunchKernelGGL(win_advance_kernel, dim3((B + 63) / 64), dim3(64), 0, st, e->d_win_t, e->d_win_rng, B,
                           (unsigned long long)B * per);
        if (hipGetLastError() != hipSuccess) { set_error("win_advance_kernel launch"); rc = -2; }
    }
    hipGraph_t graph = nullptr;
    const hipError_t ce = hipStreamEndCapture(st, &graph);
    g_prof.on = prof;
    auto drop_store = [&]() {
        if (wg.d_lists) (void)hipFree(wg.d_lists);
        if (wg.store) { for (float* p : wg.store->tens) if (p) (void)hipFree(p); for (double* p : wg.store->parts) if (p) (void)hipFree(p); delete wg.store; }
    };
    if (rc) { if (graph) (void)hipGraphDestroy(graph); drop_store(); return rc; }
    if (ce != hipSuccess) { set_error(std::string("hipStreamEndCapture: ") + hipGetErrorString(ce)); drop_store(); return -2; }
    hipGraphExec_t exec = nullptr;
    if (hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) { set_error("hipGraphInstantiate"); (void)hipGraphDestroy(graph); drop_store(); return -2; }
    wg.graph = graph; wg.exec = exec;
    if (wg.store) {
        // a prefix store holds full-size copies of every tensor before the first attention layer (over 1 GB per block at 128x128,
        // N = 160): sampling modes whose observed-frame pattern changes from window to window (hierarchy, adaptive, mixed) would
        // grow device memory without bound.  At most four graphs keep a store; the oldest one goes first.
        int cnt = 0, oldest = -1;
        for (size_t i = 0; i < e->win_graphs.size(); ++i) if (e->win_graphs[i].store) { ++cnt; if (oldest < 0) oldest = (int)i; }
        if (cnt >= 4) {
            (void)hipDeviceSynchronize();                       // its graph may still be queued on the executor's stream
            vd_engine::free_graph(e->win_graphs[oldest]);
            e->win_graphs.erase(e->win_graphs.begin() + oldest);
        }
    }
    e->win_graphs.push_back(wg);
    e->win_cur = (int)e->win_graphs.size() - 1;
    return 0;
}

int vd_set_window_prefix_cache(vd_engine* e, int on) {
    VD_REQUIRE(e, "null engine");
    e->prefix_cache_on = on != 0;
    return 0;
}

int vd_window_prefix_frames(vd_engine* e) {
    VD_REQUIRE(e, "null engine");
    return e->win_cur >= 0 && e->win_graphs[e->win_cur].store ? e->win_graphs[e->win_cur].n_inv : 0;
}

int vd_set_window_suffix_skip(vd_engine* e, int on) {
    VD_REQUIRE(e, "null engine");
    e->suffix_skip_on = on != 0;
    return 0;
}

int vd_window_suffix_frames(vd_engine* e) {
    VD_REQUIRE(e, "null engine");
    return e->win_cur >= 0 ? e->win_graphs[e->win_cur].n_suf : 0;
}

unsigned long long vd_window_generation(vd_engine* e) { return e ? e->win_gen : 0; }

int vd_window_run(vd_engine* e, int n_steps, void* stream) {
    VD_REQUIRE(e, "null engine");
    VD_REQUIRE(!e->win_lost, "window graphs invalidated (workspace growth or a new schedule since vd_window_begin): begin the window again");
    VD_REQUIRE(e->win_cur >= 0, "vd_window_begin first");
    VD_REQUIRE(n_steps >= 0 && n_steps <= e->win_left, "more steps than the window has left (t would pass 0)");
    hipStream_t st = static_cast<hipStream_t>(stream);
    for (int i = 0; i < n_steps; ++i) VD_HIP(hipGraphLaunch(e->win_graphs[e->win_cur].exec, st));
    e->win_left -= n_steps;
    return 0;
}

int vd_window_graphs(vd_engine* e) { return e ? (int)e->win_graphs.size() : -1; }

int vd_p_sample(vd_engine* e, int B, int T, const float* x, const float* obs_src, const float* obs, const float* lat,
                const float* km, const long long* fidx, const long long* t, int obs_mode, int clip, const float* noise,
                unsigned long long seed, unsigned long long offset, float* sample, float* xstart, float* eps, void* stream) {
    return sample_impl(e, 0, B, T, x, obs_src, obs, lat, km, fidx, t, obs_mode, clip, 0.f, noise, seed, offset, sample,
                       xstart, eps, stream);
}

int vd_ddim_sample(vd_engine* e, int B, int T, const float* x, const float* obs_src, const float* obs, const float* lat,
                   const float* km, const long long* fidx, const long long* t, int obs_mode, int clip, float eta,
                   const float* noise, unsigned long long seed, unsigned long long offset, float* sample, float* xstart,
                   float* eps, void* stream) {
    return sample_impl(e, 1, B, T, x, obs_src, obs, lat, km, fidx, t, obs_mode, clip, eta, noise, seed, offset, sample,
                       xstart, eps, stream);
}

int vd_posterior_update(vd_engine* e, int mode, int B, long long per, const float* x, const float* eps,
                        const long long* t, int clip, float eta, const float* noise, unsigned long long seed,
                        unsigned long long offset, float* sample, float* xstart, void* stream) {
    VD_REQUIRE(e && e->d_tab, "vd_set_schedule not called");
    VD_REQUIRE(x && eps && t && sample && (mode == 0 || mode == 1), "arguments");
    PosteriorArgs pa{x, eps, noise, reinterpret_cast<const int64_t*>(t), e->d_tab, e->num_timesteps, B, (long)per, clip,
                     mode, eta, seed, offset, sample, xstart, nullptr, nullptr};
    pa.err = e->d_err;
    return launch_posterior(pa, static_cast<hipStream_t>(stream));
}

int vd_posterior_from_xstart(vd_engine* e, int mode, int B, long long per, const float* x, const float* xstart_in,
                             const long long* t, int clip, float eta, const float* noise, unsigned long long seed,
                             unsigned long long offset, float* sample, float* xstart, float* mean, void* stream) {
    VD_REQUIRE(e && e->d_tab, "vd_set_schedule not called");
    VD_REQUIRE(x && xstart_in && t && (sample || xstart || mean) && (mode == 0 || mode == 1), "arguments");
    PosteriorArgs pa{x, nullptr, noise, reinterpret_cast<const int64_t*>(t), e->d_tab, e->num_timesteps, B, (long)per, clip,
                     mode, eta, seed, offset, sample, xstart, mean, nullptr};
    pa.x0_given = xstart_in;
    pa.err = e->d_err;
    return launch_posterior(pa, static_cast<hipStream_t>(stream));
}


// ---- return_attn_weights (unet.py:457-466, 799-836) ---------------------------------------------------------------------
int vd_attn_blocks(vd_engine* e) { return e ? (int)e->attn.size() : -1; }

// resolution (pixels per side) and channels of attention block i, in execution order (input blocks, middle, output blocks)
int vd_attn_block_info(vd_engine* e, int i, int* res, int* channels) {
    VD_REQUIRE(e && i >= 0 && i < (int)e->attn.size() && res && channels, "attention block index");
    int k = 0;
    auto scan = [&](const std::vector<Layer>& blk, int r) { for (const Layer& L : blk) if (L.type == 2 && k++ == i) { *res = r; *channels = e->attn[L.idx].C; return true; } return false; };
    int r = e->cfg.image_size;
    for (auto& blk : e->input_blocks) { if (scan(blk, r)) return 0; for (const Layer& L : blk) if (L.type == 3) r /= 2; }
    if (scan(e->middle, r)) return 0;
    for (auto& blk : e->output_blocks) { if (scan(blk, r)) return 0; for (const Layer& L : blk) if (L.type == 4) r *= 2; }
    set_error("attention block not found"); return -1;
}

// Arm (n > 0) or clear (n = 0) the capture: the next forwards write block i's head-averaged softmax weights to
// temporal[i] ([B*HW][T][T]) and spatial[i] ([B*T][HW][HW]); a NULL entry skips that block.
int vd_set_attn_capture(vd_engine* e, float* const* temporal, float* const* spatial, int n) {
    VD_REQUIRE(e && n >= 0 && n <= (int)e->attn.size() && (n == 0 || (temporal && spatial)), "attention capture");
    e->attn_cap_t.assign(temporal, temporal + n);
    e->attn_cap_s.assign(spatial, spatial + n);
    return 0;
}

// ---- use_gradient_method (gaussian_diffusion.py:264-271,350-364) ------------------------------------------------------
long long vd_bwd_weights_bytes(vd_engine* e) { return e ? (long long)(e->packed_bwd_total * sizeof(float)) : -1; }

int vd_set_bwd_weight_storage(vd_engine* e, void* buf, long long bytes, int on_host) {
    VD_REQUIRE(e && buf, "null argument");
    VD_REQUIRE(split_conv(), "use_gradient_method runs on the split arithmetic only (VD_MATH=f16x3 | bf16x6)");
    VD_REQUIRE(bytes >= (long long)(e->packed_bwd_total * sizeof(float)), "backward weight buffer too small");
    e->wbuf_bwd = static_cast<float*>(buf);
    e->wbuf_bwd_on_host = on_host != 0;
    return 0;
}

// the backward-data image of one parameter (no-op for parameters the backward pass does not read)
int vd_load_weight_bwd(vd_engine* e, const char* name, const float* host, long long numel) {
    VD_REQUIRE(e && name && host, "null argument");
    VD_REQUIRE(e->wbuf_bwd, "vd_set_bwd_weight_storage first");
    auto it = e->pidx.find(name);
    if (it == e->pidx.end()) { set_error(std::string("unexpected key in state_dict: ") + name); return -1; }
    const Param& p = e->params[it->second];
    if (p.kind_bwd < 0) return 0;
    VD_REQUIRE((long long)p.numel == numel, "size mismatch");
    std::vector<float> tmp((size_t)p.packed_bwd, 0.f);
    const int O = (int)p.shape[0], I = (int)p.shape[1];
    if (p.kind == PK_STEM) {                           // forward lin[o][k = tap*I + i]; backward matrix [k][o]
        std::vector<float> lt((size_t)STEM_KPAD * O, 0.f);
        for (int o = 0; o < O; ++o)
            for (int i = 0; i < I; ++i)
                for (int t = 0; t < 9; ++t) lt[(size_t)(t * I + i) * O + o] = host[((size_t)o * I + i) * 9 + t];
        pack_linear_split(lt.data(), reinterpret_cast<unsigned short*>(tmp.data()), STEM_KPAD, O, STEM_KPAD, 0);
    } else if (p.kind_bwd == PK_LINF) {                // W[O][I] -> W^T[I][O]
        std::vector<float> wt((size_t)O * I);
        for (int o = 0; o < O; ++o)
            for (int i = 0; i < I; ++i) wt[(size_t)i * O + o] = host[(size_t)o * I + i];
        pack_linear_split(wt.data(), reinterpret_cast<unsigned short*>(tmp.data()), I, O, I, 0);
    } else {                                           // w'[i][o][ky][kx] = w[o][i][2-ky][2-kx]
        std::vector<float> wr((size_t)O * I * 9);
        for (int o = 0; o < O; ++o)
            for (int i = 0; i < I; ++i)
                for (int t = 0; t < 9; ++t) wr[((size_t)i * O + o) * 9 + t] = host[((size_t)o * I + i) * 9 + (8 - t)];
        if (p.kind_bwd == PK_CONV3W) {
            pack_conv3_wino_split(wr.data(), reinterpret_cast<unsigned short*>(tmp.data()), I, O);
        } else {                                       // generic kernel: [tap][Cout' = I][Cin' = O]
            for (int i = 0; i < I; ++i)
                for (int o = 0; o < O; ++o)
                    for (int t = 0; t < 9; ++t) tmp[((size_t)t * I + i) * O + o] = wr[((size_t)i * O + o) * 9 + t];
        }
    }
    float* dst = e->wbuf_bwd + p.off_bwd;
    if (e->wbuf_bwd_on_host) std::memcpy(dst, tmp.data(), p.packed_bwd * sizeof(float));
    else VD_HIP(hipMemcpy(dst, tmp.data(), p.packed_bwd * sizeof(float), hipMemcpyHostToDevice));
    return 0;
}

__global__ void guided_masks_kernel(const float* obs, const float* lat, int n, float* obs_net, float* lat_net) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { obs_net[i] = 0.f; lat_net[i] = obs[i] + lat[i]; }      // gaussian_diffusion.py:268-271
}

// One guided step.  The network sees every frame as latent (obs_mask := 0, latent_mask := obs + latent); a sample of
// x_{t-1} is drawn with `noise` from the unguided posterior, its squared distance to x_t_minus_1 on the observed frames is
// back-propagated to x, and mean' = mean - 10 * alpha_t * grad / 2.  Outputs (any may be NULL): mean' , pred_xstart, grad,
// and sample = mean' + [t != 0] * sigma_t * noise2 (p_sample's own draw, gaussian_diffusion.py:438-443).
int vd_guided_step(vd_engine* e, int B, int T, const float* x, const float* obs, const float* lat, const float* km,
                   const long long* fidx, const long long* t, int clip, const float* x_t_minus_1, const float* noise,
                   const float* noise2, float* mean, float* xstart, float* grad, float* sample, void* stream) {
    int rc = check_ready(e, B, T);
    if (rc) return rc;
    VD_REQUIRE(e->d_tab, "vd_set_schedule not called");
    VD_REQUIRE(!e->cfg.learn_sigma, "learn_sigma: the reference's sampler asserts on video tensors (gaussian_diffusion.py:283)");
    VD_REQUIRE(e->mean_type == 0, "use_gradient_method: epsilon-prediction models only");
    VD_REQUIRE(e->wbuf_bwd && !e->wbuf_bwd_on_host, "use_gradient_method: the backward-data weight image is not on the device (vd_set_bwd_weight_storage / vd_load_weight_bwd)");
    VD_REQUIRE(x && obs && lat && km && fidx && t && x_t_minus_1 && noise && (sample == nullptr || noise2 != nullptr), "null tensor");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if ((rc = e->ensure_ws_guided(B, T))) return rc;
    const int N = B * T;
    const size_t per = (size_t)T * 3 * e->cfg.image_size * e->cfg.image_size, tot = (size_t)B * per;
    // tail of the workspace: t_model, the two masks, then eps, deps, dxd, mean0, dx_net, xstart0
    char* tail = e->ws + e->ws_cap - (4096 + 8 * tot * sizeof(float));
    float* tm = reinterpret_cast<float*>(tail);
    float* obs_net = tm + 256; float* lat_net = obs_net + 256;
    VD_REQUIRE(N <= 256 && B <= 256, "window of at most 256 frames");
    float* buf = reinterpret_cast<float*>(tail + 4096);
    float *eps = buf, *deps = buf + tot, *dxd = buf + 2 * tot, *mean0 = buf + 3 * tot, *dxn = buf + 4 * tot, *xs0 = buf + 5 * tot;
    hipLaunchKernelGGL(map_t_kernel, dim3((B + 63) / 64), dim3(64), 0, st, reinterpret_cast<const int64_t*>(t), e->d_tmap,
                       e->rescale, B, e->num_timesteps, tm, e->d_err);
    hipLaunchKernelGGL(guided_masks_kernel, dim3((N + 63) / 64), dim3(64), 0, st, obs, lat, N, obs_net, lat_net);
    Arena ar; ar.base = e->ws; ar.cap = (size_t)(tail - e->ws);       // forward + backward live below the step's own buffers
    Tape tp; e->tape = &tp;
    FwdIn fi{B, T, x, x, obs_net, lat_net, km, tm, reinterpret_cast<const int64_t*>(fidx), 0, eps};
    rc = e->forward(fi, st, ar);
    if (!rc) {
        GuidedArgs ga{x, eps, noise, x_t_minus_1, obs, reinterpret_cast<const int64_t*>(t), e->d_tab, e->num_timesteps, B, T, (long)per, clip,
                      deps, dxd, mean0, xstart ? xstart : xs0};
        ga.err = e->d_err;
        rc = launch_guided_grad(ga, st);
        static const bool no_gs = getenv("VD_NO_GRAD_SCALE") != nullptr;      // (A/B switch: the un-scaled backward pass of rounds 3-4)
        float* gs = lat_net + 256;                                            // 4 floats of the tail's spare kilobyte
        if (!rc && !no_gs) { rc = launch_grad_rescale(deps, tot, gs, st); ga.gscale = gs; }
        if (!rc) rc = e->backward(fi, deps, dxn, st, ar);
        if (!rc) rc = launch_guided_final(ga, dxn, noise2, grad, mean, sample, st);
    }
    e->tape = nullptr;
    return rc;
}

int vd_q_sample(vd_engine* e, int B, long long per, const float* x0, const long long* t, const float* noise, float* out,
                void* stream) {
    VD_REQUIRE(e && e->d_tab, "vd_set_schedule not called");
    VD_REQUIRE(x0 && t && noise && out, "null tensor");
    return launch_q_sample(x0, noise, reinterpret_cast<const int64_t*>(t), e->d_tab, e->num_timesteps, B, (long)per, out,
                           static_cast<hipStream_t>(stream));
}

int vd_randn(float* out, long long n, unsigned long long seed, unsigned long long offset, void* stream) {
    VD_REQUIRE(out && n >= 0, "arguments");
    return launch_randn(out, (long)n, seed, offset, static_cast<hipStream_t>(stream));
}

long long vd_split_image_u16(long long n_out, long long k_total) { return (long long)split_image_u16((size_t)n_out, (size_t)k_total); }
int vd_math_mode(void) { return math_mode(); }

int vd_pack_conv3_wino_split(const float* host_oihw, unsigned short* host_out, int O, int I) {
    VD_REQUIRE(host_oihw && host_out && O % 64 == 0 && I % 32 == 0, "vd_pack_conv3_wino_split: O multiple of 64, I of 32");
    pack_conv3_wino_split(host_oihw, host_out, O, I);
    return 0;
}

int vd_op_conv_wino_split(const float* src0, int Cin, int nfr, int Hs, int Ws, int ups, const void* w_split, const float* bias,
                          const float* res, const float* fbias, int fbias_ld, float* out, int Cout, double* gn_part,
                          void* stream) {
    IgemmArgs g{};
    g.src0 = src0; g.C0 = Cin; g.Cin = Cin; g.nfr = nfr; g.Hs = Hs; g.Ws = Ws; g.ups = ups;
    g.stride = 1; g.pad = 1; g.ksz = 3;
    g.Ho = Hs << ups; g.Wo = Ws << ups;
    g.wwino = static_cast<const float*>(w_split); g.wsplit = 2; g.bias = bias; g.res = res; g.res_ld = Cout;
    g.fbias = fbias; g.fbias_ld = fbias_ld; g.out = out; g.ldo = Cout; g.Cout = Cout; g.M = nfr * g.Ho * g.Wo;
    g.stats = gn_part; g.stats_split = conv_wino_stats_split(g.Ho);
    VD_REQUIRE(conv_wino_r64_supported(g), "vd_op_conv_wino_split: shape not covered by conv_wino_r64.hip");
    return launch_igemm(g, static_cast<hipStream_t>(stream));            // cuts big windows along frames like the engine
}

int vd_op_conv_wino_act(const float* src0, const float* src1, int C0, int Cin, int nfr, int Hs, int Ws, const void* w_split, const float* bias,
                        const float* affA, const float* affB, const float* res, float* out, int Cout, double* gn_part, void* stream) {
    IgemmArgs g{};
    VD_REQUIRE(src0 && (src1 ? C0 > 0 && C0 < Cin : C0 == Cin), "vd_op_conv_wino_act: one source of Cin channels, or two of C0 and Cin - C0");
    g.src0 = src0; g.src1 = src1; g.C0 = C0; g.Cin = Cin; g.nfr = nfr; g.Hs = Hs; g.Ws = Ws;
    g.stride = 1; g.pad = 1; g.ksz = 3; g.Ho = Hs; g.Wo = Ws;
    g.wwino = static_cast<const float*>(w_split); g.wsplit = 2; g.bias = bias; g.res = res; g.res_ld = Cout;
    g.affA = affA; g.affB = affB; g.act = 1;
    g.out = out; g.ldo = Cout; g.Cout = Cout; g.M = nfr * g.Ho * g.Wo;
    g.stats = gn_part; g.stats_split = conv_wino_stats_split(g.Ho);
    VD_REQUIRE(conv_wino_z128_act_supported(g), "vd_op_conv_wino_act: shape not covered by conv_wino_z128.hip's activating form (vd_conv_wino_act_ok)");
    return launch_igemm(g, static_cast<hipStream_t>(stream));
}

int vd_conv_wino_act_ok(int nfr, int H, int Cin, int Cout) { return f16_math() && conv_wino_z128_act_shape(nfr, H, Cin, Cout) ? 1 : 0; }

int vd_pack_conv3_wino_ups(const float* host_oihw, unsigned short* host_out, int O, int I) {
    VD_REQUIRE(host_oihw && host_out && O % 64 == 0 && I % 32 == 0, "vd_pack_conv3_wino_ups: O multiple of 64, I of 32");
    pack_conv3_wino_ups(host_oihw, host_out, O, I);
    return 0;
}

int vd_conv_ups_stats_split(int Hs) { return conv_wino_ups_stats_split(Hs); }

int vd_op_conv_wino_ups(const float* src0, int Cin, int nfr, int Hs, const void* w_ups, const float* bias, float* out, int Cout,
                        double* gn_part, void* stream) {
    IgemmArgs g{};
    g.src0 = src0; g.C0 = Cin; g.Cin = Cin; g.nfr = nfr; g.Hs = Hs; g.Ws = Hs; g.ups = 1; g.ups_phase = 1;
    g.stride = 1; g.pad = 1; g.ksz = 3;
    g.Ho = 2 * Hs; g.Wo = 2 * Hs;
    g.wwino = static_cast<const float*>(w_ups); g.wsplit = 2; g.bias = bias; g.out = out; g.ldo = Cout; g.Cout = Cout;
    g.M = nfr * g.Ho * g.Wo;
    g.stats = gn_part; g.stats_split = conv_wino_ups_stats_split(Hs);
    VD_REQUIRE(conv_wino_r64_supported(g), "vd_op_conv_wino_ups: shape not covered (square power-of-two source map >= 8, Cout % 64, Cin % 32)");
    return launch_igemm(g, static_cast<hipStream_t>(stream));            // cuts big windows along frames like the engine
}

int vd_pack_conv3_split(const float* host_oihw, unsigned short* host_out, int O, int I) {
    VD_REQUIRE(host_oihw && host_out && O % 32 == 0 && I % 32 == 0, "vd_pack_conv3_split: O, I multiples of 32");
    pack_conv3_split(host_oihw, host_out, O, I);
    return 0;
}

int vd_op_conv_split(const float* src0, int Cin, int nfr, int Hs, int Ws, int stride, const void* w_split, const float* bias,
                     const float* res, float* out, int Cout, void* stream) {
    IgemmArgs g{};
    g.src0 = src0; g.C0 = Cin; g.Cin = Cin; g.nfr = nfr; g.Hs = Hs; g.Ws = Ws;
    g.stride = stride; g.pad = 1; g.ksz = 3;
    g.Ho = (Hs + 2 - 3) / stride + 1; g.Wo = (Ws + 2 - 3) / stride + 1;
    g.wfrag = static_cast<const float*>(w_split); g.wsplit = 1; g.bias = bias; g.res = res; g.res_ld = Cout;
    g.out = out; g.ldo = Cout; g.Cout = Cout; g.M = nfr * g.Ho * g.Wo;
    VD_REQUIRE(conv_split_supported(g), "vd_op_conv_split: shape not covered by the kernel");
    return launch_igemm(g, static_cast<hipStream_t>(stream));
}

int vd_pack_linear_split(const float* host_w, unsigned short* host_out, int N, int K) {
    VD_REQUIRE(host_w && host_out && N % 32 == 0 && K % 32 == 0, "vd_pack_linear_split: N, K multiples of 32");
    pack_linear_split(host_w, host_out, N, K, N, 0);
    return 0;
}

int vd_op_linear_split(const float* a, int M, int K, const void* w_split, const float* bias, const float* res, int act,
                       float* out, int N, void* stream) {
    IgemmArgs g{};
    g.src0 = a; g.C0 = K; g.Cin = K; g.nfr = M; g.Hs = g.Ws = g.Ho = g.Wo = 1; g.stride = 1; g.ksz = 1;
    g.wfrag = static_cast<const float*>(w_split); g.wsplit = 1; g.bias = bias; g.act = act; g.res = res; g.res_ld = N;
    g.out = out; g.ldo = N; g.Cout = N; g.M = M;
    return launch_igemm(g, static_cast<hipStream_t>(stream));
}

int vd_op_linear_split_stats(const float* a, int M, int K, const void* w_split, const float* bias, const float* res, int act,
                             float* out, int N, int HW, double* gn_part, void* stream) {
    IgemmArgs g{};
    g.src0 = a; g.C0 = K; g.Cin = K; g.nfr = M; g.Hs = g.Ws = g.Ho = g.Wo = 1; g.stride = 1; g.ksz = 1;
    g.wfrag = static_cast<const float*>(w_split); g.wsplit = 1; g.bias = bias; g.act = act; g.res = res; g.res_ld = N;
    g.out = out; g.ldo = N; g.Cout = N; g.M = M;
    VD_REQUIRE(gn_part && HW > 0 && gemm_split_supported(g), "vd_op_linear_split_stats: shape not covered by gemm_split.hip");
    g.stats = gn_part; g.stats_hw = HW; g.stats_split = HW / gemm_split_stats_rows(M, N);
    return launch_igemm(g, static_cast<hipStream_t>(stream));
}

int vd_linear_stats_split(int M, int N, int HW) { return HW / gemm_split_stats_rows(M, N); }

int vd_pack_conv3_wino(const float* host_oihw, float* host_out, int O, int I) {
    VD_REQUIRE(host_oihw && host_out && O % 64 == 0 && I % 16 == 0, "O multiple of 64, I multiple of 16");
    pack_conv3_wino(host_oihw, host_out, O, I);
    return 0;
}

int vd_pack_linear_frag(const float* host_w, float* host_out, int N, int K) {
    VD_REQUIRE(host_w && host_out && N % 32 == 0 && K % 32 == 0, "N and K multiples of 32");
    pack_linear_frag(host_w, host_out, N, K, N, 0);
    return 0;
}

int vd_profile_begin(void) {
    for (auto& r : g_prof.recs) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    g_prof.recs.clear();
    g_prof.on = true;
    return 0;
}

int vd_profile_end(double* out, int cap) {
    g_prof.on = false;
    VD_REQUIRE(out && cap >= 4 * PC_COUNT, "output needs 4*vd_profile_classes() doubles");
    for (int i = 0; i < 4 * PC_COUNT; ++i) out[i] = 0.0;
    for (auto& r : g_prof.recs) {
        VD_HIP(hipEventSynchronize(r.b));
        float ms = 0.f;
        VD_HIP(hipEventElapsedTime(&ms, r.a, r.b));
        out[r.cls * 4 + 0] += 1.0; out[r.cls * 4 + 1] += ms; out[r.cls * 4 + 2] += r.flops; out[r.cls * 4 + 3] += r.bytes;
        if (getenv("VD_PROF_DUMP"))                    // per-launch listing for kernel work (tools/)
            fprintf(stderr, "[vd_prof] %-28s %-40s %8.1f us %7.1f TFLOP/s\n", prof_name(r.cls), r.tag, ms * 1e3,
                    ms > 0 ? r.flops / ms / 1e9 : 0.0);
        (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b);
    }
    g_prof.recs.clear();
    return 0;
}

int vd_profile_classes(void) { return PC_COUNT; }
const char* vd_profile_class_name(int i) { return prof_name(i); }

// ---- single-operator entry points ---------------------------------------------------------------
int vd_op_conv(const float* src0, const float* src1, int C0, int Cin, int nfr, int Hs, int Ws, int ups, int stride,
               int pad, int ksz, const float* w, const float* w_frag, const float* w_wino, const float* bias, const float* affA,
               const float* affB, int act,
               const float* res, const float* fbias, int fbias_ld, float* out, int Cout, void* stream) {
    IgemmArgs g{};
    g.src0 = src0; g.src1 = src1; g.C0 = C0; g.Cin = Cin; g.nfr = nfr; g.Hs = Hs; g.Ws = Ws; g.ups = ups;
    g.stride = stride; g.pad = pad; g.ksz = ksz;
    g.Ho = ((Hs << ups) + 2 * pad - ksz) / stride + 1;
    g.Wo = ((Ws << ups) + 2 * pad - ksz) / stride + 1;
    g.w = w; g.wfrag = w_frag; g.wwino = w_wino; g.bias = bias; g.affA = affA; g.affB = affB; g.act = act; g.res = res; g.res_ld = Cout;
    g.fbias = fbias; g.fbias_ld = fbias_ld; g.out = out; g.ldo = Cout; g.Cout = Cout; g.M = nfr * g.Ho * g.Wo;
    return launch_igemm(g, static_cast<hipStream_t>(stream));
}

int vd_conv_stats_split(int Hout) { return conv_wino_stats_split(Hout); }
int vd_conv_wino_block_couts(int nfr, int H, int Cin, int Cout) { return conv_wino_z128_shape(nfr, H, Cin, Cout) ? 128 : 64; }

int vd_op_conv_stats(const float* src0, int Cin, int nfr, int Hs, int Ws, int ups, const float* w_wino, const float* bias,
                     const float* res, const float* fbias, int fbias_ld, float* out, int Cout, double* gn_part,
                     void* stream) {
    IgemmArgs g{};
    g.src0 = src0; g.C0 = Cin; g.Cin = Cin; g.nfr = nfr; g.Hs = Hs; g.Ws = Ws; g.ups = ups;
    g.stride = 1; g.pad = 1; g.ksz = 3;
    g.Ho = Hs << ups; g.Wo = Ws << ups;
    g.wwino = w_wino; g.bias = bias; g.res = res; g.res_ld = Cout; g.fbias = fbias; g.fbias_ld = fbias_ld;
    g.out = out; g.ldo = Cout; g.Cout = Cout; g.M = nfr * g.Ho * g.Wo;
    g.stats = gn_part; g.stats_split = conv_wino_stats_split(g.Ho);
    VD_REQUIRE(conv_wino_supported(g), "vd_op_conv_stats: shape not covered by the Winograd kernel");
    return launch_igemm(g, static_cast<hipStream_t>(stream));
}

int vd_op_gn_affine(const double* part0, int split0, int C0, const double* part1, int split1, int C, int nfr, int HW,
                    const float* gamma, const float* beta, const float* film, int film_ld, float* affA, float* affB,
                    void* stream) {
    return launch_gn_affine(part0, split0, C0, part1, split1, (double)HW * (C / 32), gamma, beta, film, film_ld, nfr, C, affA,
                            affB, static_cast<hipStream_t>(stream));
}

int vd_op_gn_fold(const float* src0, const float* src1, int C0, int C, int nfr, int HW, const float* gamma,
                  const float* beta, const float* film, int film_ld, float* affA, float* affB, void* stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int split = gn_stats_split(nfr, HW, C);
    double* part;
    VD_HIP(hipMalloc(reinterpret_cast<void**>(&part), (size_t)nfr * split * C * 2 * sizeof(double)));
    int rc = launch_gn_stats(src0, src1, C0, C, nfr, HW, part, split, st);
    if (!rc) rc = launch_gn_affine(part, split, C, nullptr, 0, (double)HW * (C / 32), gamma, beta, film, film_ld, nfr, C, affA, affB, st);
    (void)hipStreamSynchronize(st);
    (void)hipFree(part);
    return rc;
}

int vd_op_affine_act(const float* src0, const float* src1, int C0, int C, const float* affA, const float* affB, int nfr,
                     int HW, int act, float* y, void* stream) {
    return launch_affine_act(src0, src1, C0, C, affA, affB, nfr, HW, act, y, static_cast<hipStream_t>(stream));
}

int vd_op_affine_apply(const float* x, const float* affA, const float* affB, int nfr, int HW, int C, float* y, void* stream) {
    return launch_affine_apply(x, affA, affB, nfr, HW, C, y, static_cast<hipStream_t>(stream));
}

int vd_op_gn_temporal(const float* x, const float* gamma, const float* beta, int B, int T, int HW, int C, float* y,
                      void* stream) {
    return launch_gn_temporal(x, gamma, beta, B, T, HW, C, y, static_cast<hipStream_t>(stream));
}

int vd_op_attn_spatial(const float* qkv, int nfr, int L, int C, int heads, float* out, void* stream) {
    AttnSpatialArgs a{qkv, out, nfr, L, C, heads, 1.0f / sqrtf((float)(C / heads))};
    return launch_attn_spatial(a, static_cast<hipStream_t>(stream));
}

int vd_op_attn_temporal(const float* qkv, const float* Rk, const float* Rq, const float* Rv, const float* mask, int B,
                        int T, int HW, int C, int heads, int allow_pad, float* out, void* stream) {
    AttnTemporalArgs a{qkv, Rk, Rq, Rv, mask, out, B, T, HW, C, heads, allow_pad, 1.0f / sqrtf((float)(C / heads))};
    return launch_attn_temporal(a, static_cast<hipStream_t>(stream));
}

int vd_op_out_conv(const float* x, const float* affA, const float* affB, const float* w, const float* bias, int nfr,
                   int H, int W, int C, int Cout, float* out, void* stream) {
    return launch_out_conv(x, affA, affB, w, bias, nfr, H, W, C, Cout, out, static_cast<hipStream_t>(stream));
}

}  // extern "C"
