// libeffconf host side: weight packing, workspace layout, the encoder forward schedule and the C ABI
// declared in include/effconf.h.  Everything the forward path does is "enqueue kernels on the
// caller's stream": no allocation, no synchronisation, no host<->device copies (graph-capturable).
//
// The forward schedule follows ConformerEncoder.forward (reference models/encoders.py:97-142) and
// ConformerBlock.forward (models/blocks.py:119-137); see DESIGN.md for the kernel map.
#include "kernels.h"
#include "../../include/effconf.h"
#ifdef EFFCONF_DEBUG_ABI
#include "../../include/effconf_debug.h"
#endif

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace {

thread_local std::string g_err;
int fail(const std::string& m) { g_err = m; return -1; }
#define EC_TRY(expr) do { int _rc = (expr); if (_rc != 0) return fail(std::string(#expr) + " failed rc=" + std::to_string(_rc)); } while (0)

inline int ld8(int d) { return ec_round_up(d, 8); }

// Timing-only ablation build (tools/build_ablate.py: -DEFFCONF_ABLATE into a SEPARATE library, never the product): EFFCONF_SKIP = bit mask of kernel
// families whose launches are dropped (1 attention, 2 chain A, 4 chain B, 8 depthwise conv, 16 mel, 32 subsampling, 64 glue) - what a family costs the
// STEP when three row ranges overlap on three streams (results are wrong by construction)
#ifdef EFFCONF_ABLATE
static int ablate_mask() { static const int m = getenv("EFFCONF_SKIP") ? atoi(getenv("EFFCONF_SKIP")) : 0; return m; }
#define EC_ABL(bit, stmt) do { if (!(ablate_mask() & (bit))) { stmt; } } while (0)
#else
#define EC_ABL(bit, stmt) do { stmt; } while (0)
#endif

uint16_t h_f2bf(float f) {
    uint32_t u; memcpy(&u, &f, 4);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

struct HostTensor { std::vector<float> data; std::vector<int64_t> shape; };

struct PackedLinear { const bf16_t* w = nullptr; const float* bias = nullptr; int N = 0, K = 0, ldw = 0; std::vector<float> hbias; /* host copy of the padded bias */ };
struct LNp { const float* g = nullptr; const float* b = nullptr; };

struct BlockW {
    LNp ln_ffn1, ln_att, ln_conv, ln_ffn2, ln_out;
    PackedLinear ffn1_a, ffn1_b, qkv, qkv_nat, pos, outp, pw1, pw2, res, ffn2_a, ffn2_b;
    const bf16_t *ffn1_bp = nullptr, *ffn2_bp = nullptr;   // W2 with the hidden index permuted per 16 (rsgemm.hip)
    const float *u = nullptr, *v = nullptr, *dw_w = nullptr, *dw_b = nullptr;
    const uint16_t* dw_a = nullptr;              // pack_dwconv_mfma: Toeplitz rows of the depthwise taps for dwconv_mfma_kernel (stride-1 layers)
    const float* dvu = nullptr; int dvu_ld = 0;   // (v - u) per head column [H][dvu_ld], zero beyond d (attention derives Q + v from Q + u)
    const bf16_t* pos_table = nullptr;   // [2*max_pos-1][ld8(D)], row r <-> position max_pos-1-r
    // fused row-local chains (chain.hip): every weight with its K index permuted per 16; FFN second weight / bias pre-scaled by 1/2
    bool chain_in = false, chain_out = false;          // chain-packed weights exist for the D-wide / De-wide parts of the block
    PackedLinear c_outp, c_pw1, c_pw2, c_qkv, c_f1a, c_f2a;
    const bf16_t *c_f1b_cm = nullptr, *c_f2b_cm = nullptr;      // the same second weights chunk-major (chain2.hip): every 32-hidden-unit slab contiguous, in LDS slot order
    const bf16_t *c_f1b = nullptr, *c_f2b = nullptr; const float *c_f1b2 = nullptr, *c_f2b2 = nullptr;
    int c_qkv_chunks = 0, c_pw1_chunks = 0;
    std::vector<float> h_ln_out_g, h_ln_out_b, h_u, h_v, h_f1b2, h_f2b2;     // host copies for the chains' constant blocks
    // split mode (sxf_ffn.hip): weight images of the two feed-forward modules, b2 / 2, hidden chunks
    const uint16_t *xf_img[2] = {nullptr, nullptr}; const float* xf_b2[2] = {nullptr, nullptr}; int xf_nch[2] = {0, 0};
    // split mode (sxf_chain.hip): images in the accumulator layout's k order - out-proj / pointwise-2 (F2), pointwise-1 with GLU row pairs / Q | K | V (F1, pre-norm
    // folded), the two feed-forward modules; biases of the F2 products
    const uint16_t *xc_wo = nullptr, *xc_p1 = nullptr, *xc_p2 = nullptr, *xc_qkv = nullptr, *xc_f[2] = {nullptr, nullptr};
    const float *xc_bo = nullptr, *xc_bp2 = nullptr; int xc_nch_p1 = 0;
    bool xc_in = false, xc_out = false;          // the D-wide (out-proj, pointwise-1, FFN1, Q K V) / De-wide (pointwise-2, FFN2) images exist
    const float *cc_b = nullptr, *cc_head = nullptr, *cc_tail = nullptr, *cc_full = nullptr;   // constant blocks (chain_const_layout)
};

struct TraceEntry { char name[64]; int64_t offset, rows, cols, ld; int32_t dtype; };
struct ProfRec { int cls; double flops, bytes; };

}  // namespace

int ec_fail(const char* msg) { return fail(msg ? msg : "error"); }   // shared with rnnt.hip


struct EcEncoder {
    EcConfig cfg;
    std::vector<EcBlock> blocks;
    std::map<std::string, HostTensor> host;
    bool finalized = false;
    std::vector<void*> allocs;
    int wide_gemm = 0;                    // option "wide_gemm": GemmParams::wide of every tiled GEMM (0 by shape, 1 never, 2 / 3 forced)
    size_t guard_bytes = 0;               // EFFCONF_POISON_GUARDS (test hook): NaN-filled guard regions around every parameter buffer
    // packed
    const float *sub_w9 = nullptr, *sub_b = nullptr;
    PackedLinear lin;
    const bf16_t* lin_fused = nullptr; int lin_fused_ld = 0;   // Linear weight in the fused kernel's K order (sublinear.hip)
    const uint16_t *sub3_cimg = nullptr, *sub3_wimg = nullptr; const float* sub3_bias = nullptr; int sub3_ncb = 0, sub3_fo = 0;   // sublinear3.hip (kernels.h: SubLin3Params)
    const bf16_t* lin_rs = nullptr; const float* conv_tab = nullptr;   // sublinear2.hip: Linear weight [F/2][32 NT][32 CG] (K-permuted per 16), conv taps [32 CG][16]
    int fuse_subsample = 2;                  // 0: separate conv + GEMM kernels, 1: sublinear.hip, 2: sublinear2.hip where it supports the shape (else 1; wide front ends: sublinear3.hip, option sub3_auto), 3: sublinear3.hip
    bool fuse_chain = true;                  // row-local chains (chain.hip) where supported
    int ctc_mfma = 2;                        // CTC head: 2 split-bf16 operands on the bf16 MFMA (bf16 path; fp32 mode falls back to 1), 1 fp32 MFMA (bit-identical to 0), 0 the VALU kernel
    int attention_v2 = 1;                    // 0: attention.hip; 1 (default) / 2: attention2.hip variants where they support the head width (padded <= 160)
    // tuning / test options that used to be process-global environment switches (effconf_encoder_set_option)
    int chain_count_stores = 0;              // measurement only: the unsafe counted waits of rounds 3 - 6 (kernels.h: ChainParams::count_stores)
    int chain_variant = 1, chain_full_max = 192, attn_waves = 4, rs_variant = 0, ffn_variant = 0;
    int chain_pair_min_d = 193, chain_nt = 0, chain_w2cm = 1;   // round 5 defaults: the column-pair kernels at padded width 256 (D = 240: 147 -> 119 us per tail + head; at 192 they cost more per row than chain.hip's 256-row workgroups)
    int dwconv_mfma = 1;                         // stride-1 depthwise convolutions on the matrix pipe (conv.hip dwconv_mfma_kernel): 1 = kernel size 15 (the Efficient Conformer
                                                 // family), 2 = also 31 / 7 (equally accurate, profiles/r5_35_dw_accuracy.txt, but ConformerCTC-Small's 5-frame test utterance sits ON the
                                                 // stated tolerance with either kernel and crosses it with this one's rounding: 0.0608 against 0.06), 0 = dwconv_kernel (VALU) everywhere
    int chain_pair = 5, chain_pair_min_m = 0;   // chain2.hip (column-pair chains at padded width 192 / 256): 0 off, 1 burst refills, 2 hooked; launches below chain_pair_min_m rows stay on chain.hip
    int chain_small_m = 4096;                // chain launches of at most this many rows run as 2-wave workgroups (small-batch latency; bit-identical rows)
    int chain_max_dim = 256;                 // fused chains only for stage widths <= this (tuning: wider stages on the per-GEMM / tiled kernels)
    int tiled_auto = 1;                      // wide_gemm = 0: configurations whose widest stage lies in (tiled_min_k, 384] (EfficientConformer Medium: D = 360) send that stage to LayerNorm + the tiled
                                             // GEMMs (+ 2.3 % on Medium, profiles/r6_100_*; neutral where wider stages exist - Large - which keep the row-stationary kernels there); 0 = as before round 6's last session
    bool tiled_auto_on = false;              // the rule's outcome for this configuration (finalize)
    int tiled_min_k = 256;                   // with wide_gemm >= 2: layers with K > this leave the row-stationary kernels for LayerNorm + tiled GEMMs
    std::vector<float*> att_out;             // per block: device buffer [B][H][Tg][Tg] for the softmax maps of the next forward, or null
    int split_chain = 1;                     // split mode: the row-local work of a block as two kernels (sxf_chain.hip) where the width is built; 0 = per-module kernels (tests)
    int sub3_auto = 1;                       // with fuse_subsample = 2: front ends wider than 128 channels / columns on sublinear3.hip (0: sublinear2.hip / conv + GEMM as before round 6)
    int split_sublin = 1;                    // split mode: Conv2dSubsampling + Linear as one kernel (sxf_sub.hip) for the one-layer subsampler; 0 = conv kernel + GEMM [+ row gather] (tests)
    int split_ffn = 1;                       // split mode: the feed-forward modules as one kernel each (sxf_ffn.hip) where the width is built; 0 = LayerNorm + two GEMMs (tests)
    int exact_attention = 0;                 // fp32 mode: 0 tiled attention kernel (2: its 16-row shape), 1 one wave per query row (round 2's); bit-identical
    bool head_major_odd = false;             // odd grouped head widths on the head-major Q/K/V layout (tests; the default reads the natural layout unaligned)
    // two-layer subsampler (plain Conformer configs): layer-2 implicit-GEMM weight [N][9*Cp] (tap, c_in), folded bias, Cp
    const bf16_t* sub2_w = nullptr; const float* sub2_b = nullptr; int sub2_cp = 0;
    std::vector<BlockW> bw;
    const float *fc_wt = nullptr, *fc_b = nullptr;
    const bf16_t *fc_hi = nullptr, *fc_lo = nullptr;      // fc.weight as split-bf16 MFMA B fragments (launch_ctc_split)
    const int* block_stride = nullptr;
    const int *block_group = nullptr, *block_heads = nullptr;     // ragged batches: attention group size / heads per block (device)
    MelTables mel{};
    // trace
    char* trace_arena = nullptr; size_t trace_bytes = 0, trace_used = 0;
    std::vector<TraceEntry> trace;
    // positional-embedding cache: E = pos_layer(R) depends only on (block, T); when the caller keeps the SAME workspace
    // untouched between forwards (opt-in), the 15-18 small E projections are skipped for an unchanged T
    // One tag per workspace: callers that alternate workspaces (one per stream) keep every one of them warm.
    bool e_cache_on = false;
    struct ECacheTag { const void* ws; int batch, tm; size_t layout; };   // layout: offset of the first E buffer (ragged batches: it moves with the row totals)
    std::vector<ECacheTag> e_cache;          // most recently used last; at most E_CACHE_MAX entries
    static constexpr size_t E_CACHE_MAX = 16;
    bool e_cache_hit(const void* ws, int batch, int tm, size_t layout) const {   // the workspace layout depends on (batch, tm) [+ the row totals]
        for (const ECacheTag& t : e_cache) if (t.ws == ws) return t.batch == batch && t.tm == tm && t.layout == layout;
        return false;
    }
    void e_cache_put(const void* ws, int batch, int tm, size_t layout) {
        e_cache_drop(ws);
        if (e_cache.size() >= E_CACHE_MAX) e_cache.erase(e_cache.begin());
        e_cache.push_back({ws, batch, tm, layout});
    }
    void e_cache_drop(const void* ws) {
        for (size_t i = 0; i < e_cache.size(); ++i) if (e_cache[i].ws == ws) { e_cache.erase(e_cache.begin() + i); break; }
    }
    // fp32-operand "exact" mode (exact.hip): raw fp32 state-dict tensors on the device by key, fp32 sinusoid tables,
    // per-layer BatchNorm scale / shift of the subsampling convs
    bool exact_pack = false, exact_on = false;
    // exact_fp32 = 2: the same schedule with every GEMM / the attention products on the fp16 matrix pipe with split operands (split.hip)
    bool exact_split = false;
    struct SplitW { const uint16_t *hi, *lo; int ldh; };
    std::map<std::string, SplitW> xsplit;    // Linear / 1x1 conv weights by state-dict prefix (+ the stacked "...mhsa.qkv_layer")
    std::map<std::string, const float*> xw;
    std::map<std::pair<int, int>, const float*> xtab;
    const float *xsub_scale[2] = {nullptr, nullptr}, *xsub_shift[2] = {nullptr, nullptr};
    // split mode: images of the fused front end (sxf_sub.hip; kernels.h: SxfSubParams) - one-layer subsampler only
    const uint16_t *xsub_cimg = nullptr, *xsub_wimg = nullptr; const float* xsub_bias = nullptr; int xsub_ncb = 0, xsub_fo = 0;
    // per-launch event profiler (bench / tuning only; off by default)
    bool prof_on = false;
    std::vector<hipEvent_t> prof_ev;          // pairs
    std::vector<ProfRec> prof_rec;
    size_t prof_next = 0;
};

namespace {

template <class T>
const T* upload(EcEncoder* e, const std::vector<T>& v) {
    // guard > 0 (EFFCONF_POISON_GUARDS, a test hook read once at create): every parameter buffer sits between two guard regions of 0xFF
    // bytes (NaN as bf16 and as fp32), so a read past either end of a packed weight shows up in the output instead of depending on
    // what the allocator happened to place next to it
    const size_t guard = e->guard_bytes, used = v.size() * sizeof(T);
    void* d = nullptr;
    size_t bytes = std::max<size_t>((used + 15) / 16 * 16, 16);
    if (hipMalloc(&d, bytes + 2 * guard) != hipSuccess) return nullptr;
    e->allocs.push_back(d);
    char* base = static_cast<char*>(d) + guard;
    if (guard && (hipMemset(d, 0xFF, guard) != hipSuccess || hipMemset(base + used, 0xFF, bytes - used + guard) != hipSuccess)) return nullptr;
    if (!v.empty() && hipMemcpy(base, v.data(), used, hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    return reinterpret_cast<const T*>(base);
}

const HostTensor* find(EcEncoder* e, const std::string& k) {
    auto it = e->host.find(k);
    return it == e->host.end() ? nullptr : &it->second;
}

// [N][K] fp32 (row-major, possibly a gather of rows given by `rows`) -> padded bf16 + padded bias
// kperm: K index permuted inside every group of 16 (packed position 8h+e holds column 4h + 8(e>>2) + (e&3)): the B fragments of
// chain.hip are LayerNorm-ed accumulator registers in MFMA C order
// ln_g / ln_b: a LayerNorm in front of this linear layer folded in: W diag(gamma), b + W beta (fp32, before the bf16 rounding)
bool pack_linear(EcEncoder* e, const std::vector<const float*>& row_ptr, const std::vector<float>& bias, int K, PackedLinear* out,
                 bool kperm = false, const float* ln_g = nullptr, const float* ln_b = nullptr, int min_ldw = 0) {
    const int N = (int)row_ptr.size();
    const int Np = ec_round_up(N, 128), Kp = ec_round_up(K > min_ldw ? K : min_ldw, 64);
    std::vector<uint16_t> w((size_t)Np * Kp, 0);
    std::vector<float> b(Np, 0.f);
    for (int n = 0; n < N && n < (int)bias.size(); ++n) b[n] = bias[n];
    for (int n = 0; n < N; ++n) {
        if (!row_ptr[n]) continue;
        for (int k = 0; k < Kp; ++k) {
            int src = k;
            if (kperm) { const int g = k / 16, pp = k % 16, hh = pp >> 3, ee = pp & 7; src = g * 16 + 4 * hh + 8 * (ee >> 2) + (ee & 3); }
            if (src < K) w[(size_t)n * Kp + k] = h_f2bf(row_ptr[n][src] * (ln_g ? ln_g[src] : 1.0f));
        }
        if (ln_b) {
            double acc = 0.0;
            for (int k = 0; k < K; ++k) acc += (double)row_ptr[n][k] * ln_b[k];
            b[n] += (float)acc;
        }
    }
    out->w = upload(e, w);
    out->bias = upload(e, b);
    out->hbias = b;
    out->N = N; out->K = K; out->ldw = Kp;
    return out->w && out->bias;
}

// min_ldw: row pitch floor in elements.  The whole-row kernels of rsgemm.hip (RS_F32 / RS_RESID) pick their k-step class from
// max(K, N) and DMA that many columns of every weight row: an expanding layer (K < N) must be packed at least N wide, or the last
// row's DMA runs past the buffer (found with EFFCONF_POISON_GUARDS: conv_res 180 -> 256 of EfficientConformer Medium)
bool pack_named_linear(EcEncoder* e, const std::string& prefix, int N, int K, PackedLinear* out, std::string* err, bool kperm = false,
                       const std::string& fold_ln = "", int min_ldw = 0) {
    const HostTensor* w = find(e, prefix + ".weight");
    const HostTensor* b = find(e, prefix + ".bias");
    if (!w || !b) { *err = "missing tensor " + prefix + ".weight/.bias"; return false; }
    if ((int64_t)w->data.size() != (int64_t)N * K || (int)b->data.size() != N) { *err = "shape mismatch for " + prefix; return false; }
    std::vector<const float*> rows(N);
    for (int n = 0; n < N; ++n) rows[n] = w->data.data() + (size_t)n * K;
    const HostTensor *lg = fold_ln.empty() ? nullptr : find(e, fold_ln + ".weight"), *lb = fold_ln.empty() ? nullptr : find(e, fold_ln + ".bias");
    if (!fold_ln.empty() && (!lg || !lb || (int)lg->data.size() != K || (int)lb->data.size() != K)) { *err = "missing LayerNorm " + fold_ln; return false; }
    return pack_linear(e, rows, b->data, K, out, kperm, lg ? lg->data.data() : nullptr, lb ? lb->data.data() : nullptr, min_ldw);
}

// second FFN weight [D][F] with the hidden (K) index permuted inside every group of 16 so that the first GEMM's
// accumulator registers are directly the second GEMM's B fragments: position 8h+e <-> 4h + 8(e>>2) + (e&3)
const bf16_t* pack_ffn2_permuted(EcEncoder* e, const std::string& prefix, int D, int F, float scale = 1.0f) {
    const HostTensor* w = find(e, prefix + ".weight");
    if (!w || (int64_t)w->data.size() != (int64_t)D * F) return nullptr;
    const int Np = ec_round_up(D, 128), Kp = ec_round_up(F, 64);
    std::vector<uint16_t> out((size_t)Np * Kp, 0);
    for (int n = 0; n < D; ++n)
        for (int k = 0; k < Kp; ++k) {
            const int g = k / 16, pp = k % 16, hh = pp >> 3, ee = pp & 7;
            const int src = g * 16 + 4 * hh + 8 * (ee >> 2) + (ee & 3);
            if (src < F) out[(size_t)n * Kp + k] = h_f2bf(scale * w->data[(size_t)n * F + src]);
        }
    return upload(e, out);
}

// The same weight chunk-major for chain2.hip: slab c (hidden units 32c .. 32c + 31 of all DP output rows, 64 B per row) is contiguous and already in
// the LDS image's slot order (piece pc of row n at slot 4n + ((pc + (n >> 2)) & 3), rowstat.h dma_w2_off), so a wave-DMA reads 1 KiB of consecutive
// bytes (row-major: sixteen 64-byte half lines per wave-DMA, the other half of every line belonging to the next slab)
const bf16_t* pack_ffn2_chunkmajor(EcEncoder* e, const std::string& prefix, int D, int F, float scale, int DP) {
    const HostTensor* w = find(e, prefix + ".weight");
    if (!w || (int64_t)w->data.size() != (int64_t)D * F) return nullptr;
    const int nch = ec_round_up(F, 32) / 32;
    std::vector<uint16_t> out((size_t)nch * DP * 32, 0);
    for (int c = 0; c < nch; ++c)
        for (int n = 0; n < DP; ++n)
            for (int pc = 0; pc < 4; ++pc) {
                const int slot = 4 * n + ((pc + (n >> 2)) & 3);
                for (int i = 0; i < 8; ++i) {
                    const int k = 32 * c + 8 * pc + i;
                    const int g = k / 16, pp = k % 16, hh = pp >> 3, ee = pp & 7;
                    const int src = g * 16 + 4 * hh + 8 * (ee >> 2) + (ee & 3);
                    if (n < D && src < F) out[((size_t)c * DP * 4 + slot) * 8 + i] = h_f2bf(scale * w->data[(size_t)n * F + src]);
                }
            }
    return upload(e, out);
}

bool get_ln(EcEncoder* e, const std::string& prefix, int D, LNp* out, std::string* err) {
    const HostTensor* g = find(e, prefix + ".weight");
    const HostTensor* b = find(e, prefix + ".bias");
    if (!g || !b || (int)g->data.size() != D || (int)b->data.size() != D) { *err = "missing/mis-shaped LayerNorm " + prefix; return false; }
    out->g = upload(e, g->data);
    out->b = upload(e, b->data);
    return out->g && out->b;
}

// BatchNorm(eval) fold: y = (x - mean) / sqrt(var + 1e-5) * gamma + beta  -> per-channel scale / shift
bool bn_fold(EcEncoder* e, const std::string& prefix, int C, std::vector<float>* scale, std::vector<float>* shift, std::string* err) {
    const HostTensor *g = find(e, prefix + ".weight"), *b = find(e, prefix + ".bias");
    const HostTensor *m = find(e, prefix + ".running_mean"), *v = find(e, prefix + ".running_var");
    if (!g || !b || !m || !v || (int)g->data.size() != C) { *err = "missing/mis-shaped BatchNorm " + prefix; return false; }
    scale->resize(C); shift->resize(C);
    for (int c = 0; c < C; ++c) {
        const float s = g->data[c] / std::sqrt(v->data[c] + 1e-5f);
        (*scale)[c] = s;
        (*shift)[c] = b->data[c] - m->data[c] * s;
    }
    return true;
}

// Relative sinusoid table, fp32 operation order of the reference (attentions.py:1219-1226 / 1275-1284):
// angle = pos / 10000^(2i/D) in fp32, row r <-> position max_pos-1-r, even cols sin, odd cols cos.
const bf16_t* build_pos_table(EcEncoder* e, int max_pos, int D) {
    const int rows = 2 * max_pos - 1, ld = ld8(D);
    std::vector<uint16_t> t((size_t)rows * ld, 0);
    std::vector<float> denom(D / 2);
    for (int i = 0; i < D / 2; ++i) denom[i] = std::pow(10000.0f, (2.0f * (float)i) / (float)D);
    for (int r = 0; r < rows; ++r) {
        const float pos = (float)(max_pos - 1 - r);
        for (int i = 0; i < D / 2; ++i) {
            const float a = pos / denom[i];
            t[(size_t)r * ld + 2 * i] = h_f2bf(std::sin(a));
            t[(size_t)r * ld + 2 * i + 1] = h_f2bf(std::cos(a));
        }
    }
    return upload(e, t);
}

bool build_mel_tables(EcEncoder* e, std::string* err) {
    const EcConfig& c = e->cfg;
    if (c.n_fft != 512) { *err = "only n_fft = 512 is native"; return false; }
    // Hann(win_length, periodic) centred in n_fft (torch.stft pads the window on both sides)
    std::vector<float> win(c.n_fft, 0.f);
    const int off = (c.n_fft - c.win_length) / 2;
    for (int n = 0; n < c.win_length; ++n) win[off + n] = (float)(0.5 - 0.5 * std::cos(2.0 * M_PI * n / c.win_length));
    std::vector<float2> tw(c.n_fft / 2);
    for (int k = 0; k < c.n_fft / 2; ++k) {
        const double a = -2.0 * M_PI * k / c.n_fft;
        tw[k] = make_float2((float)std::cos(a), (float)std::sin(a));
    }
    // HTK triangular filterbank, f in [0, 8000], no area normalisation (torchaudio melscale_fbanks, modules.py:82)
    const int nf = c.n_fft / 2 + 1, nm = c.n_mels;
    const double fmin = 0.0, fmax = 8000.0;
    auto hz2mel = [](double f) { return 2595.0 * std::log10(1.0 + f / 700.0); };
    auto mel2hz = [](double m) { return 700.0 * (std::pow(10.0, m / 2595.0) - 1.0); };
    std::vector<double> fpts(nm + 2);
    for (int i = 0; i < nm + 2; ++i) fpts[i] = mel2hz(hz2mel(fmin) + (hz2mel(fmax) - hz2mel(fmin)) * i / (nm + 1));
    std::vector<int> start(nm), count(nm), offs(nm);
    std::vector<float> wts;
    for (int m = 0; m < nm; ++m) {
        int s0 = -1, cnt = 0;
        std::vector<float> row;
        for (int k = 0; k < nf; ++k) {
            const double f = (double)(c.sample_rate / 2) * k / (nf - 1);
            const double down = (f - fpts[m]) / (fpts[m + 1] - fpts[m]);
            const double up = (fpts[m + 2] - f) / (fpts[m + 2] - fpts[m + 1]);
            const double w = std::max(0.0, std::min(down, up));
            if (w > 0.0) {
                if (s0 < 0) s0 = k;
                row.resize(k - s0 + 1, 0.f);
                row[k - s0] = (float)w;
                cnt = k - s0 + 1;
            }
        }
        start[m] = s0 < 0 ? 0 : s0; count[m] = cnt; offs[m] = (int)wts.size();
        wts.insert(wts.end(), row.begin(), row.begin() + cnt);
    }
    e->mel.window = upload(e, win);
    e->mel.twiddle = upload(e, tw);
    e->mel.fb_start = upload(e, start);
    e->mel.fb_count = upload(e, count);
    e->mel.fb_offset = upload(e, offs);
    e->mel.fb_nnz = (int)wts.size();
    e->mel.fb_weight = upload(e, wts);
    return e->mel.window && e->mel.twiddle && e->mel.fb_weight;
}

// ------------------------------------------------------------------ shapes + workspace layout
struct Shapes {
    int B, Tm, T1;                 // Tm / T1: mel frames / frames after the subsampling (of the LONGEST utterance when ragged)
    std::vector<int> Tin, Tout;    // frames entering / leaving each block (longest utterance when ragged)
    // rows of the residual stream entering / leaving each block and of the Q / K / V buffers: B * T (B * Tp for Q / K / V), or - ragged -
    // the sum over the utterances of their frames rounded up to the block's attention group size
    std::vector<long long> Min, Mout, Mq;
    bool ragged = false;
    std::vector<int> wgs, tiles;   // ragged: attention workgroups (heads x 64-query tiles) and depthwise-conv tiles (128 frames) per block
    std::vector<double> tg2;       // ragged: sum over the utterances of (grouped length)^2 per block (attention flop accounting)
    long long Mfinal = 0;          // ragged: rows of the encoder output (sum of the utterances' output frames)
};

Shapes make_shapes(const EcEncoder* e, int B, int Tm) {
    Shapes s; s.B = B; s.Tm = Tm;
    int t = Tm;
    for (int i = 0; i < e->cfg.sub_layers; ++i) t = (t - 1) / 2 + 1;
    s.T1 = t;
    for (const EcBlock& b : e->blocks) {
        s.Tin.push_back(t);
        s.Min.push_back((long long)B * t);
        s.Mq.push_back((long long)B * ec_round_up(t, b.group_size));
        if (b.conv_stride > 1) t = (t - 1) / b.conv_stride + 1;
        s.Tout.push_back(t);
        s.Mout.push_back((long long)B * t);
    }
    return s;
}

// ragged batch: `tm[b]` mel frames of every utterance (host).  The same length chain as lengths_ragged_kernel (floor divisions of positive
// numbers), accumulated into the totals the host needs for grids and the workspace.
Shapes make_shapes_ragged(const EcEncoder* e, const std::vector<int>& tm) {
    const int B = (int)tm.size(), nb = (int)e->blocks.size();
    int tmax = 0;
    for (int v : tm) tmax = std::max(tmax, v);
    Shapes s = make_shapes(e, B, tmax);
    s.ragged = true;
    s.Min.assign(nb, 0); s.Mout.assign(nb, 0); s.Mq.assign(nb, 0); s.wgs.assign(nb, 0); s.tiles.assign(nb, 0); s.tg2.assign(nb, 0.0);
    for (int b = 0; b < B; ++b) {
        int t = tm[b];
        for (int i = 0; i < e->cfg.sub_layers; ++i) t = (t - 1) / 2 + 1;
        for (int k = 0; k < nb; ++k) {
            const EcBlock& bk = e->blocks[k];
            const int G = bk.group_size, Gn = k + 1 < nb ? e->blocks[k + 1].group_size : 1;
            const int tp = ec_round_up(t, G);
            s.Min[k] += tp; s.Mq[k] += tp;
            s.wgs[k] += bk.num_heads * ec_cdiv(tp / G, 64);
            s.tg2[k] += (double)(tp / G) * (tp / G);
            if (bk.conv_stride > 1) t = (t - 1) / bk.conv_stride + 1;
            const int top = ec_round_up(t, Gn);
            s.Mout[k] += top;
            s.tiles[k] += ec_cdiv(top, 128);
        }
        s.Mfinal += t;
    }
    return s;
}

struct Workspace {
    size_t total = 0;
    size_t mel, sub, sub1, x0, x1, a, hbuf, qu, kh, vt, eh, o, gbuf, cbuf, xs, lens, preds;
    size_t mel_len = 0, row_off = 0, wg_off = 0, tile_off = 0;     // ragged descriptors (ints)
    size_t xrect = 0;                                              // ragged + unfused front end: rectangular Linear output before the gather
    std::vector<size_t> eh_blk;   // per-block E (kept across forwards for the cache)
};

inline size_t al(size_t x) { return (x + 255) & ~(size_t)255; }

Workspace make_workspace(const EcEncoder* e, const Shapes& s, bool from_audio) {
    Workspace w;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += al(bytes); return o; };
    const size_t B = s.B;
    size_t mx = 0, ma = 0, mh = 0, mq = 0, mvt = 0, me = 0, mg = 0, mc = 0;
    std::vector<size_t> esz;
    for (size_t k = 0; k < e->blocks.size(); ++k) {
        const EcBlock& b = e->blocks[k];
        const size_t T = s.Tin[k], D = b.dim_model, De = b.dim_expand;
        const size_t Tp = ec_round_up((int)T, b.group_size), Tg = Tp / b.group_size;
        const size_t d = (size_t)b.group_size * D / b.num_heads, dpad = ec_round_up((int)d, 32);
        const size_t Mi = (size_t)s.Min[k], Mo = (size_t)s.Mout[k], Mqk = (size_t)s.Mq[k];     // rows in, rows out, Q / K / V rows
        mx = std::max(mx, std::max(Mi * D, Mo * De) * 4);
        ma = std::max(ma, std::max(Mi * ld8(D), Mo * ld8(De)) * 2);
        mh = std::max(mh, std::max(Mi * D, Mo * De) * b.ff_ratio * 2);
        // Q / K / V: natural layout Mq rows x D; the head-major test layout needs B * H * Tg * dpad (rectangular batches only)
        const size_t qkv = std::max(Mqk * D, s.ragged ? (size_t)0 : B * b.num_heads * Tg * dpad);
        mq = std::max(mq, qkv * 2 + 512);     // + slack: 16-byte chunk loads may run past a row's head span
        mvt = std::max(mvt, qkv * 2 + 512);
        me = std::max(me, (size_t)b.num_heads * (2 * Tg - 1) * dpad * 2);
        esz.push_back((size_t)b.num_heads * (2 * Tg - 1) * dpad * 2 + 512);
        mg = std::max(mg, Mi * ld8(De) * 2);
        mc = std::max(mc, Mo * ld8(De) * 2);
    }
    w.mel = take(from_audio ? B * e->cfg.n_mels * s.Tm * 4 : 0);
    const int C = e->cfg.sub_filters[e->cfg.sub_layers - 1];
    int F = e->cfg.n_mels; for (int i = 0; i < e->cfg.sub_layers; ++i) F /= 2;
    const bool rag_unfused = s.ragged && !(e->fuse_subsample == 2 && e->lin_rs);        // (s.Tm = the input's row pitch in ragged batches)
    size_t T1r = (size_t)s.T1;
    if (s.ragged) { T1r = (size_t)s.Tm; for (int i = 0; i < e->cfg.sub_layers; ++i) T1r = (T1r - 1) / 2 + 1; }      // rows per utterance of the rectangular image
    // scratch of the unfused front ends; ragged rows: every utterance's frames rounded up to the first block's group size
    w.sub = take(s.ragged && !rag_unfused ? 0 : B * (T1r + (s.ragged ? e->blocks[0].group_size - 1 : 0)) * C * F * 2);
    w.xrect = take(rag_unfused ? B * T1r * e->blocks[0].dim_model * 4 : 0);
    {   // two-layer subsampler: channel-last layer-1 activation [B][F/2][T after layer 1][Cp]
        const size_t tl1 = (s.Tm - 1) / 2 + 1;
        w.sub1 = take(e->cfg.sub_layers == 2 ? B * (e->cfg.n_mels / 2) * tl1 * ec_round_up(e->cfg.sub_filters[0], 64) * 2 : 0);
    }
    w.x0 = take(mx); w.x1 = take(mx);
    w.a = take(ma); w.hbuf = take(mh);
    w.qu = take(mq); w.kh = take(mq); w.vt = take(mvt); w.eh = take(me);
    w.o = take(ma); w.gbuf = take(mg); w.cbuf = take(mc); w.xs = take(ma);
    w.lens = take((e->blocks.size() + 1) * B * 4);
    if (s.ragged) {
        const size_t nbk = e->blocks.size();
        w.mel_len = take(B * 4);
        w.row_off = take((nbk + 1) * (B + 1) * 4);
        w.wg_off = take(nbk * (B + 1) * 4);
        w.tile_off = take(nbk * (B + 1) * 4);
    }
    for (size_t k = 0; k < esz.size(); ++k) w.eh_blk.push_back(take(esz[k]));
    w.preds = take(0);
    w.total = off;
    return w;
}

// ------------------------------------------------------------------ per-launch profiler
struct ProfScope {
    EcEncoder* e; hipStream_t st; bool on;
    ProfScope(EcEncoder* e_, hipStream_t st_, int cls, double flops, double bytes) : e(e_), st(st_), on(e_->prof_on) {
        if (!on) return;
        if (e->prof_next + 2 > e->prof_ev.size()) {
            for (int i = 0; i < 2; ++i) { hipEvent_t ev; (void)hipEventCreate(&ev); e->prof_ev.push_back(ev); }
        }
        e->prof_rec.push_back(ProfRec{cls, flops, bytes});
        (void)hipEventRecord(e->prof_ev[e->prof_next], st);
    }
    ~ProfScope() {
        if (!on) return;
        (void)hipEventRecord(e->prof_ev[e->prof_next + 1], st);
        e->prof_next += 2;
    }
};
#define PROF(cls, flops, bytes) ProfScope _prof_scope(e, st, (cls), (double)(flops), (double)(bytes))

// ------------------------------------------------------------------ forward
void trace_add(EcEncoder* e, hipStream_t st, const char* name, const void* ptr, int64_t rows, int64_t cols, int64_t ld, int dtype) {
    if (!e->trace_arena) return;
    const size_t esz = dtype == 1 ? 2 : 4;
    const size_t bytes = (size_t)rows * ld * esz;
    const size_t off = al(e->trace_used);
    if (off + bytes > e->trace_bytes) return;
    (void)hipMemcpyAsync(e->trace_arena + off, ptr, bytes, hipMemcpyDeviceToDevice, st);
    TraceEntry t{};
    snprintf(t.name, sizeof(t.name), "%s", name);
    t.offset = (int64_t)off; t.rows = rows; t.cols = cols; t.ld = ld; t.dtype = dtype;
    e->trace.push_back(t);
    e->trace_used = off + bytes;
}

enum ProfClass { PC_MEL = 0, PC_SUBCONV = 1, PC_GEMM_FFN = 2, PC_GEMM_OTHER = 3, PC_LAYERNORM = 4, PC_ATTENTION = 5,
                 PC_DWCONV = 6, PC_MISC = 7, PC_COUNT = 8 };

int run_gemm(EcEncoder* e, int cls, hipStream_t st, const bf16_t* A, int lda, int M, const PackedLinear& L, int epi, void* C, int ldc,
             const float* R = nullptr, int ldr = 0, float alpha = 1.f) {
    const double out_b = (epi == EPI_F32) ? 4.0 : (epi == EPI_RESID_F32 ? 8.0 : 2.0);
    PROF(cls, 2.0 * M * (double)L.N * L.K, (double)M * L.K * 2 + (double)L.N * L.K * 2 + (double)M * L.N * out_b);
    GemmParams p{};
    p.A = A; p.lda = lda; p.W = L.w; p.ldw = L.ldw; p.bias = L.bias;
    p.M = M; p.N = L.N; p.K = L.K; p.C = C; p.ldc = ldc; p.R = R; p.ldr = ldr; p.alpha = alpha;
    p.wide = e->wide_gemm;
    return launch_gemm(p, epi, st);
}

// x += alpha * FFN(a)  — fused row-stationary kernel when the width allows, else two tiled GEMMs
// ln != null: the pre-norm is computed inside the fused kernel's prologue (a is not read); the tiled fallback needs `a`
inline int F1c(const EcBlock& b) { return ec_round_up(b.dim_model * b.ff_ratio, 32); }

// POST half of a chain A: the block's FFN1 (pre-norm ln[2]), attention pre-norm ln[3] and stacked Q/K/V projection
void fill_chain_head(ChainParams& cp, const BlockW& W, int D, int Fp, int T, int Tp, const GemmParams& q) {
    cp.D = D;
    cp.ln[2] = ChainLn{W.ln_ffn1.g, W.ln_ffn1.b};
    cp.ln[3] = ChainLn{W.ln_att.g, W.ln_att.b};
    cp.f[1] = ChainFfn{W.c_f1a.w, W.c_f1a.ldw, W.c_f1a.bias, W.c_f1b, W.ffn1_b.ldw, W.c_f1b2, Fp, W.c_f1b_cm};
    cp.g1 = ChainGemm{W.c_qkv.w, W.c_qkv.ldw, W.c_qkv.bias, W.c_qkv_chunks};
    cp.qu = q.qu; cp.kh = q.kh; cp.vt = q.vt; cp.u = W.u; cp.v = W.v; cp.T = T; cp.Tp = Tp;
}

bool prefer_tiled(const EcEncoder* e, int M, int N, int K);

int run_ffn(EcEncoder* e, hipStream_t st, const bf16_t* a, int M, int D, const PackedLinear& L1, const PackedLinear& L2,
            const bf16_t* w2p, float* x, bf16_t* hbuf, const LNp* ln = nullptr) {
    const int F = L1.N;
    if (ffn_fused_supported(D) && !(prefer_tiled(e, M, F, D) && !ln)) {
        PROF(PC_GEMM_FFN, 4.0 * M * (double)D * F, (double)M * D * 10 + 4.0 * D * F);
        FfnParams p{};
        p.A = a; p.lda = ld8(D); p.X = x; p.ldx = D; p.Y = x; p.ldy = D;
        p.W1 = L1.w; p.ldw1 = L1.ldw; p.b1 = L1.bias; p.W2 = w2p; p.ldw2 = L2.ldw; p.b2 = L2.bias;
        p.M = M; p.D = D; p.Fp = ec_round_up(F, 32); p.alpha = 0.5f; p.variant = e->ffn_variant;
        if (ln) { p.ln_g = ln->g; p.ln_b = ln->b; }
        return launch_ffn_fused(p, st);
    }
    int rc = run_gemm(e, PC_GEMM_FFN, st, a, ld8(D), M, L1, EPI_SWISH_BF16, hbuf, F);
    if (rc) return rc;
    return run_gemm(e, PC_GEMM_FFN, st, hbuf, F, M, L2, EPI_RESID_F32, x, D, x, D, 0.5f);
}

// Widths 257 .. 384 (EfficientConformer Large stage 1, Medium stage 3) fit the row-stationary kernels, but at 24 k-steps those run one
// wave per SIMD and stream every weight per 32-row tile: 60 - 200 TFLOP/s (profiles/r2_03_large_kernel_stats.txt).  `wide_gemm` 2 / 3
// sends these layers to LayerNorm + the tiled GEMMs instead.  Only when forced: chosen by row count (gemm256.hip's tiles filling the
// chip) it was -3 % kernel time on Large and wall-neutral, and it made a forward's bits depend on how the batch is split into row ranges
// (the two paths round differently; tools/robustness_sweep.py) - the gemm.hip / gemm256.hip choice does not (bit-identical kernels).
bool prefer_tiled(const EcEncoder* e, int M, int N, int K) {
    (void)M; (void)N;
    return (e->wide_gemm == 2 || e->wide_gemm == 3 || (e->wide_gemm == 0 && (e->tiled_auto == 2 || (e->tiled_auto && e->tiled_auto_on)))) && K > e->tiled_min_k && K % 8 == 0;
}

// row-stationary single GEMM when K <= 384, else the tiled kernel
int run_rs_or_tiled(EcEncoder* e, int cls, hipStream_t st, const bf16_t* A, int lda, int M, const PackedLinear& L, int rs_epi,
                    int tiled_epi, void* C, int ldc, const float* R = nullptr, int ldr = 0, float alpha = 1.f,
                    const float* lnX = nullptr, const LNp* ln = nullptr) {
    const bool ok = (rs_epi == 0 || rs_epi == 1) ? rs_gemm_resident_supported(L.K, L.N) : rs_gemm_supported(L.K);
    if (!ok) return run_gemm(e, cls, st, A, lda, M, L, tiled_epi, C, ldc, R, ldr, alpha);
    if (prefer_tiled(e, M, L.N, L.K)) {
        if (lnX && ln) { PROF(PC_LAYERNORM, 0, (double)M * L.K * 6); EC_TRY(launch_layernorm(lnX, M, L.K, ln->g, ln->b, nullptr, const_cast<bf16_t*>(A), lda, nullptr, nullptr, st)); }
        return run_gemm(e, cls, st, A, lda, M, L, tiled_epi, C, ldc, R, ldr, alpha);
    }
    const double out_b = (tiled_epi == EPI_F32) ? 4.0 : (tiled_epi == EPI_RESID_F32 ? 8.0 : 2.0);
    PROF(cls, 2.0 * M * (double)L.N * L.K, (double)M * L.K * 2 + (double)L.N * L.K * 2 + (double)M * L.N * out_b);
    GemmParams p{};
    p.A = A; p.lda = lda; p.W = L.w; p.ldw = L.ldw; p.bias = L.bias;
    p.M = M; p.N = L.N; p.K = L.K; p.C = C; p.ldc = ldc; p.R = R; p.ldr = ldr; p.alpha = alpha;
    p.rs_variant = e->rs_variant;
    if (lnX && ln) { p.X = lnX; p.ldx = L.K; p.ln_g = ln->g; p.ln_b = ln->b; }
    return launch_rs_gemm(p, rs_epi, st);
}

// Conv2dSubsampling (modules.py:232-249) + transpose + Linear (encoders.py:113-116): mel (B, n_mels, Tm) -> x fp32 (B * T1, D0).
// sub / act1: bf16 scratch of the unfused variants (the subsampler's output rows / the two-layer subsampler's layer-1 image).
// sublinear3.hip: option fuse_subsample = 3 (every one-layer front end it is built for) or 2 = the default where sublinear2.hip has no instance or runs one workgroup per
// CU (channel counts / widths above 128); a debug trace keeps the kernels that write the "subsample" activation only when fuse_subsample = 0
bool use_sublinear3(const EcEncoder* e) {
    if (!e->sub3_wimg || e->cfg.sub_layers != 1) return false;
    if (e->fuse_subsample == 3) return true;
    return e->fuse_subsample == 2 && e->sub3_auto && (e->cfg.sub_filters[0] > 128 || e->blocks[0].dim_model > 128);
}

int run_subsample_linear(EcEncoder* e, hipStream_t st, const float* mel, int B, int Tm, int T1, bf16_t* sub, bf16_t* act1, float* x) {
    const EcConfig& c = e->cfg;
    const int C0 = c.sub_filters[0], F2 = c.n_mels / 2, Ksub = C0 * F2;
    if (c.sub_layers == 2) {
        const int Tl1 = (Tm - 1) / 2 + 1, F1 = c.n_mels / 2, F2q = c.n_mels / 4, C1 = c.sub_filters[1];
        { PROF(PC_SUBCONV, 2.0 * 9 * B * Tl1 * (double)C0 * F1, (double)B * c.n_mels * Tm * 4 + (double)B * Tl1 * F1 * e->sub2_cp * 2);
          EC_TRY(launch_subsample_conv_cl(mel, B, c.n_mels, Tm, Tl1, e->sub_w9, e->sub_b, C0, e->sub2_cp, act1, st)); }
        { PROF(PC_GEMM_OTHER, 2.0 * 9 * (double)B * F2q * T1 * C0 * C1, (double)B * Tl1 * F1 * e->sub2_cp * 2 + (double)B * T1 * F2q * C1 * 2);
          EC_TRY(launch_conv2_igemm(act1, B, F1, Tl1, e->sub2_cp, e->sub2_w, 9 * e->sub2_cp, e->sub2_b, C1, F2q, T1, sub, st)); }
        trace_add(e, st, "subsample", sub, (int64_t)B * T1, F2q * C1, F2q * C1, 1);
        EC_TRY(run_gemm(e, PC_GEMM_OTHER, st, sub, F2q * C1, B * T1, e->lin, EPI_F32, x, e->lin.N));
    } else if (use_sublinear3(e)) {
        PROF(PC_SUBCONV, 2.0 * 9 * B * T1 * (double)Ksub + 2.0 * B * T1 * (double)Ksub * e->lin.N,
             (double)B * c.n_mels * Tm * 4 + (double)B * T1 * e->lin.N * 4);
        SubLin3Params sp{};
        sp.mel = mel; sp.B = B; sp.F = c.n_mels; sp.Tm = Tm; sp.To = T1; sp.rows_max = T1;
        sp.cimg = e->sub3_cimg; sp.wimg = e->sub3_wimg; sp.bias = e->sub3_bias; sp.y = x; sp.ldy = e->lin.N; sp.N = e->lin.N; sp.ncb = e->sub3_ncb; sp.Fo = e->sub3_fo;
        EC_TRY(launch_sublinear3(sp, st));
    } else if (e->fuse_subsample >= 2 && e->lin_rs) {
        PROF(PC_SUBCONV, 2.0 * 9 * B * T1 * (double)Ksub + 2.0 * B * T1 * (double)Ksub * e->lin.N,
             (double)B * c.n_mels * Tm * 4 + (double)B * T1 * e->lin.N * 4);
        EC_TRY(launch_sublinear2(mel, B, c.n_mels, Tm, T1, e->conv_tab, e->lin_rs, e->lin.bias, C0, e->lin.N, x, e->lin.N, st));
    } else if (e->fuse_subsample && e->lin_fused) {
        PROF(PC_SUBCONV, 2.0 * 9 * B * T1 * (double)Ksub + 2.0 * B * T1 * (double)Ksub * e->lin.N,
             (double)B * c.n_mels * Tm * 4 + (double)B * T1 * e->lin.N * 4);
        EC_TRY(launch_sublinear_fused(mel, B, c.n_mels, Tm, T1, e->sub_w9, e->sub_b, C0, e->lin_fused, e->lin_fused_ld,
                                      e->lin.bias, e->lin.N, x, e->lin.N, st));
    } else {
        { PROF(PC_SUBCONV, 2.0 * 9 * B * T1 * (double)Ksub, (double)B * c.n_mels * Tm * 4 + (double)B * T1 * Ksub * 2); EC_TRY(launch_subsample_conv(mel, B, c.n_mels, Tm, T1, e->sub_w9, e->sub_b, C0, sub, Ksub, st)); }
        trace_add(e, st, "subsample", sub, (int64_t)B * T1, Ksub, Ksub, 1);
        EC_TRY(run_gemm(e, PC_GEMM_OTHER, st, sub, Ksub, B * T1, e->lin, EPI_F32, x, e->lin.N));
    }
    return 0;
}

// chain launches of width D and M rows that go to chain2.hip (launch_chain's rule): there the tail and the next head of chain A are one kernel up to D = 256
static const void* dw_mfma_table(const EcEncoder* e, const uint16_t* t, int ks) { return (e->dwconv_mfma == 2 || (e->dwconv_mfma == 1 && ks == 15)) ? t : nullptr; }
static bool pair_on(const EcEncoder* e, int D, int M) { return e->chain_pair && chain2_supported(D) && D >= e->chain_pair_min_d && M >= e->chain_pair_min_m; }

// Ragged batches (s.ragged): every utterance runs at its own length in one concatenated row space (kernels.h: RaggedRows) - the row-local
// kernels (chains, GEMMs, LayerNorms) just see M rows; the frame-mixing ones (subsampling, attention, depthwise conv, conv_res decimation)
// index utterances through the descriptor arrays lengths_ragged_kernel leaves in the workspace.  out: (B, out_frames, D_last), zero filled
// behind every utterance's own last frame.
int forward_core(EcEncoder* e, const float* mel, const int64_t* in_len, int from_audio, const Shapes& s, const Workspace& w,
                 char* ws, float* out, int64_t* out_len, hipStream_t st, int out_frames = 0) {
    const EcConfig& c = e->cfg;
    const int B = s.B, nb = (int)e->blocks.size();
    const bool rg = s.ragged;
    e->trace.clear(); e->trace_used = 0;
    int* lens = reinterpret_cast<int*>(ws + w.lens);
    const int *mel_len = nullptr, *row_off = nullptr, *wg_off = nullptr, *tile_off = nullptr;
    if (rg) {
        int* ml = reinterpret_cast<int*>(ws + w.mel_len); int* ro = reinterpret_cast<int*>(ws + w.row_off);
        int* wo = reinterpret_cast<int*>(ws + w.wg_off); int* to = reinterpret_cast<int*>(ws + w.tile_off);
        PROF(PC_MISC, 0, 0);
        EC_TRY(launch_lengths_ragged(in_len, B, from_audio, c.hop_length, c.sub_layers, e->block_stride, e->block_group, e->block_heads, nb, lens, ml,
                                     ro, wo, to, out_len, st));
        mel_len = ml; row_off = ro; wg_off = wo; tile_off = to;
    } else {
        PROF(PC_MISC, 0, 0); EC_TRY(launch_lengths(in_len, B, from_audio, c.hop_length, c.sub_layers, e->block_stride, nb, lens, out_len, st));
    }
    if (from_audio) trace_add(e, st, "mel", mel, (int64_t)B * c.n_mels, s.Tm, s.Tm, 0);
    auto rows_at = [&](int k) { RaggedRows r{}; r.off = row_off + (size_t)k * (B + 1); r.len = lens + (size_t)k * B; r.n = B;
                                r.rows = (int)(k < nb ? s.Min[k] : s.Mfinal); r.tmax = k < nb ? s.Tin[k] : s.Tout[nb - 1]; return r; };

    // ---- Conv2dSubsampling (modules.py:232-249) + transpose + Linear (encoders.py:113-116)
    float* x = reinterpret_cast<float*>(ws + w.x0);
    float* xalt = reinterpret_cast<float*>(ws + w.x1);
    if (rg) {
        const int C0 = c.sub_filters[0], Ksub = C0 * (c.n_mels / 2);
        const RaggedRows r0 = rows_at(0);
        if (c.sub_layers == 2) {
            // two-layer subsampler (the plain Conformer configurations): both convolutions and the Linear on the RECTANGULAR image - layer 1
            // zero-fills every utterance's image behind its own last frame, so layer 2 sees the zero padding of the utterance run alone -
            // then the valid rows are gathered into the ragged row space
            bf16_t* sub = reinterpret_cast<bf16_t*>(ws + w.sub);
            bf16_t* act1 = reinterpret_cast<bf16_t*>(ws + w.sub1);
            float* xrect = reinterpret_cast<float*>(ws + w.xrect);
            const int Tl1 = (s.Tm - 1) / 2 + 1, T1r = (Tl1 - 1) / 2 + 1, F1 = c.n_mels / 2, F2q = c.n_mels / 4, C1 = c.sub_filters[1];
            { PROF(PC_SUBCONV, 2.0 * 9 * B * Tl1 * (double)C0 * F1, (double)B * c.n_mels * s.Tm * 4 + (double)B * Tl1 * F1 * e->sub2_cp * 2);
              EC_TRY(launch_subsample_conv_cl(mel, B, c.n_mels, s.Tm, Tl1, e->sub_w9, e->sub_b, C0, e->sub2_cp, act1, st, mel_len)); }
            { PROF(PC_GEMM_OTHER, 2.0 * 9 * (double)B * F2q * T1r * C0 * C1, (double)B * Tl1 * F1 * e->sub2_cp * 2 + (double)B * T1r * F2q * C1 * 2);
              EC_TRY(launch_conv2_igemm(act1, B, F1, Tl1, e->sub2_cp, e->sub2_w, 9 * e->sub2_cp, e->sub2_b, C1, F2q, T1r, sub, st)); }
            EC_TRY(run_gemm(e, PC_GEMM_OTHER, st, sub, F2q * C1, B * T1r, e->lin, EPI_F32, xrect, e->lin.N));
            { PROF(PC_MISC, 0, (double)s.Min[0] * e->lin.N * 8); EC_TRY(launch_gather_rows(xrect, e->lin.N, T1r, r0, x, st)); }
        } else if (use_sublinear3(e)) {                            // sublinear3.hip: workgroup = (utterance, 128 frames)
            PROF(PC_SUBCONV, 2.0 * 9 * (double)s.Min[0] * Ksub + 2.0 * (double)s.Min[0] * Ksub * e->lin.N, (double)B * c.n_mels * s.Tm * 4 + (double)s.Min[0] * e->lin.N * 4);
            SubLin3Params sp{};
            sp.mel = mel; sp.B = B; sp.F = c.n_mels; sp.Tm = s.Tm; sp.mel_len = mel_len; sp.off = r0.off; sp.len = r0.len;
            sp.rows_max = ec_round_up(s.Tin[0], e->blocks[0].group_size);
            sp.cimg = e->sub3_cimg; sp.wimg = e->sub3_wimg; sp.bias = e->sub3_bias; sp.y = x; sp.ldy = e->lin.N; sp.N = e->lin.N; sp.ncb = e->sub3_ncb; sp.Fo = e->sub3_fo;
            EC_TRY(launch_sublinear3(sp, st));
        } else if (e->fuse_subsample >= 2 && e->lin_rs) {        // sublinear2.hip indexes the ragged rows itself
            PROF(PC_SUBCONV, 2.0 * 9 * (double)s.Min[0] * Ksub + 2.0 * (double)s.Min[0] * Ksub * e->lin.N, (double)B * c.n_mels * s.Tm * 4 + (double)s.Min[0] * e->lin.N * 4);
            EC_ABL(32, EC_TRY(launch_sublinear2(mel, B, c.n_mels, s.Tm, s.T1, e->conv_tab, e->lin_rs, e->lin.bias, C0, e->lin.N, x, e->lin.N, st, &r0, mel_len)));
        } else {
            // wide front ends (Large: 360 filters): conv (zero padding at every utterance's own last mel frame) + Linear on the RECTANGULAR
            // (B, T1 of the longest) rows, then the valid rows are gathered into the ragged row space (pad rows are computed and dropped:
            // the subsampler is a few percent of the step)
            // Round 4: the conv writes the RAGGED rows itself (tiles behind an utterance's own end exit; group-padding rows = zeros) and the
            // Linear runs on those rows only - until round 3 both ran on the (B, longest) rectangle (24 % padding on the bench batch) and a
            // gather pass copied the valid rows.  Group-padding rows of x = the Linear's bias (finite; no kernel mixes them into valid rows).
            bf16_t* sub = reinterpret_cast<bf16_t*>(ws + w.sub);
            const int T1r = (s.Tm - 1) / 2 + 1;            // rows per utterance of the rectangular image (pitch of the input)
            const int Tcover = T1r + e->blocks[0].group_size - 1;      // >= every utterance's frames rounded up to the group size
            { PROF(PC_SUBCONV, 2.0 * 9 * (double)s.Min[0] * Ksub, (double)B * c.n_mels * s.Tm * 4 + (double)s.Min[0] * Ksub * 2);
              EC_TRY(launch_subsample_conv(mel, B, c.n_mels, s.Tm, Tcover, e->sub_w9, e->sub_b, C0, sub, Ksub, st, mel_len, &r0)); }
            EC_TRY(run_gemm(e, PC_GEMM_OTHER, st, sub, Ksub, (int)s.Min[0], e->lin, EPI_F32, x, e->lin.N));
        }
    } else {
        EC_TRY(run_subsample_linear(e, st, mel, B, s.Tm, s.T1, reinterpret_cast<bf16_t*>(ws + w.sub), reinterpret_cast<bf16_t*>(ws + w.sub1), x));
    }
    trace_add(e, st, "linear", x, s.Min[0], e->lin.N, e->lin.N, 0);

    bf16_t* a = reinterpret_cast<bf16_t*>(ws + w.a);
    bf16_t* hbuf = reinterpret_cast<bf16_t*>(ws + w.hbuf);
    bf16_t* o = reinterpret_cast<bf16_t*>(ws + w.o);
    bf16_t* gbuf = reinterpret_cast<bf16_t*>(ws + w.gbuf);
    bf16_t* cbuf = reinterpret_cast<bf16_t*>(ws + w.cbuf);
    bf16_t* xs = reinterpret_cast<bf16_t*>(ws + w.xs);
    bool have_a = false, head_done = false;
    char nm[64];
    // While the stream is being CAPTURED into a hipGraph nothing executes: the positional projections must be part of the graph (a replay
    // recomputes them) and the workspace must not be tagged warm (an eager forward before the first replay would read E nobody wrote)
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(st, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone;
    // E of block k depends on the frame count entering it (the LONGEST utterance's in a ragged batch, where s.Tm is only the input's row pitch)
    const int e_tag = rg ? -(s.Tin[0] + 1) : s.Tm;
    const bool e_cached = !capturing && e->e_cache_on && e->e_cache_hit(ws, B, e_tag, w.eh_blk[0]);
    if (!e_cached) e->e_cache_drop(ws);       // re-tagged only after every projection of this forward was enqueued

    int mask_stride = 1;                       // product of the strides of the blocks before block k
    for (int k = 0; k < nb; ++k) {
        const EcBlock& b = e->blocks[k];
        const BlockW& W = e->bw[k];
        const int T = s.Tin[k], To = s.Tout[k], D = b.dim_model, De = b.dim_expand;      // ragged: the LONGEST utterance's frames
        const int M = (int)s.Min[k], Mo = (int)s.Mout[k];
        const int G = b.group_size, H = b.num_heads;
        const int Tp = ec_round_up(T, G), Tg = Tp / G, Tgp = ec_round_up(Tg, 8);
        const int d = G * D / H, dpad = ec_round_up(d, 32);
        // Q/K/V/E layout: "natural" row-major [B*Tp][D] (16-byte row stores from the GEMM, head split = pointer arithmetic in
        // the attention kernel).  An odd grouped head width (d = 135: Medium / Large stage 0) makes the head spans only 2-byte
        // aligned; gfx950 global loads are alignment-free, so the attention kernel reads them as they are - the head-major
        // fallback (a scatter epilogue of 2-byte stores, 9 % of Medium's step) is kept behind the option "head_major_odd" for tests.
        const bool head_major_odd = e->head_major_odd;
        const bool nat = (d % 2) == 0 || !head_major_odd;
        if (rg && !nat) return fail("ragged batches use the natural Q / K / V layout (option head_major_odd = 0)");
        // rows (b, t) -> Q / K / V rows b * Tp + t; a ragged batch keeps every utterance's rows group-padded in the residual stream itself,
        // so the map is the identity: ONE "utterance" of M rows
        const int qT = rg ? M : T, qTp = rg ? M : Tp;
        const bool chain_head = e->fuse_chain && W.chain_in && nat && chain_head_supported(D) && D <= e->chain_max_dim;          // FFN1 + QKV of this block as a fused chain
        const bool chain_b = e->fuse_chain && W.chain_in && D <= e->chain_max_dim;                      // out-proj + LN + pointwise-1/GLU
        const bool chain_tail = e->fuse_chain && W.chain_out && chain_tail_supported(De) && De <= e->chain_max_dim;                  // pointwise-2 + FFN2 + block norm (+ next block's head)
        GemmParams p{};
        p.A = a; p.lda = ld8(D); p.W = W.qkv.w; p.ldw = W.qkv.ldw; p.bias = W.qkv.bias;
        p.M = M; p.N = 3 * D; p.K = D;
        p.T = qT; p.G = rg ? 1 : G; p.H = H; p.D = D; p.d = d; p.dpad = dpad; p.Tg = rg ? M : Tg; p.Tgp = Tgp;
        p.qu = reinterpret_cast<bf16_t*>(ws + w.qu);
        p.kh = reinterpret_cast<bf16_t*>(ws + w.kh); p.vt = reinterpret_cast<bf16_t*>(ws + w.vt);
        p.u = W.u; p.v = W.v; p.rs_variant = e->rs_variant;
        if (head_done) {
            // FFN1 and the Q/K/V projection of this block already ran inside the previous block's tail chain
        } else if (chain_head) {
            ChainParams cp{};
            cp.variant = e->chain_variant; cp.count_stores = e->chain_count_stores; cp.small_m = e->chain_small_m; cp.pair = e->chain_pair; cp.pair_small_max = e->chain_pair_min_m - 1; cp.pair_min_d = e->chain_pair_min_d; cp.nt = e->chain_nt; cp.w2cm = e->chain_w2cm;
            fill_chain_head(cp, W, D, F1c(b), qT, qTp, p);
            cp.M = M; cp.X = x; cp.ldx = D; cp.Y = x; cp.ldy = D; cp.consts = W.cc_head;
            PROF(PC_GEMM_FFN, 2.0 * M * (double)D * (2.0 * D * b.ff_ratio + 3.0 * D), (double)M * D * 16 + 22.0 * D * D);
            EC_ABL(2, EC_TRY(launch_chain(cp, CHAIN_A_HEAD, st)));
        } else {
            // ---- x += 1/2 FFN1(x)   (blocks.py:122; modules.py:385-392)
            { PROF(PC_LAYERNORM, 0, (double)M * D * 6); if (!have_a) EC_TRY(launch_layernorm(x, M, D, W.ln_ffn1.g, W.ln_ffn1.b, nullptr, a, ld8(D), nullptr, nullptr, st)); }
            EC_TRY(run_ffn(e, st, a, M, D, W.ffn1_a, W.ffn1_b, W.ffn1_bp, x, hbuf));
            // ---- Q/K/V of LN(x)   (modules.py:472-488; attentions.py:651-686)
            const bool qkv_tiled = nat && prefer_tiled(e, M, 3 * D, D);
            const bool ln_fused = rs_gemm_supported(D) && !qkv_tiled;      // pre-norm computed in the QKV kernel's prologue
            if (!ln_fused) { PROF(PC_LAYERNORM, 0, (double)M * D * 6); EC_TRY(launch_layernorm(x, M, D, W.ln_att.g, W.ln_att.b, nullptr, a, ld8(D), nullptr, nullptr, st)); }
            { PROF(PC_GEMM_OTHER, 2.0 * M * 3.0 * D * D, (double)M * D * 2 + 3.0 * D * D * 2 + (double)M * D * 8);
              if (ln_fused) {
                  if (nat) { p.W = W.qkv_nat.w; p.ldw = W.qkv_nat.ldw; p.bias = W.qkv_nat.bias; }
                  p.X = x; p.ldx = D; p.ln_g = W.ln_att.g; p.ln_b = W.ln_att.b;
                  EC_TRY(launch_rs_gemm(p, nat ? 4 : 3, st));
              } else {
                  p.wide = e->wide_gemm;
                  EC_TRY(launch_gemm(p, nat ? EPI_QKV_NAT : EPI_QKV, st));
              } }
        }
        snprintf(nm, sizeof(nm), "blocks.%d.x_ffn1", k); trace_add(e, st, nm, x, M, D, D, 0);

        // ---- x += MHSA(LN(x))   (blocks.py:125-126; attentions.py:549-718)
        {
            { PROF(PC_MISC, 0, 0);
              if (rg) EC_ABL(64, EC_TRY(launch_attn_pad_rows_ragged(p.qu, p.kh, p.vt, W.u, D, G, rows_at(k), st)));
              else EC_TRY(nat ? launch_attn_pad_rows_nat(p, B, st) : launch_attn_pad_rows(p, B, st)); }
            // positional embeddings E = pos_layer(R) (attentions.py:588 / 678): input independent, tiny (2Tp-G rows)
            GemmParams pe{};
            // relative tables: R[m] = sinusoid(Tp - 1 - G/2 - m), m < 2 Tp - G; causal: R[m] = sinusoid(Tp - 1 - m), m < Tp (attentions.py:1243-1251, 1296-1309)
            const int erows = c.causal ? Tp : 2 * Tp - G;
            pe.A = W.pos_table + (size_t)(b.max_pos - Tp + (c.causal ? 0 : G / 2)) * ld8(D); pe.lda = ld8(D);
            pe.W = W.pos.w; pe.ldw = W.pos.ldw; pe.bias = W.pos.bias;
            pe.M = erows; pe.N = D; pe.K = D;
            pe.T = erows; pe.G = G; pe.H = H; pe.D = D; pe.d = d; pe.dpad = dpad; pe.Tg = c.causal ? Tg : 2 * Tg - 1; pe.Tgp = 0;
            pe.kh = reinterpret_cast<bf16_t*>(ws + w.eh_blk[k]);
            pe.C = pe.kh; pe.ldc = D;
            if (Tp > b.max_pos) return fail("sequence longer than max_pos_encoding");
            if (!e_cached) { PROF(PC_GEMM_OTHER, 2.0 * erows * (double)D * D, (double)erows * D * 4 + (double)D * D * 2);
                             EC_TRY(launch_gemm(pe, nat ? EPI_BF16 : EPI_HEADS, st)); }
            AttnParams ap{};
            ap.qu = p.qu; ap.kh = p.kh; ap.vt = p.vt; ap.eh = pe.kh;
            ap.dvu = W.dvu; ap.dvu_ld = W.dvu_ld;
            ap.lens = lens + (size_t)k * B;
            ap.B = B; ap.H = H; ap.T = T; ap.G = G; ap.D = D; ap.d = d; ap.dpad = dpad; ap.Tg = Tg; ap.Tgp = Tgp;
            if (nat) { ap.q_bstride = (long long)Tp * D; ap.q_hstride = d; ap.q_rowstride = G * D; ap.e_hstride = d; ap.e_rowstride = G * D; }
            else { ap.q_bstride = (long long)H * Tg * dpad; ap.q_hstride = (long long)Tg * dpad; ap.q_rowstride = dpad;
                   ap.e_hstride = (long long)(2 * Tg - 1) * dpad; ap.e_rowstride = dpad; }
            ap.out = o; ap.ldo = ld8(D); ap.scale = 1.0f / std::sqrt((float)d); ap.force_waves = e->attn_waves;
            // streaming mask of this block: built after the subsampling, sliced ::stride after every strided block before this one and ::G in
            // grouped attention (encoders.py:132-136, attentions.py:698): grouped positions compare (mask_stride * G) * (j - i) with the contexts
            const long long unit = (long long)mask_stride * G;
            ap.band_l = (int)std::min<long long>(c.left_context / unit, 1 << 30); ap.band_r = (int)std::min<long long>(c.right_context / unit, 1 << 30);
            ap.causal = c.causal;
            const bool streaming = c.causal || ap.band_l < Tg || ap.band_r < Tg;
            if (streaming && !(nat && relpos_attention2_supported(dpad) && e->attention_v2))
                return fail("streaming contexts / causal attention run on attention2.hip (natural layout, padded head width <= 160, option attention_v2 != 0)");
            if (rg) {
                if (!relpos_attention2_supported(dpad)) return fail("ragged batches need attention2.hip (padded head width <= 160)");
                ap.rag_off = row_off + (size_t)k * (B + 1); ap.rag_wg = wg_off + (size_t)k * (B + 1); ap.rag_nwg = s.wgs[k]; ap.rag_tgmax = Tg;
            }
            { PROF(PC_ATTENTION, 2.0 * H * (rg ? s.tg2[k] : (double)B * Tg * Tg) * d * 3.0, (double)M * D * 2 * 5);
              if (rg || streaming) EC_ABL(1, EC_TRY(launch_relpos_attention2(ap, 1, st))); else
              // attention2.hip reads the natural layout only (its column masks assume the next head's finite data behind a head span); the
              // head-major test layout of odd head widths (EFFCONF_HEAD_MAJOR_ODD) stays on attention.hip
              if (e->attention_v2 && nat && relpos_attention2_supported(dpad)) EC_TRY(launch_relpos_attention2(ap, e->attention_v2, st));
              else EC_TRY(launch_relpos_attention(ap, st)); }
            if ((int)e->att_out.size() == nb && e->att_out[k]) {       // opt-in: the reference's att_w of this block (encoders.py:129)
                // ragged batches: (B, H, Tg of the LONGEST utterance, same) per block, an utterance's own Tg x Tg block = its map run alone, zeros elsewhere
                EC_TRY(launch_attention_probs(ap, e->att_out[k], st));
            }
            snprintf(nm, sizeof(nm), "blocks.%d.att_o", k); trace_add(e, st, nm, o, M, D, ld8(D), 1);
            if (chain_b) {
                ChainParams cp{};
                cp.variant = e->chain_variant; cp.count_stores = e->chain_count_stores; cp.small_m = e->chain_small_m; cp.pair = e->chain_pair; cp.pair_small_max = e->chain_pair_min_m - 1; cp.pair_min_d = e->chain_pair_min_d; cp.nt = e->chain_nt; cp.w2cm = e->chain_w2cm;
                cp.M = M; cp.D = D; cp.X = x; cp.ldx = D; cp.Y = x; cp.ldy = D; cp.A = o; cp.lda = ld8(D);
                cp.g0 = ChainGemm{W.c_outp.w, W.c_outp.ldw, W.c_outp.bias, 0};
                cp.ln[0] = ChainLn{W.ln_conv.g, W.ln_conv.b};
                cp.g1 = ChainGemm{W.c_pw1.w, W.c_pw1.ldw, W.c_pw1.bias, W.c_pw1_chunks};
                cp.glu = gbuf; cp.ldg = ld8(De); cp.Ng = De; cp.T = qT; cp.Tp = qTp; cp.consts = W.cc_b;
                PROF(PC_GEMM_OTHER, 2.0 * M * (double)D * (D + 2.0 * De), (double)M * D * 10 + (double)M * De * 2 + 2.0 * D * (D + 2.0 * De));
                EC_ABL(4, EC_TRY(launch_chain(cp, CHAIN_B, st)));
            } else {
                EC_TRY(run_rs_or_tiled(e, PC_GEMM_OTHER, st, o, ld8(D), M, W.outp, 0, EPI_RESID_F32, x, D, x, D, 1.0f));
            }
            snprintf(nm, sizeof(nm), "blocks.%d.x_mhsa", k); trace_add(e, st, nm, x, M, D, D, 0);
        }

        // ---- x = conv_res(x) + ConvModule(x)   (blocks.py:129; modules.py:511-522)
        if (chain_b) {
        } else if (rs_gemm_supported(D)) {
            EC_TRY(run_rs_or_tiled(e, PC_GEMM_OTHER, st, a, ld8(D), M, W.pw1, 2, EPI_GLU_BF16, gbuf, ld8(De), nullptr, 0, 1.f, x, &W.ln_conv));
        } else {
            { PROF(PC_LAYERNORM, 0, (double)M * D * 6); EC_TRY(launch_layernorm(x, M, D, W.ln_conv.g, W.ln_conv.b, nullptr, a, ld8(D), nullptr, nullptr, st)); }
            EC_TRY(run_rs_or_tiled(e, PC_GEMM_OTHER, st, a, ld8(D), M, W.pw1, 2, EPI_GLU_BF16, gbuf, ld8(De)));
        }
        RaggedConv rc{};
        if (rg) { rc.in_off = row_off + (size_t)k * (B + 1); rc.in_len = lens + (size_t)k * B; rc.out_off = row_off + (size_t)(k + 1) * (B + 1);
                  rc.out_len = lens + (size_t)(k + 1) * B; rc.tile_off = tile_off + (size_t)k * (B + 1); rc.tiles = s.tiles[k]; rc.n = B; rc.out_rows = Mo; }
        { PROF(PC_DWCONV, 2.0 * Mo * (double)De * b.kernel_size, (double)M * De * 2 + (double)Mo * De * 2); EC_ABL(8, EC_TRY(launch_dwconv(gbuf, B, T, To, De, ld8(De), W.dw_w, W.dw_b, b.kernel_size, b.conv_stride, cbuf, st, rg ? &rc : nullptr, c.causal, dw_mfma_table(e, W.dw_a, b.kernel_size)))); }
        mask_stride *= b.conv_stride;
        snprintf(nm, sizeof(nm), "blocks.%d.dw", k); trace_add(e, st, nm, cbuf, Mo, De, ld8(De), 1);
        if (D != De) {   // 1x1 strided conv on frames 0, s, 2s, ...  (blocks.py:106-110)
            { PROF(PC_MISC, 0, (double)Mo * D * 6); EC_ABL(64, EC_TRY(launch_cast_rows(x, D, T, b.conv_stride, To, B, xs, ld8(D), st, rg ? &rc : nullptr))); }
            EC_TRY(run_rs_or_tiled(e, PC_GEMM_OTHER, st, xs, ld8(D), Mo, W.res, 1, EPI_F32, xalt, De));
            std::swap(x, xalt);
        } else if (b.conv_stride > 1) {
            return fail("strided block without expansion is not native (no shipped config uses it)");
        }
        const bool last = (k == nb - 1);
        float* xo = (last && !rg) ? out : x;        // ragged: the last block writes its rows in place; emit_rows pads them into `out` below
        if (chain_tail) {
            bool next_head = false;
            if (!last) {
                const EcBlock& nbk = e->blocks[k + 1];
                next_head = W.cc_full && nbk.dim_model <= e->chain_max_dim && e->bw[k + 1].chain_in && chain_full_supported(De, pair_on(e, De, Mo) ? 256 : e->chain_full_max) && (((nbk.group_size * nbk.dim_model / nbk.num_heads) % 2) == 0 || !head_major_odd) && nbk.dim_model == De;
            }
            ChainParams cp{};
            cp.variant = e->chain_variant; cp.count_stores = e->chain_count_stores; cp.small_m = e->chain_small_m; cp.pair = e->chain_pair; cp.pair_small_max = e->chain_pair_min_m - 1; cp.pair_min_d = e->chain_pair_min_d; cp.nt = e->chain_nt; cp.w2cm = e->chain_w2cm;
            cp.M = Mo; cp.D = De; cp.X = x; cp.ldx = De; cp.Y = xo; cp.ldy = De; cp.A = cbuf; cp.lda = ld8(De);
            cp.g0 = ChainGemm{W.c_pw2.w, W.c_pw2.ldw, W.c_pw2.bias, 0};
            cp.ln[0] = ChainLn{W.ln_ffn2.g, W.ln_ffn2.b};
            cp.ln[1] = ChainLn{W.ln_out.g, W.ln_out.b};
            cp.f[0] = ChainFfn{W.c_f2a.w, W.c_f2a.ldw, W.c_f2a.bias, W.c_f2b, W.ffn2_b.ldw, W.c_f2b2, ec_round_up(De * b.ff_ratio, 32), W.c_f2b_cm};
            double fl = 2.0 * Mo * (double)De * (De + 2.0 * De * b.ff_ratio), by = (double)Mo * De * 10 + 2.0 * De * De * (1 + 2.0 * b.ff_ratio);
            if (next_head) {
                const EcBlock& nbk = e->blocks[k + 1];
                const int Gn = nbk.group_size;
                const int Tn = rg ? Mo : s.Tin[k + 1], Tpn = rg ? Mo : ec_round_up(Tn, Gn);
                GemmParams pn{};
                pn.qu = reinterpret_cast<bf16_t*>(ws + w.qu);
                pn.kh = reinterpret_cast<bf16_t*>(ws + w.kh); pn.vt = reinterpret_cast<bf16_t*>(ws + w.vt);
                fill_chain_head(cp, e->bw[k + 1], De, F1c(nbk), Tn, Tpn, pn);
                fl += 2.0 * Mo * (double)De * (2.0 * De * nbk.ff_ratio + 3.0 * De); by += (double)Mo * De * 8 + 2.0 * De * De * (3 + 2.0 * nbk.ff_ratio);
            }
            cp.consts = next_head ? W.cc_full : W.cc_tail;
            { PROF(PC_GEMM_FFN, fl, by); EC_ABL(2, EC_TRY(launch_chain(cp, next_head ? CHAIN_A_FULL : CHAIN_A_TAIL, st))); }
            head_done = next_head;
            have_a = false;
            if (last) { snprintf(nm, sizeof(nm), "blocks.%d.out", k); trace_add(e, st, nm, xo, Mo, De, De, 0); }
            continue;
        }
        head_done = false;
        EC_TRY(run_rs_or_tiled(e, PC_GEMM_OTHER, st, cbuf, ld8(De), Mo, W.pw2, 0, EPI_RESID_F32, x, De, x, De, 1.0f));
        snprintf(nm, sizeof(nm), "blocks.%d.x_conv", k); trace_add(e, st, nm, x, Mo, De, De, 0);

        // ---- x += 1/2 FFN2(x); x = LN(x)   (blocks.py:132-135)
        if (ffn_fused_supported(De) && !prefer_tiled(e, Mo, De * b.ff_ratio, De)) {
            EC_TRY(run_ffn(e, st, a, Mo, De, W.ffn2_a, W.ffn2_b, W.ffn2_bp, x, hbuf, &W.ln_ffn2));
        } else {
            { PROF(PC_LAYERNORM, 0, (double)Mo * De * 6); EC_TRY(launch_layernorm(x, Mo, De, W.ln_ffn2.g, W.ln_ffn2.b, nullptr, a, ld8(De), nullptr, nullptr, st)); }
            EC_TRY(run_ffn(e, st, a, Mo, De, W.ffn2_a, W.ffn2_b, W.ffn2_bp, x, hbuf));
        }
        // block-final norm fused with the next block's FFN1 pre-norm (both read the same rows)
        { PROF(PC_LAYERNORM, 0, (double)Mo * De * 10); EC_TRY(launch_layernorm(x, Mo, De, W.ln_out.g, W.ln_out.b, xo, last ? nullptr : a, ld8(De),
                                last ? nullptr : e->bw[k + 1].ln_ffn1.g, last ? nullptr : e->bw[k + 1].ln_ffn1.b, st)); }
        have_a = !last;
        snprintf(nm, sizeof(nm), "blocks.%d.out", k); trace_add(e, st, nm, xo, Mo, De, De, 0);
    }
    if (!capturing) e->e_cache_put(ws, B, e_tag, w.eh_blk[0]);
    if (rg) {
        const RaggedRows rl = rows_at(nb);
        PROF(PC_MISC, 0, (double)s.Mfinal * e->blocks.back().dim_expand * 4 + (double)B * out_frames * e->blocks.back().dim_expand * 4);
        EC_TRY(launch_emit_rows(x, e->blocks.back().dim_expand, rl.off, rl.len, B, out_frames, out, st));
    }
    return 0;
}

// ------------------------------------------------------------------ fp32-operand "exact" forward (kernels: exact.hip)
struct XWorkspace { size_t total = 0, conv1, sub, x0, x1, a, h, q, k, v, e, o, p1, g, c, lens, scores = 0; size_t qkv_stride = 0;
                    std::vector<size_t> ep_blk;      // sxf.hip forward: the E image of every block (input-independent: kept warm between forwards, as the bf16 path's)
                    size_t kp = 0, vp = 0, ep = 0, xs = 0, xrect = 0, mel_len = 0, row_off = 0, wg_off = 0, tile_off = 0; };     // sxf.hip forward: operand images (bytes / 4), decimated rows, ragged descriptors

// rows come from the Shapes totals: B * T for rectangular batches, the sums over the utterances for ragged ones (s.Tm = the input's row pitch there)
XWorkspace make_xworkspace(const EcEncoder* e, const Shapes& s) {
    XWorkspace w;
    size_t off = 0;
    auto take = [&](size_t floats) { size_t o = off; off += al(floats * 4); return o; };
    const size_t B = s.B;
    size_t mx = 0, mh = 0, mq = 0, me = 0, mp = 0, mg = 0, mc = 0, mkp = 0, mvp = 0, mep = 0, mxs = 0;
    for (size_t k = 0; k < e->blocks.size(); ++k) {
        const EcBlock& b = e->blocks[k];
        const size_t T = s.Tin[k], D = b.dim_model, De = b.dim_expand;
        const size_t Mi = (size_t)s.Min[k], Mo = (size_t)s.Mout[k], Mqk = (size_t)s.Mq[k];
        const size_t Tp = ec_round_up((int)T, b.group_size);
        mx = std::max(mx, std::max(Mi * D, Mo * De));
        mh = std::max(mh, std::max(Mi * D, Mo * De) * b.ff_ratio);
        mq = std::max(mq, Mqk * D);
        me = std::max(me, (2 * Tp - b.group_size) * D);
        {   // operand images of the fused split attention (kernels.h: SxfAttnParams), in floats
            const size_t dh = b.group_size * D / b.num_heads, pk = sxf_attention_pk((int)dh), vx = sxf_attention_vx((int)dh), Tg = Tp / b.group_size;
            mkp = std::max(mkp, (Mqk / b.group_size + 64) * b.num_heads * 2 * pk / 2);
            mvp = std::max(mvp, B * b.num_heads * 2 * vx * (size_t)ec_round_up((int)Tg, 64) / 2);
            mep = std::max(mep, 2 * Tg * b.num_heads * 2 * pk / 2);
        }
        mp = std::max(mp, Mi * 2 * De);
        mg = std::max(mg, Mi * De);
        mc = std::max(mc, Mo * De);
        if (D != De) mxs = std::max(mxs, Mo * D);
    }
    const int L = e->cfg.sub_layers;
    size_t T1r = s.Tm; for (int i = 0; i < L; ++i) T1r = (T1r - 1) / 2 + 1;          // rows per utterance of the rectangular front end (ragged: at the input's pitch)
    const size_t F1 = (e->cfg.n_mels - 1) / 2 + 1, Tl1 = (s.Tm - 1) / 2 + 1;
    w.conv1 = take(L == 2 ? B * e->cfg.sub_filters[0] * F1 * Tl1 : 0);
    int F = e->cfg.n_mels; for (int i = 0; i < L; ++i) F = (F - 1) / 2 + 1;
    w.sub = take(B * T1r * (size_t)e->cfg.sub_filters[L - 1] * F);
    w.x0 = take(mx); w.x1 = take(mx); w.a = take(mx); w.h = take(mh);
    w.q = take(mq); w.k = take(mq); w.v = take(mq); w.e = take(me); w.o = take(mq);
    w.p1 = take(mp); w.g = take(mg); w.c = take(mc);
    w.lens = take((e->blocks.size() + 1) * B);
    w.qkv_stride = (w.k - w.q) / 4;                 // floats between the Q, K and V buffers (the stacked projection writes all three)
    if (e->exact_split) {                           // split.hip: (B, H, Tg, Tg) attention scores of one block (rectangular batches with attention maps)
        size_t ms = 0;
        if (!s.ragged)
            for (size_t k = 0; k < e->blocks.size(); ++k) {
                const EcBlock& b = e->blocks[k];
                const int Tg = ec_round_up(s.Tin[k], b.group_size) / b.group_size;
                ms = std::max(ms, sx_attention_scores_bytes(s.B, b.num_heads, Tg) / 4);
            }
        w.scores = take(ms);
        w.kp = take(mkp); w.vp = take(mvp); w.ep = take(mep); w.xs = take(mxs);
        for (size_t k = 0; k < e->blocks.size(); ++k) {
            const EcBlock& b = e->blocks[k];
            const size_t dh = b.group_size * b.dim_model / b.num_heads, pk = sxf_attention_pk((int)dh), Tg = ec_round_up(s.Tin[k], b.group_size) / b.group_size;
            w.ep_blk.push_back(take(2 * Tg * b.num_heads * 2 * pk / 2));
        }
        if (s.ragged) {
            const size_t nbk = e->blocks.size();
            w.xrect = take(B * T1r * e->blocks[0].dim_model);
            w.mel_len = take(B);
            w.row_off = take((nbk + 1) * (B + 1));
            w.wg_off = take(nbk * (B + 1));
            w.tile_off = take(nbk * (B + 1));
        }
    }
    w.total = off;
    return w;
}

const float* xget(EcEncoder* e, const std::string& k) {
    auto it = e->xw.find(k);
    return it == e->xw.end() ? nullptr : it->second;
}

int xgemm(EcEncoder* e, hipStream_t st, const float* A, int lda, int M, const std::string& prefix, int N, int K, float* C, int ldc, int epi = 0,
          const float* R = nullptr, float alpha = 1.f, int a_rows = 0, int a_pitch = 0, int a_stride = 0, int c_rows = 0, int c_pitch = 0,
          int split_cols = 0, size_t split_stride = 0, int cls = PC_GEMM_OTHER) {
    ExGemmParams p{};
    p.A = A; p.lda = lda; p.a_rows = a_rows; p.a_pitch = a_pitch; p.a_stride = a_stride;
    p.W = xget(e, prefix + ".weight"); p.ldw = K; p.bias = xget(e, prefix + ".bias");
    if (!p.bias) return fail("exact mode: missing " + prefix);
    p.M = M; p.N = N; p.K = K; p.C = C; p.ldc = ldc; p.c_rows = c_rows; p.c_pitch = c_pitch;
    p.split_cols = split_cols; p.split_stride = split_stride;
    p.R = R; p.ldr = ldc; p.alpha = alpha; p.epi = epi;
    // flop: the algorithmic 2 M N K (the split kernels issue three MFMAs per product)
    PROF(cls, 2.0 * M * (double)N * K, 4.0 * ((double)M * K + (double)N * K + (double)M * N));
    if (e->exact_split) {
        auto it = e->xsplit.find(prefix);
        if (it != e->xsplit.end()) {
            SxGemmParams q{};
            q.g = p; q.Whi = it->second.hi; q.Wlo = it->second.lo; q.ldh = it->second.ldh;
            const int rc = launch_sx_gemm(q, st);
            // -2 = a shape the split kernel does not take (N, lda or ldc not a multiple of 4, rows * lda >= 2^32 elements, more than 65535 column tiles):
            // the fp32-MFMA kernel handles those, and its weights are uploaded in split mode too (advisor, round 4)
            if (rc != -2 || !p.W) return rc;
        }
    }
    if (!p.W) return fail("exact mode: missing " + prefix);
    return launch_ex_gemm(p, st);
}

int forward_core_exact(EcEncoder* e, const float* mel, const int64_t* in_len, int from_audio, const Shapes& s, const XWorkspace& w,
                       char* ws, float* out, int64_t* out_len, hipStream_t st) {
    const EcConfig& c = e->cfg;
    const int B = s.B, nb = (int)e->blocks.size();
    // finite left / right contexts: the band mask is part of the label-exact attention kernels since round 4; `causal` (causal relative tables,
    // causal depthwise padding: attentions.py:506, 1243-1247; layers.py:97-101) stays on the bf16 path
    if (c.causal) return fail("the label-exact modes have no causal kernels (causal relative tables / depthwise pre-padding): bf16 path only");
    e->trace.clear(); e->trace_used = 0;
    // this forward lays its own buffers over the caller's workspace: a positional-embedding cache the bf16 path left there is gone
    // (fp32 -> bf16 -> fp32 -> bf16 on one workspace otherwise ends with attention reading fp32 activations as E)
    e->e_cache_drop(ws);
    auto F32 = [&](size_t off) { return reinterpret_cast<float*>(ws + off); };
    int* lens = reinterpret_cast<int*>(ws + w.lens);
    EC_TRY(launch_lengths(in_len, B, from_audio, c.hop_length, c.sub_layers, e->block_stride, nb, lens, out_len, st));
    if (from_audio) trace_add(e, st, "mel", mel, (int64_t)B * c.n_mels, s.Tm, s.Tm, 0);
    // ---- Conv2dSubsampling (modules.py:232-249) + transpose + Linear (encoders.py:113-116)
    float* sub = F32(w.sub);
    const int C0 = c.sub_filters[0];
    int Fl = c.n_mels, Cl = C0;
    if (c.sub_layers == 1) {
        EC_TRY(launch_ex_conv2d(mel, B, 1, c.n_mels, s.Tm, xget(e, "subsampling_module.layers.0.0.weight"), e->xsub_scale[0], e->xsub_shift[0], C0, sub, 1, st));
        Fl = (c.n_mels - 1) / 2 + 1;
    } else {
        float* img = F32(w.conv1);
        const int F1 = (c.n_mels - 1) / 2 + 1, T1 = (s.Tm - 1) / 2 + 1, C1 = c.sub_filters[1];
        EC_TRY(launch_ex_conv2d(mel, B, 1, c.n_mels, s.Tm, xget(e, "subsampling_module.layers.0.0.weight"), e->xsub_scale[0], e->xsub_shift[0], C0, img, 0, st));
        EC_TRY(launch_ex_conv2d(img, B, C0, F1, T1, xget(e, "subsampling_module.layers.1.0.weight"), e->xsub_scale[1], e->xsub_shift[1], C1, sub, 1, st));
        Fl = (F1 - 1) / 2 + 1; Cl = C1;
    }
    const int Ksub = Cl * Fl;
    trace_add(e, st, "subsample", sub, (int64_t)B * s.T1, Ksub, Ksub, 0);
    float* x = F32(w.x0);
    float* xalt = F32(w.x1);
    const int D0 = e->blocks[0].dim_model;
    EC_TRY(xgemm(e, st, sub, Ksub, B * s.T1, "linear", D0, Ksub, x, D0));
    trace_add(e, st, "linear", x, (int64_t)B * s.T1, D0, D0, 0);
    float *a = F32(w.a), *hb = F32(w.h), *q = F32(w.q), *kk = F32(w.k), *v = F32(w.v), *eb = F32(w.e), *o = F32(w.o), *p1 = F32(w.p1), *g = F32(w.g),
          *cb = F32(w.c);
    char nm[64];
    int xmask_stride = 1;                      // product of the strides of the blocks before block k
    for (int k = 0; k < nb; ++k) {
        const EcBlock& b = e->blocks[k];
        const BlockW& W = e->bw[k];
        const int T = s.Tin[k], To = s.Tout[k], D = b.dim_model, De = b.dim_expand, M = B * T, Mo = B * To;
        const int G = b.group_size, H = b.num_heads, Tp = ec_round_up(T, G), Tg = Tp / G, d = G * D / H;
        const std::string p = "blocks." + std::to_string(k);
        // ---- x += 1/2 FFN1(LN(x))   (blocks.py:122; modules.py:385-392)
        EC_TRY(launch_layernorm(x, M, D, W.ln_ffn1.g, W.ln_ffn1.b, a, nullptr, 0, nullptr, nullptr, st));
        EC_TRY(xgemm(e, st, a, D, M, p + ".feed_forward_module1.layers.1", D * b.ff_ratio, D, hb, D * b.ff_ratio, 1, nullptr, 1.f, 0, 0, 0, 0, 0, 0, 0, PC_GEMM_FFN));
        EC_TRY(xgemm(e, st, hb, D * b.ff_ratio, M, p + ".feed_forward_module1.layers.4", D, D * b.ff_ratio, x, D, 2, x, 0.5f, 0, 0, 0, 0, 0, 0, 0, PC_GEMM_FFN));
        snprintf(nm, sizeof(nm), "blocks.%d.x_ffn1", k); trace_add(e, st, nm, x, M, D, D, 0);
        // ---- x += MHSA(LN(x))   (blocks.py:125-126; attentions.py:549-718)
        const std::string m = p + ".multi_head_self_attention_module";
        EC_TRY(launch_layernorm(x, M, D, W.ln_att.g, W.ln_att.b, a, nullptr, 0, nullptr, nullptr, st));
        if (Tp != T) {      // chunk padding: zero rows AFTER the projections (attentions.py:107-138, 671)
            if (hipMemsetAsync(q, 0, (size_t)B * Tp * D * 4, st) != hipSuccess || hipMemsetAsync(kk, 0, (size_t)B * Tp * D * 4, st) != hipSuccess ||
                hipMemsetAsync(v, 0, (size_t)B * Tp * D * 4, st) != hipSuccess) return fail("memset failed");
        }
        if (e->exact_split && e->xsplit.count(m + ".mhsa.qkv_layer")) {      // one stacked projection: column n -> buffer n / D (q | k | v), column n % D
            EC_TRY(xgemm(e, st, a, D, M, m + ".mhsa.qkv_layer", 3 * D, D, q, D, 0, nullptr, 1.f, 0, 0, 0, T, Tp, D, w.qkv_stride));
        } else {
            EC_TRY(xgemm(e, st, a, D, M, m + ".mhsa.query_layer", D, D, q, D, 0, nullptr, 1.f, 0, 0, 0, T, Tp));
            EC_TRY(xgemm(e, st, a, D, M, m + ".mhsa.key_layer", D, D, kk, D, 0, nullptr, 1.f, 0, 0, 0, T, Tp));
            EC_TRY(xgemm(e, st, a, D, M, m + ".mhsa.value_layer", D, D, v, D, 0, nullptr, 1.f, 0, 0, 0, T, Tp));
        }
        if (Tp > b.max_pos) return fail("sequence longer than max_pos_encoding");
        const float* tab = e->xtab[std::make_pair(b.max_pos, D)];
        EC_TRY(xgemm(e, st, tab + (size_t)(b.max_pos - Tp + G / 2) * D, D, 2 * Tp - G, m + ".mhsa.pos_layer", D, D, eb, D));
        ExAttnParams ap{};
        ap.q = q; ap.k = kk; ap.v = v; ap.e = eb; ap.u = W.u; ap.vb = W.v; ap.lens = lens + (size_t)k * B;
        ap.B = B; ap.H = H; ap.T = T; ap.Tp = Tp; ap.G = G; ap.D = D; ap.d = d; ap.Tg = Tg; ap.out = o; ap.variant = e->exact_attention;
        ap.att = (int)e->att_out.size() == nb ? e->att_out[k] : nullptr;
        {   // streaming mask of this block (encoders.py:132-136, attentions.py:698): contexts in frames after the subsampling, sliced ::stride after
            // every strided block before this one and ::G in grouped attention
            const long long unit = (long long)xmask_stride * G;
            ap.band_l = (int)std::min<long long>(c.left_context / unit, 1 << 30); ap.band_r = (int)std::min<long long>(c.right_context / unit, 1 << 30);
        }
        { PROF(PC_ATTENTION, 2.0 * B * H * (double)Tg * Tg * d * 3.0, (double)M * D * 4 * 5);
          if (e->exact_split && sx_attention_supported(d) && (long long)B * H <= 65535) EC_TRY(launch_sx_attention(ap, F32(w.scores), st));
          else EC_TRY(launch_ex_attention(ap, st)); }
        EC_TRY(xgemm(e, st, o, D, M, m + ".mhsa.output_layer", D, D, x, D, 2, x, 1.0f, T, Tp, 1));
        snprintf(nm, sizeof(nm), "blocks.%d.x_mhsa", k); trace_add(e, st, nm, x, M, D, D, 0);
        // ---- x = conv_res(x) + ConvModule(x)   (blocks.py:129; modules.py:511-522)
        const std::string cm = p + ".convolution_module.layers";
        EC_TRY(launch_layernorm(x, M, D, W.ln_conv.g, W.ln_conv.b, a, nullptr, 0, nullptr, nullptr, st));
        EC_TRY(xgemm(e, st, a, D, M, cm + ".2", 2 * De, D, p1, 2 * De));
        EC_TRY(launch_ex_glu(p1, M, De, g, st));
        EC_TRY(launch_ex_dwconv(g, B, T, To, De, W.dw_w, W.dw_b, b.kernel_size, b.conv_stride, cb, st));
        xmask_stride *= b.conv_stride;
        if (D != De) {      // 1x1 strided conv on frames 0, s, 2s, ...  (blocks.py:106-110)
            EC_TRY(xgemm(e, st, x, D, Mo, p + ".conv_res.1", De, D, xalt, De, 0, nullptr, 1.f, To, T, b.conv_stride));
            std::swap(x, xalt);
        } else if (b.conv_stride > 1) {
            return fail("strided block without expansion is not native (no shipped config uses it)");
        }
        EC_TRY(xgemm(e, st, cb, De, Mo, cm + ".7", De, De, x, De, 2, x, 1.0f));
        snprintf(nm, sizeof(nm), "blocks.%d.x_conv", k); trace_add(e, st, nm, x, Mo, De, De, 0);
        // ---- x += 1/2 FFN2(LN(x)); x = LN(x)   (blocks.py:132-135)
        EC_TRY(launch_layernorm(x, Mo, De, W.ln_ffn2.g, W.ln_ffn2.b, a, nullptr, 0, nullptr, nullptr, st));
        EC_TRY(xgemm(e, st, a, De, Mo, p + ".feed_forward_module2.layers.1", De * b.ff_ratio, De, hb, De * b.ff_ratio, 1, nullptr, 1.f, 0, 0, 0, 0, 0, 0, 0, PC_GEMM_FFN));
        EC_TRY(xgemm(e, st, hb, De * b.ff_ratio, Mo, p + ".feed_forward_module2.layers.4", De, De * b.ff_ratio, x, De, 2, x, 0.5f, 0, 0, 0, 0, 0, 0, 0, PC_GEMM_FFN));
        float* xo = (k == nb - 1) ? out : xalt;
        EC_TRY(launch_layernorm(x, Mo, De, W.ln_out.g, W.ln_out.b, xo, nullptr, 0, nullptr, nullptr, st));
        if (k != nb - 1) std::swap(x, xalt);
        snprintf(nm, sizeof(nm), "blocks.%d.out", k); trace_add(e, st, nm, xo, Mo, De, De, 0);
    }
    return 0;
}


// ------------------------------------------------------------------ split-precision forward on the fused kernels of sxf.hip (round 6)
// The schedule of forward_core_exact with (i) ONE attention kernel per block that keeps the scores on the CU, (ii) ragged batches - every utterance at its own
// length in the concatenated, group-padded row space of the bf16 path (the row-local GEMMs / LayerNorms see M rows; attention, depthwise conv and the conv_res
// decimation index utterances through the descriptors of lengths_ragged_kernel), (iii) causal relative tables / causal depthwise padding and streaming
// contexts.  Reference: encoders.py:97-142, blocks.py:119-137, attentions.py:506-529, 549-718, 1243-1247, layers.py:97-101.
bool split_fused_ok(const EcEncoder* e) {
    if (!e->exact_split) return false;
    for (const EcBlock& b : e->blocks)
        if (!sxf_attention_supported(b.group_size * b.dim_model / b.num_heads) || b.dim_model % 4 || b.dim_expand % 4) return false;
    return true;
}

int forward_core_split(EcEncoder* e, const float* mel, const int64_t* in_len, int from_audio, const Shapes& s, const XWorkspace& w,
                       char* ws, float* out, int64_t* out_len, hipStream_t st, int out_frames = 0) {
    const EcConfig& c = e->cfg;
    const int B = s.B, nb = (int)e->blocks.size();
    const bool rg = s.ragged;
    e->trace.clear(); e->trace_used = 0;
    // E = pos_layer(R) and its operand image depend on the block and the frame count entering it only: with the caller's workspace left untouched between
    // forwards (option cache_pos_embeddings, as on the bf16 path) the 2 x blocks small launches are skipped for an unchanged shape
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(st, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone;
    const int e_tag = rg ? -(s.Tin[0] + 1) : s.Tm;
    const size_t e_layout = w.ep_blk.empty() ? 0 : w.ep_blk[0] ^ ((size_t)1 << 62) ^ ((size_t)c.causal << 61);      // never equal to a bf16 forward's tag on the same workspace
    const bool e_cached = !capturing && e->e_cache_on && !e->trace_arena && e->e_cache_hit(ws, B, e_tag, e_layout);
    if (!e_cached) e->e_cache_drop(ws);
    auto F32 = [&](size_t off) { return reinterpret_cast<float*>(ws + off); };
    int* lens = reinterpret_cast<int*>(ws + w.lens);
    const int *mel_len = nullptr, *row_off = nullptr;
    if (rg) {
        int* ml = reinterpret_cast<int*>(ws + w.mel_len); int* ro = reinterpret_cast<int*>(ws + w.row_off);
        PROF(PC_MISC, 0, 0);
        EC_TRY(launch_lengths_ragged(in_len, B, from_audio, c.hop_length, c.sub_layers, e->block_stride, e->block_group, e->block_heads, nb, lens, ml,
                                     ro, reinterpret_cast<int*>(ws + w.wg_off), reinterpret_cast<int*>(ws + w.tile_off), out_len, st));
        mel_len = ml; row_off = ro;
    } else {
        PROF(PC_MISC, 0, 0);
        EC_TRY(launch_lengths(in_len, B, from_audio, c.hop_length, c.sub_layers, e->block_stride, nb, lens, out_len, st));
    }
    if (from_audio) trace_add(e, st, "mel", mel, (int64_t)B * c.n_mels, s.Tm, s.Tm, 0);
    auto rows_at = [&](int k) { RaggedRows r{}; r.off = row_off + (size_t)k * (B + 1); r.len = lens + (size_t)k * B; r.n = B;
                                r.rows = (int)(k < nb ? s.Min[k] : s.Mfinal); r.tmax = k < nb ? s.Tin[k] : s.Tout[nb - 1]; return r; };
    // ---- Conv2dSubsampling (modules.py:232-249) + transpose + Linear (encoders.py:113-116).  Ragged: on the rectangular image at the input's pitch with every
    //      utterance's frames behind its own end read / written as zeros (the zero padding it sees when run alone), then the valid rows are gathered
    float* sub = F32(w.sub);
    const int C0 = c.sub_filters[0];
    int Fl = c.n_mels, Cl = C0;
    int T1r = s.Tm; for (int i = 0; i < c.sub_layers; ++i) T1r = (T1r - 1) / 2 + 1;
    // one-layer subsampler: convolution + Swish + Linear as ONE kernel on the frames that exist (sxf_sub.hip) - the (frames, C F') activation stays in registers.
    // A debug trace wants that activation ("subsample"): per-module kernels then
    const bool sublin = e->split_sublin && e->xsub_wimg && c.sub_layers == 1 && !e->trace_arena;
    if (sublin) {
        const int D0 = e->blocks[0].dim_model;
        SxfSubParams sp{};
        sp.mel = mel; sp.B = B; sp.F = c.n_mels; sp.Tm = s.Tm; sp.mel_len = mel_len;
        if (rg) { const RaggedRows r0 = rows_at(0); sp.off = r0.off; sp.len = r0.len; sp.rows_max = ec_round_up(s.Tin[0], e->blocks[0].group_size); }
        else { sp.To = T1r; sp.rows_max = T1r; }
        sp.cimg = e->xsub_cimg; sp.wimg = e->xsub_wimg; sp.bias = e->xsub_bias; sp.y = F32(w.x0); sp.N = D0; sp.ncb = e->xsub_ncb; sp.Fo = e->xsub_fo;
        const double Mr = (double)s.Min[0], Ks = (double)C0 * e->xsub_fo;
        PROF(PC_SUBCONV, 2.0 * Mr * Ks * (9.0 + D0), (double)B * c.n_mels * s.Tm * 4 + Mr * D0 * 4);
        EC_TRY(launch_sxf_sublin(sp, st));
    } else {
        PROF(PC_SUBCONV, 0, (double)B * c.n_mels * s.Tm * 4);
        if (c.sub_layers == 1) {
            EC_TRY(launch_ex_conv2d(mel, B, 1, c.n_mels, s.Tm, xget(e, "subsampling_module.layers.0.0.weight"), e->xsub_scale[0], e->xsub_shift[0], C0, sub, 1, st, mel_len));
            Fl = (c.n_mels - 1) / 2 + 1;
        } else {
            float* img = F32(w.conv1);
            const int F1 = (c.n_mels - 1) / 2 + 1, Tl1 = (s.Tm - 1) / 2 + 1, C1 = c.sub_filters[1];
            EC_TRY(launch_ex_conv2d(mel, B, 1, c.n_mels, s.Tm, xget(e, "subsampling_module.layers.0.0.weight"), e->xsub_scale[0], e->xsub_shift[0], C0, img, 0, st, mel_len));
            EC_TRY(launch_ex_conv2d(img, B, C0, F1, Tl1, xget(e, "subsampling_module.layers.1.0.weight"), e->xsub_scale[1], e->xsub_shift[1], C1, sub, 1, st));
            Fl = (F1 - 1) / 2 + 1; Cl = C1;
        }
    }
    const int Ksub = Cl * Fl;
    float* x = F32(w.x0);
    float* xalt = F32(w.x1);
    const int D0 = e->blocks[0].dim_model;
    if (sublin) {
        // x holds the rows already
    } else if (rg) {
        float* xrect = F32(w.xrect);
        EC_TRY(xgemm(e, st, sub, Ksub, B * T1r, "linear", D0, Ksub, xrect, D0));
        PROF(PC_MISC, 0, (double)s.Min[0] * D0 * 8);
        EC_TRY(launch_gather_rows(xrect, D0, T1r, rows_at(0), x, st));
    } else {
        trace_add(e, st, "subsample", sub, (int64_t)B * s.T1, Ksub, Ksub, 0);
        EC_TRY(xgemm(e, st, sub, Ksub, B * s.T1, "linear", D0, Ksub, x, D0));
    }
    trace_add(e, st, "linear", x, s.Min[0], D0, D0, 0);
    float *a = F32(w.a), *hb = F32(w.h), *q = F32(w.q), *kk = F32(w.k), *v = F32(w.v), *eb = F32(w.e), *o = F32(w.o), *p1 = F32(w.p1), *g = F32(w.g),
          *cbuf = F32(w.c), *xs = F32(w.xs);
    uint16_t *kpk = reinterpret_cast<uint16_t*>(ws + w.kp), *vpk = reinterpret_cast<uint16_t*>(ws + w.vp);
    char nm[64];
    int xmask_stride = 1;                      // product of the strides of the blocks before block k
    auto layernorm = [&](const float* in, int rows, int dim, const LNp& ln, float* dst) {
        PROF(PC_LAYERNORM, 0, (double)rows * dim * 8);
        return launch_layernorm(in, rows, dim, ln.g, ln.b, dst, nullptr, 0, nullptr, nullptr, st);
    };
    // the row-local work between attention and the depthwise convolution as two kernels per block (sxf_chain.hip); a debug trace wants the intermediate
    // states, which the chains never write: per-module kernels then
    const bool chains = e->split_chain && e->split_ffn && !e->trace_arena;
    bool head_done = false;                    // this block's FFN1 + Q / K / V projections ran at the end of the previous block's chain A
    for (int k = 0; k < nb; ++k) {
        const EcBlock& b = e->blocks[k];
        const BlockW& W = e->bw[k];
        const int T = s.Tin[k], To = s.Tout[k], D = b.dim_model, De = b.dim_expand;          // ragged: the LONGEST utterance's frames
        const int M = (int)s.Min[k], Mo = (int)s.Mout[k];
        const int G = b.group_size, H = b.num_heads, Tp = ec_round_up(T, G), Tg = Tp / G, d = G * D / H;
        const std::string p = "blocks." + std::to_string(k);
        const std::string m = p + ".multi_head_self_attention_module";
        const int qr = rg ? 0 : T, qp = rg ? 0 : Tp;
        const bool chain_in = chains && W.xc_in, chain_out = chains && W.xc_out;
        if (head_done) {
            // nothing: x is the stream after FFN1, Q / K / V are written
        } else if (chain_in) {
            SxcAParams cp{};
            cp.head = 1; cp.y = x; cp.M = M; cp.D = D; cp.w_f1 = W.xc_f[0]; cp.nch_f1 = W.xf_nch[0]; cp.b_f1 = W.xf_b2[0]; cp.w_qkv = W.xc_qkv;
            cp.q = q; cp.qkv_stride = w.qkv_stride; cp.q_rows = qr; cp.q_pitch = qp; cp.qkv_bytes = (2 * w.qkv_stride + (size_t)s.Mq[k] * D) * 4;
            PROF(PC_GEMM_FFN, M * (double)D * D * (4.0 * b.ff_ratio + 6.0), (double)M * D * 24 + D * (double)D * (16.0 * b.ff_ratio + 12.0));
            EC_TRY(launch_sxc_a(cp, st));
        } else {
        // ---- x += 1/2 FFN1(LN(x))   (blocks.py:122; modules.py:385-392): one kernel where the width is built (sxf_ffn.hip), else LayerNorm + two GEMMs
        if (W.xf_img[0] && e->split_ffn) {
            SxfFfnParams fp{};
            fp.X = x; fp.ldx = D; fp.Y = x; fp.ldy = D; fp.wimg = W.xf_img[0]; fp.b2 = W.xf_b2[0]; fp.M = M; fp.D = D; fp.nchunk = W.xf_nch[0];
            PROF(PC_GEMM_FFN, 4.0 * M * (double)D * D * b.ff_ratio, (double)M * D * 8 + 16.0 * D * D * b.ff_ratio);
            EC_TRY(launch_sxf_ffn(fp, st));
        } else {
            EC_TRY(layernorm(x, M, D, W.ln_ffn1, a));
            EC_TRY(xgemm(e, st, a, D, M, p + ".feed_forward_module1.layers.1", D * b.ff_ratio, D, hb, D * b.ff_ratio, 1, nullptr, 1.f, 0, 0, 0, 0, 0, 0, 0, PC_GEMM_FFN));
            EC_TRY(xgemm(e, st, hb, D * b.ff_ratio, M, p + ".feed_forward_module1.layers.4", D, D * b.ff_ratio, x, D, 2, x, 0.5f, 0, 0, 0, 0, 0, 0, 0, PC_GEMM_FFN));
        }
        snprintf(nm, sizeof(nm), "blocks.%d.x_ffn1", k); trace_add(e, st, nm, x, M, D, D, 0);
        // ---- x += MHSA(LN(x))   (blocks.py:125-126; attentions.py:549-718).  Rows of Q / K / V: rectangular (b, t) -> b Tp + t, ragged: the identity (the
        //      residual stream keeps every utterance group-padded); chunk-padding rows are never written - the attention kernel substitutes them
        EC_TRY(layernorm(x, M, D, W.ln_att, a));
        if (e->xsplit.count(m + ".mhsa.qkv_layer")) {
            EC_TRY(xgemm(e, st, a, D, M, m + ".mhsa.qkv_layer", 3 * D, D, q, D, 0, nullptr, 1.f, 0, 0, 0, qr, qp, D, w.qkv_stride));
        } else {
            EC_TRY(xgemm(e, st, a, D, M, m + ".mhsa.query_layer", D, D, q, D, 0, nullptr, 1.f, 0, 0, 0, qr, qp));
            EC_TRY(xgemm(e, st, a, D, M, m + ".mhsa.key_layer", D, D, kk, D, 0, nullptr, 1.f, 0, 0, 0, qr, qp));
            EC_TRY(xgemm(e, st, a, D, M, m + ".mhsa.value_layer", D, D, v, D, 0, nullptr, 1.f, 0, 0, 0, qr, qp));
        }
        }
        head_done = false;
        if (e->trace_arena) {       // rows (b, t) -> b Tp + t of the projections (chunk-padding rows are never written: whatever the workspace held)
            snprintf(nm, sizeof(nm), "blocks.%d.q", k); trace_add(e, st, nm, q, s.Mq[k], D, D, 0);
            snprintf(nm, sizeof(nm), "blocks.%d.k", k); trace_add(e, st, nm, kk, s.Mq[k], D, D, 0);
            snprintf(nm, sizeof(nm), "blocks.%d.v", k); trace_add(e, st, nm, v, s.Mq[k], D, D, 0);
        }
        if (Tp > b.max_pos) return fail("sequence longer than max_pos_encoding");
        // relative tables: R[m] = sinusoid(Tp - 1 - G/2 - m), m < 2 Tp - G; causal: R[m] = sinusoid(Tp - 1 - m), m < Tp (attentions.py:1243-1251, 1296-1309)
        const float* tab = e->xtab[std::make_pair(b.max_pos, D)];
        const int erows = c.causal ? Tp : 2 * Tp - G;
        uint16_t* epk = reinterpret_cast<uint16_t*>(ws + w.ep_blk[k]);
        if (!e_cached) {
            EC_TRY(xgemm(e, st, tab + (size_t)(b.max_pos - Tp + (c.causal ? 0 : G / 2)) * D, D, erows, m + ".mhsa.pos_layer", D, D, eb, D));
            PROF(PC_MISC, 0, (double)erows * D * 8); EC_TRY(launch_sxf_pack_e(eb, W.u, W.v, erows / G, H, G, D, d, epk, st));
        }
        SxfAttnParams ap{};
        ap.q = q; ap.k = kk; ap.v = v; ap.kp = kpk; ap.vp = vpk; ap.ep = epk; ap.vpitch = ec_round_up(Tg, 64); ap.u = W.u; ap.lens = lens + (size_t)k * B;
        ap.off = rg ? row_off + (size_t)k * (B + 1) : nullptr;
        ap.B = B; ap.H = H; ap.G = G; ap.D = D; ap.d = d; ap.T = T; ap.Tp = Tp; ap.Tg = Tg; ap.out = o; ap.causal = c.causal;
        {   // streaming mask of this block (encoders.py:132-136, attentions.py:698): contexts in frames after the subsampling, sliced ::stride after every
            // strided block before this one and ::G in grouped attention
            const long long unit = (long long)xmask_stride * G;
            ap.band_l = (int)std::min<long long>(c.left_context / unit, 1 << 30); ap.band_r = (int)std::min<long long>(c.right_context / unit, 1 << 30);
        }
        { PROF(PC_MISC, 0, (double)s.Mq[k] * D * 16); EC_TRY(launch_sxf_pack_kv(ap, st)); }
        { PROF(PC_ATTENTION, 2.0 * H * (rg ? s.tg2[k] : (double)B * Tg * Tg) * d * 3.0, (double)s.Mq[k] * D * 4 * 4);
          EC_TRY(launch_sxf_attention(ap, st)); }
        snprintf(nm, sizeof(nm), "blocks.%d.att_o", k); trace_add(e, st, nm, o, s.Mq[k], D, D, 0);
        const std::string cm = p + ".convolution_module.layers";
        if (chain_in) {       // x += O Wo^T + bo;  g = GLU(LN(x) Wp1^T + bp1)
            SxcBParams cp{};
            cp.o = o; cp.o_rows = qr; cp.o_pitch = qp; cp.x = x; cp.g = g; cp.M = M; cp.D = D; cp.De = De;
            cp.w_o = W.xc_wo; cp.b_o = W.xc_bo; cp.w_p1 = W.xc_p1; cp.nch_p1 = W.xc_nch_p1;
            PROF(PC_GEMM_OTHER, 2.0 * M * D * ((double)D + 2.0 * De), (double)M * (D * 12.0 + De * 4.0) + 4.0 * D * ((double)D + 2.0 * De));
            EC_TRY(launch_sxc_b(cp, st));
        } else {
        EC_TRY(xgemm(e, st, o, D, M, m + ".mhsa.output_layer", D, D, x, D, 2, x, 1.0f, qr, qp, 1));
        snprintf(nm, sizeof(nm), "blocks.%d.x_mhsa", k); trace_add(e, st, nm, x, M, D, D, 0);
        // ---- x = conv_res(x) + ConvModule(x)   (blocks.py:129; modules.py:511-522)
        EC_TRY(layernorm(x, M, D, W.ln_conv, a));
        EC_TRY(xgemm(e, st, a, D, M, cm + ".2", 2 * De, D, p1, 2 * De));
        { PROF(PC_MISC, 0, (double)M * De * 12); EC_TRY(launch_sxf_glu(p1, M, De, g, st)); }
        }
        RaggedConv rc{};
        int tcap = To;
        if (rg) { rc.in_off = row_off + (size_t)k * (B + 1); rc.in_len = lens + (size_t)k * B; rc.out_off = row_off + (size_t)(k + 1) * (B + 1);
                  rc.out_len = lens + (size_t)(k + 1) * B; rc.n = B; rc.out_rows = Mo;
                  tcap = ec_round_up(To, k + 1 < nb ? e->blocks[k + 1].group_size : 1); }
        { PROF(PC_DWCONV, 2.0 * Mo * (double)De * b.kernel_size, (double)M * De * 4 + (double)Mo * De * 4);
          EC_TRY(launch_sxf_dwconv(g, B, T, To, De, W.dw_w, W.dw_b, b.kernel_size, b.conv_stride, cbuf, st, rg ? &rc : nullptr, c.causal, tcap)); }
        snprintf(nm, sizeof(nm), "blocks.%d.dw", k); trace_add(e, st, nm, cbuf, Mo, De, De, 0);
        xmask_stride *= b.conv_stride;
        if (D != De) {      // 1x1 strided conv on frames 0, s, 2s, ...  (blocks.py:106-110)
            if (rg) {
                { PROF(PC_MISC, 0, (double)Mo * D * 8); EC_TRY(launch_sxf_decimate(x, D, b.conv_stride, rc, xs, st)); }
                EC_TRY(xgemm(e, st, xs, D, Mo, p + ".conv_res.1", De, D, xalt, De));
            } else {
                EC_TRY(xgemm(e, st, x, D, Mo, p + ".conv_res.1", De, D, xalt, De, 0, nullptr, 1.f, To, T, b.conv_stride));
            }
            std::swap(x, xalt);
        } else if (b.conv_stride > 1) {
            return fail("strided block without expansion is not native (no shipped config uses it)");
        }
        float* xo = (k == nb - 1 && !rg) ? out : xalt;
        if (chain_out) {      // x = xres + C Wp2^T + bp2;  x += 1/2 FFN2(LN(x));  xo = LN(x);  [next block: xo += 1/2 FFN1(LN(xo)); Q | K | V]
            SxcAParams cp{};
            cp.tail = 1; cp.c = cbuf; cp.xres = x; cp.y = xo; cp.M = Mo; cp.D = De;
            cp.w_p2 = W.xc_p2; cp.b_p2 = W.xc_bp2; cp.w_f2 = W.xc_f[1]; cp.nch_f2 = W.xf_nch[1]; cp.b_f2 = W.xf_b2[1]; cp.ln_g = W.ln_out.g; cp.ln_b = W.ln_out.b;
            double fl = Mo * (double)De * De * (2.0 + 4.0 * b.ff_ratio), by = (double)Mo * De * 16 + De * (double)De * (4.0 + 16.0 * b.ff_ratio);
            if (k + 1 < nb && e->bw[k + 1].xc_in && e->blocks[k + 1].dim_model == De && (int)s.Min[k + 1] == Mo) {
                const EcBlock& bn = e->blocks[k + 1];
                const BlockW& Wn = e->bw[k + 1];
                const int Tn = s.Tin[k + 1];
                cp.head = 1; cp.w_f1 = Wn.xc_f[0]; cp.nch_f1 = Wn.xf_nch[0]; cp.b_f1 = Wn.xf_b2[0]; cp.w_qkv = Wn.xc_qkv;
                cp.q = q; cp.qkv_stride = w.qkv_stride; cp.q_rows = rg ? 0 : Tn; cp.q_pitch = rg ? 0 : ec_round_up(Tn, bn.group_size); cp.qkv_bytes = (2 * w.qkv_stride + (size_t)s.Mq[k + 1] * De) * 4;
                fl += Mo * (double)De * De * (4.0 * bn.ff_ratio + 6.0); by += (double)Mo * De * 20 + De * (double)De * (16.0 * bn.ff_ratio + 12.0);
                head_done = true;
            }
            PROF(PC_GEMM_FFN, fl, by);
            EC_TRY(launch_sxc_a(cp, st));
        } else {
        EC_TRY(xgemm(e, st, cbuf, De, Mo, cm + ".7", De, De, x, De, 2, x, 1.0f));
        snprintf(nm, sizeof(nm), "blocks.%d.x_conv", k); trace_add(e, st, nm, x, Mo, De, De, 0);
        // ---- x += 1/2 FFN2(LN(x)); x = LN(x)   (blocks.py:132-135)
        if (W.xf_img[1] && e->split_ffn) {
            SxfFfnParams fp{};
            fp.X = x; fp.ldx = De; fp.Y = xo; fp.ldy = De; fp.wimg = W.xf_img[1]; fp.b2 = W.xf_b2[1]; fp.M = Mo; fp.D = De; fp.nchunk = W.xf_nch[1];
            fp.ln_g = W.ln_out.g; fp.ln_b = W.ln_out.b;
            PROF(PC_GEMM_FFN, 4.0 * Mo * (double)De * De * b.ff_ratio, (double)Mo * De * 8 + 16.0 * De * De * b.ff_ratio);
            EC_TRY(launch_sxf_ffn(fp, st));
        } else {
            EC_TRY(layernorm(x, Mo, De, W.ln_ffn2, a));
            EC_TRY(xgemm(e, st, a, De, Mo, p + ".feed_forward_module2.layers.1", De * b.ff_ratio, De, hb, De * b.ff_ratio, 1, nullptr, 1.f, 0, 0, 0, 0, 0, 0, 0, PC_GEMM_FFN));
            EC_TRY(xgemm(e, st, hb, De * b.ff_ratio, Mo, p + ".feed_forward_module2.layers.4", De, De * b.ff_ratio, x, De, 2, x, 0.5f, 0, 0, 0, 0, 0, 0, 0, PC_GEMM_FFN));
            EC_TRY(layernorm(x, Mo, De, W.ln_out, xo));
        }
        }
        if (!(k == nb - 1 && !rg)) std::swap(x, xalt);
        snprintf(nm, sizeof(nm), "blocks.%d.out", k); trace_add(e, st, nm, xo, Mo, De, De, 0);
    }
    if (!capturing && !e->trace_arena) e->e_cache_put(ws, B, e_tag, e_layout);
    if (rg) {
        const RaggedRows rl = rows_at(nb);
        PROF(PC_MISC, 0, (double)s.Mfinal * e->blocks.back().dim_expand * 4 + (double)B * out_frames * e->blocks.back().dim_expand * 4);
        EC_TRY(launch_emit_rows(x, e->blocks.back().dim_expand, rl.off, rl.len, B, out_frames, out, st));
    }
    return 0;
}

}  // namespace

// =================================================================== C ABI
extern "C" {

int effconf_abi_version(void) { return EFFCONF_ABI_VERSION; }
const char* effconf_last_error(void) { return g_err.c_str(); }

EcEncoder* effconf_encoder_create(const EcConfig* cfg) {
    if (!cfg || cfg->num_blocks <= 0 || !cfg->blocks) { fail("null / empty config"); return nullptr; }
    if (cfg->sub_layers < 1 || cfg->sub_layers > 2) { fail("Conv2dSubsampling with 1 or 2 layers is native"); return nullptr; }
    if (cfg->sub_layers == 2 && (cfg->sub_filters[0] % 8 || cfg->sub_filters[1] % 8 || cfg->n_mels % 16)) { fail("two-layer subsampler needs filters % 8 == 0, n_mels % 16 == 0"); return nullptr; }
    if (cfg->n_mels % 4 || cfg->n_mels > 128) { fail("n_mels must be a multiple of 4, <= 128"); return nullptr; }
    for (int i = 0; i < cfg->num_blocks; ++i) {
        const EcBlock& b = cfg->blocks[i];
        if (b.dim_model % 4 || b.dim_expand % 4 || b.kernel_size > 31 || !(b.kernel_size & 1) || !(b.group_size & 1) ||
            (b.group_size * b.dim_model) % b.num_heads || b.conv_stride < 1 || b.conv_stride > 2 ||
            ec_round_up(b.group_size * b.dim_model / b.num_heads, 32) > 192) {
            fail("unsupported block hyper-parameters at block " + std::to_string(i)); return nullptr;
        }
    }
    if (cfg->left_context < 0 || cfg->right_context < 0) { fail("left_context / right_context must be >= 0"); return nullptr; }
    EcEncoder* e = new EcEncoder();
    e->cfg = *cfg;
    e->blocks.assign(cfg->blocks, cfg->blocks + cfg->num_blocks);
    e->cfg.blocks = e->blocks.data();
    if (const char* g = getenv("EFFCONF_POISON_GUARDS")) e->guard_bytes = atoi(g) > 0 ? (size_t)(atoi(g) > 16 ? atoi(g) : 16) * 1024 : 0;   // value = guard size in KiB (at least 16)   // once per encoder, never on the forward path
    return e;
}

void effconf_encoder_destroy(EcEncoder* e) {
    if (!e) return;
    for (void* p : e->allocs) (void)hipFree(p);
    delete e;
}

int effconf_encoder_load_tensor(EcEncoder* e, const char* key, const float* host, const int64_t* shape, int32_t ndim) {
    if (!e || !key || !host) return fail("null argument");
    HostTensor t;
    int64_t n = 1;
    for (int i = 0; i < ndim; ++i) { t.shape.push_back(shape[i]); n *= shape[i]; }
    t.data.assign(host, host + n);
    e->host[key] = std::move(t);
    e->finalized = false;
    return 0;
}

int effconf_encoder_finalize(EcEncoder* e) {
    if (!e) return fail("null encoder");
    for (void* p : e->allocs) (void)hipFree(p);
    e->allocs.clear();
    e->bw.assign(e->blocks.size(), BlockW());
    std::string err;
    const EcConfig& c = e->cfg;
    // ---- subsampling conv (C,1,3,3) + BatchNorm2d fold
    {
        const int C = c.sub_filters[0];
        const HostTensor* w = find(e, "subsampling_module.layers.0.0.weight");
        const HostTensor* b = find(e, "subsampling_module.layers.0.0.bias");
        std::vector<float> sc, sh;
        if (!w || !b || (int)w->data.size() != C * 9) return fail("missing subsampling conv weights");
        if (!bn_fold(e, "subsampling_module.layers.0.1", C, &sc, &sh, &err)) return fail(err);
        std::vector<float> w9(C * 9), bb(C);
        for (int ch = 0; ch < C; ++ch) {
            for (int j = 0; j < 9; ++j) w9[ch * 9 + j] = w->data[ch * 9 + j] * sc[ch];
            bb[ch] = b->data[ch] * sc[ch] + sh[ch];
        }
        e->sub_w9 = upload(e, w9); e->sub_b = upload(e, bb);
        if (c.sub_layers == 1) {
            if (!pack_named_linear(e, "linear", e->blocks[0].dim_model, C * (c.n_mels / 2), &e->lin, &err)) return fail(err);
        } else {
            // ---- layer 2: (C1, C, 3, 3) conv + BatchNorm2d fold -> implicit-GEMM weight [C1][9*Cp], K order (tap, c_in)
            const int C1 = c.sub_filters[1], Cp = ec_round_up(C, 64), F2 = c.n_mels / 4;
            const HostTensor* w2 = find(e, "subsampling_module.layers.1.0.weight");
            const HostTensor* b2 = find(e, "subsampling_module.layers.1.0.bias");
            std::vector<float> sc2, sh2;
            if (!w2 || !b2 || (int64_t)w2->data.size() != (int64_t)C1 * C * 9) return fail("missing subsampling layer-2 conv weights");
            if (!bn_fold(e, "subsampling_module.layers.1.1", C1, &sc2, &sh2, &err)) return fail(err);
            const int Np = ec_round_up(C1, 128);
            std::vector<uint16_t> wp((size_t)Np * 9 * Cp, 0);
            std::vector<float> bp(Np, 0.f);
            for (int n = 0; n < C1; ++n) {
                for (int ci = 0; ci < C; ++ci)
                    for (int tap = 0; tap < 9; ++tap)
                        wp[(size_t)n * 9 * Cp + (size_t)tap * Cp + ci] = h_f2bf(w2->data[((size_t)n * C + ci) * 9 + tap] * sc2[n]);
                bp[n] = b2->data[n] * sc2[n] + sh2[n];
            }
            e->sub2_w = upload(e, wp); e->sub2_b = upload(e, bp); e->sub2_cp = Cp;
            // ---- Linear with its K axis in (f2, c) order (reference feature index c*F2 + f2, modules.py:247)
            const HostTensor* lw = find(e, "linear.weight");
            const HostTensor* lb = find(e, "linear.bias");
            const int N = e->blocks[0].dim_model, K = C1 * F2;
            if (!lw || !lb || (int64_t)lw->data.size() != (int64_t)N * K) return fail("missing / mis-shaped linear.weight");
            std::vector<float> perm((size_t)N * K);
            for (int n = 0; n < N; ++n)
                for (int f2 = 0; f2 < F2; ++f2)
                    for (int ch = 0; ch < C1; ++ch) perm[(size_t)n * K + (size_t)f2 * C1 + ch] = lw->data[(size_t)n * K + (size_t)ch * F2 + f2];
            std::vector<const float*> rows(N);
            for (int n = 0; n < N; ++n) rows[n] = perm.data() + (size_t)n * K;
            if (!pack_linear(e, rows, lb->data, K, &e->lin)) return fail("upload failed");
        }
        if (c.sub_layers == 1 && sublinear_fused_supported(c.n_mels, e->blocks[0].dim_model)) {
            // K' = (fc*Cp + ch)*8 + e  <->  reference feature ch*(F/2) + 8*fc + e   (sublinear.hip)
            const HostTensor* lw = find(e, "linear.weight");
            const int N = e->blocks[0].dim_model, F2 = c.n_mels / 2, Cp = ec_round_up(C, 8), Kp = (F2 / 8) * Cp * 8;
            const int Np = ec_round_up(N, 128);
            std::vector<uint16_t> wf((size_t)Np * Kp, 0);
            for (int n = 0; n < N; ++n)
                for (int fc = 0; fc < F2 / 8; ++fc)
                    for (int ch = 0; ch < C; ++ch)
                        for (int ee = 0; ee < 8; ++ee)
                            wf[(size_t)n * Kp + ((size_t)fc * Cp + ch) * 8 + ee] = h_f2bf(lw->data[(size_t)n * (C * F2) + ch * F2 + fc * 8 + ee]);
            e->lin_fused = upload(e, wf); e->lin_fused_ld = Kp;
        }
        e->lin_rs = nullptr; e->conv_tab = nullptr;
        if (c.sub_layers == 1) {
            const int N = e->blocks[0].dim_model, F2 = c.n_mels / 2;
            const int CGr = sublinear2_groups(c.n_mels, C, N);
            if (CGr > 0) {
                // sublinear2.hip: per output frequency f a slab [32 NT rows n][32 CG columns]: natural column = channel c (reference feature
                // c*(F/2) + f, modules.py:247), K index permuted inside every group of 16 (packed position 8h+e <-> column 4h + 8(e>>2) + (e&3)):
                // the Swish-ed accumulator registers of the conv MFMA are the B fragments directly (chain.hip's register hand-off)
                const HostTensor* lw = find(e, "linear.weight");
                const int NT = CGr, Cp = 32 * CGr, rows = 32 * NT;
                std::vector<uint16_t> wr((size_t)F2 * rows * Cp, 0);
                for (int f = 0; f < F2; ++f)
                    for (int n = 0; n < N; ++n)
                        for (int k = 0; k < Cp; ++k) {
                            const int g16 = k / 16, pp = k % 16, hh = pp >> 3, ee = pp & 7;
                            const int ch = g16 * 16 + 4 * hh + 8 * (ee >> 2) + (ee & 3);
                            if (ch < C) wr[((size_t)f * rows + n) * Cp + k] = h_f2bf(lw->data[(size_t)n * (C * F2) + (size_t)ch * F2 + f]);
                        }
                std::vector<float> tab((size_t)Cp * 16, 0.f);
                for (int ch = 0; ch < C; ++ch) {
                    for (int j = 0; j < 9; ++j) tab[(size_t)ch * 16 + j] = w9[ch * 9 + j];
                    tab[(size_t)ch * 16 + 9] = bb[ch];
                }
                e->lin_rs = upload(e, wr); e->conv_tab = upload(e, tab);
            }
            // sublinear3.hip: chunks of (output frequency, 32 channels), any channel count / width up to 384: conv taps as bf16 hi (truncation) + lo, the
            // Linear's weight [chunk][32 nt rows][32 k] with the k order of the accumulator layout
            e->sub3_cimg = e->sub3_wimg = nullptr; e->sub3_bias = nullptr;
            const int nt3 = sublinear3_tiles(N);
            const HostTensor *lw3 = find(e, "linear.weight"), *lb3 = find(e, "linear.bias");
            if (nt3 && lw3 && lb3 && (int64_t)lw3->data.size() == (int64_t)N * C * F2 && (int)lb3->data.size() == N) {
                const int ncb = (C + 31) / 32, DP2 = 32 * nt3;
                std::vector<uint16_t> cimg((size_t)ncb * 2 * 32 * 16, 0), wimg((size_t)F2 * ncb * DP2 * 32, 0);
                auto put_tap = [&](int ch, int tap, float v) {
                    uint32_t u; memcpy(&u, &v, 4);
                    const uint32_t hb = u & 0xFFFF0000u; float hf; memcpy(&hf, &hb, 4);
                    const size_t base = (size_t)(ch / 32) * 2 * 32 * 16 + (size_t)(ch % 32) * 16 + tap;
                    cimg[base] = (uint16_t)(hb >> 16); cimg[base + 32 * 16] = h_f2bf(v - hf);
                };
                for (int ch = 0; ch < C; ++ch) {
                    for (int j = 0; j < 9; ++j) put_tap(ch, j, w9[ch * 9 + j]);
                    put_tap(ch, 9, bb[ch]);
                }
                for (int f = 0; f < F2; ++f)
                    for (int cb = 0; cb < ncb; ++cb) {
                        const size_t base = (size_t)(f * ncb + cb) * DP2 * 32;
                        for (int n = 0; n < N; ++n)
                            for (int pos = 0; pos < 32; ++pos) {
                                const int sstep = pos >> 4, khh = (pos >> 3) & 1, ee = pos & 7;
                                const int ch = 32 * cb + 16 * sstep + 8 * (ee >> 2) + 4 * khh + (ee & 3);      // accumulator register 8 s + e of lane half kh holds this channel
                                if (ch < C) wimg[base + (size_t)n * 32 + pos] = h_f2bf(lw3->data[(size_t)n * (C * F2) + (size_t)ch * F2 + f]);
                            }
                    }
                std::vector<float> bp(DP2, 0.f);
                for (int n = 0; n < N; ++n) bp[n] = lb3->data[n];
                e->sub3_cimg = upload(e, cimg); e->sub3_wimg = upload(e, wimg); e->sub3_bias = upload(e, bp); e->sub3_ncb = ncb; e->sub3_fo = F2;
            }
        }
    }
    std::map<std::pair<int, int>, const bf16_t*> tables;
    std::vector<int> strides;
    for (size_t k = 0; k < e->blocks.size(); ++k) {
        const EcBlock& b = e->blocks[k];
        BlockW& W = e->bw[k];
        const int D = b.dim_model, De = b.dim_expand, F1 = D * b.ff_ratio, F2 = De * b.ff_ratio;
        const std::string p = "blocks." + std::to_string(k);
        strides.push_back(b.conv_stride);
        bool ok = get_ln(e, p + ".feed_forward_module1.layers.0", D, &W.ln_ffn1, &err) &&
                  pack_named_linear(e, p + ".feed_forward_module1.layers.1", F1, D, &W.ffn1_a, &err) &&
                  pack_named_linear(e, p + ".feed_forward_module1.layers.4", D, F1, &W.ffn1_b, &err) &&
                  get_ln(e, p + ".feed_forward_module2.layers.0", De, &W.ln_ffn2, &err) &&
                  pack_named_linear(e, p + ".feed_forward_module2.layers.1", F2, De, &W.ffn2_a, &err) &&
                  pack_named_linear(e, p + ".feed_forward_module2.layers.4", De, F2, &W.ffn2_b, &err) &&
                  get_ln(e, p + ".norm", De, &W.ln_out, &err);
        if (!ok) return fail(err);
        W.ffn1_bp = pack_ffn2_permuted(e, p + ".feed_forward_module1.layers.4", D, F1);
        W.ffn2_bp = pack_ffn2_permuted(e, p + ".feed_forward_module2.layers.4", De, F2);
        if (!W.ffn1_bp || !W.ffn2_bp) return fail("upload failed");
        const std::string m = p + ".multi_head_self_attention_module";
        if (!get_ln(e, m + ".norm", D, &W.ln_att, &err)) return fail(err);
        {   // Q, K, V stacked into one [3D][D] weight
            std::vector<const float*> rows; std::vector<float> bias;
            for (const char* n : {"query_layer", "key_layer", "value_layer"}) {
                const HostTensor* w = find(e, m + ".mhsa." + n + ".weight");
                const HostTensor* bb = find(e, m + ".mhsa." + n + ".bias");
                if (!w || !bb || (int)w->data.size() != D * D) return fail("missing " + m + ".mhsa." + n);
                for (int r = 0; r < D; ++r) rows.push_back(w->data.data() + (size_t)r * D);
                bias.insert(bias.end(), bb->data.begin(), bb->data.end());
            }
            if (!pack_linear(e, rows, bias, D, &W.qkv)) return fail("upload failed");
            // natural-layout variant for the row-stationary kernel: rows permuted inside every chunk of 32 so that a lane's
            // accumulators are row-contiguous runs of 8 columns (chunk row j <-> column 16(j>>4) + 8((j>>2)&1) + 4((j>>3)&1) + (j&3))
            const int np = ec_round_up(3 * D, 32);
            std::vector<const float*> prow(np, nullptr); std::vector<float> pbias(np, 0.f);
            for (int n = 0; n < np; ++n) {
                const int c = n / 32, j = n % 32;
                const int src = c * 32 + 16 * (j >> 4) + 8 * ((j >> 2) & 1) + 4 * ((j >> 3) & 1) + (j & 3);
                if (src < 3 * D) { prow[n] = rows[src]; pbias[n] = bias[src]; }
            }
            if (!pack_linear(e, prow, pbias, D, &W.qkv_nat)) return fail("upload failed");
            W.qkv_nat.N = 3 * D;
            if (chain_supported(D)) {
                const HostTensor *lg = find(e, m + ".norm.weight"), *lb = find(e, m + ".norm.bias");      // attention pre-norm folded in
                if (!lg || !lb || !pack_linear(e, prow, pbias, D, &W.c_qkv, true, lg->data.data(), lb->data.data())) return fail("upload failed");
                W.c_qkv_chunks = ec_cdiv(3 * D, 64);
            }
        }
        if (chain_supported(D)) {      // D-wide part of the block: FFN1, attention output projection (pointwise-1 below)
            const HostTensor* b2 = find(e, p + ".feed_forward_module1.layers.4.bias");
            if (!b2 || !pack_named_linear(e, p + ".feed_forward_module1.layers.1", F1, D, &W.c_f1a, &err, true, p + ".feed_forward_module1.layers.0") ||
                !pack_named_linear(e, m + ".mhsa.output_layer", D, D, &W.c_outp, &err, true)) return fail("chain packing failed: " + err);
            W.c_f1b = pack_ffn2_permuted(e, p + ".feed_forward_module1.layers.4", D, F1, 0.5f);
            if (chain2_supported(D)) W.c_f1b_cm = pack_ffn2_chunkmajor(e, p + ".feed_forward_module1.layers.4", D, F1, 0.5f, chain_padded_width(D));
            std::vector<float> hb(b2->data); for (float& x : hb) x *= 0.5f;
            W.c_f1b2 = upload(e, hb); W.h_f1b2 = hb;
            if (!W.c_f1b || !W.c_f1b2) return fail("upload failed");
            W.chain_in = true;
        }
        if (chain_supported(De)) {     // De-wide part: pointwise-2, FFN2
            const HostTensor* b2 = find(e, p + ".feed_forward_module2.layers.4.bias");
            if (!b2 || !pack_named_linear(e, p + ".feed_forward_module2.layers.1", F2, De, &W.c_f2a, &err, true, p + ".feed_forward_module2.layers.0") ||
                !pack_named_linear(e, p + ".convolution_module.layers.7", De, De, &W.c_pw2, &err, true)) return fail("chain packing failed: " + err);
            W.c_f2b = pack_ffn2_permuted(e, p + ".feed_forward_module2.layers.4", De, F2, 0.5f);
            if (chain2_supported(De)) W.c_f2b_cm = pack_ffn2_chunkmajor(e, p + ".feed_forward_module2.layers.4", De, F2, 0.5f, chain_padded_width(De));
            std::vector<float> hb(b2->data); for (float& x : hb) x *= 0.5f;
            W.c_f2b2 = upload(e, hb); W.h_f2b2 = hb;
            const HostTensor *og = find(e, p + ".norm.weight"), *ob = find(e, p + ".norm.bias");
            if (!og || !ob) return fail("missing " + p + ".norm");
            W.h_ln_out_g = og->data; W.h_ln_out_b = ob->data;
            if (!W.c_f2b || !W.c_f2b2) return fail("upload failed");
            W.chain_out = true;
        }
        if (!pack_named_linear(e, m + ".mhsa.pos_layer", D, D, &W.pos, &err)) return fail(err);
        if (!pack_named_linear(e, m + ".mhsa.output_layer", D, D, &W.outp, &err)) return fail(err);
        const HostTensor *u = find(e, m + ".mhsa.u"), *v = find(e, m + ".mhsa.v");
        if (!u || !v || (int)u->data.size() != D) return fail("missing " + m + ".mhsa.u/v");
        W.u = upload(e, u->data); W.v = upload(e, v->data);
        W.h_u = u->data; W.h_v = v->data;
        {   // head column x of head h is feature (h*d + x) % D of the un-grouped row (group = view, attentions.py:677-686)
            const int H = b.num_heads, d = b.group_size * D / H;
            W.dvu_ld = ec_round_up(d, 32);
            std::vector<float> t((size_t)H * W.dvu_ld, 0.f);
            for (int h = 0; h < H; ++h)
                for (int x = 0; x < d; ++x) { const int n = (h * d + x) % D; t[(size_t)h * W.dvu_ld + x] = v->data[n] - u->data[n]; }
            W.dvu = upload(e, t);
        }
        auto key = std::make_pair(b.max_pos, D);
        if (!tables.count(key)) tables[key] = build_pos_table(e, b.max_pos, D);
        W.pos_table = tables[key];
        // ---- convolution module
        const std::string cm = p + ".convolution_module.layers";
        if (!get_ln(e, cm + ".0", D, &W.ln_conv, &err)) return fail(err);
        {   // pointwise-1 (2De, D, 1): GLU halves interleaved in blocks of 32 output channels (a | b)
            const HostTensor* w = find(e, cm + ".2.weight");
            const HostTensor* bb = find(e, cm + ".2.bias");
            if (!w || !bb || (int)w->data.size() != 2 * De * D) return fail("missing " + cm + ".2");
            const int nblk = ec_cdiv(De, 32);
            std::vector<const float*> rows(nblk * 64, nullptr); std::vector<float> bias(nblk * 64, 0.f);
            for (int j = 0; j < De; ++j) {
                const int jb = j / 32, jj = j % 32;
                rows[jb * 64 + jj] = w->data.data() + (size_t)j * D;            bias[jb * 64 + jj] = bb->data[j];
                rows[jb * 64 + 32 + jj] = w->data.data() + (size_t)(De + j) * D; bias[jb * 64 + 32 + jj] = bb->data[De + j];
            }
            if (!pack_linear(e, rows, bias, D, &W.pw1)) return fail("upload failed");
            if (chain_supported(D)) {
                const HostTensor *lg = find(e, cm + ".0.weight"), *lb = find(e, cm + ".0.bias");          // conv-module pre-norm folded in
                if (!lg || !lb || !pack_linear(e, rows, bias, D, &W.c_pw1, true, lg->data.data(), lb->data.data())) return fail("upload failed");
                W.c_pw1_chunks = nblk;
            }
        }
        {   // depthwise (De, 1, k) + BatchNorm1d fold -> [k][De] fp32
            const int ks = b.kernel_size;
            const HostTensor* w = find(e, cm + ".4.weight");
            const HostTensor* bb = find(e, cm + ".4.bias");
            std::vector<float> sc, sh;
            if (!w || !bb || (int)w->data.size() != De * ks) return fail("missing " + cm + ".4");
            if (!bn_fold(e, cm + ".5", De, &sc, &sh, &err)) return fail(err);
            std::vector<float> wk((size_t)ks * De), bz(De);
            for (int ch = 0; ch < De; ++ch) {
                for (int j = 0; j < ks; ++j) wk[(size_t)j * De + ch] = w->data[(size_t)ch * ks + j] * sc[ch];
                bz[ch] = bb->data[ch] * sc[ch] + sh[ch];
            }
            W.dw_w = upload(e, wk); W.dw_b = upload(e, bz);
            if (dwconv_mfma_supported(ks, b.conv_stride)) {
                std::vector<uint16_t> ta((size_t)De * 4 * dwconv_mfma_groups(ks) * 8);
                pack_dwconv_mfma(wk.data(), ks, De, ta.data());
                W.dw_a = upload(e, ta);
                if (!W.dw_a) return fail("upload failed");
            }
        }
        if (!pack_named_linear(e, cm + ".7", De, De, &W.pw2, &err)) return fail(err);
        if (D != De && !pack_named_linear(e, p + ".conv_res.1", De, D, &W.res, &err, false, "", De)) return fail(err);
    }
    // ---- constant blocks of the fused chains (one LDS-DMA per workgroup instead of a dozen small strided copies)
    for (size_t k = 0; k < e->blocks.size(); ++k) {
        BlockW& W = e->bw[k];
        const EcBlock& b = e->blocks[k];
        const int D = b.dim_model, De = b.dim_expand;
        auto build = [&](int kind, int dim, const BlockW* pre, const BlockW* post, const EcBlock* pb, const EcBlock* qb) -> const float* {
            ChainParams cp{};
            cp.D = dim;
            const bool isb = kind == CHAIN_B;
            if (pre && !isb) cp.f[0].Fp = ec_round_up(pb->dim_expand * pb->ff_ratio, 32);
            if (post) cp.f[1].Fp = ec_round_up(qb->dim_model * qb->ff_ratio, 32);
            cp.g1.nchunks = isb ? pre->c_pw1_chunks : (post ? post->c_qkv_chunks : 0);
            int nf[8];
            const int nfl = chain_const_layout(cp, kind, nf);
            const int DP = chain_padded_width(dim);
            std::vector<float> blk(nfl, 0.f);
            auto put = [&](int off, const std::vector<float>& src, int n) { for (int i = 0; i < n && i < (int)src.size(); ++i) blk[off + i] = src[i]; };
            if (isb) {
                put(nf[0], pre->c_outp.hbias, dim);
                put(nf[6], pre->c_pw1.hbias, 64 * cp.g1.nchunks);
            } else {
                if (pre) {
                    put(nf[0], pre->c_pw2.hbias, dim);
                    put(nf[1], pre->h_ln_out_g, dim); put(nf[1] + DP, pre->h_ln_out_b, dim);
                    put(nf[2], pre->c_f2a.hbias, cp.f[0].Fp); put(nf[3], pre->h_f2b2, dim);
                }
                if (post) {
                    put(nf[4], post->c_f1a.hbias, cp.f[1].Fp); put(nf[5], post->h_f1b2, dim);
                    put(nf[6], post->c_qkv.hbias, 64 * cp.g1.nchunks);
                    put(nf[7], post->h_u, dim); put(nf[7] + DP, post->h_v, dim);
                }
            }
            return upload(e, blk);
        };
        if (W.chain_in) {
            W.cc_b = build(CHAIN_B, D, &W, nullptr, &b, nullptr);
            if (chain_head_supported(D)) W.cc_head = build(CHAIN_A_HEAD, D, nullptr, &W, nullptr, &b);
        }
        if (W.chain_out && chain_tail_supported(De)) {
            W.cc_tail = build(CHAIN_A_TAIL, De, &W, nullptr, &b, nullptr);
            if (chain_full_supported(De, std::max(e->chain_full_max, chain2_supported(De) ? 256 : 0)) && k + 1 < e->blocks.size() && e->bw[k + 1].chain_in && e->blocks[k + 1].dim_model == De)
                W.cc_full = build(CHAIN_A_FULL, De, &W, &e->bw[k + 1], &b, &e->blocks[k + 1]);
        }
    }
    e->block_stride = upload(e, strides);
    {
        std::vector<int> gs, hs;
        for (const EcBlock& b : e->blocks) { gs.push_back(b.group_size); hs.push_back(b.num_heads); }
        e->block_group = upload(e, gs); e->block_heads = upload(e, hs);
    }
    if (c.vocab_size > 0) {
        const HostTensor *w = find(e, "fc.weight"), *b = find(e, "fc.bias");
        const int D = e->blocks.back().dim_expand, V = c.vocab_size;
        if (w && b) {
            if ((int)w->data.size() != V * D) return fail("fc.weight shape mismatch");
            std::vector<float> wt((size_t)D * V);
            for (int v = 0; v < V; ++v) for (int k = 0; k < D; ++k) wt[(size_t)k * V + v] = w->data[(size_t)v * D + k];
            e->fc_wt = upload(e, wt); e->fc_b = upload(e, b->data);
            // split-bf16 images: W = hi + lo (hi = bf16(W), lo = bf16(W - hi)), fragment (k-step s, column v, k-half h) = W[v][16 s + 8 h .. + 7]
            const int Kp = ec_round_up(D, 16), Vp = ec_round_up(V, 256);   // whole 256-column passes of ctc_argmax_bf16x3_kernel (4 waves x 64): every wave's fragment loads stay inside the image
            std::vector<uint16_t> hi((size_t)(Kp / 16) * Vp * 16, 0), lo(hi.size(), 0);
            for (int v = 0; v < V; ++v)
                for (int k = 0; k < D; ++k) {
                    const float wv = w->data[(size_t)v * D + k];
                    const uint16_t h = h_f2bf(wv);
                    uint32_t hb = (uint32_t)h << 16; float hf; memcpy(&hf, &hb, 4);
                    const size_t idx = (((size_t)(k / 16) * Vp + v) * 2 + (k % 16) / 8) * 8 + k % 8;
                    hi[idx] = h; lo[idx] = h_f2bf(wv - hf);
                }
            e->fc_hi = upload(e, hi); e->fc_lo = upload(e, lo);
        }
    }
    if (!build_mel_tables(e, &err)) return fail(err);
    e->xw.clear(); e->xtab.clear();
    if (e->exact_pack) {       // fp32-operand mode: the reference-layout fp32 tensors themselves, fp32 sinusoid tables, BatchNorm scale / shift
        for (auto& kv : e->host) e->xw[kv.first] = upload(e, kv.second.data);
        std::vector<float> sub_sc0, sub_sh0;
        for (int l = 0; l < c.sub_layers; ++l) {
            const std::string sp = "subsampling_module.layers." + std::to_string(l);
            const int C = c.sub_filters[l];
            const HostTensor* cbias = find(e, sp + ".0.bias");
            std::vector<float> sc, sh;
            if (!cbias || (int)cbias->data.size() != C || !bn_fold(e, sp + ".1", C, &sc, &sh, &err)) return fail("exact mode: " + err);
            for (int ch = 0; ch < C; ++ch) sh[ch] += cbias->data[ch] * sc[ch];
            e->xsub_scale[l] = upload(e, sc); e->xsub_shift[l] = upload(e, sh);
            if (l == 0) { sub_sc0 = sc; sub_sh0 = sh; }
        }
        e->xsplit.clear();
        e->xsub_cimg = e->xsub_wimg = nullptr; e->xsub_bias = nullptr;
        if (e->exact_split && c.sub_layers == 1) {
            // images of the fused front end (sxf_sub.hip): conv taps with the BatchNorm scale folded in (shift + scaled conv bias in tap 9), the Linear's weight in
            // chunks of (output frequency f', 32 channels) with the k order of the accumulator layout; same-scale halves at 2^10 as for the other fused kernels
            const int Co = c.sub_filters[0], Fo = (c.n_mels - 1) / 2 + 1, N = e->blocks[0].dim_model, nt = sxf_sublin_tiles(N);
            const HostTensor *cw = find(e, "subsampling_module.layers.0.0.weight"), *lw = find(e, "linear.weight"), *lb = find(e, "linear.bias");
            if (nt && cw && lw && lb && (int64_t)cw->data.size() == (int64_t)Co * 9 && (int64_t)lw->data.size() == (int64_t)N * Co * Fo && (int)lb->data.size() == N) {
                auto half_bits = [](float f) { _Float16 h = (_Float16)f; uint16_t u; memcpy(&u, &h, 2); return u; };
                auto put = [&](std::vector<uint16_t>& img, size_t hi_at, size_t lo_at, double wv) {
                    float ws = (float)(wv * 1024.0);
                    ws = ws > 65000.f ? 65000.f : (ws < -65000.f ? -65000.f : ws);
                    const _Float16 hh = (_Float16)ws;
                    img[hi_at] = half_bits((float)hh);
                    img[lo_at] = half_bits(ws - (float)hh);
                };
                const int ncb = (Co + 31) / 32, DP2 = 32 * nt;
                std::vector<uint16_t> cimg((size_t)ncb * 2 * 32 * 16, 0), wimg((size_t)Fo * ncb * 2 * DP2 * 32, 0);
                for (int co = 0; co < Co; ++co) {
                    const size_t base = (size_t)(co / 32) * 2 * 32 * 16 + (size_t)(co % 32) * 16;
                    for (int tap = 0; tap < 9; ++tap) put(cimg, base + tap, base + 32 * 16 + tap, (double)cw->data[(size_t)co * 9 + tap] * sub_sc0[co]);
                    put(cimg, base + 9, base + 32 * 16 + 9, sub_sh0[co]);
                }
                for (int fo = 0; fo < Fo; ++fo)
                    for (int cb = 0; cb < ncb; ++cb) {
                        const size_t base = (size_t)(fo * ncb + cb) * 2 * DP2 * 32;
                        for (int n = 0; n < N; ++n)
                            for (int pos = 0; pos < 32; ++pos) {
                                const int sstep = pos >> 4, khh = (pos >> 3) & 1, ee = pos & 7;
                                const int co = 32 * cb + 16 * sstep + 8 * (ee >> 2) + 4 * khh + (ee & 3);      // accumulator register 8 s + e of lane half kh holds this channel
                                if (co >= Co) continue;
                                put(wimg, base + (size_t)n * 32 + pos, base + (size_t)DP2 * 32 + (size_t)n * 32 + pos, lw->data[(size_t)n * Co * Fo + (size_t)co * Fo + fo]);
                            }
                    }
                std::vector<float> bp(DP2, 0.f);
                for (int n = 0; n < N; ++n) bp[n] = lb->data[n];
                e->xsub_cimg = upload(e, cimg); e->xsub_wimg = upload(e, wimg); e->xsub_bias = upload(e, bp); e->xsub_ncb = ncb; e->xsub_fo = Fo;
            }
        }
        if (e->exact_split) {
            // every 2-D weight (nn.Linear [N][K], 1x1 Conv1d [N][K][1]) as two fp16 images h = fp16(w), l = fp16((w - h) * 2048), K padded
            // with zeros to whole 32-wide k-tiles and stored k-tile major; the three attention projections of a block additionally stacked (q | k | v)
            auto half_bits = [](float f) { _Float16 h = (_Float16)f; uint16_t u; memcpy(&u, &h, 2); return u; };
            auto clampf = [](float f) { return f > 65000.f ? 65000.f : (f < -65000.f ? -65000.f : f); };
            auto add_split = [&](const std::string& prefix, const std::vector<const float*>& rows, int K) {
                const int N = (int)rows.size(), ldh = ec_round_up(K, 32);
                std::vector<uint16_t> hi((size_t)N * ldh, 0), lo(hi.size(), 0);
                for (int n = 0; n < N; ++n)
                    for (int k = 0; k < K; ++k) {
                        const float wv = rows[n][k];
                        const _Float16 h = (_Float16)clampf(wv);
                        const size_t at = ((size_t)(k / 32) * N + n) * 32 + k % 32;       // k-tile major (kernels.h: SxGemmParams)
                        hi[at] = half_bits((float)h);
                        lo[at] = half_bits(clampf((wv - (float)h) * 2048.0f));
                    }
                e->xsplit[prefix] = EcEncoder::SplitW{upload(e, hi), upload(e, lo), ldh};
            };
            for (auto& kv : e->host) {
                const std::string& key = kv.first;
                const HostTensor& t = kv.second;
                if (key.size() < 8 || key.compare(key.size() - 7, 7, ".weight") != 0) continue;
                const bool lin = t.shape.size() == 2, pw = t.shape.size() == 3 && t.shape[2] == 1 && key.find("subsampling") == std::string::npos;
                if (!lin && !pw) continue;
                const int N = (int)t.shape[0], K = (int)t.shape[1];
                if (K % 4 || key == "fc.weight") continue;
                std::vector<const float*> rows(N);
                for (int n = 0; n < N; ++n) rows[n] = t.data.data() + (size_t)n * K;
                add_split(key.substr(0, key.size() - 7), rows, K);
            }
            for (size_t k = 0; k < e->blocks.size(); ++k) {
                const std::string m = "blocks." + std::to_string(k) + ".multi_head_self_attention_module.mhsa.";
                const int D = e->blocks[k].dim_model;
                std::vector<const float*> rows;
                std::vector<float> bias;
                bool ok = true;
                for (const char* nm : {"query_layer", "key_layer", "value_layer"}) {
                    const HostTensor *w = find(e, m + nm + ".weight"), *b = find(e, m + nm + ".bias");
                    if (!w || !b || (int)w->data.size() != D * D || (int)b->data.size() != D) { ok = false; break; }
                    for (int n = 0; n < D; ++n) rows.push_back(w->data.data() + (size_t)n * D);
                    bias.insert(bias.end(), b->data.begin(), b->data.end());
                }
                if (!ok) continue;
                add_split(m + "qkv_layer", rows, D);
                e->xw[m + "qkv_layer.bias"] = upload(e, bias);
            }
            // weight images of the fused FFN kernel (sxf_ffn.hip; kernels.h: SxfFfnParams)
            for (size_t k = 0; k < e->blocks.size(); ++k)
                for (int which = 0; which < 2; ++which) {
                    const int D = which ? e->blocks[k].dim_expand : e->blocks[k].dim_model, F = D * e->blocks[k].ff_ratio;
                    if (!sxf_ffn_supported(D)) continue;
                    const std::string pf = "blocks." + std::to_string(k) + (which ? ".feed_forward_module2.layers." : ".feed_forward_module1.layers.");
                    const HostTensor *g = find(e, pf + "0.weight"), *bt = find(e, pf + "0.bias"), *w1 = find(e, pf + "1.weight"), *b1 = find(e, pf + "1.bias"),
                                     *w2 = find(e, pf + "4.weight"), *b2 = find(e, pf + "4.bias");
                    if (!g || !bt || !w1 || !b1 || !w2 || !b2 || (int64_t)w1->data.size() != (int64_t)F * D || (int64_t)w2->data.size() != (int64_t)F * D ||
                        (int)g->data.size() != D || (int)bt->data.size() != D || (int)b1->data.size() != F || (int)b2->data.size() != D) continue;
                    int ks1, nt2; sxf_ffn_shape(D, &ks1, &nt2);
                    const int DP1 = 16 * ks1, DP2 = 32 * nt2, nch = (F + 31) / 32;
                    const size_t per = (size_t)64 * (DP1 + DP2);
                    std::vector<uint16_t> img((size_t)nch * per, 0);
                    auto put = [&](size_t hi_at, size_t lo_at, float wv) {      // same-scale halves at the weight scale 2^10 (sx_common.h split2s; sxf_ffn.hip SW)
                        const float ws = clampf(wv * 1024.0f);
                        const _Float16 hh = (_Float16)ws;
                        img[hi_at] = half_bits((float)hh);
                        img[lo_at] = half_bits(ws - (float)hh);
                    };
                    for (int c = 0; c < nch; ++c) {
                        const size_t base = (size_t)c * per;
                        for (int r = 0; r < 32; ++r) {
                            const int hrow = 32 * c + r;
                            if (hrow >= F) continue;
                            double bias = b1->data[hrow];
                            for (int kk = 0; kk < D; ++kk) {
                                const double wv = w1->data[(size_t)hrow * D + kk];
                                bias += wv * bt->data[kk];                                       // W1 beta folded into the bias column
                                put(base + (size_t)r * DP1 + kk, base + (size_t)32 * DP1 + (size_t)r * DP1 + kk, (float)(wv * g->data[kk]));
                            }
                            put(base + (size_t)r * DP1 + D, base + (size_t)32 * DP1 + (size_t)r * DP1 + D, (float)bias);
                        }
                        for (int n = 0; n < D; ++n)
                            for (int pos = 0; pos < 32; ++pos) {
                                const int sstep = pos >> 4, khh = (pos >> 3) & 1, ee = pos & 7;
                                const int hid = 32 * c + 16 * sstep + 8 * (ee >> 2) + 4 * khh + (ee & 3);      // accumulator register 8 s + e of lane half kh holds this hidden unit
                                if (hid >= F) continue;
                                put(base + (size_t)64 * DP1 + (size_t)n * 32 + pos, base + (size_t)64 * DP1 + (size_t)32 * DP2 + (size_t)n * 32 + pos,
                                    0.5f * w2->data[(size_t)n * F + hid]);
                            }
                    }
                    std::vector<float> b2h(DP2, 0.f);
                    for (int n = 0; n < D; ++n) b2h[n] = 0.5f * b2->data[n];
                    e->bw[k].xf_img[which] = upload(e, img); e->bw[k].xf_b2[which] = upload(e, b2h); e->bw[k].xf_nch[which] = nch;
                }
            // weight images of the split chains (sxf_chain.hip; kernels.h: SxcBParams / SxcAParams).  ONE k order everywhere - the accumulator layout's: inside a
            // 16-block, position 8 kh + e holds feature 8 (e >> 2) + 4 kh + (e & 3) - because every operand of a product is a converted accumulator tile.
            {
                auto perm16 = [](int pos) { const int khh = pos >> 3, ee = pos & 7; return 8 * (ee >> 2) + 4 * khh + (ee & 3); };
                auto put = [&](std::vector<uint16_t>& img, size_t hi_at, size_t lo_at, double wv) {
                    const float ws = clampf((float)(wv * 1024.0));
                    const _Float16 hh = (_Float16)ws;
                    img[hi_at] = half_bits((float)hh);
                    img[lo_at] = half_bits(ws - (float)hh);
                };
                // F1 chunk at `base`: 32 output rows (null = padding) x DP1 columns, h plane then l plane; gamma folded into the weights, W beta + bias in column D
                auto f1_chunk = [&](std::vector<uint16_t>& img, size_t base, int DP1, int K, const float* const* wrow, const float* bias, const float* g, const float* beta) {
                    for (int r = 0; r < 32; ++r) {
                        if (!wrow[r]) continue;
                        double bsum = bias[r];
                        for (int col = 0; col < DP1; ++col) {
                            const int f = (col & ~15) + perm16(col & 15);
                            if (f < K) {
                                const double wv = wrow[r][f];
                                if (beta) bsum += wv * beta[f];
                                put(img, base + (size_t)r * DP1 + col, base + (size_t)32 * DP1 + (size_t)r * DP1 + col, g ? wv * g[f] : wv);
                            }
                        }
                        for (int col = 0; col < DP1; ++col)
                            if ((col & ~15) + perm16(col & 15) == K) put(img, base + (size_t)r * DP1 + col, base + (size_t)32 * DP1 + (size_t)r * DP1 + col, bsum);
                    }
                };
                // F2 chunk c at `base`: DP2 output rows x 32 input positions (inputs 32 c ..), h plane then l plane
                auto f2_chunk = [&](std::vector<uint16_t>& img, size_t base, int DP2, int c, const float* Wm, int N, int K, int ldw, double scale) {
                    for (int n = 0; n < N; ++n)
                        for (int pos = 0; pos < 32; ++pos) {
                            const int kk = 32 * c + (pos & ~15) + perm16(pos & 15);
                            if (kk < K) put(img, base + (size_t)n * 32 + pos, base + (size_t)32 * DP2 + (size_t)n * 32 + pos, scale * Wm[(size_t)n * ldw + kk]);
                        }
                };
                auto padded = [&](const std::vector<float>& v, int n, float scale) { std::vector<float> o(n, 0.f); for (size_t i = 0; i < v.size() && (int)i < n; ++i) o[i] = scale * v[i]; return o; };
                for (size_t k = 0; k < e->blocks.size(); ++k) {
                    const EcBlock& b = e->blocks[k];
                    BlockW& W = e->bw[k];
                    const int D = b.dim_model, De = b.dim_expand;
                    const std::string pb = "blocks." + std::to_string(k);
                    const std::string mh = pb + ".multi_head_self_attention_module.", cm = pb + ".convolution_module.layers.";
                    auto ffn_image = [&](int which, int Dw) -> const uint16_t* {       // FeedForwardModule `which` at width Dw: per 32 hidden units an F1 chunk + an F2 chunk
                        const int F = Dw * b.ff_ratio;
                        const std::string pf = pb + (which ? ".feed_forward_module2.layers." : ".feed_forward_module1.layers.");
                        const HostTensor *g = find(e, pf + "0.weight"), *bt = find(e, pf + "0.bias"), *w1 = find(e, pf + "1.weight"), *b1 = find(e, pf + "1.bias"),
                                         *w2 = find(e, pf + "4.weight");
                        if (!g || !bt || !w1 || !b1 || !w2 || (int64_t)w1->data.size() != (int64_t)F * Dw || (int64_t)w2->data.size() != (int64_t)F * Dw ||
                            (int)g->data.size() != Dw || (int)bt->data.size() != Dw || (int)b1->data.size() != F || !W.xf_b2[which]) return nullptr;
                        int ks, nt; sxf_ffn_shape(Dw, &ks, &nt);
                        const int DP1 = 16 * ks, DP2 = 32 * nt, nch = (F + 31) / 32;
                        const size_t per = (size_t)64 * (DP1 + DP2);
                        std::vector<uint16_t> img((size_t)nch * per, 0);
                        for (int c = 0; c < nch; ++c) {
                            const float* rows[32]; float bias[32];
                            for (int r = 0; r < 32; ++r) { const int h = 32 * c + r; rows[r] = h < F ? w1->data.data() + (size_t)h * Dw : nullptr; bias[r] = h < F ? b1->data[h] : 0.f; }
                            f1_chunk(img, (size_t)c * per, DP1, Dw, rows, bias, g->data.data(), bt->data.data());
                            f2_chunk(img, (size_t)c * per + (size_t)64 * DP1, DP2, c, w2->data.data(), Dw, F, F, 0.5);
                        }
                        return upload(e, img);
                    };
                    if (sxc_supported(D) && sxf_ffn_supported(D)) {
                        int ks, nt; sxf_ffn_shape(D, &ks, &nt);
                        const int DP1 = 16 * ks, DP2 = 32 * nt, nte = (De + 31) / 32;
                        const HostTensor *wo = find(e, mh + "mhsa.output_layer.weight"), *bo = find(e, mh + "mhsa.output_layer.bias"), *lg = find(e, cm + "0.weight"), *lb = find(e, cm + "0.bias"),
                                         *w1 = find(e, cm + "2.weight"), *b1 = find(e, cm + "2.bias"), *ag = find(e, mh + "norm.weight"), *ab = find(e, mh + "norm.bias");
                        const HostTensor *wq[3] = {find(e, mh + "mhsa.query_layer.weight"), find(e, mh + "mhsa.key_layer.weight"), find(e, mh + "mhsa.value_layer.weight")};
                        const HostTensor *bq[3] = {find(e, mh + "mhsa.query_layer.bias"), find(e, mh + "mhsa.key_layer.bias"), find(e, mh + "mhsa.value_layer.bias")};
                        bool ok = wo && bo && lg && lb && w1 && b1 && ag && ab && (int)wo->data.size() == D * D && (int)bo->data.size() == D && (int)lg->data.size() == D && (int)lb->data.size() == D &&
                                  (int64_t)w1->data.size() == (int64_t)2 * De * D && (int)b1->data.size() == 2 * De && (int)ag->data.size() == D && (int)ab->data.size() == D;
                        for (int i = 0; i < 3; ++i) ok = ok && wq[i] && bq[i] && (int)wq[i]->data.size() == D * D && (int)bq[i]->data.size() == D;
                        if (ok) {
                            std::vector<uint16_t> io((size_t)nt * 64 * DP2, 0);
                            for (int c = 0; c < nt; ++c) f2_chunk(io, (size_t)c * 64 * DP2, DP2, c, wo->data.data(), D, D, D, 1.0);
                            std::vector<uint16_t> ip((size_t)2 * nte * 64 * DP1, 0);
                            for (int c = 0; c < 2 * nte; ++c) {          // chunk 2 j: value rows 32 j .., chunk 2 j + 1: their gate rows De + 32 j ..
                                const float* rows[32]; float bias[32];
                                for (int r = 0; r < 32; ++r) { const int f = 32 * (c >> 1) + r, n = (c & 1) * De + f; rows[r] = f < De ? w1->data.data() + (size_t)n * D : nullptr; bias[r] = f < De ? b1->data[n] : 0.f; }
                                f1_chunk(ip, (size_t)c * 64 * DP1, DP1, D, rows, bias, lg->data.data(), lb->data.data());
                            }
                            std::vector<uint16_t> iq((size_t)3 * nt * 64 * DP1, 0);
                            for (int c = 0; c < 3 * nt; ++c) {
                                const int which = c / nt, cc = c % nt;
                                const float* rows[32]; float bias[32];
                                for (int r = 0; r < 32; ++r) { const int n = 32 * cc + r; rows[r] = n < D ? wq[which]->data.data() + (size_t)n * D : nullptr; bias[r] = n < D ? bq[which]->data[n] : 0.f; }
                                f1_chunk(iq, (size_t)c * 64 * DP1, DP1, D, rows, bias, ag->data.data(), ab->data.data());
                            }
                            W.xc_wo = upload(e, io); W.xc_bo = upload(e, padded(bo->data, DP2, 1.f)); W.xc_p1 = upload(e, ip); W.xc_nch_p1 = 2 * nte; W.xc_qkv = upload(e, iq);
                            W.xc_f[0] = ffn_image(0, D);
                            W.xc_in = W.xc_f[0] != nullptr;
                        }
                    }
                    if (sxc_supported(De) && sxf_ffn_supported(De)) {
                        int ks, nt; sxf_ffn_shape(De, &ks, &nt);
                        const int DP2 = 32 * nt;
                        const HostTensor *w2 = find(e, cm + "7.weight"), *b2 = find(e, cm + "7.bias");
                        if (w2 && b2 && (int64_t)w2->data.size() == (int64_t)De * De && (int)b2->data.size() == De) {
                            std::vector<uint16_t> i2((size_t)nt * 64 * DP2, 0);
                            for (int c = 0; c < nt; ++c) f2_chunk(i2, (size_t)c * 64 * DP2, DP2, c, w2->data.data(), De, De, De, 1.0);
                            W.xc_p2 = upload(e, i2); W.xc_bp2 = upload(e, padded(b2->data, DP2, 1.f));
                            W.xc_f[1] = ffn_image(1, De);
                            W.xc_out = W.xc_f[1] != nullptr;
                        }
                    }
                }
            }
        }
        for (const EcBlock& b : e->blocks) {
            auto key = std::make_pair(b.max_pos, b.dim_model);
            if (e->xtab.count(key)) continue;
            const int rows = 2 * b.max_pos - 1, D = b.dim_model;
            std::vector<float> t((size_t)rows * D, 0.f), denom(D / 2);
            for (int i = 0; i < D / 2; ++i) denom[i] = std::pow(10000.0f, (2.0f * (float)i) / (float)D);
            for (int r = 0; r < rows; ++r) {
                const float pos = (float)(b.max_pos - 1 - r);
                for (int i = 0; i < D / 2; ++i) { const float a = pos / denom[i]; t[(size_t)r * D + 2 * i] = std::sin(a); t[(size_t)r * D + 2 * i + 1] = std::cos(a); }
            }
            e->xtab[key] = upload(e, t);
        }
    }
    for (void* p : e->allocs) if (!p) return fail("device allocation failed");
    if (hipDeviceSynchronize() != hipSuccess) return fail("upload failed");
    e->host.clear();
    e->e_cache.clear();
    {   // tiled_auto: the widest stage of the configuration decides (a property of the configuration, never of the batch: one path per handle)
        int dmax = 0;
        for (const EcBlock& b : e->blocks) dmax = std::max(dmax, std::max(b.dim_model, b.dim_expand));
        e->tiled_auto_on = dmax > e->tiled_min_k && dmax <= 384;
    }
    e->finalized = true;
    return 0;
}

size_t effconf_encoder_workspace_bytes(const EcEncoder* e, int32_t batch, int32_t n, int32_t from_audio) {
    if (!e || batch <= 0 || n <= 0) return 0;
    const int Tm = from_audio ? n / e->cfg.hop_length + 1 : n;
    const Shapes s = make_shapes(e, batch, Tm);
    size_t bytes = make_workspace(e, s, from_audio != 0).total;
    if (e->exact_pack) bytes = std::max(bytes, make_xworkspace(e, s).total + (from_audio ? al((size_t)batch * e->cfg.n_mels * Tm * 4) : 0));
    return bytes;
}

int32_t effconf_encoder_out_frames(const EcEncoder* e, int32_t n, int32_t from_audio) {
    if (!e || n <= 0) return 0;
    const int Tm = from_audio ? n / e->cfg.hop_length + 1 : n;
    return make_shapes(e, 1, Tm).Tout.back();
}

int effconf_encoder_forward_mel(EcEncoder* e, const float* mel, const int64_t* mel_len, int32_t batch, int32_t n_frames,
                                float* out, int64_t* out_len, void* workspace, size_t workspace_bytes, void* stream) {
    if (!e || !e->finalized) return fail("encoder not finalized");
    if (!mel || !mel_len || !out || !workspace || batch <= 0 || n_frames <= 0) return fail("bad argument");
    const Shapes s = make_shapes(e, batch, n_frames);
    if (e->exact_on) {
        const XWorkspace xw = make_xworkspace(e, s);
        if (workspace_bytes < xw.total) return fail("workspace too small");
        // split mode: the fused kernels (sxf.hip); attention maps are a by-product of split.hip's scores-in-memory kernels only
        if (split_fused_ok(e) && e->att_out.empty())
            return forward_core_split(e, mel, mel_len, 0, s, xw, reinterpret_cast<char*>(workspace), out, out_len, (hipStream_t)stream);
        return forward_core_exact(e, mel, mel_len, 0, s, xw, reinterpret_cast<char*>(workspace), out, out_len, (hipStream_t)stream);
    }
    const Workspace w = make_workspace(e, s, false);
    if (workspace_bytes < w.total) return fail("workspace too small");
    return forward_core(e, mel, mel_len, 0, s, w, reinterpret_cast<char*>(workspace), out, out_len, (hipStream_t)stream);
}

int effconf_encoder_forward(EcEncoder* e, const float* audio, const int64_t* x_len, int32_t batch, int32_t n_samples,
                            float* out, int64_t* out_len, void* workspace, size_t workspace_bytes, void* stream) {
    if (!e || !e->finalized) return fail("encoder not finalized");
    if (!audio || !x_len || !out || !workspace || batch <= 0 || n_samples <= e->cfg.n_fft / 2) return fail("bad argument");
    const int Tm = n_samples / e->cfg.hop_length + 1;
    const Shapes s = make_shapes(e, batch, Tm);
    if (e->exact_on) {       // mel at the tail of the exact workspace (the mel kernel is fp32 in both modes)
        const XWorkspace xw = make_xworkspace(e, s);
        if (workspace_bytes < xw.total + al((size_t)batch * e->cfg.n_mels * Tm * 4)) return fail("workspace too small");
        char* ws = reinterpret_cast<char*>(workspace);
        float* mel = reinterpret_cast<float*>(ws + xw.total);
        hipStream_t st = (hipStream_t)stream;
        { PROF(PC_MEL, 0, (double)batch * n_samples * 4 + (double)batch * e->cfg.n_mels * Tm * 4);
          EC_TRY(launch_mel(audio, batch, n_samples, e->mel, e->cfg.n_fft, e->cfg.hop_length, e->cfg.n_mels, Tm, e->cfg.normalize, e->cfg.mean, e->cfg.std, mel, st)); }
        if (split_fused_ok(e) && e->att_out.empty()) return forward_core_split(e, mel, x_len, 1, s, xw, ws, out, out_len, st);
        return forward_core_exact(e, mel, x_len, 1, s, xw, ws, out, out_len, st);
    }
    const Workspace w = make_workspace(e, s, true);
    if (workspace_bytes < w.total) return fail("workspace too small");
    char* ws = reinterpret_cast<char*>(workspace);
    float* mel = reinterpret_cast<float*>(ws + w.mel);
    hipStream_t st = (hipStream_t)stream;
    { PROF(PC_MEL, 0, (double)batch * n_samples * 4 + (double)batch * e->cfg.n_mels * Tm * 4); EC_TRY(launch_mel(audio, batch, n_samples, e->mel, e->cfg.n_fft, e->cfg.hop_length, e->cfg.n_mels, Tm,
                      e->cfg.normalize, e->cfg.mean, e->cfg.std, mel, st)); }
    return forward_core(e, mel, x_len, 1, s, w, ws, out, out_len, st);
}

// ---- ragged batches: every utterance at its own length (the reference's result for that utterance ALONE: no pad frames exist)
static bool ragged_host_lengths(const EcEncoder* e, const int64_t* host_len, int32_t batch, int32_t n, int32_t from_audio, std::vector<int>* tm) {
    tm->resize(batch);
    for (int b = 0; b < batch; ++b) {
        const int64_t l = host_len[b];
        if (l > n || (from_audio ? l <= e->cfg.n_fft / 2 : l < 1)) return false;
        (*tm)[b] = from_audio ? (int)(l / e->cfg.hop_length + 1) : (int)l;
    }
    return true;
}

size_t effconf_encoder_workspace_bytes_ragged(const EcEncoder* e, const int64_t* x_len_host, int32_t batch, int32_t n, int32_t from_audio) {
    if (!e || !x_len_host || batch <= 0 || n <= 0) return 0;
    std::vector<int> tm;
    if (!ragged_host_lengths(e, x_len_host, batch, n, from_audio, &tm)) return 0;
    const Shapes s = make_shapes_ragged(e, tm);
    Shapes full = s; full.Tm = from_audio ? n / e->cfg.hop_length + 1 : n;      // the mel image keeps the input's row pitch
    size_t bytes = make_workspace(e, full, from_audio != 0).total;
    if (e->exact_pack) bytes = std::max(bytes, make_xworkspace(e, full).total + (from_audio ? al((size_t)batch * e->cfg.n_mels * full.Tm * 4) : 0));
    return bytes;
}

int effconf_encoder_forward_ragged(EcEncoder* e, const float* x, const int64_t* x_len, const int64_t* x_len_host, int32_t batch, int32_t n,
                                   int32_t from_audio, float* out, int32_t out_frames, int64_t* out_len, void* workspace, size_t workspace_bytes,
                                   void* stream) {
    if (!e || !e->finalized) return fail("encoder not finalized");
    if (e->exact_on && !split_fused_ok(e))
        return fail("ragged batches run on the bf16 path and in the split mode (exact_fp32 = 2, head widths <= 144); exact_fp32 = 1 keeps rectangular batches");
    if (e->exact_on && !e->att_out.empty()) return fail("attention maps of a ragged batch: bf16 path only");
    if (!x || !x_len || !x_len_host || !out || !workspace || batch <= 0 || n <= 0 || out_frames <= 0) return fail("bad argument");
    std::vector<int> tm;
    if (!ragged_host_lengths(e, x_len_host, batch, n, from_audio, &tm)) return fail("ragged lengths out of range (audio: n_fft / 2 < len <= n; mel: 1 <= len <= n)");
    Shapes s = make_shapes_ragged(e, tm);
    if (s.Tout.back() > out_frames) return fail("out_frames smaller than the longest utterance's output");
    s.Tm = from_audio ? n / e->cfg.hop_length + 1 : n;        // pitch of the mel image = the input's row pitch (every utterance masks at its own length)
    if (e->exact_on) {       // split mode on the fused kernels: mel at the tail of the exact workspace
        const XWorkspace xw = make_xworkspace(e, s);
        if (workspace_bytes < xw.total + (from_audio ? al((size_t)batch * e->cfg.n_mels * s.Tm * 4) : 0)) return fail("workspace too small");
        char* wsx = reinterpret_cast<char*>(workspace);
        hipStream_t stx = (hipStream_t)stream;
        const float* melx = x;
        if (from_audio) {
            float* m = reinterpret_cast<float*>(wsx + xw.total);
            hipStream_t st = stx;
            PROF(PC_MEL, 0, (double)batch * n * 4 + (double)batch * e->cfg.n_mels * s.Tm * 4);
            EC_TRY(launch_mel(x, batch, n, e->mel, e->cfg.n_fft, e->cfg.hop_length, e->cfg.n_mels, s.Tm, e->cfg.normalize, e->cfg.mean, e->cfg.std, m, st, x_len));
            melx = m;
        }
        return forward_core_split(e, melx, x_len, from_audio, s, xw, wsx, out, out_len, stx, out_frames);
    }
    const Workspace w = make_workspace(e, s, from_audio != 0);
    if (workspace_bytes < w.total) return fail("workspace too small");
    char* ws = reinterpret_cast<char*>(workspace);
    hipStream_t st = (hipStream_t)stream;
    const float* mel = x;
    if (from_audio) {
        float* m = reinterpret_cast<float*>(ws + w.mel);
        PROF(PC_MEL, 0, (double)batch * n * 4 + (double)batch * e->cfg.n_mels * s.Tm * 4);
        EC_ABL(16, EC_TRY(launch_mel(x, batch, n, e->mel, e->cfg.n_fft, e->cfg.hop_length, e->cfg.n_mels, s.Tm, e->cfg.normalize, e->cfg.mean, e->cfg.std, m, st, x_len)));
        mel = m;
    }
    return forward_core(e, mel, x_len, from_audio, s, w, ws, out, out_len, st, out_frames);
}

int effconf_mel_frontend(EcEncoder* e, const float* audio, int32_t batch, int32_t n_samples, float* mel, void* stream) {
    if (!e || !e->finalized) return fail("encoder not finalized");
    const int Tm = n_samples / e->cfg.hop_length + 1;
    EC_TRY(launch_mel(audio, batch, n_samples, e->mel, e->cfg.n_fft, e->cfg.hop_length, e->cfg.n_mels, Tm,
                      e->cfg.normalize, e->cfg.mean, e->cfg.std, mel, (hipStream_t)stream));
    return 0;
}

#ifdef EFFCONF_DEBUG_ABI        // libeffconf_debug.so only (include/effconf_debug.h)
int effconf_debug_mel(EcEncoder* e, int32_t variant, int32_t extra_lds, const float* audio, int32_t batch, int32_t n_samples, float* mel,
                      uint32_t* counters, void* stream) {
    if (!e || !e->finalized) return fail("encoder not finalized");
    const int Tm = n_samples / e->cfg.hop_length + 1;
    EC_TRY(launch_mel_debug(variant, extra_lds, audio, batch, n_samples, e->mel, e->cfg.n_fft, e->cfg.hop_length, e->cfg.n_mels, Tm,
                            e->cfg.normalize, e->cfg.mean, e->cfg.std, mel, counters, (hipStream_t)stream));
    return 0;
}

int effconf_debug_neighbour(int32_t kind, int32_t blocks, int32_t lds_bytes, int32_t iters, float* buf, size_t n_floats, void* stream) {
    EC_TRY(launch_debug_neighbour(kind, blocks, lds_bytes, iters, buf, n_floats, (hipStream_t)stream));
    return 0;
}

#endif

int effconf_relpos_attention(const uint16_t* qu, const uint16_t* k, const uint16_t* v, const uint16_t* e, const float* dvu, int32_t dvu_ld,
                             const int32_t* lens, int32_t batch, int32_t heads, int32_t frames, int32_t group, int32_t dim, uint16_t* out,
                             int32_t ld_out, int32_t variant, void* stream) {
    if (!qu || !k || !v || !e || !dvu || !lens || !out) return fail("null argument");
    if (batch <= 0 || heads <= 0 || frames <= 0 || group <= 0 || !(group & 1) || dim <= 0 || (group * dim) % heads) return fail("bad attention shape");
    AttnParams ap{};
    const int Tp = ec_round_up(frames, group), Tg = Tp / group, d = group * dim / heads, dpad = ec_round_up(d, 32);
    if (dpad > 192 || dvu_ld < dpad || ld_out < dim) return fail("unsupported head width / leading dimension");
    ap.qu = qu; ap.kh = k; ap.vt = v; ap.eh = e; ap.dvu = dvu; ap.dvu_ld = dvu_ld; ap.lens = lens;
    ap.B = batch; ap.H = heads; ap.T = frames; ap.G = group; ap.D = dim; ap.d = d; ap.dpad = dpad; ap.Tg = Tg; ap.Tgp = ec_round_up(Tg, 8);
    ap.q_bstride = (long long)Tp * dim; ap.q_hstride = d; ap.q_rowstride = group * dim; ap.e_hstride = d; ap.e_rowstride = group * dim;
    ap.out = out; ap.ldo = ld_out; ap.scale = 1.0f / std::sqrt((float)d);
    ap.band_l = ap.band_r = 1 << 30;          // full context (the streaming variants are tested end to end against the reference goldens)
    if (variant == 0) { EC_TRY(launch_relpos_attention(ap, (hipStream_t)stream)); return 0; }
    if ((variant != 1 && variant != 2) || !relpos_attention2_supported(dpad)) return fail("attention variant not available for this head width");
    EC_TRY(launch_relpos_attention2(ap, variant, (hipStream_t)stream));
    return 0;
}

// ---- per-kernel entry points: one module of a block on the product kernels, unit-testable against the reference's
// per-module outputs (tests/golden/tiny_*.npz: trace/blocks.N.ffn1 | conv | out, trace/linear)
size_t effconf_module_workspace_bytes(const EcEncoder* e, int32_t batch, int32_t frames) {
    if (!e || batch <= 0 || frames <= 0) return 0;
    size_t mx = 0;
    for (const EcBlock& b : e->blocks) {
        const size_t D = (size_t)std::max(b.dim_model, b.dim_expand);
        mx = std::max(mx, D * ((size_t)b.ff_ratio + 4) * 2 + 64);
    }
    const size_t rows = (size_t)batch * frames;
    size_t sub = 0;
    {   // subsampler scratch: frames = mel frames here
        const int L = e->cfg.sub_layers, C = e->cfg.sub_filters[L - 1];
        int F = e->cfg.n_mels; for (int i = 0; i < L; ++i) F /= 2;
        const size_t t1 = (frames - 1) / 2 + 1;
        sub = al((size_t)batch * t1 * C * F * 2) + (L == 2 ? al((size_t)batch * (e->cfg.n_mels / 2) * t1 * ec_round_up(e->cfg.sub_filters[0], 64) * 2) : 0);
    }
    return std::max(al(rows * mx) + 4 * 256, sub + 256);
}

int effconf_ffn(EcEncoder* e, int32_t block, int32_t which, const float* x, int32_t rows, float* y, void* workspace, size_t workspace_bytes,
                void* stream) {
    if (!e || !e->finalized) return fail("encoder not finalized");
    if (block < 0 || block >= (int)e->blocks.size() || (which != 1 && which != 2) || !x || !y || rows <= 0 || !workspace) return fail("bad argument");
    const EcBlock& b = e->blocks[block];
    const BlockW& W = e->bw[block];
    const int D = which == 1 ? b.dim_model : b.dim_expand, F = D * b.ff_ratio;
    hipStream_t st = (hipStream_t)stream;
    char* ws = reinterpret_cast<char*>(workspace);
    const size_t a_bytes = al((size_t)rows * ld8(D) * 2), h_bytes = al((size_t)rows * F * 2);
    if (workspace_bytes < a_bytes + h_bytes) return fail("workspace too small");
    bf16_t* a = reinterpret_cast<bf16_t*>(ws);
    bf16_t* hbuf = reinterpret_cast<bf16_t*>(ws + a_bytes);
    if (y != x && hipMemcpyAsync(y, x, (size_t)rows * D * 4, hipMemcpyDeviceToDevice, st) != hipSuccess) return fail("copy failed");
    const LNp& ln = which == 1 ? W.ln_ffn1 : W.ln_ffn2;
    const PackedLinear &L1 = which == 1 ? W.ffn1_a : W.ffn2_a, &L2 = which == 1 ? W.ffn1_b : W.ffn2_b;
    const bf16_t* w2p = which == 1 ? W.ffn1_bp : W.ffn2_bp;
    if (ffn_fused_supported(D)) return run_ffn(e, st, a, rows, D, L1, L2, w2p, y, hbuf, &ln);       // pre-norm in the kernel's prologue
    EC_TRY(launch_layernorm(y, rows, D, ln.g, ln.b, nullptr, a, ld8(D), nullptr, nullptr, st));
    return run_ffn(e, st, a, rows, D, L1, L2, w2p, y, hbuf);
}

int effconf_conv_module(EcEncoder* e, int32_t block, const float* x, int32_t batch, int32_t frames, float* y, void* workspace,
                        size_t workspace_bytes, void* stream) {
    if (!e || !e->finalized) return fail("encoder not finalized");
    if (block < 0 || block >= (int)e->blocks.size() || !x || !y || batch <= 0 || frames <= 0 || !workspace) return fail("bad argument");
    const EcBlock& b = e->blocks[block];
    const BlockW& W = e->bw[block];
    const int D = b.dim_model, De = b.dim_expand, T = frames, To = (T - 1) / b.conv_stride + 1, M = batch * T, Mo = batch * To;
    hipStream_t st = (hipStream_t)stream;
    char* ws = reinterpret_cast<char*>(workspace);
    const size_t a_bytes = al((size_t)M * ld8(D) * 2), g_bytes = al((size_t)M * ld8(De) * 2), c_bytes = al((size_t)Mo * ld8(De) * 2);
    if (workspace_bytes < a_bytes + g_bytes + c_bytes) return fail("workspace too small");
    bf16_t* a = reinterpret_cast<bf16_t*>(ws);
    bf16_t* gbuf = reinterpret_cast<bf16_t*>(ws + a_bytes);
    bf16_t* cbuf = reinterpret_cast<bf16_t*>(ws + a_bytes + g_bytes);
    // LayerNorm -> pointwise-1 + GLU (modules.py:511-514)
    if (rs_gemm_supported(D)) {
        EC_TRY(run_rs_or_tiled(e, PC_GEMM_OTHER, st, a, ld8(D), M, W.pw1, 2, EPI_GLU_BF16, gbuf, ld8(De), nullptr, 0, 1.f, x, &W.ln_conv));
    } else {
        EC_TRY(launch_layernorm(x, M, D, W.ln_conv.g, W.ln_conv.b, nullptr, a, ld8(D), nullptr, nullptr, st));
        EC_TRY(run_rs_or_tiled(e, PC_GEMM_OTHER, st, a, ld8(D), M, W.pw1, 2, EPI_GLU_BF16, gbuf, ld8(De)));
    }
    // depthwise conv + BatchNorm + Swish (modules.py:516-518), pointwise-2 (modules.py:519)
    EC_TRY(launch_dwconv(gbuf, batch, T, To, De, ld8(De), W.dw_w, W.dw_b, b.kernel_size, b.conv_stride, cbuf, st, nullptr, e->cfg.causal, dw_mfma_table(e, W.dw_a, b.kernel_size)));
    return run_rs_or_tiled(e, PC_GEMM_OTHER, st, cbuf, ld8(De), Mo, W.pw2, 1, EPI_F32, y, De);
}

int effconf_subsample(EcEncoder* e, const float* mel, int32_t batch, int32_t n_frames, float* y, void* workspace, size_t workspace_bytes,
                      void* stream) {
    if (!e || !e->finalized) return fail("encoder not finalized");
    if (!mel || !y || batch <= 0 || n_frames <= 0 || !workspace) return fail("bad argument");
    const Shapes s = make_shapes(e, batch, n_frames);
    const int L = e->cfg.sub_layers, C = e->cfg.sub_filters[L - 1];
    int F = e->cfg.n_mels; for (int i = 0; i < L; ++i) F /= 2;
    const size_t sub_bytes = al((size_t)batch * s.T1 * C * F * 2);
    const size_t tl1 = (n_frames - 1) / 2 + 1;
    const size_t act_bytes = L == 2 ? al((size_t)batch * (e->cfg.n_mels / 2) * tl1 * ec_round_up(e->cfg.sub_filters[0], 64) * 2) : 0;
    if (workspace_bytes < sub_bytes + act_bytes) return fail("workspace too small");
    char* ws = reinterpret_cast<char*>(workspace);
    return run_subsample_linear(e, (hipStream_t)stream, mel, batch, n_frames, s.T1, reinterpret_cast<bf16_t*>(ws), reinterpret_cast<bf16_t*>(ws + sub_bytes), y);
}

int effconf_layernorm_residual(EcEncoder* e, int32_t block, int32_t which, const float* x, const float* r, float alpha, int32_t rows, float* y,
                               void* stream) {
    if (!e || !e->finalized) return fail("encoder not finalized");
    if (block < 0 || block >= (int)e->blocks.size() || which < 0 || which > 4 || !x || !y || rows <= 0) return fail("bad argument");
    const EcBlock& b = e->blocks[block];
    const BlockW& W = e->bw[block];
    const LNp* ln[5] = {&W.ln_ffn1, &W.ln_att, &W.ln_conv, &W.ln_ffn2, &W.ln_out};
    const int D = which >= 3 ? b.dim_expand : b.dim_model;
    EC_TRY(launch_layernorm_residual(x, r, alpha, rows, D, ln[which]->g, ln[which]->b, y, (hipStream_t)stream));
    return 0;
}

#ifdef EFFCONF_DEBUG_ABI        // libeffconf_debug.so only (include/effconf_debug.h)
int effconf_debug_gemm(const uint16_t* a, int32_t lda, const uint16_t* w, int32_t ldw, const float* bias, int32_t m, int32_t n, int32_t k,
                       int32_t epi, int32_t wide, void* c, int32_t ldc, const float* r, int32_t ldr, float alpha, void* stream) {
    if (!a || !w || !bias || !c) return fail("null argument");
    if (epi < EPI_F32 || epi > EPI_GLU_BF16 || (epi == EPI_RESID_F32 && !r)) return fail("epilogue: 0 f32, 1 bf16, 2 swish bf16, 3 residual f32, 4 GLU bf16");
    if (wide < 0 || wide > 3) return fail("wide: 0 .. 3");
    GemmParams p{};
    p.A = a; p.lda = lda; p.W = w; p.ldw = ldw; p.bias = bias; p.M = m; p.N = n; p.K = k;
    p.C = c; p.ldc = ldc; p.R = r; p.ldr = ldr; p.alpha = alpha; p.wide = wide;
    if (wide >= 2 && !gemm256_supported(p, epi)) return fail("gemm256 does not take this shape / alignment");
    EC_TRY(launch_gemm(p, epi, (hipStream_t)stream));
    return 0;
}

int effconf_debug_sx_gemm(const float* a, int32_t lda, const uint16_t* w_hi, const uint16_t* w_lo, int32_t ldh, const float* bias, int32_t m, int32_t n,
                          int32_t k, int32_t epi, float* c, int32_t ldc, const float* r, int32_t ldr, float alpha, void* stream) {
    SxGemmParams q{};
    q.g.A = a; q.g.lda = lda; q.g.bias = bias; q.g.M = m; q.g.N = n; q.g.K = k; q.g.C = c; q.g.ldc = ldc; q.g.R = r; q.g.ldr = ldr; q.g.alpha = alpha; q.g.epi = epi;
    q.Whi = w_hi; q.Wlo = w_lo; q.ldh = ldh;
    const int rc = launch_sx_gemm(q, reinterpret_cast<hipStream_t>(stream));
    return rc ? fail("sx_gemm launch failed rc=" + std::to_string(rc)) : 0;
}

int effconf_debug_pack_dwconv_mfma(const float* w_kc, int32_t ksize, int32_t channels, uint16_t* dst, size_t dst_elems) {
    if (!w_kc || !dst || channels <= 0) return fail("null argument");
    if (!dwconv_mfma_supported(ksize, 1)) return fail("kernel size: 15, 31 or 7");
    if (dst_elems != (size_t)channels * 4 * dwconv_mfma_groups(ksize) * 8) return fail("dst: channels * 4 * groups * 8 bf16");
    pack_dwconv_mfma(w_kc, ksize, channels, dst);          // host memory in, host memory out
    return 0;
}

int effconf_debug_dwconv(const uint16_t* g, int32_t batch, int32_t frames, int32_t channels, int32_t ld, const float* w_kc_host, const float* bias_host,
                         int32_t ksize, int32_t stride, int32_t use_mfma, int32_t causal, uint16_t* out, void* stream) {
    if (!g || !w_kc_host || !bias_host || !out || batch <= 0 || frames <= 0 || channels <= 0 || ld < channels || ld % 8) return fail("bad argument");
    if (use_mfma && !dwconv_mfma_supported(ksize, stride)) return fail("dwconv_mfma_kernel: stride 1, kernel size 15, 31 or 7");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    float *dw = nullptr, *db = nullptr; uint16_t* dt = nullptr;
    std::vector<uint16_t> tab;
    if (use_mfma) { tab.resize((size_t)channels * 4 * dwconv_mfma_groups(ksize) * 8); pack_dwconv_mfma(w_kc_host, ksize, channels, tab.data()); }
    // test-only entry: temporary device copies of the taps, synchronous
    if (hipMalloc(&dw, (size_t)ksize * channels * 4) != hipSuccess || hipMalloc(&db, (size_t)channels * 4) != hipSuccess ||
        (use_mfma && hipMalloc(&dt, tab.size() * 2) != hipSuccess)) return fail("hipMalloc failed");
    (void)hipMemcpy(dw, w_kc_host, (size_t)ksize * channels * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(db, bias_host, (size_t)channels * 4, hipMemcpyHostToDevice);
    if (use_mfma) (void)hipMemcpy(dt, tab.data(), tab.size() * 2, hipMemcpyHostToDevice);
    const int to = (frames - 1) / stride + 1;
    const int rc = launch_dwconv(g, batch, frames, to, channels, ld, dw, db, ksize, stride, out, st, nullptr, causal, use_mfma ? dt : nullptr);
    (void)hipStreamSynchronize(st);
    (void)hipFree(dw); (void)hipFree(db); if (dt) (void)hipFree(dt);
    return rc ? fail("launch_dwconv failed rc=" + std::to_string(rc)) : 0;
}

int effconf_debug_sxf_ffn(EcEncoder* e, int32_t block, int32_t which, const float* x, int32_t rows, float* y, int32_t with_norm, int32_t ablate, void* stream) {
    if (!e || !e->finalized || block < 0 || block >= (int)e->blocks.size() || which < 1 || which > 2 || !x || !y || rows <= 0) return fail("bad argument");
    const BlockW& W = e->bw[block];
    if (!W.xf_img[which - 1]) return fail("no fused split FFN image for this block (finalize with exact_fp32 = 2; width not built)");
    SxfFfnParams fp{};
    const int D = which == 2 ? e->blocks[block].dim_expand : e->blocks[block].dim_model;
    fp.X = x; fp.ldx = D; fp.Y = y; fp.ldy = D; fp.wimg = W.xf_img[which - 1]; fp.b2 = W.xf_b2[which - 1]; fp.M = rows; fp.D = D; fp.nchunk = W.xf_nch[which - 1];
    if (with_norm) { fp.ln_g = W.ln_out.g; fp.ln_b = W.ln_out.b; }
    fp.ablate = ablate;
    EC_TRY(launch_sxf_ffn(fp, reinterpret_cast<hipStream_t>(stream)));
    return 0;
}

int effconf_debug_spin(double microseconds, void* stream) {
    if (launch_debug_spin(microseconds, reinterpret_cast<hipStream_t>(stream)) != 0) return fail("spin launch failed");
    return 0;
}

int effconf_debug_lds_fill(int32_t mode, int32_t blocks, int32_t waves, const void* src, size_t window, int32_t kib_per_wave, int32_t passes, uint64_t* out, void* stream) {
    EC_TRY(launch_debug_lds_fill(mode, blocks, waves, reinterpret_cast<const char*>(src), window, kib_per_wave, passes, reinterpret_cast<unsigned long long*>(out), (hipStream_t)stream));
    return 0;
}

int effconf_debug_victim(int32_t kind, int32_t blocks, int32_t iters, float* out, void* stream) {
    EC_TRY(launch_debug_victim(kind, blocks, iters, out, (hipStream_t)stream));
    return 0;
}

#endif

int effconf_ctc_greedy(EcEncoder* e, const float* enc_out, const int64_t* out_len, int32_t batch, int32_t t_out,
                       int32_t* labels, int32_t* label_len, float* logits, void* workspace, size_t workspace_bytes, void* stream) {
    if (!e || !e->finalized) return fail("encoder not finalized");
    if (!e->fc_wt) return fail("no CTC head (fc.weight / fc.bias) loaded");
    if (workspace_bytes < (size_t)batch * t_out * 4) return fail("workspace too small");
    int* preds = reinterpret_cast<int*>(workspace);
    hipStream_t st = (hipStream_t)stream;
    // bf16 path: split-bf16 operands on the bf16 matrix pipe (ctc_mfma = 2, the default); fp32-operand mode: the fp32 matrix pipe, bit-identical
    // to the VALU kernel (the label-exact mode keeps the reference's fp32 head)
    const int mode = e->exact_on && e->ctc_mfma == 2 ? 1 : e->ctc_mfma;
    if (mode == 2 && e->fc_hi && launch_ctc_split(enc_out, batch * t_out, e->blocks.back().dim_expand, e->fc_hi, e->fc_lo, e->fc_b, e->cfg.vocab_size,
                                                  preds, logits, st) == 0) {
    } else
    EC_TRY(launch_ctc_argmax(enc_out, batch * t_out, e->blocks.back().dim_expand, e->fc_wt, e->fc_b, e->cfg.vocab_size,
                             preds, logits, st, mode != 0));
    EC_TRY(launch_ctc_collapse(preds, out_len, batch, t_out, labels, label_len, st));
    return 0;
}

int effconf_ctc_greedy_bf16(EcEncoder* e, const uint16_t* enc_out_bf16, const int64_t* out_len, int32_t batch, int32_t t_out,
                            int32_t* labels, int32_t* label_len, float* logits, void* workspace, size_t workspace_bytes, void* stream) {
    if (!e || !e->finalized) return fail("encoder not finalized");
    if (e->exact_on) return fail("effconf_ctc_greedy_bf16 belongs to the bf16 path (the label-exact modes keep fp32 rows and the fp32 head)");
    if (!e->fc_hi || !e->fc_lo) return fail("no CTC head (fc.weight / fc.bias) loaded");
    if (workspace_bytes < (size_t)batch * t_out * 4) return fail("workspace too small");
    int* preds = reinterpret_cast<int*>(workspace);
    hipStream_t st = (hipStream_t)stream;
    EC_TRY(launch_ctc_split(reinterpret_cast<const float*>(enc_out_bf16), batch * t_out, e->blocks.back().dim_expand, e->fc_hi, e->fc_lo, e->fc_b,
                            e->cfg.vocab_size, preds, logits, st, 1));
    EC_TRY(launch_ctc_collapse(preds, out_len, batch, t_out, labels, label_len, st));
    return 0;
}

/* Attention maps (reference encoders.py:126-142): maps[k] = device buffer of batch x heads[k] x tg[k] x tg[k] floats for block k, or null. */
int effconf_encoder_attention_dims(EcEncoder* e, int32_t n, int32_t from_audio, int32_t* heads, int32_t* tg) {
    if (!e || !e->finalized || !heads || !tg) return fail("effconf_encoder_attention_dims: null argument / encoder not finalized");
    if (n < 0) return fail("effconf_encoder_attention_dims: negative length");
    const Shapes s = make_shapes(e, 1, from_audio ? n / e->cfg.hop_length + 1 : n);
    for (size_t k = 0; k < e->blocks.size(); ++k) {
        const EcBlock& b = e->blocks[k];
        heads[k] = b.num_heads;
        tg[k] = ec_round_up(s.Tin[k], b.group_size) / b.group_size;
    }
    return 0;
}

int effconf_encoder_set_attention_outputs(EcEncoder* e, float* const* maps, int32_t n_blocks) {
    if (!e) return fail("null encoder");
    if (!maps || n_blocks == 0) { e->att_out.clear(); return 0; }
    if (n_blocks != (int)e->blocks.size()) return fail("effconf_encoder_set_attention_outputs: one pointer per block (null = skip that block)");
    e->att_out.assign(maps, maps + n_blocks);
    return 0;
}

int effconf_encoder_set_option(EcEncoder* e, const char* name, int32_t value) {
    if (!e || !name) return fail("null argument");
    if (!strcmp(name, "fuse_subsample")) { if (value < 0 || value > 3) return fail("fuse_subsample: 0, 1, 2 or 3"); e->fuse_subsample = value; return 0; }
    if (!strcmp(name, "fuse_chain")) { e->fuse_chain = value != 0; return 0; }
    if (!strcmp(name, "ctc_mfma")) { if (value < 0 || value > 2) return fail("ctc_mfma: 0 (VALU), 1 (fp32 MFMA) or 2 (split-bf16 MFMA)"); e->ctc_mfma = value; return 0; }
    if (!strcmp(name, "wide_gemm")) { if (value < 0 || (value > 3 && value < 16)) return fail("wide_gemm: 0 (by shape), 1 (never), 2 (256-column tile), 3 (128-column tile), >= 16 (by shape with this many 256 x 256 tiles as the threshold)"); e->wide_gemm = value; return 0; }
    if (!strcmp(name, "attention_v2")) { if (value != 0 && value != 1 && value != 2) return fail("attention_v2: 0, 1 or 2"); e->attention_v2 = value; return 0; }
    // former EFFCONF_* environment switches (process-global statics): per-handle options now
    if (!strcmp(name, "chain_variant")) { if (value != 0 && value != 1) return fail("chain_variant: 0 (8-wave chain workgroups) or 1 (4-wave, two per CU, at 65..128-wide stages)"); e->chain_variant = value; return 0; }
    if (!strcmp(name, "chain_full_max")) { if (value < 0 || value > 256) return fail("chain_full_max: widest stage (0..256) that runs chain A as ONE kernel"); e->chain_full_max = value; return 0; }   // widening takes effect at the next finalize (the combined constant blocks are built there)
    if (!strcmp(name, "attn_waves")) { if (value != 4 && value != 8 && value != 2) return fail("attn_waves: 4 or 8 (attention.hip workgroup size); 2: attention2.hip with two staging sets at head widths 64 / 96 (two workgroups per CU at 64: tuning)"); e->attn_waves = value; return 0; }
    if (!strcmp(name, "rs_variant")) { if (value != 0 && value != 1) return fail("rs_variant: 0 or 1 (8-wave row-stationary GEMM workgroups)"); e->rs_variant = value; return 0; }
    if (!strcmp(name, "chain_max_dim")) { e->chain_max_dim = value; return 0; }
    if (!strcmp(name, "chain_small_m")) { e->chain_small_m = value; return 0; }
    if (!strcmp(name, "chain_pair")) { if (value < 0 || value > 5) return fail("chain_pair: 0 (chain.hip everywhere), 5 (chain3.hip for chain A at padded width 256, chain2.hip mode 4 elsewhere), 1 .. 4 (chain2.hip's column-pair kernels at padded width 192 / 256; refill modes, see launch_chain2_kind)"); e->chain_pair = value; return 0; }
    if (!strcmp(name, "chain_pair_min_m")) { e->chain_pair_min_m = value; return 0; }
    if (!strcmp(name, "dwconv_mfma")) { if (value < 0 || value > 2) return fail("dwconv_mfma: 0, 1 (kernel size 15) or 2 (15 / 31 / 7)"); e->dwconv_mfma = value; return 0; }
    if (!strcmp(name, "chain_pair_min_d")) { e->chain_pair_min_d = value; return 0; }
    if (!strcmp(name, "chain_nt")) { e->chain_nt = value; return 0; }
    if (!strcmp(name, "chain_w2cm")) { e->chain_w2cm = value; return 0; }
    if (!strcmp(name, "tiled_min_k")) { e->tiled_min_k = value; return 0; }
    if (!strcmp(name, "chain_count_stores")) { e->chain_count_stores = value != 0; return 0; }
    if (!strcmp(name, "tiled_auto")) { e->tiled_auto = value; return 0; }      // 2: every layer with K in (tiled_min_k, 384] whatever the widest stage (tuning)
    if (!strcmp(name, "ffn_variant")) { if (value < 0 || value > 2) return fail("ffn_variant: 0, 1 or 2 (fused-FFN workgroup shapes)"); e->ffn_variant = value; return 0; }
    if (!strcmp(name, "head_major_odd")) { e->head_major_odd = value != 0; return 0; }
    if (!strcmp(name, "split_ffn")) { e->split_ffn = value != 0; return 0; }
    if (!strcmp(name, "split_sublin")) { e->split_sublin = value != 0; return 0; }
    if (!strcmp(name, "sub3_auto")) { e->sub3_auto = value != 0; return 0; }
    if (!strcmp(name, "split_chain")) { e->split_chain = value != 0; return 0; }
    if (!strcmp(name, "exact_attention")) { if (value < 0 || value > 2) return fail("exact_attention: 0 (tiled), 2 (tiled, 16-row workgroups) or 1 (one wave per query row)"); e->exact_attention = value; return 0; }
    if (!strcmp(name, "cache_pos_embeddings")) { e->e_cache_on = value != 0; e->e_cache.clear(); return 0; }
    if (!strcmp(name, "exact_fp32")) {
        // 0 = bf16 path, 1 = fp32 operands on the fp32 matrix pipe (exact.hip), 2 = split fp16 operand pairs on the fp16 matrix pipe (split.hip)
        if (!e->finalized) { e->exact_pack = e->exact_on = value != 0; e->exact_split = value == 2; return 0; }
        if (value && !e->exact_pack) return fail("exact_fp32 must be requested before effconf_encoder_finalize (the fp32 weights are uploaded there)");
        if (value == 2 && e->xsplit.empty()) return fail("exact_fp32 = 2 must be requested before effconf_encoder_finalize (the split weight images are built there)");
        e->exact_on = value != 0; e->exact_split = value == 2;
        return 0;
    }
    return fail(std::string("unknown option ") + name);
}

int effconf_profile_enable(EcEncoder* e, int32_t enable) {
    if (!e) return fail("null encoder");
    e->prof_on = enable != 0; e->prof_next = 0; e->prof_rec.clear();
    return 0;
}

int effconf_profile_read(EcEncoder* e, int32_t cls, double* total_ms, int64_t* launches, double* flops, double* bytes) {
    if (!e || cls < 0 || cls >= PC_COUNT) return fail("bad profile class");
    if (hipDeviceSynchronize() != hipSuccess) return fail("sync failed");
    double ms = 0, fl = 0, by = 0; int64_t n = 0;
    for (size_t i = 0; i < e->prof_rec.size(); ++i) {
        if (e->prof_rec[i].cls != cls) continue;
        float t = 0.f;
        if (hipEventElapsedTime(&t, e->prof_ev[2 * i], e->prof_ev[2 * i + 1]) != hipSuccess) return fail("event read failed");
        ms += t; fl += e->prof_rec[i].flops; by += e->prof_rec[i].bytes; ++n;
    }
    *total_ms = ms; *launches = n; *flops = fl; *bytes = by;
    return 0;
}

int effconf_encoder_set_trace(EcEncoder* e, void* dev_arena, size_t bytes) {
    if (!e) return fail("null encoder");
    e->trace_arena = reinterpret_cast<char*>(dev_arena); e->trace_bytes = bytes; e->trace_used = 0; e->trace.clear();
    return 0;
}
int32_t effconf_encoder_trace_count(const EcEncoder* e) { return e ? (int32_t)e->trace.size() : 0; }
int effconf_encoder_trace_entry(const EcEncoder* e, int32_t i, char* name64, int64_t* offset, int64_t* rows, int64_t* cols,
                                int64_t* ld, int32_t* dtype) {
    if (!e || i < 0 || i >= (int)e->trace.size()) return fail("bad trace index");
    const TraceEntry& t = e->trace[i];
    memcpy(name64, t.name, 64);
    *offset = t.offset; *rows = t.rows; *cols = t.cols; *ld = t.ld; *dtype = t.dtype;
    return 0;
}

}  // extern "C"
