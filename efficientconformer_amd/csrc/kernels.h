// Internal launch API of libeffconf: one function per HIP kernel family.  All pointers are device
// memory owned by the caller; every launch goes to the stream passed in; nothing allocates or syncs.
#pragma once
#include "common.h"

// ---------------------------------------------------------------- ragged batches
// Every utterance at its own length in ONE concatenated row space (no pad frames: the reference's result for that utterance alone).
// Descriptor of the rows at one position of the network: device arrays in the workspace (lengths_ragged_kernel fills them) + host totals.
struct RaggedRows {
    const int* off;    // dev [n + 1]: first row of every utterance; an utterance's rows are padded to a multiple of the attention group size
    const int* len;    // dev [n]: valid frames per utterance
    int n;             // utterances
    int rows;          // off[n]
    int tmax;          // longest utterance (frames)
};

// ragged batch view of a frame-mixing kernel with a stride (depthwise conv, conv_res decimation): per-utterance lengths and first rows on
// the input and on the output side (dev arrays), the prefix sums of the utterances' 128-frame output tiles and their host total
struct RaggedConv {
    const int *in_off, *in_len, *out_off, *out_len;   // [n + 1], [n], [n + 1], [n]
    const int* tile_off; int tiles;                   // [n + 1] prefix of ceil(padded output rows / 128); tile_off[n]
    int n, out_rows;                                  // utterances; out_off[n]
};

// ---------------------------------------------------------------- GEMM  (gemm.hip)
// C[m, n] = sum_k A[m, k] * W[n, k] + bias[n]   (A bf16 row-major, W bf16 "Linear weight" layout [N][K])
// W must be packed by pack_weight_bf16(): [round_up(N,128)][round_up(K,64)] zero padded; bias [round_up(N,128)].
enum GemmEpilogue {
    EPI_F32 = 0,         // C f32 [M][ldc]            = acc + bias
    EPI_BF16 = 1,        // C bf16 [M][ldc]           = acc + bias
    EPI_SWISH_BF16 = 2,  // C bf16 [M][ldc]           = swish(acc + bias)
    EPI_RESID_F32 = 3,   // C f32 [M][ldc]            = R[m][n] + alpha * (acc + bias)   (R may alias C)
    EPI_GLU_BF16 = 4,    // C bf16 [M][ldc], N/2 cols = a * sigmoid(b); W rows interleaved in blocks of 32 (a|b)
    EPI_QKV = 5,         // scatter to head-major Qu, Qv, K and key-major V^T (see GemmParams)
    EPI_HEADS = 6,       // scatter a single matrix to head-major [H][rows/G][dpad]   (positional E)
    EPI_QKV_NAT = 7,     // Q+u, Q+v, K, V as plain row-major [B*Tp][D] bf16 ("natural" layout; attention does the head split)
};

struct GemmParams {
    const bf16_t* A; int lda;          // lda multiple of 8 (16-byte rows); columns >= K are ignored (masked)
    int a_rows, a_pitch, a_stride;     // row map: src_row(m) = (m / a_rows) * a_pitch + (m % a_rows) * a_stride; a_rows==0: identity
    const bf16_t* W; int ldw;          // packed weight, ldw = round_up(K, 64)
    const float* bias;
    int M, N, K;
    void* C; int ldc;
    const float* R; int ldr; float alpha;
    // EPI_QKV / EPI_HEADS: rows m = (b, t) with t < T; grouped attention view
    int T, G, H, D, d, dpad, Tg, Tgp;
    bf16_t *qu, *kh, *vt;              // Q + u, K, V: [B][H][Tg][dpad] (head major) or rows (b*Tp + t)*D (natural); Q + v is derived in the attention kernel
    const float *u, *v;                // [D] content / position bias (attentions.py:474-475)
    // row-stationary kernels only: if X != null the A operand is LayerNorm(X[m][0..K)) (eps 1e-6, fp32 two-pass statistics)
    // computed in the prologue from the fp32 residual stream instead of being read from A
    const float* X; int ldx; const float *ln_g, *ln_b;
    int wide;                          // launch_gemm only: 0 pick by shape, 1 never gemm256.hip, 2 / 3 force its 256- / 128-column tile
    int rs_variant;                    // launch_rs_gemm only (option "rs_variant"): 1 = 8-wave workgroups (256 rows) at KS = 8 / 16
};
int launch_gemm(const GemmParams& p, int epi, hipStream_t s);
// gemm256.hip: 256 x bn x 64 tiles, both operands by LDS-DMA (bn = 256 or 128); the same operands and epilogues as launch_gemm
bool gemm256_supported(const GemmParams& p, int epi);
int launch_gemm256(const GemmParams& p, int epi, int bn, hipStream_t s);

// ---------------------------------------------------------------- row-stationary fused kernels  (rsgemm.hip)
// y = x + alpha * (Swish(a W1^T + b1) W2^T + b2);  a = bf16 LayerNorm(x) [M][lda], x/y fp32 [M][ld] (may alias)
// W1 : bf16 [Fp][ldw1] (rows = hidden units, zero padded to Fp = round_up(F, 32) and ldw1 >= round_up(D, 64))
// W2 : bf16 [>= round_up(D,128)][ldw2 = Fp], hidden index permuted within groups of 16 (see rsgemm.hip)
struct FfnParams {
    const bf16_t* A; int lda;
    const float* X; int ldx;
    float* Y; int ldy;
    const bf16_t* W1; int ldw1; const float* b1;
    const bf16_t* W2; int ldw2; const float* b2;
    int M, D, Fp;
    float alpha;
    const float *ln_g, *ln_b;          // if non-null: a = LayerNorm(X) computed in the prologue (A is ignored)
    int variant;                       // option "ffn_variant": workgroup shape overrides (tuning)
};
// single row-stationary GEMM (K <= 384); epi: 0 residual fp32, 1 fp32, 2 GLU bf16 (N = packed a|b rows), 3 QKV head-major
// scatter, 4 QKV natural layout (weight rows permuted inside every 32-row chunk, see pack_linear_chunkperm)
// (GemmParams fields as for launch_gemm; `vt` receives V in the SAME head-major row-major layout as K)
bool rs_gemm_supported(int K);
bool rs_gemm_resident_supported(int K, int N);   // residual / fp32 epilogues keep the whole output row in registers
int launch_rs_gemm(const GemmParams& p, int epi, hipStream_t s);
bool ffn_fused_supported(int D);
int launch_ffn_fused(const FfnParams& p, hipStream_t s);


// ---------------------------------------------------------------- fused row-local chains  (chain.hip)
// One kernel runs a whole row-local stretch of the Conformer block on a 32-row tile per wave, the fp32 residual row staying in
// registers between GEMMs (see chain.hip).  All weights are K-permuted per 16 (pack_linear_kperm), rows padded to 64.
struct ChainGemm { const bf16_t* w; int ldw; const float* bias; int nchunks; };        // nchunks = 64-row chunks
struct ChainFfn { const bf16_t* w1; int ldw1; const float* b1; const bf16_t* w2; int ldw2; const float* b2; int Fp; const bf16_t* w2cm; };   // w2cm: w2 chunk-major (chain2.hip; may be null)   // w2, b2 pre-scaled by 1/2
struct ChainLn { const float* g; const float* b; };
struct ChainParams {
    int M, D;
    const float* X; int ldx;            // residual stream in
    float* Y; int ldy;                  // residual stream out (may alias X)
    const bf16_t* A; int lda;           // bf16 activation feeding g0 (depthwise-conv output / attention output)
    ChainGemm g0;                       // pointwise-2 or attention output projection: x += g0(A)
    ChainLn ln[4];                      // only ln[1] (block norm, applied in fp32) is read: the pre-norms' gamma / beta are folded into the following weights at pack time
    ChainFfn f[2];                      // chain A: ffn2, ffn1 (of the next block)
    ChainGemm g1;                       // chain A: stacked QKV (natural layout, chunk-permuted rows); chain B: pointwise-1 (a|b interleaved per 32)
    bf16_t *qu, *kh, *vt; const float *u, *v; int T, Tp;          // QKV outputs (Q + u, K, V): rows (b, t) -> (b*Tp + t)*D
    bf16_t* glu; int ldg, Ng;           // GLU output [M][ldg], Ng channels
    const float* consts;                // biases / block-norm gamma, beta / u, v as ONE zero padded block laid out by chain_const_layout
    int variant;                        // option "chain_variant": 1 = 4-wave workgroups (two per CU) at KS = 8
    int count_stores;                   // option "chain_count_stores" (measurement only, default 0): 1 = the counted ring waits of rounds 3 - 6 that also allow the global stores
                                        // since the last barrier to stay outstanding - UNSAFE: a store can retire before an older LDS-DMA (chain.hip, advance())
    int small_m;                        // option "chain_small_m": launches of at most this many rows use 2-wave workgroups (64 rows): see launch_chain_kind
    int pair_min_d;                     // option "chain_pair_min_d": narrower stages stay on chain.hip
    int w2cm;                           // option "chain_w2cm": chain2.hip streams the FFN second weights from their chunk-major images
    int nt;                             // option "chain_nt": non-temporal hints on the activation loads (1) / stores (2) of chain2.hip (3 = both)
    int pair_small_max;                 // option "chain_pair_min_m" - 1: launches of at most this many rows stay on chain.hip's shapes (-1: none)
    int pair;                           // option "chain_pair": != 0 = the column-pair kernels of chain2.hip where they exist (padded width 192 / 256); 1 = burst refills, 2 = hooked
};
enum { CHAIN_B = 0, CHAIN_A_FULL = 1, CHAIN_A_HEAD = 2, CHAIN_A_TAIL = 3 };   // HEAD: first block (no previous tail); TAIL: last block (no next head)
bool chain_supported(int D);
int chain_padded_width(int D);       // padded row width of the kernel instance used for width D (units of the constant block)
int chain_const_layout(const ChainParams& p, int kind, int (&nf)[8]);   // float offsets of the constant block; returns its size in floats
bool chain_head_supported(int D);   // FFN1 + Q/K/V half (chain A head / full)
bool chain_tail_supported(int D);   // pointwise-2 + FFN2 + block norm half
bool chain_full_supported(int D, int dmax = 192);   // tail + next block's head in one kernel (dmax: option "chain_full_max")
int launch_chain(const ChainParams& p, int kind, hipStream_t s);
// chain2.hip: the same chains with a PAIR of waves per 32 rows (column halves), 8-wave 128-row workgroups at two waves per SIMD; rows bit-identical to chain.hip's
bool chain2_supported(int D);
int launch_chain2(const ChainParams& p, int kind, hipStream_t s);
// chain3.hip: chain A at padded width 256 with the row tile spread over THREE waves (first GEMM + Swish | second GEMM on a column half + the weight stream,
// twice): 12-wave 128-row workgroups, three waves per SIMD; needs the chunk-major second FFN weights (ChainFfn::w2cm); rows bit-identical to chain.hip's
bool chain3_supported(int D);
int launch_chain3(const ChainParams& p, int kind, hipStream_t s);

// ---------------------------------------------------------------- normalisation / casts  (norm.hip)
// y = LayerNorm(x) over the last dim (eps 1e-6), two-pass fp32 statistics, one wave per row.
// out_bf16 (ld = round_up(D,8), pad columns zeroed) and/or out_f32 (ld = D) may be null.
// If gamma2 != null a second LayerNorm is applied to the first result and *that* goes to out_bf16
// (block-final norm fused with the next block's FFN1 pre-norm).
int launch_layernorm(const float* x, int M, int D, const float* gamma, const float* beta,
                     float* out_f32, bf16_t* out_bf16, int ld_bf16,
                     const float* gamma2, const float* beta2, hipStream_t s);
// y = LayerNorm(x + alpha * r) (r may be null), fp32 in / out, eps 1e-6
int launch_layernorm_residual(const float* x, const float* r, float alpha, int M, int D, const float* gamma, const float* beta, float* y, hipStream_t s);
// out[m][:] = bf16(x[src_row(m)][:])   (strided frame decimation + cast for conv_res, blocks.py:106-110)
int launch_cast_rows(const float* x, int D, int rows_per_batch, int stride, int out_rows_per_batch, int batch,
                     bf16_t* out, int ld_out, hipStream_t s, const RaggedConv* rc = nullptr);
// ragged rows -> the caller's (B, t_out, D) fp32 output, zero filled behind every utterance's own last frame
int launch_emit_rows(const float* x, int D, const int* off, const int* len, int batch, int t_out, float* out, hipStream_t s);

// ---------------------------------------------------------------- attention  (attention.hip)
struct AttnParams {
    const bf16_t *qu, *kh, *vt, *eh;        // see GemmParams; eh [H][2Tg-1][dpad]
    const float* dvu; int dvu_ld;           // (v - u) per head column [H][dvu_ld] (zero beyond d): Q + v = (Q + u) + (v - u)
    // element strides: row (b, h, tq) of Q/K/V starts at b*q_bstride + h*q_hstride + tq*q_rowstride; E row r of head h at
    // h*e_hstride + r*e_rowstride.  head-major: (H*Tg*dpad, Tg*dpad, dpad) / ((2Tg-1)*dpad, dpad);
    // natural [B*Tp][D]: (Tp*D, d, G*D) / (d, G*D) — a grouped row is G consecutive rows, head h is the span [h*d, (h+1)*d)
    long long q_bstride, q_hstride, e_hstride; int q_rowstride, e_rowstride;
    const int* lens;                        // [B] valid frames at this stage (keys j with G*j >= lens[b] are masked)
    int B, H, T, G, D, d, dpad, Tg, Tgp;
    bf16_t* out; int ldo;                   // [B*T][ldo] un-grouped attention output (rows t >= T dropped)
    float scale;                            // 1/sqrt(d)
    int force_waves;                        // attention.hip only (option "attn_waves"): 8 = one 128-query workgroup per CU where LDS allows
    // ragged batch (attention2.hip, natural layout): rag_off [B + 1] first row of every utterance in the Q / K / V / out row space (rows
    // padded to the group size per utterance), lens[b] = its frames (all valid), rag_wg [B + 1] prefix sums of H x ceil(Tg_b / 64)
    // workgroups in the kernel's utterance order (0, 8, 16, .. | 1, 9, ..), rag_nwg their total, rag_tgmax = Tg of the longest utterance
    // (`eh` holds the positional rows for THAT length); T / Tg are then upper bounds only
    const int *rag_off, *rag_wg; int rag_nwg, rag_tgmax;
    // streaming contexts / causal (attention2.hip): key j of query i (grouped positions) is visible iff -band_l <= j - i <= band_r
    // (>= Tg: unlimited); causal: `eh` holds the Tg rows of the non-negative distances only (index Tg - 1 + j - i, j <= i)
    int band_l, band_r, causal;
    int ablate;                             // timing-only ablations of attention2.hip (EFFCONF_ATTN_ABLATE, tools only): 1 no compute, 2 no K / V / E loads after the first block, 4 no epilogue stores, 8 no utterance search
};
int launch_relpos_attention(const AttnParams& p, hipStream_t s);
// opt-in side output: att [B][H][Tg][Tg] fp32 softmax maps (the reference's att_w), recomputed from the same bf16 operands; rectangular batches
int launch_attention_probs(const AttnParams& p, float* att, hipStream_t s);
// second generation (attention2.hip): 32 queries per wave, transposing LDS reads for V; waves = 2 (64-query workgroups) or 4
bool relpos_attention2_supported(int dpad);
int launch_relpos_attention2(const AttnParams& p, int waves, hipStream_t s);
// rows t in [T, Tp) of the grouped view: Qu=u, Qv=v, K=V=0  (attentions.py:107-138, 671-675)
int launch_attn_pad_rows(const GemmParams& p, int B, hipStream_t s);       // head-major buffers
int launch_attn_pad_rows_nat(const GemmParams& p, int B, hipStream_t s);   // natural [B*Tp][D] buffers
int launch_attn_pad_rows_ragged(bf16_t* qu, bf16_t* kh, bf16_t* vt, const float* u, int D, int G, const RaggedRows& rg, hipStream_t s);

// ---------------------------------------------------------------- convolutions  (conv.hip)
// mel (B, F, Tm) f32 -> (B*T1, C*F/2) bf16, feature index c*(F/2)+f; 3x3 s2 p1 conv (Cin=1) + folded BN + Swish
// rag_tm != null (dev [B]): per-utterance mel frames (the zero padding starts there); rows stay rectangular (b, t < T1)
// rg != null (with rag_tm): rows written straight into the ragged row space (utterance b: rows rg->off[b] .., rg->len[b] frames + zero
// group-padding rows); T1 is then only an upper bound of an utterance's padded row count (tiles behind an utterance's end exit)
int launch_subsample_conv(const float* mel, int B, int F, int Tm, int T1, const float* w9, const float* bias, int C,
                          bf16_t* out, int ldo, hipStream_t s, const int* rag_tm = nullptr, const RaggedRows* rg = nullptr);
// rectangular fp32 rows (b, t) of B x t_pitch -> the ragged row space (row off[b] + t for t < len[b]; group-padding rows = zeros)
int launch_gather_rows(const float* x, int D, int t_pitch, const RaggedRows& rg, float* out, hipStream_t s);
// fused subsampling conv + Linear (sublinear.hip): mel (B, F, Tm) -> out fp32 (B*T1, N); W packed in (f-chunk, channel, f) K order
bool sublinear_fused_supported(int F, int N);
int launch_sublinear_fused(const float* mel, int B, int F, int Tm, int T1, const float* w9, const float* cbias, int C,
                           const bf16_t* W, int ldw, const float* bias, int N, float* out, int ldc, hipStream_t s);
// second generation (sublinear2.hip): row-stationary, the 3x3 conv on the MFMA pipe.  groups = 32-channel groups (0: shape not supported)
int sublinear2_groups(int F, int C, int N);
// rg != null: ragged rows (rag_tm: dev [n] mel frames per utterance; Tm stays the pitch of the mel image; pad rows are written as zeros)
int launch_sublinear2(const float* mel, int B, int F, int Tm, int T1, const float* ctab, const bf16_t* Wp, const float* bias, int C, int N,
                      float* out, int ldc, hipStream_t s, const RaggedRows* rg = nullptr, const int* rag_tm = nullptr);
// two-layer subsampling (conv2.hip): layer 1 channel-last, layer 2 implicit GEMM
int launch_subsample_conv_cl(const float* mel, int B, int F, int Tm, int T1, const float* w9, const float* bias, int C, int Cp,
                             bf16_t* out, hipStream_t s, const int* rag_tm = nullptr);
int launch_conv2_igemm(const bf16_t* act1, int B, int F1, int T1, int Cp, const bf16_t* W, int ldw, const float* bias,
                       int N, int F2, int T2, bf16_t* out, hipStream_t s);
// g (B, T, ld) bf16 -> (B, To, ld) bf16: depthwise conv k taps ("same" zero pad), stride s, folded BN, Swish
int launch_dwconv(const bf16_t* g, int B, int T, int To, int C, int ld, const float* w_kc, const float* bias,
                  int ksize, int stride, bf16_t* out, hipStream_t s, const RaggedConv* rc = nullptr, int causal = 0,
                  const void* w_mfma = nullptr);      // w_mfma: pack_dwconv_mfma's table -> dwconv_mfma_kernel (stride 1)
int dwconv_mfma_groups(int ksize);
bool dwconv_mfma_supported(int ksize, int stride);
void pack_dwconv_mfma(const float* w_kc, int ksize, int C, uint16_t* dst);     // dst: C * 4 * dwconv_mfma_groups(ksize) * 8 bf16   // causal: pre-padding (k - 1, 0)

// ---------------------------------------------------------------- fp32-operand "exact" mode  (exact.hip)
struct ExGemmParams {              // C = epi(A W^T + bias), everything fp32 (v_mfma_f32_32x32x2_f32)
    const float* A; int lda;
    int a_rows, a_pitch, a_stride; // src_row(m) = a_rows ? (m / a_rows) * a_pitch + (m % a_rows) * a_stride : m
    const float* W; int ldw;       // [N][K] row-major (reference nn.Linear / 1x1 Conv1d layout)
    const float* bias;             // [N] or null
    int M, N, K;
    float* C; int ldc;
    int c_rows, c_pitch;           // dst_row(m) = c_rows ? (m / c_rows) * c_pitch + m % c_rows : m
    int split_cols; size_t split_stride;   // split_cols > 0: column n -> buffer C + (n / split_cols) * split_stride, column n % split_cols
    const float* R; int ldr; float alpha;  // epi 2: C = R + alpha * (acc + bias)   (R may alias C)
    int epi;                       // 0 plain, 1 Swish, 2 residual
};
int launch_ex_gemm(const ExGemmParams& p, hipStream_t s);
// tlen != null (dev [B]): valid input frames per utterance - input frames behind them read as ZERO (the utterance run alone sees the conv's zero padding there)
// and, in the image layout (flat = 0), outputs behind (tlen - 1) / 2 + 1 are written as zero (the next layer's zero padding)
int launch_ex_conv2d(const float* in, int B, int Cin, int F, int T, const float* w, const float* scale, const float* shift, int Co,
                     float* out, int flat, hipStream_t s, const int* tlen = nullptr);
int launch_ex_glu(const float* in, long long M, int N, float* out, hipStream_t s);
int launch_ex_dwconv(const float* g, int B, int T, int To, int C, const float* w_kc, const float* bias, int ks, int stride, float* out, hipStream_t s);
struct ExAttnParams {              // natural-layout fp32 Q, K, V [B*Tp][D], E [2Tp-G][D]; out [B*Tp][D]
    const float *q, *k, *v, *e, *u, *vb;
    const int* lens;
    int B, H, T, Tp, G, D, d, Tg;
    float* out;
    float* att;                    // optional [B][H][Tg][Tg] softmax maps (null: not written)
    int variant;                   // 0: tiled kernel (default), 2: its 16-row shape, 1: one wave per query row (round 2's kernel); all agree bit for bit
    int band_l, band_r;            // streaming contexts in grouped positions of the stage (key j of query i visible iff -band_l <= j - i <= band_r; >= Tg: unlimited)
};
int launch_ex_attention(const ExAttnParams& p, hipStream_t s);

// ---------------------------------------------------------------- split-precision mode  (split.hip): fp32 tensors, fp16 (h, l) operand pairs on the matrix pipe
struct SxGemmParams {              // the ExGemmParams contract with W replaced by its two fp16 images, k-tile major: [ldh / 32][N][32] (ldh = round_up(K, 32),
    ExGemmParams g;                // zero padded): element (n, k) at ((k / 32) * N + n) * 32 + k % 32
    const uint16_t *Whi, *Wlo; int ldh;
};
int launch_sx_gemm(const SxGemmParams& p, hipStream_t s);
struct SxAttnParams { ExAttnParams a; float* scores; int TgP; };
bool sx_attention_supported(int d);
size_t sx_attention_scores_bytes(int B, int H, int Tg);
int launch_sx_attention(const ExAttnParams& p, float* scores, hipStream_t s);     // scores: sx_attention_scores_bytes(B, H, Tg) of scratch

// ---------------------------------------------------------------- split-precision mode on fused kernels  (sxf.hip, round 6): ragged batches, causal / streaming
struct SxfAttnParams {             // natural-layout fp32 rows [rows][D]: Q, K, V straight from the projection (no + u / + v, no pad-row pass), out likewise
    const float *q, *k, *v;        // k, v: read by launch_sxf_pack_kv only
    const uint16_t *kp, *vp, *ep;  // operand images (sxf.hip): K [grouped row][H][2][PK], V^T [B][H][2][VX][vpitch], E [causal ? Tg : 2 Tg - 1][H][2][PK] fp16 (h | l)
    int vpitch;                    // keys per V^T row: a multiple of 64 >= Tg
    const float* u;                // [D] content bias (attentions.py:474)
    const int* lens;               // [B] valid frames per utterance at this stage (key mask; ragged: the utterance's frames)
    const int* off;                // ragged: [B + 1] first row of every utterance (rows padded to the group size); null: utterance b = rows b Tp ..
    int B, H, G, D, d;
    int T, Tp, Tg;                 // rectangular: frames / padded frames / grouped frames of every utterance; ragged: Tg of the longest one (E and the grid)
    float* out;
    int band_l, band_r, causal;    // key j of query i visible iff -band_l <= j - i <= band_r (grouped positions); causal: E holds the non-negative distances
};
bool sxf_attention_supported(int d);
int sxf_attention_pk(int d);       // columns per K / E image row half
int sxf_attention_vx(int d);       // rows per V^T image plane
int launch_sxf_pack_kv(const SxfAttnParams& p, hipStream_t s);       // K / V (fp32, this block's projections) -> their images
// E = pos_layer(R) fp32 [erows_grouped * G][D] -> its image, the positional bias (v - u) E^T in column d
int launch_sxf_pack_e(const float* e, const float* u, const float* vb, int erows_grouped, int H, int G, int D, int d, uint16_t* ep, hipStream_t s);
int launch_sxf_attention(const SxfAttnParams& p, hipStream_t s);
// FeedForwardModule + half-step residual (+ optional block-final LayerNorm) as one split-precision kernel (sxf_ffn.hip)
struct SxfFfnParams {
    const float* X; int ldx;       // residual stream in [M][ldx]
    float* Y; int ldy;             // out (may alias X): x + 1/2 FFN(LN(x)), or LayerNorm of that when ln_g != null
    const uint16_t* wimg;          // weight image, chunk-major: per 32 hidden units [W1 hi 32 x DP1 | W1 lo | W2 hi DP2 x 32 | W2 lo] fp16 (pack_sxf_ffn in encoder.hip:
                                   // pre-norm gamma / beta and b1 folded into W1 / its column D, W2 pre-scaled by 1/2 with its k order permuted to the accumulator layout)
    const float* b2;               // [DP2] b2 / 2, zero padded
    const float *ln_g, *ln_b;      // optional LayerNorm applied to the result (blocks.py:135), [D]
    int M, D, nchunk;
    int ablate;                    // timing-only ablations (tools/sxf_ffn_probe.py through the diagnostic library; 0 in the product): 1 no first product, 2 no Swish,
                                   // 4 no second product, 8 no weight stream after the first chunk, 16 no per-chunk barrier
};
// Conv2dSubsampling (one layer) + transpose / flatten + Linear as one bf16 kernel, chunked over (output frequency, 32 channels) (sublinear3.hip)
struct SubLin3Params {
    const float* mel; int B, F, Tm;   // (B, F, Tm) fp32 mel image, Tm = row pitch
    const int* mel_len;               // dev [B]: the utterance's own mel frames (ragged batches), or null: Tm
    const int* off;                   // dev [B + 1]: first output row of every utterance (ragged: group padded), or null: b To
    const int* len;                   // dev [B]: output frames that exist, or null: To
    int To;                           // output frames per utterance of a rectangular batch
    int rows_max;                     // grid: rows of the longest utterance (ragged: incl. its group padding; rectangular: To)
    const uint16_t* cimg;             // conv taps [ncb][hi | lo][32 channels][16 taps] bf16 (hi by truncation, lo = bf16(w - hi)): tap 3 i + j = folded weight, tap 9 = folded bias
    const uint16_t* wimg;             // Linear image [Fo ncb chunks][DP2 outputs][32 k] bf16: chunk f' ncb + cb, k position 16 s + 8 kh + e <-> channel
                                      // 32 cb + 16 s + 8 (e >> 2) + 4 kh + (e & 3) (the accumulator layout), weight column channel Fo + f' (encoders.py:114)
    const float* bias;                // [DP2] Linear bias, zero padded
    float* y; int ldy, N;             // out [rows][ldy], N columns
    int ncb, Fo;                      // channel blocks of 32, output frequencies
};
int sublinear3_tiles(int N);          // 32-column output tiles of the instance that serves width N (DP2 = 32 x this); 0 = not built
int launch_sublinear3(const SubLin3Params& p, hipStream_t s);
// Conv2dSubsampling (one layer) + transpose / flatten + Linear as one split-precision kernel (sxf_sub.hip)
struct SxfSubParams {
    const float* mel; int B, F, Tm;   // (B, F, Tm) fp32 mel image, Tm = row pitch
    const int* mel_len;               // dev [B]: the utterance's own mel frames (ragged batches), or null: Tm
    const int* off;                   // dev [B + 1]: first output row of every utterance (ragged: group padded), or null: b To
    const int* len;                   // dev [B]: output frames that exist, or null: To
    int To;                           // output frames per utterance of a rectangular batch
    int rows_max;                     // grid: rows of the longest utterance (ragged: incl. its group padding; rectangular: To)
    const uint16_t* cimg;             // conv taps [ncb][hi | lo][32 channels][16 taps] fp16 at 2^10: tap 3 i + j (frequency, time) x BN scale, tap 9 = BN shift + scaled conv bias
    const uint16_t* wimg;             // Linear image [Fo ncb chunks][hi | lo][DP2 outputs][32 k] fp16 at 2^10: chunk f' ncb + cb, k position 16 s + 8 kh + e <->
                                      // channel 32 cb + 16 s + 8 (e >> 2) + 4 kh + (e & 3) (the accumulator layout), weight column channel Fo + f' (encoders.py:114)
    const float* bias;                // [DP2] Linear bias, zero padded
    float* y; int N;                  // out [rows][N]
    int ncb, Fo;                      // channel blocks of 32, output frequencies
};
int sxf_sublin_tiles(int N);          // 32-column output tiles of the instance that serves width N (DP2 = 32 x this); 0 = not built
int launch_sxf_sublin(const SxfSubParams& p, hipStream_t s);
void sxf_ffn_shape(int D, int* ks1, int* nt2);
bool sxf_ffn_supported(int D);
size_t sxf_ffn_image_halfs(int D, int F);
int launch_sxf_ffn(const SxfFfnParams& p, hipStream_t s);
// Split-precision row-local chains (sxf_chain.hip): the work of a block between attention and the depthwise convolution as one kernel each.  Weight images: fp16
// (h | l) planes at the weight scale 2^10, chunk-major, k order = the accumulator layout's (pack_sxc_* in encoder.hip); "F1" = 32 outputs x all inputs per chunk with
// the bias in column D, "F2" = all outputs x 32 inputs per chunk, an FFN image = both per chunk of 32 hidden units.
struct SxcBParams {                // chain B: x += O Wo^T + bo;  g = GLU(LN(x) Wp1^T + bp1)
    const float* o; int o_rows, o_pitch;       // attention output [.][D]; o_rows > 0: row m of x is row (m / o_rows) o_pitch + m % o_rows of o (rectangular batches: frames / padded frames)
    float* x;                      // residual stream [M][D], updated in place
    float* g;                      // [M][De]
    int M, D, De;
    const uint16_t* w_o; const float* b_o;     // F2 image of the output projection (ceil(D / 32) chunks), bias padded to 32 ceil(D / 32)
    const uint16_t* w_p1; int nch_p1;          // F1 image of pointwise-1 (conv-module LayerNorm folded in): chunk pairs (32 value rows, their 32 gate rows)
};
struct SxcAParams {                // chain A.  tail: x = xres + C Wp2^T + bp2; x += 1/2 FFN2(LN(x)); y = LN(x).  head: y += 1/2 FFN1(LN(y)); Q | K | V = LN(y) Wqkv^T + b
    int tail, head;                // at least one
    const float* c;                // tail: conv-module activations [M][D] (depthwise conv output)
    const float* xres;             // tail: residual input [M][D] (the stream, or conv_res of it)
    float* y;                      // [M][D]: tail: the block's output (scratch for x on the way); head: updated in place (head alone: the input stream)
    int M, D;
    const uint16_t* w_p2; const float* b_p2;   // F2 image of pointwise-2 + bias
    const uint16_t* w_f2; int nch_f2; const float* b_f2;     // FFN2 image (pre-norm folded), b2 / 2
    const float *ln_g, *ln_b;      // block-final LayerNorm [D]
    const uint16_t* w_f1; int nch_f1; const float* b_f1;     // head: FFN1 image of the NEXT block, b2 / 2
    const uint16_t* w_qkv;         // head: F1 image of Q | K | V (each padded to 32 ceil(D / 32) rows; attention pre-norm folded)
    float* q; size_t qkv_stride;   // head: Q at q, K at q + qkv_stride, V at q + 2 qkv_stride (fp32 [.][D])
    size_t qkv_bytes;              // head: bytes from q to the end of V's rows (< 4 GB: the stores are range-checked buffer stores)
    int q_rows, q_pitch;           // head: row remap of the Q / K / V rows (as o_rows / o_pitch)
};
bool sxc_supported(int D);
int launch_sxc_b(const SxcBParams& p, hipStream_t s);
int launch_sxc_a(const SxcAParams& p, hipStream_t s);
// fp32 depthwise conv + folded BatchNorm + Swish; rc != null: per-utterance row ranges (tcap_max = the longest utterance's padded output rows)
int launch_sxf_dwconv(const float* g, int B, int T, int To, int C, const float* w_kc, const float* bias, int ks, int stride, float* out, hipStream_t s,
                      const RaggedConv* rc = nullptr, int causal = 0, int tcap_max = 0);
int launch_sxf_decimate(const float* x, int D, int stride, const RaggedConv& rc, float* out, hipStream_t s);
int launch_sxf_glu(const float* in, long long M, int N, float* out, hipStream_t s);

// ---------------------------------------------------------------- mel frontend  (mel.hip)
struct MelTables {                 // device tables built once per encoder
    const float* window;           // [n_fft] Hann(win_length) centred in n_fft
    const float2* twiddle;         // [n_fft/2] exp(-2 pi i k / n_fft)
    const int* fb_start;           // [n_mels] first non-zero bin
    const int* fb_count;           // [n_mels]
    const int* fb_offset;          // [n_mels] offset into fb_weight
    const float* fb_weight;        // packed non-zero triangular weights
    int fb_nnz;                    // number of packed weights
};
// ragged_len != null (dev i64 [B], samples): every row at its own length (reflect padding at its own ends, len / hop + 1 frames); L and
// Tm stay the row pitches of `audio` and of the mel image
int launch_mel(const float* audio, int B, int L, const MelTables& t, int n_fft, int hop, int n_mels, int Tm,
               int normalize, float mean, float std, float* mel, hipStream_t s, const int64_t* ragged_len = nullptr);
// diagnostics (tools/mel_repro.py): kernel variant (mel.hip), unused dynamic LDS per workgroup, counters dbg[8]
int launch_mel_debug(int variant, int extra_lds, const float* audio, int B, int L, const MelTables& t, int n_fft, int hop, int n_mels,
                     int Tm, int normalize, float mean, float std, float* mel, unsigned int* dbg, hipStream_t s);
// synthetic single-resource neighbour kernels (debug.hip)
int launch_debug_neighbour(int kind, int blocks, int lds_bytes, int iters, float* buf, size_t n, hipStream_t s);
int launch_debug_spin(double microseconds, hipStream_t s);
// tools/lds_fill_rate_probe.py: L2 -> LDS fill rate of a CU (mode 0 LDS-DMA, 1 loads + ds_write); out[2 * blocks] = {cycles, bytes} per workgroup
int launch_debug_lds_fill(int mode, int blocks, int waves, const char* src, size_t window, int kib_per_wave, int passes, unsigned long long* out, hipStream_t s);
int launch_debug_victim(int kind, int blocks, int iters, float* out, hipStream_t s);
// mel.hip compiled a second time WITH packed-fp32 VALU instructions (diagnostics: variant bit 8 = the hazardous round-1 kernel)
int launch_mel_debug_pk(int variant, int extra_lds, const float* audio, int B, int L, const MelTables& t, int n_fft, int hop, int n_mels,
                          int Tm, int normalize, float mean, float std, float* mel, unsigned int* dbg, hipStream_t s);

// ---------------------------------------------------------------- lengths + CTC head  (ctc.hip)
// stage lengths: mel frames -> per-stage frame counts (int32), final int64 out_len
int launch_lengths(const int64_t* x_len, int B, int from_audio, int hop, int sub_layers, const int* block_stride,
                   int n_blocks, int* stage_lens /*[n_blocks+1][B]*/, int64_t* out_len, hipStream_t s);
// ragged batches: the same lengths plus, per block position k = 0 .. n_blocks (position n_blocks = the encoder output):
//   mel_len [B]                 mel frames per utterance
//   row_off [k][B + 1]          prefix sums of the utterances' rows, every utterance padded to a multiple of group[k] (group[n_blocks] = 1)
//   wg_off  [k][B + 1] (k < n_blocks)  prefix sums of heads[k] x ceil(ceil(len / group[k]) / 64) attention workgroups in the attention kernels'
//                               utterance order (0, 8, 16, .. | 1, 9, ..)
//   tile_off[k][B + 1] (k < n_blocks)  prefix sums of ceil(padded output rows of block k / 128) depthwise-conv tiles
// One workgroup; B <= 4096.
int launch_lengths_ragged(const int64_t* x_len, int B, int from_audio, int hop, int sub_layers, const int* block_stride, const int* group,
                          const int* heads, int n_blocks, int* stage_lens, int* mel_len, int* row_off, int* wg_off, int* tile_off,
                          int64_t* out_len, hipStream_t s);
// logits = x W^T + b in fp32 (Wt is [D][V]), argmax per frame (first max), optional logits out
int launch_ctc_argmax(const float* x, int M, int D, const float* Wt, const float* bias, int V,
                      int* preds, float* logits_or_null, hipStream_t s, int use_mfma = 1);   // use_mfma: fp32-MFMA kernel where the frame tile fits LDS
// the same with split-bf16 operands on the bf16 matrix pipe (x_hi W_hi + x_hi W_lo + x_lo W_hi): Whi / Wlo [ceil(D / 16)][round_up(V, 64)][2][8]
// bf16 (MFMA B fragments, packed at finalize); logits within ~2^-16 relative of the fp32 head
// x_is_bf16: x points to bf16 rows [M][D] (gathered encoder outputs, bf16 wire): x_lo = 0, two MFMAs per 16 k, logits bit-identical to the fp32-input
// kernel on the same values
int launch_ctc_split(const float* x, int M, int D, const bf16_t* Whi, const bf16_t* Wlo, const float* bias, int V, int* preds, float* logits_or_null,
                     hipStream_t s, int x_is_bf16 = 0);
// drop blanks (0), collapse repeats, stop at len[b]  (model_ctc.py:99-133)
int launch_ctc_collapse(const int* preds, const int64_t* lens, int B, int T, int* labels, int* label_len, hipStream_t s);
