/* libeffconf — MI355X (gfx950) native Efficient Conformer encoder forward path.  C ABI.
 *
 * The reference (burchim/EfficientConformer) has no plugin/FFI layer: its seam for this path is the
 * Python class models.encoders.ConformerEncoder (reference models/encoders.py:44), constructed from
 * the `encoder_params` dict (encoders.py:46-95) and called as `encoder(x, x_len)` at
 * models/model_ctc.py:63, 93, 159, models/transducer.py:94, 145, 206 and models/model.py:552, 639.
 * This header is what a native replacement of that seam exports; the Python class
 * efficientconformer_amd.ConformerEncoder binds it with ctypes.
 *
 * Conventions
 *   - every `dev` pointer is caller-owned DEVICE memory (HIP); the library never frees it;
 *   - packed weights are the only memory the library owns (allocated in effconf_encoder_finalize,
 *     released by effconf_encoder_destroy);
 *   - all work is enqueued on the `stream` argument (a hipStream_t passed as void*; NULL = default
 *     stream); no call synchronises the device or allocates in the forward path (graph-capturable);
 *   - return value 0 = ok, negative = error, message via effconf_last_error() (thread-local);
 *   - no C++ exceptions or torch types cross this boundary.
 */
#ifndef EFFCONF_H
#define EFFCONF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is compiled with -fvisibility=hidden: the declarations of this header (and of effconf_debug.h) are its ONLY dynamic symbols
 * (tests/test_abi_and_host.py checks `nm -D`); the internal launch_* / pack_* C++ functions of csrc/kernels.h are not linkable. */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

/* 2: EcConfig gained causal / left_context / right_context
 * 3: the effconf_debug_* entries left libeffconf.so for libeffconf_debug.so (round 5), hidden visibility for everything this header does not
 *    declare, the label-exact "split" mode takes ragged batches and causal / streaming configurations (round 6) */
#define EFFCONF_ABI_VERSION 3

/* Per-block hyper-parameters, already resolved from the per-stage lists exactly as
 * ConformerEncoder.__init__ does (reference encoders.py:80-95). */
typedef struct EcBlock {
    int32_t dim_model;    /* D   : input / attention width                                  */
    int32_t dim_expand;   /* De  : output width (== D except in the two transition blocks)  */
    int32_t ff_ratio;     /* FFN hidden = ff_ratio * dim                                    */
    int32_t num_heads;    /* H                                                              */
    int32_t kernel_size;  /* depthwise conv taps (odd, <= 31)                               */
    int32_t group_size;   /* attention group size G (odd)                                   */
    int32_t max_pos;      /* max_pos_encoding of this block's relative table                */
    int32_t conv_stride;  /* 1 or 2 (progressive downsampling, blocks.py:106-117)           */
} EcBlock;

/* Mirrors the reference `encoder_params` keys that reach the hot path (encoders.py:50-94). */
typedef struct EcConfig {
    int32_t n_mels, sample_rate, n_fft, win_length, hop_length;  /* modules.py:77-85           */
    int32_t normalize; float mean, std;                          /* modules.py:103-104         */
    int32_t sub_layers;                                          /* Conv2dSubsampling layers   */
    int32_t sub_filters[4];                                      /* channels per layer         */
    int32_t num_blocks;
    const EcBlock* blocks;
    int32_t vocab_size;                                          /* >0: CTC head `fc` present  */
    /* streaming / causal (encoders.py:68, 94): contexts in frames AFTER the subsampling, as StreamingMask gets them - key j of query i
     * masked iff j - i > right_context or j - i < -left_context; >= 2^30 = unlimited (the shipped configs pass max_pos_encoding);
     * causal = 1: causal relative tables (attentions.py:506, 1243-1247), depthwise convs pre-padded (k - 1, 0) (layers.py:97-101), and
     * the caller passes right_context = 0 (encoders.py:68).  bf16 path, attention2.hip head widths (<= 160 padded) only. */
    int32_t causal, left_context, right_context;
} EcConfig;

typedef struct EcEncoder EcEncoder;

int effconf_abi_version(void);
const char* effconf_last_error(void);

/* ---- lifetime + weights ------------------------------------------------------------------- */
/* Test hook: with EFFCONF_POISON_GUARDS=<KiB> (>0; at least 16 KiB is used) in the environment at create time, finalize places every
 * packed parameter buffer between two guard regions of 0xFF bytes (NaN as bf16 and fp32), so an out-of-bounds parameter read shows
 * up in the output (tests/test_gpu_exact_and_sweep.py::test_no_kernel_reads_past_a_parameter_buffer, tools/poison_sweep.py).  Read
 * once here, never on the forward path.  The Python wrapper has the matching EFFCONF_POISON_WORKSPACE=<byte> hook for the
 * caller-owned workspace (efficientconformer_amd/encoders.py). */
EcEncoder* effconf_encoder_create(const EcConfig* cfg);
void effconf_encoder_destroy(EcEncoder* enc);
/* Hand over one reference state_dict tensor (HOST fp32, contiguous, reference layout) by its key
 * without the "encoder." prefix, e.g. "blocks.3.feed_forward_module1.layers.1.weight",
 * "subsampling_module.layers.0.1.running_var", "fc.weight" (reference models/model.py:361-384,
 * model_ctc.py:77-88 decide which keys exist).  Keys the packer does not know are kept and ignored;
 * effconf_encoder_finalize reports the first REQUIRED key that is missing or mis-shaped. */
int effconf_encoder_load_tensor(EcEncoder* enc, const char* key, const float* host, const int64_t* shape, int32_t ndim);
/* Fold BatchNorm(eval) into the convolutions, cast to bf16, pad to MFMA-friendly shapes, build the
 * sinusoid / window / filterbank tables and upload.  Fails (-1) if a required tensor is missing. */
int effconf_encoder_finalize(EcEncoder* enc);

/* ---- forward ------------------------------------------------------------------------------ */
/* Bytes of scratch a forward of this shape needs (n = samples per row if from_audio, else mel frames). */
size_t effconf_encoder_workspace_bytes(const EcEncoder* enc, int32_t batch, int32_t n, int32_t from_audio);
/* Output frames T_out for an input of n samples / mel frames (encoders.py:139 bookkeeping). */
int32_t effconf_encoder_out_frames(const EcEncoder* enc, int32_t n, int32_t from_audio);

/* ConformerEncoder.forward(x, x_len) (reference encoders.py:97-142), eval mode.
 *   audio   dev f32 (batch, n_samples)   zero-padded rows (utils/preprocessing.py:38)
 *   x_len   dev i64 (batch)              valid samples per row
 *   out     dev f32 (batch, T_out, D_last)
 *   out_len dev i64 (batch)
 * The third value of the reference's return tuple (attention maps no caller consumes) is not produced. */
int effconf_encoder_forward(EcEncoder* enc, const float* audio, const int64_t* x_len, int32_t batch, int32_t n_samples,
                            float* out, int64_t* out_len, void* workspace, size_t workspace_bytes, void* stream);
/* Same, entered after AudioPreprocessing: mel dev f32 (batch, n_mels, n_frames), mel_len in frames.
 * This is the parity boundary ("identical mel inputs"). */
int effconf_encoder_forward_mel(EcEncoder* enc, const float* mel, const int64_t* mel_len, int32_t batch, int32_t n_frames,
                                float* out, int64_t* out_len, void* workspace, size_t workspace_bytes, void* stream);

/* Ragged batch: the same forward with every utterance at ITS OWN length - no pad frames exist, so utterance b's output is what the
 * reference computes for that utterance alone (batch size 1; the reference's batched output differs from it by the pad-frame leakage of
 * SURVEY.md 8a), independent of what else is in the batch.  All utterances share one concatenated row space and one set of launches.
 *   x          dev f32 (batch, n) audio rows or (batch, n_mels, n) mel (from_audio = 0); n = the row pitch (content behind x_len unused)
 *   x_len      dev i64 (batch); x_len_host: the same lengths on the HOST (grids and the workspace are sized from them)
 *              CONTRACT: x_len_host[b] == x_len[b] for every b.  The library cannot compare them without a synchronisation and does not:
 *              the kernels index rows with the device copy inside grids / a workspace sized from the host copy, so a device length that
 *              exceeds its host twin is an out-of-bounds access.  (The Python wrapper checks on request: ConformerEncoder.check_host_lengths.)
 *   out        dev f32 (batch, out_frames, D_last): utterance b's T_out(b) frames, zeros behind them; out_frames >= the longest T_out
 * Workspace: effconf_encoder_workspace_bytes_ragged(enc, x_len_host, batch, n, from_audio).  Front ends: sublinear2.hip indexes the ragged
 * rows itself (one subsampling layer, <= 192 filters and <= 192-wide first stage); wider one-layer and the two-layer subsamplers run on the
 * rectangular image (zero padding at every utterance's own end) and gather the valid rows.  Needs head widths <= 160 (attention2.hip);
 * bf16 path only. */
size_t effconf_encoder_workspace_bytes_ragged(const EcEncoder* enc, const int64_t* x_len_host, int32_t batch, int32_t n, int32_t from_audio);
int effconf_encoder_forward_ragged(EcEncoder* enc, const float* x, const int64_t* x_len, const int64_t* x_len_host, int32_t batch, int32_t n,
                                   int32_t from_audio, float* out, int32_t out_frames, int64_t* out_len, void* workspace,
                                   size_t workspace_bytes, void* stream);

/* AudioPreprocessing.forward alone (reference modules.py:87-106): audio -> (batch, n_mels, n//hop+1). */
int effconf_mel_frontend(EcEncoder* enc, const float* audio, int32_t batch, int32_t n_samples, float* mel, void* stream);

/* CTC head + greedy decode (reference model_ctc.py:49, 90-133): fc, per-frame argmax, drop blanks,
 * collapse repeats, stop at out_len.  labels dev i32 (batch, T_out) zero-filled tail, label_len dev i32 (batch).
 * logits (dev f32 (batch, T_out, vocab)) may be NULL.  Needs batch*T_out*4 bytes of workspace. */
int effconf_ctc_greedy(EcEncoder* enc, const float* enc_out, const int64_t* out_len, int32_t batch, int32_t t_out,
                       int32_t* labels, int32_t* label_len, float* logits, void* workspace, size_t workspace_bytes, void* stream);
/* The same head on bf16 rows (dev bf16 (batch, T_out, D_last)): what a rank holds after the all-gather of encoder outputs on a bf16 wire
 * (reference sharding: main.py:33-35, 217-220; the head itself: model_ctc.py:49, 90-133).  bf16 path only.  x = x_hi exactly, so the split head issues
 * two MFMAs per 16 k instead of three and skips the conversion pass; logits and labels are bit-identical to effconf_ctc_greedy on the same values
 * widened to fp32. */
int effconf_ctc_greedy_bf16(EcEncoder* enc, const uint16_t* enc_out_bf16, const int64_t* out_len, int32_t batch, int32_t t_out,
                            int32_t* labels, int32_t* label_len, float* logits, void* workspace, size_t workspace_bytes, void* stream);

/* (Grouped)RelPosMultiHeadSelfAttention core alone (reference attentions.py:549-718 between the input projections and the output
 * projection): natural-layout bf16 device buffers qu = Q + u, k, v of (batch * Tp, dim) rows (Tp = frames rounded up to the group
 * size; pad rows: qu = u, k = v = 0), e = pos_layer(R) of (2 Tp - group, dim) rows, dvu = (v - u) per head column as fp32
 * [heads][dvu_ld] (dvu_ld >= head width rounded up to 32, zero beyond the head width), lens = valid frames per utterance (i32).
 * out: bf16 (batch * frames, ld_out) un-grouped attention output.  variant 0 = attention.hip, 1 / 2 = attention2.hip (see the
 * "attention_v2" option).  Needs 512 bytes of readable slack behind qu / k / v / e (16-byte chunk loads may run past a head span). */
int effconf_relpos_attention(const uint16_t* qu, const uint16_t* k, const uint16_t* v, const uint16_t* e, const float* dvu, int32_t dvu_ld,
                             const int32_t* lens, int32_t batch, int32_t heads, int32_t frames, int32_t group, int32_t dim, uint16_t* out,
                             int32_t ld_out, int32_t variant, void* stream);

/* ---- per-kernel entry points (unit tests of ONE module on the product kernels) -------------------------------- */
/* Each runs one module of Conformer block `block` of a finalized encoder on caller-owned device buffers and is tested against the
 * reference's per-module outputs (tests/golden/tiny_*.npz `trace/blocks.N.*`).  `workspace`: effconf_module_workspace_bytes(enc,
 * batch, frames) bytes (for effconf_subsample: frames = mel frames).  fp32 rows are dense: (rows, D) row-major. */
size_t effconf_module_workspace_bytes(const EcEncoder* enc, int32_t batch, int32_t frames);
/* FeedForwardModule + half-step residual (reference modules.py:385-395, blocks.py:122, 132): y = x + 1/2 FFN_which(LayerNorm(x));
 * which = 1 (feed_forward_module1, width dim_model) or 2 (feed_forward_module2, width dim_expand).  y may alias x. */
int effconf_ffn(EcEncoder* enc, int32_t block, int32_t which, const float* x, int32_t rows, float* y, void* workspace, size_t workspace_bytes,
                void* stream);
/* ConvolutionModule (reference modules.py:511-525; layers.py:122-136): LayerNorm -> pointwise-1 + GLU -> depthwise conv (stride of the
 * block) + BatchNorm(eval) + Swish -> pointwise-2; NO residual.  x f32 (batch * frames, dim_model) -> y f32 (batch * frames_out, dim_expand),
 * frames_out = (frames - 1) / conv_stride + 1. */
int effconf_conv_module(EcEncoder* enc, int32_t block, const float* x, int32_t batch, int32_t frames, float* y, void* workspace,
                        size_t workspace_bytes, void* stream);
/* Conv2dSubsampling + transpose + Linear (reference modules.py:232-249, encoders.py:113-116): mel f32 (batch, n_mels, n_frames) ->
 * y f32 (batch * T1, dim_model of block 0), T1 = frames after the subsampling layers.  Honours the "fuse_subsample" option. */
int effconf_subsample(EcEncoder* enc, const float* mel, int32_t batch, int32_t n_frames, float* y, void* workspace, size_t workspace_bytes,
                      void* stream);
/* y = LayerNorm_which(x + alpha * r), r may be NULL: the residual / norm glue of ConformerBlock.forward (reference blocks.py:119-137).
 * which: 0 FFN1 pre-norm, 1 attention pre-norm, 2 conv-module pre-norm (width dim_model); 3 FFN2 pre-norm, 4 block-final norm (dim_expand). */
int effconf_layernorm_residual(EcEncoder* enc, int32_t block, int32_t which, const float* x, const float* r, float alpha, int32_t rows,
                               float* y, void* stream);

/* ---- RNN-T greedy decode (next row after the encoder: BASELINE.json configs[3]) ------------------ */
/* Replaces Transducer.gready_search_decoding's per-utterance Python loop (reference models/transducer.py:139-186) for
 * the shipped Transducer configs: RnnDecoder = Embedding + 1-layer LSTM (models/decoders.py:41-70) and
 * JointNetwork(joint_mode "sum", act "tanh", models/joint_networks.py:35-104).  All arithmetic fp32. */
typedef struct EcRnntConfig {
    int32_t dim_encoder;          /* encoder_params["dim_model"][-1]                (transducer.py:76)  */
    int32_t dim_decoder;          /* decoder_params["dim_model"] (embedding = LSTM hidden, decoders.py:46-47) */
    int32_t dim_joint;            /* joint_params["dim_model"]                      (joint_networks.py:41) */
    int32_t vocab_size;           /* decoder_params["vocab_size"]; token 0 = blank / start (transducer.py:151, 167) */
    int32_t num_layers;           /* decoder_params["num_layers"]: 1 is native                                   */
    int32_t max_consec_dec_step;  /* decoder_params.get("max_consec_dec_step", 5)  (transducer.py:83)            */
    int32_t joint_mode;           /* 0 = "sum" (native); "concat" is rejected                                    */
    int32_t joint_act;            /* 0 = "tanh" (native); others rejected                                        */
} EcRnntConfig;

typedef struct EcRnnt EcRnnt;

EcRnnt* effconf_rnnt_create(const EcRnntConfig* cfg);
void effconf_rnnt_destroy(EcRnnt* r);
/* state_dict tensors (HOST fp32) by reference key: "decoder.embedding.weight", "decoder.rnn.{weight,bias}_{ih,hh}_l0",
 * "joint_network.linear_{encoder,decoder,joint}.{weight,bias}". */
int effconf_rnnt_load_tensor(EcRnnt* r, const char* key, const float* host, const int64_t* shape, int32_t ndim);
/* Builds the per-token LSTM input table Gin[y] = W_ih emb[y] + b_ih + b_hh and the k-major weight images; uploads. */
int effconf_rnnt_finalize(EcRnnt* r);
size_t effconf_rnnt_workspace_bytes(const EcRnnt* r, int32_t batch, int32_t t_out);
/* Upper bound of emitted tokens per utterance: max_consec_dec_step * T_out (transducer.py:167-176). */
int32_t effconf_rnnt_max_tokens(const EcRnnt* r, int32_t t_out);
/* enc_out dev f32 (batch, T_out, dim_encoder), out_len dev i64 (batch) -> tokens dev i32 (batch, max_tokens), zero-filled
 * tails (the leading start token of the reference's `y` is not included: transducer.py:179 decodes y[:, 1:]),
 * token_len dev i32 (batch). */
/* "cluster_decode": -1 auto (default: batches of 16..256 utterances decode in clusters of 8 workgroups x 8 utterances that share the
 * weight streams and run their three mat-vec phases on the fp32 matrix pipe), 0 one workgroup per utterance, 1 force (batch <= 256).
 * A cluster's workgroups synchronise through global memory.  "cluster_by_slice" (default 1): a cluster is 8 CONSECUTIVE workgroups, so
 * slice k of the weights is only ever read through XCD k's L2 and any resident part of a launch consists of whole clusters (decodes that
 * share the GPU cannot starve each other); 0: the members of a cluster share an XCD and must all be resident - auto then stops at 128
 * utterances.  The spin is bounded: a starved cluster gives up with undefined tokens instead of hanging.  All paths produce identical
 * tokens.  "cluster_shape" (round 6): 0 (default) clusters of 8 workgroups x 8 utterances x 2 encoder frames per joint pass, 1 = 16 x 16 x 1 where the
 * decoder / joint widths leave LDS for it (<= 768; half the weight bytes per round - measured no faster: profiles/r6_33_rnnt_cluster_shapes.txt). */
int effconf_rnnt_set_option(EcRnnt* r, const char* name, int32_t value);
int effconf_rnnt_greedy(EcRnnt* r, const float* enc_out, const int64_t* out_len, int32_t batch, int32_t t_out,
                        int32_t* tokens, int32_t* token_len, int32_t max_tokens, void* workspace, size_t workspace_bytes,
                        void* stream);

/* Options.  "cache_pos_embeddings" = 1: the positional projections E = pos_layer(R) (reference attentions.py:588, 678)
 * are input independent; they live in the workspace and are recomputed only when that workspace last held a forward of
 * another batch size or number of frames.  The library remembers one tag per workspace pointer (the 16 most recently used),
 * so a caller that alternates workspaces (one per stream) keeps all of them warm.  Enable ONLY if the caller leaves the
 * workspaces untouched between forwards; setting the option again (to any value) forgets every tag — do that when a
 * workspace is freed and its address may be reused.  Host calls on one handle must not run concurrently (they only enqueue).
 * "fuse_subsample" (default 2): conv-subsampling + Linear as 0 = separate conv and GEMM kernels, 1 = sublinear.hip (tiled GEMM whose A tile is
 *   produced by a VALU convolution), 2 = sublinear2.hip (row-stationary, the convolution on the MFMA pipe with a bf16 hi / lo operand
 *   split; falls back to 1 where the shape is not supported; front ends wider than 128 channels / columns - EfficientConformer Medium, Large - run
 *   sublinear3.hip instead: option "sub3_auto", default 1), 3 = sublinear3.hip (round 6: the same fusion in chunks of (output frequency, 32 channels), any
 *   channel count / width up to 384; bit-identical to sublinear2.hip where both exist, slower there).
 * "chain_count_stores" (default 0; measurement only): 1 = the weight-ring waits of rounds 3 - 6, which also allowed global stores to stay outstanding - UNSAFE (a store can
 *   retire before an older LDS-DMA; profiles/r6_108_ring_wait_fix.txt): kept to reproduce the finding, never for production.
 * "tiled_auto" (default 1; round 6): with "wide_gemm" = 0, a configuration whose widest stage lies in ("tiled_min_k" = 256, 384] - EfficientConformer Medium's
 *   D = 360 - runs that stage as LayerNorm + tiled GEMMs instead of the row-stationary kernels (+ 2.3 % on Medium; decided per configuration at finalize,
 *   never per batch: one rounding path per handle).  0 = the row-stationary kernels as before.
 * "fuse_chain" (default 1): 0 runs every GEMM of a block as its own kernel instead of the fused row-local chains (chain.hip);
 *   with the debug trace this exposes the intermediate residual-stream states that otherwise only exist in registers.
 * "attention_v2" (default 1): 0 = attention.hip; 1 = attention2.hip (same tiling; V read through ds_read_b64_tr_b16 instead of being
 *   transposed on the way into LDS, hoisted staging addresses, no column masks off the buffer tails, deferred accumulator rescale);
 *   2 = attention2.hip with 32 queries per wave (one wave per SIMD: measured slower, kept for experiments).  Head widths above 160 (padded)
 *   always use attention.hip.  Results agree to fp32 summation order (variant 1's deferred rescale: to bf16 rounding of P).
 * "wide_gemm" (default 0): tile choice of the tiled GEMM layers (the widths the row-stationary kernels do not hold): 0 by shape,
 *     1 always csrc/gemm.hip (128 x 128, register staged), 2 / 3 csrc/gemm256.hip with 256 x 256 / 256 x 128 tiles wherever it applies.
 * "ctc_mfma" (default 2): the CTC head (fc + argmax).  2: split-bf16 operands on the bf16 MFMA (x_hi W_hi + x_hi W_lo + x_lo W_hi, fp32
 *   accumulation: logits within ~2^-16 relative of the fp32 head) - bf16 path only, the fp32-operand mode runs 1; 1: the fp32 MFMA; 0: the VALU
 *   kernel.  1 and 0 are k-ordered fp32 fma chains: bit-identical logits and labels.
 * "chain_small_m" (default 4096): chain launches (csrc/chain.hip) of at most this many rows run 2-wave workgroups (64 rows) instead of 8-wave ones:
 *   small-batch latency (B <= 8 utterances of 10 s); a wave computes its 32 rows identically in both shapes, so outputs do not depend on it.  0 = off.
 * "dwconv_mfma" (default 1; round 5): stride-1 depthwise convolutions on the matrix pipe (csrc/conv.hip dwconv_mfma_kernel: 4 x 4 x 4 Toeplitz blocks, one per
 *   channel, fp32 taps as bf16 hi + lo halves): 1 = kernel size 15, 2 = also 31 / 7, 0 = the VALU kernel everywhere.  Outputs agree with the VALU kernel's
 *   up to the summation order (a bf16 output may move by one ulp).
 * "attn_waves" = 2 (round 5): attention2.hip with two staging sets (two workgroups per CU at head width 64; the default is one set, three workgroups).
 * "chain_pair" (default 5), "chain_pair_min_d", "chain_w2cm", "chain_nt": kernel selection of the chains at padded width 256 (csrc/chain2.hip, chain3.hip; bit-identical).
 * "chain_variant" (0 / 1), "chain_full_max" (widest stage that runs chain A as one kernel; set before finalize to widen), "attn_waves"
 *   (4 / 8, attention.hip), "rs_variant" (0 / 1), "ffn_variant" (0 .. 2), "head_major_odd" (0 / 1), "exact_attention" (0 tiled / 2 tiled with 16-row workgroups / 1 one wave per query row; fp32 mode, bit-identical): tuning / test switches of the kernel launchers that were
 *   process-global EFFCONF_* environment variables until round 2; per handle now.  (Still read from the environment, once, as
 *   profiling / test hooks: EFFCONF_POISON_GUARDS at create, EFFCONF_{CHAIN,ATTN,FFN}_PHASES for the in-kernel phase profilers of chain.hip / attention.hip /
 *   rsgemm.hip.  EFFCONF_CHAIN2_PHASES, EFFCONF_CHAIN3_PHASES and EFFCONF_ATTN_ABLATE exist in the tuning library of tools/build_ablate.py only - round 6.)
 * "exact_fp32" (default 0): fp32-operand precision mode (csrc/exact.hip): every GEMM on fp32 MFMA, fp32 attention / convolutions,
 *   so that greedy CTC label sequences equal the reference's CPU fp32 path (model_ctc.py:99-133) wherever its top-2 logit margins
 *   exceed fp32 summation-order noise; ~10x slower than the default bf16-operand path.  Set to 1 BEFORE effconf_encoder_finalize
 *   (the fp32 tensors are uploaded there; the workspace query then covers both modes); afterwards it toggles the mode per handle.
 *   2 (round 4, csrc/split.hip; round 6, csrc/sxf.hip + sxf_ffn.hip + sxf_chain.hip): fp32 tensors with every GEMM and the attention products on the fp16
 *   matrix pipe, each operand split into two fp16 numbers (three MFMAs per product, products accurate to ~2^-21): label sequences identical to the
 *   reference's on every golden and on the bench's oracle samples of Small / Medium / Large.  Round 6: ONE attention kernel per block with the scores on the
 *   CU, the row-local work of a block as two kernels, the front end as one kernel, ragged batches, causal and streaming configurations - 2.6x the bf16 step on
 *   EfficientConformerCTCSmall (5.7x in round 5), 2.4x / 2.7x on Medium / Large.  Set to 2 BEFORE finalize (the split weight images are built there; such a handle then serves 2, 1 and 0);
 *   a handle finalized with 1 refuses 2.
 * "split_chain" (default 1), "split_ffn" (default 1): split mode on the fused row-local kernels (csrc/sxf_chain.hip; csrc/sxf_ffn.hip when split_chain = 0);
 *   0 / 0 = LayerNorm, split GEMM, GLU kernels per module (tests: the same results within a few 1e-6, another summation order).  A debug trace
 *   (effconf_encoder_trace_*) runs the per-module kernels: the chains never write the intermediate states.
 * "split_sublin" (default 1; round 6, csrc/sxf_sub.hip): split mode, one-layer subsampler (the EfficientConformer configurations): Conv2d + BatchNorm + Swish + flatten
 *   + Linear as ONE kernel whose (frames, C F') activation stays in registers (the conv as a 16-tap MFMA product whose accumulators are the B fragments of the
 *   Linear's product), ragged batches on the frames that exist; 0 = fp32 VALU convolution + split GEMM (+ row gather) - the same results within a few 1e-6.
 *   Small split step 14.6 -> 12.9 ms.  A debug trace runs the per-module kernels (it wants the "subsample" activation).
 * MODE MATRIX (what a forward accepts; everything else returns an error, never a silent fallback):
 *   bf16 path (exact_fp32 = 0): rectangular batches (effconf_encoder_forward / _forward_mel), ragged batches (effconf_encoder_forward_ragged; head
 *     widths <= 160 padded), streaming contexts / causal configurations (EcConfig; natural Q / K / V layout, head widths <= 160 padded), attention
 *     maps (effconf_encoder_set_attention_outputs; ragged batches since round 4: rectangles sized for the LONGEST utterance, utterance b's own
 *     Tg(b) x Tg(b) block = its map run alone, zeros elsewhere).
 *   split mode (exact_fp32 = 2; round 6): rectangular AND ragged batches, finite left_context / right_context AND causal configurations (causal relative
 *     tables, causal depthwise padding); attention maps on rectangular batches (they run round 4's scores-in-memory kernels of csrc/split.hip).
 *   fp32 mode (exact_fp32 = 1): rectangular batches, attention maps, finite left_context / right_context; NOT ragged batches, NOT causal configurations
 *     (the forward fails). */
int effconf_encoder_set_option(EcEncoder* enc, const char* name, int32_t value);

/* ---- per-launch event profiler (bench / tuning only) --------------------------------------- */
/* When enabled every kernel launch of the forward is bracketed by two hipEventRecord on the caller's stream.
 * Classes: 0 mel, 1 subsample conv, 2 FFN GEMMs, 3 other GEMMs, 4 LayerNorm, 5 attention, 6 depthwise conv, 7 misc.
 * effconf_profile_read synchronises the device and returns, for one class, the summed kernel time, the number of
 * launches and the summed ALGORITHMIC flops / bytes of those launches since the last enable. */
int effconf_profile_enable(EcEncoder* enc, int32_t enable);
int effconf_profile_read(EcEncoder* enc, int32_t cls, double* total_ms, int64_t* launches, double* flops, double* bytes);

/* ---- debug trace (tests only) ------------------------------------------------------------- */
/* When a trace arena is set, every forward copies its intermediate tensors into it (device-to-device,
 * same stream).  Entries describe name / byte offset / rows / cols / leading dim / dtype (0=f32, 1=bf16, 2=i32). */
int effconf_encoder_set_trace(EcEncoder* enc, void* dev_arena, size_t bytes);
int32_t effconf_encoder_trace_count(const EcEncoder* enc);
int effconf_encoder_trace_entry(const EcEncoder* enc, int32_t i, char* name64, int64_t* offset, int64_t* rows, int64_t* cols,
                                int64_t* ld, int32_t* dtype);

/* ---- diagnostics: include/effconf_debug.h, exported by the SEPARATE library libeffconf_debug.so (tests and tools only); libeffconf.so has none of them */

/* ---- attention maps: the third return value of the reference's ConformerEncoder.forward (encoders.py:126-142: att_w of every block,
 * (batch, heads, Tg, Tg) softmax rows; no caller on the hot path reads them, so they are opt-in).  effconf_encoder_attention_dims fills
 * heads[k] / tg[k] per block for inputs of n samples (from_audio) or mel frames; effconf_encoder_set_attention_outputs registers one device
 * buffer of batch * heads[k] * tg[k] * tg[k] floats per block (null entries skip a block; maps = null or n_blocks = 0 clears the
 * registration) - every following effconf_encoder_forward / _forward_ragged writes them (ragged: dims for n = the longest utterance; bf16 path: recomputed in fp32 from the bf16
 * Q / K / E operands; fp32 path: the kernel's own probabilities). */
int effconf_encoder_attention_dims(EcEncoder* enc, int32_t n, int32_t from_audio, int32_t* heads, int32_t* tg);
int effconf_encoder_set_attention_outputs(EcEncoder* enc, float* const* maps, int32_t n_blocks);

/* ---- host helper of the batching front door (reference: utils/preprocessing.py:33-45, collate_fn_pad zero-pads a sorted batch) -------
 * Copies n host rows src[i][0 .. len[i]) to dst + i * pitch (floats) on `threads` host threads, zero-filling dst[i][len[i] .. pitch) when
 * zero_pad != 0 (ragged batches never read the pad samples: pass 0).  dst is typically pinned memory, so that ONE H2D copy follows.  No
 * device work, no stream; returns when the copy is complete. */
int effconf_host_pack_rows(const float* const* src, const int64_t* len, int32_t n, float* dst, int64_t pitch, int32_t zero_pad,
                           int32_t threads);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* EFFCONF_H */
