/* effconf_debug.h - diagnostic entry points of libeffconf_debug.so (efficientconformer_amd/_build.py builds it beside the product library).
 *
 * libeffconf_debug.so = every object of libeffconf.so + csrc/debug.hip (hazard reproducer, spin, LDS-fill probe), the packed-fp32 diagnostic build of
 * csrc/mel.hip, and the entry points below.  It exports the whole product ABI (include/effconf.h) as well, so a test can run on it alone; nothing of the
 * package loads it (tests / tools: efficientconformer_amd._lib.load_debug()).  The product library carries none of these symbols and no packed-fp32 kernel.
 */
#ifndef EFFCONF_DEBUG_H
#define EFFCONF_DEBUG_H
#include "effconf.h"
#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

/* effconf_mel_frontend with a diagnostic variant of mel_kernel (csrc/mel.hip: 1 canaries, 2 self-verifying hand-offs,
 * 4 workgroup barriers, 8 the build WITH packed-fp32 VALU instructions = round 1's hazardous kernel; bits combine), `extra_lds` bytes of unused dynamic LDS per workgroup, and 8 u32 counters (dev). */
int effconf_debug_mel(EcEncoder* enc, int32_t variant, int32_t extra_lds, const float* audio, int32_t batch, int32_t n_samples,
                      float* mel, uint32_t* counters, void* stream);
/* One synthetic kernel that loads a single compute-unit resource (csrc/debug.hip): kind 0 LDS 16-byte hammer, 1 VALU +
 * transcendental, 2 global loads, 3 global stores, 4 MFMA, 5 LDS publish + barrier loop, 6 LDS 4-byte hammer. */
int effconf_debug_neighbour(int32_t kind, int32_t blocks, int32_t lds_bytes, int32_t iters, float* buf, size_t n_floats, void* stream);
/* One Linear layer on the tiled bf16 GEMM kernels alone (kernel-level tests and tuning of csrc/gemm.hip and csrc/gemm256.hip; reference
 * models/layers.py:57-67): c = epilogue(a w^T + bias).  a: bf16 [m][lda]; w: bf16 [>= round_up(n, 128)][ldw = round_up(k, 64)], zero
 * padded (for GLU: rows interleaved per 32 channels, a | b); bias: [>= round_up(n, 128)].  epi: 0 fp32 [m][ldc], 1 bf16, 2 Swish
 * bf16, 3 fp32 r + alpha * (...), 4 GLU bf16 [m][n / 2].  wide: 0 tile picked by shape, 1 the 128 x 128 kernel, 2 / 3 the LDS-DMA
 * kernel with 256 x 256 / 256 x 128 tiles.  All pointers are device pointers. */
int effconf_debug_gemm(const uint16_t* a, int32_t lda, const uint16_t* w, int32_t ldw, const float* bias, int32_t m, int32_t n, int32_t k,
                       int32_t epi, int32_t wide, void* c, int32_t ldc, const float* r, int32_t ldr, float alpha, void* stream);

/* One self-contained victim: 16 chains per lane of a single instruction class (0 v_fma_f32, 1 v_pk_fma_f32, 2 v_pk_mul/add_f32,
 * 3 v_log/v_exp_f32, 4 v_mul/v_add_f32, 5 integer, 6 wave-local LDS exchange); out dev f32 (blocks * 256 * 16). */
int effconf_debug_victim(int32_t kind, int32_t blocks, int32_t iters, float* out, void* stream);
/* One Linear layer on the split-precision GEMM kernel alone (csrc/split.hip; kernel-level tests and tools/sx_gemm_bench.py): c = epilogue(a w^T
 * + bias) with a fp32 [m][lda], w given as its two fp16 images h = fp16(w), l = fp16((w - h) * 2048), each packed k-tile major [ldh / 32][n][32]
 * (ldh = round_up(k, 32), zero padded: element (row, col) at ((col / 32) * n + row) * 32 + col % 32), products accurate to ~2^-21.  epi: 0 plain, 1 Swish, 2 c = r + alpha * (...).  All pointers are device pointers. */
int effconf_debug_sx_gemm(const float* a, int32_t lda, const uint16_t* w_hi, const uint16_t* w_lo, int32_t ldh, const float* bias, int32_t m, int32_t n,
                          int32_t k, int32_t epi, float* c, int32_t ldc, const float* r, int32_t ldr, float alpha, void* stream);
/* HOST function (no GPU): the A-operand table of dwconv_mfma_kernel (csrc/conv.hip) for folded depthwise taps w_kc [ksize][channels] (fp32):
 * dst [channels][4 rows i][groups q = (ksize + 6) / 4][bf16 hi (4 taps) | bf16 lo (4 taps)] of w[4 q + k - i] (zero outside the taps); tests/test_abi_and_host.py. */
int effconf_debug_pack_dwconv_mfma(const float* w_kc, int32_t ksize, int32_t channels, uint16_t* dst, size_t dst_elems);
/* The depthwise-convolution stage of the ConvolutionModule alone (reference modules.py:516-518 with BatchNorm folded into the taps; csrc/conv.hip): g dev bf16
 * [batch * frames][ld] (GLU output) -> out dev bf16 [batch * ((frames - 1) / stride + 1)][ld] = Swish(sum_j w[j][c] gpad[s t + j][c] + bias[c]); w_kc_host
 * [ksize][channels] / bias_host [channels]: HOST fp32 (the entry uploads them, builds the Toeplitz table for use_mfma = 1 and synchronises: tests only).
 * use_mfma: 1 = dwconv_mfma_kernel (stride 1, kernel size 15 / 31 / 7), 0 = dwconv_kernel (VALU).  causal: pre-padding (k - 1, 0) instead of "same". */
int effconf_debug_dwconv(const uint16_t* g, int32_t batch, int32_t frames, int32_t channels, int32_t ld, const float* w_kc_host, const float* bias_host,
                         int32_t ksize, int32_t stride, int32_t use_mfma, int32_t causal, uint16_t* out, void* stream);
/* The fused split-precision FeedForwardModule kernel alone (csrc/sxf_ffn.hip; reference modules.py:385-395, blocks.py:122, 132-135) on a handle finalized with
 * exact_fp32 = 2: y = x + 1/2 FFN_which(LayerNorm(x)) on fp32 rows (rows, D), with_norm != 0: followed by the block-final LayerNorm.  ablate: timing-only switches
 * of the diagnostic build (1 no first product, 2 no Swish, 4 no second product, 8 no weight stream, 16 no per-chunk barrier; results are wrong by construction);
 * tools/sxf_ffn_probe.py. */
int effconf_debug_sxf_ffn(EcEncoder* enc, int32_t block, int32_t which, const float* x, int32_t rows, float* y, int32_t with_norm, int32_t ablate, void* stream);
/* One idle wave that occupies `stream` for `microseconds` (tools/overlap_probe.py: the duration of an xGMI transfer a single GPU cannot make). */
int effconf_debug_spin(double microseconds, void* stream);
/* diagnostics (tools/lds_fill_rate_probe.py): `blocks` workgroups of `waves` waves each walk the same `window` bytes of `src` (dev) into LDS, `kib_per_wave`
 * (multiple of 8) 1-KiB wave-instructions per wave and pass; mode 0 = LDS-DMA (global_load_lds_dwordx4), 1 = loads + ds_write_b128.
 * out (dev, 2 * blocks uint64): {cycles, bytes} per workgroup. */
int effconf_debug_lds_fill(int32_t mode, int32_t blocks, int32_t waves, const void* src, size_t window, int32_t kib_per_wave, int32_t passes, uint64_t* out, void* stream);


#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif
