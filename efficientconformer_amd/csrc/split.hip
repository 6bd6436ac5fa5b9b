// Split-precision ("exact_fp32" = 2) kernels: fp32 tensors in HBM, products on the fp16 matrix pipe at close to fp32 accuracy.
//
// The label-exact mode of round 2 / 3 (exact.hip) multiplies on v_mfma_f32_32x32x2_f32, 1 / 16 of the bf16 / fp16 MFMA rate, and ran the
// attention dot products on the VALU: 58 ms per Small step, 10.6 x the bf16 path.  Here every operand x is split into two fp16 numbers
//     x = h + l / 2048,   h = fp16(x),   l = fp16((x - h) * 2048)
// (h carries 11 significant bits, l the next 11: x - h is exact in fp32 and |x - h| <= 2^-11 |x|, so l has x's own magnitude and
// stays a NORMAL fp16 number wherever h is one - no dependence on how the matrix pipe treats fp16 subnormals) and a product sum becomes
//     sum a w = sum a_h w_h + (sum a_h w_l + sum a_l w_h) / 2048          (a_l w_l / 2^22 dropped: 2^-22 relative)
// on v_mfma_f32_32x32x16_f16 with fp32 accumulation in TWO accumulators (main, correction): 3 MFMAs where the bf16 path issues 1, products
// accurate to ~2^-21, against 2^-24 for fp32 and 2^-9 for bf16 operands.  Everything else of the mode (LayerNorm, softmax, GLU, depthwise and
// subsampling convolutions, residual stream) stays fp32 as in exact.hip; reference: models/encoders.py:97-142, blocks.py:119-137,
// attentions.py:549-718, modules.py:385-395, 511-525.
//
//   sx_gemm_kernel    C = epi(A W^T + b): A fp32 (split while it is staged into LDS), W pre-split at finalize into two fp16 images packed
//                     k-tile major ([K / 32][N][32]: with row-major images a k-tile touched half a cache line per row);
//                     128 x 128 x 32 tiles, 4 waves x (2 x 2) 32 x 32 MFMA tiles, register-staged prefetch of the next k-tile.
//   sx_scores_kernel  one 64 x 64 (query, key) tile of S = ((Q + u) K^T + rel_to_abs((Q + v) E^T)) / sqrt(d) + mask: the positional product
//                     on the 127-row band the tile touches, realigned through LDS (PE[i][j - i + 63]); scores go to a global fp32 buffer.
//   sx_pv_kernel      row softmax of those scores (reference order: exp(s - max) / sum) and O = P V on the matrix pipe, V transposed in LDS.
#include "kernels.h"
#include "sx_common.h"

namespace {

using namespace sx;

// ------------------------------------------------------------------------------------------------ GEMM
constexpr int SBM = 128, SBN = 128, SBK = 32, SROW = SBK * 2;          // 64-byte rows, the 16-byte chunk index XOR-ed with (row >> 2) & 3: rows r, r + 4, r + 8,
// r + 12 start on the same banks and get four different chunk positions, so the fragment reads of a 16-lane group cover all 64 banks once.
// (80-byte padded rows were conflict-free too, but two 80 KiB stage pairs are exactly the CU's 160 KiB: the second workgroup per CU did not
// get in - 64 KiB leaves room for it.)
__device__ __forceinline__ int sx_off(int row, int chunk) { return row * SROW + ((chunk ^ ((row >> 2) & 3)) << 4); }

constexpr int SX_STAGE = 4 * SBM * SROW;                           // bytes of one LDS stage: A_hi | A_lo | W_hi | W_lo, [128][SROW] each
constexpr int SX_CLD = 132;                                        // floats per row of the epilogue's staging tile

__global__ __launch_bounds__(256, 2) void sx_gemm_kernel(const SxGemmParams q) {
    const ExGemmParams& p = q.g;
    extern __shared__ __attribute__((aligned(16))) char sm[];       // two stages (double buffer: one barrier per k-tile)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, lr = lane & 31, kh = lane >> 5;
    // N tiles fastest: the workgroups that are resident together share their A rows (read once from HBM, then L2 hits) and walk the small W
    // images; with M fastest every column of N tiles re-read the whole of A from HBM / MALL (2.3 GB per Large FFN GEMM at 3.3 TB/s: the bound)
    const int m0 = blockIdx.y * SBM, n0 = blockIdx.x * SBN;
    // staging roles: A rows ar + 32 i (float4 at column akq of the k-tile), W rows wr + 64 i (8 halfs at wch) of both images
    const int ar = tid >> 3, akq = (tid & 7) * 4, wr = tid >> 2, wch = (tid & 3) * 8;
    // 32-bit element offsets from the (wave-uniform) base pointers instead of 64-bit pointers per row: 6 registers instead of 16 - the kernel
    // sits at the 256-register budget of two waves per SIMD (launch check: the operands stay below 2^32 elements)
    uint32_t ao[4], wo[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int am = m0 + ar + 32 * i;
        am = am < p.M ? am : p.M - 1;
        const long long arow = p.a_rows ? (long long)(am / p.a_rows) * p.a_pitch + (long long)(am % p.a_rows) * p.a_stride : am;
        ao[i] = (uint32_t)(arow * p.lda + akq);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int bn = n0 + wr + 64 * i;
        bn = bn < p.N ? bn : p.N - 1;
        wo[i] = (uint32_t)(bn * SBK + wch);              // images are packed k-tile major: [K / 32][N][32] - a k-tile's slice of 16 consecutive rows is 1 KiB contiguous
    }
    f32x16 acc[2][2], acx[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; acx[i][j][r] = 0.f; }
    // staging registers as native vector types (HIP's float4 / uint4 structs in an array are copied by memcpy and end up in scratch)
    typedef float f4v __attribute__((ext_vector_type(4)));
    typedef uint32_t u4v __attribute__((ext_vector_type(4)));
    f4v ra[4];
    u4v rh[2], rl[2];
    bool keep = true;
    auto gload_a = [&](int k0) __attribute__((always_inline)) {
        const bool ok = k0 + akq < p.K;                          // K % 4 == 0: a float4 is inside or outside as a whole
        const int ko = ok ? k0 : 0;                              // always a mapped address (the row's own first k-tile; with K < 32 the NEXT row's, behind the last row whatever the buffer holds): replaced by zeros
        keep = ok;                                               // applied as a SELECT when the registers are published (a use here would wait for the loads inside the loop; a multiply by 0 would pass a NaN on); branch-free: the loop body stays ONE basic block
#pragma unroll
        for (int i = 0; i < 4; ++i) ra[i] = *reinterpret_cast<const f4v*>(p.A + (ao[i] + (uint32_t)ko));
    };
    auto gload_w = [&](int k0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            rh[i] = *reinterpret_cast<const u4v*>(q.Whi + (wo[i] + (uint32_t)k0 * (uint32_t)p.N));       // k-tile k0 / 32 starts at (k0 / 32) * N * 32
            rl[i] = *reinterpret_cast<const u4v*>(q.Wlo + (wo[i] + (uint32_t)k0 * (uint32_t)p.N));
        }
    };
    auto gload = [&](int k0) __attribute__((always_inline)) { gload_a(k0); gload_w(k0); };
    typedef uint32_t u2v __attribute__((ext_vector_type(2)));
    u2v ph[4], pl[4];                                            // the A registers of the next k-tile, split
    auto split_regs = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            uint32_t h0, l0, h1, l1;
            split2(keep ? ra[i][0] : 0.f, keep ? ra[i][1] : 0.f, h0, l0);
            split2(keep ? ra[i][2] : 0.f, keep ? ra[i][3] : 0.f, h1, l1);
            ph[i] = u2v{h0, h1}; pl[i] = u2v{l0, l1};
        }
    };
    auto write_regs = [&](char* st) __attribute__((always_inline)) {     // split A registers + W image registers -> LDS stage
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int o = sx_off(ar + 32 * i, akq >> 3) + (akq & 7) * 2;
            *reinterpret_cast<u2v*>(st + o) = ph[i];
            *reinterpret_cast<u2v*>(st + SBM * SROW + o) = pl[i];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int o = sx_off(wr + 64 * i, wch >> 3);
            *reinterpret_cast<u4v*>(st + 2 * SBM * SROW + o) = rh[i];
            *reinterpret_cast<u4v*>(st + 3 * SBM * SROW + o) = rl[i];
        }
    };
    auto compute = [&](const char* st, int ks) __attribute__((always_inline)) {
        const int kc = ks * 2 + kh;                              // 16-byte chunk of the k-tile row
        f16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int o = sx_off(wm * 64 + 32 * i + lr, kc);
            ah[i] = *reinterpret_cast<const f16x8*>(st + o);
            al[i] = *reinterpret_cast<const f16x8*>(st + SBM * SROW + o);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int o = sx_off(wn * 64 + 32 * j + lr, kc);
            bh[j] = *reinterpret_cast<const f16x8*>(st + 2 * SBM * SROW + o);
            bl[j] = *reinterpret_cast<const f16x8*>(st + 3 * SBM * SROW + o);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
                acx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acx[i][j], 0, 0, 0);
                acx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acx[i][j], 0, 0, 0);
            }
    };
    // Pipeline: while the MFMAs of k-tile t run on stage t & 1, the SAME wave splits k-tile t + 1 (already in registers) - its VALU
    // instructions issue between the matrix instructions - writes it to the other stage and requests k-tile t + 2; one barrier per k-tile.
    // The fragment reads come first in program order: the compiler cannot tell the two stages apart, and LDS writes in front of the
    // reads would order every MFMA behind the whole split.
    const int nk = (p.K + SBK - 1) / SBK;
    gload(0);
    split_regs();
    write_regs(sm);
    if (nk > 1) gload(SBK);
    __syncthreads();
    for (int t = 0; t + 1 < nk; ++t) {                           // steady state: one basic block (no branch inside)
        const char* st = sm + (t & 1) * SX_STAGE;
        const int t2 = t + 2 < nk ? t + 2 : nk - 1;              // past the end: the last k-tile again (never published)
        compute(st, 0);
        split_regs();
        gload_a(t2 * SBK);                                       // the A registers are free as soon as they are split: a whole iteration for these loads to land
        compute(st, 1);
        write_regs(sm + ((t + 1) & 1) * SX_STAGE);
        gload_w(t2 * SBK);
#pragma unroll
        for (int g = 0; g < 12; ++g) {                           // one matrix instruction (32 cycles in the pipe), then a few of the split's VALU instructions
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 9, 0);
        }
        __syncthreads();
    }
    compute(sm + ((nk - 1) & 1) * SX_STAGE, 0);
    compute(sm + ((nk - 1) & 1) * SX_STAGE, 1);
    __syncthreads();
    // Epilogue through LDS: 64 scalar 4-byte stores (and residual loads) per lane are store-ISSUE bound (they were most of the kernel at K = 120);
    // each half of the tile (64 rows) is staged as fp32 and leaves as 16-byte row pieces: 8 per thread and half.
    float* sC = reinterpret_cast<float*>(sm);
    const int ec = (tid & 31) * 4, er = tid >> 5;                  // this thread's 4 columns / first row of a half
    const int n = n0 + ec;
    const bool nok = n < p.N;                                      // N % 4 == 0: a quad is inside or outside as a whole
    float4 bz = make_float4(0.f, 0.f, 0.f, 0.f);
    if (nok && p.bias) bz = *reinterpret_cast<const float4*>(p.bias + n);
    float* cb = p.C;
    int ncol = n;
    if (nok && p.split_cols > 0) { cb += (size_t)(n / p.split_cols) * p.split_stride; ncol = n % p.split_cols; }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if (wm == half) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        sC[(32 * i + (r & 3) + 8 * (r >> 2) + 4 * kh) * SX_CLD + wn * 64 + 32 * j + lr] = fmaf(acx[i][j][r], LO_INV, acc[i][j][r]);
        }
        __syncthreads();
        if (nok) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int row = er + 8 * i, m = m0 + 64 * half + row;
                if (m >= p.M) continue;
                const float4 a = *reinterpret_cast<const float4*>(sC + row * SX_CLD + ec);
                float v[4] = {a.x + bz.x, a.y + bz.y, a.z + bz.z, a.w + bz.w};
                if (p.epi == 1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] * sx_rcp(1.0f + sx_expf(fminf(-v[e], 87.0f)));   // Swish (modules.py:389); the clamp: exp2 overflows beyond ~88.7, inf then turns sx_rcp's Newton step into NaN - x sigmoid(x) is -x e^x -> 0 there (the reference's value to fp32)
                } else if (p.epi == 2) {
                    const float4 rr = *reinterpret_cast<const float4*>(p.R + (size_t)m * p.ldr + n);
                    v[0] = fmaf(p.alpha, v[0], rr.x); v[1] = fmaf(p.alpha, v[1], rr.y); v[2] = fmaf(p.alpha, v[2], rr.z); v[3] = fmaf(p.alpha, v[3], rr.w);
                }
                const long long crow = p.c_rows ? (long long)(m / p.c_rows) * p.c_pitch + m % p.c_rows : m;
                *reinterpret_cast<float4*>(cb + crow * p.ldc + ncol) = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------ attention scores
constexpr int PE_LD = 132;                                       // floats per query row of the band product in LDS

template <int KS>                                                // 16-wide k-steps of the padded head width PK = 16 KS
__global__ __launch_bounds__(256) void sx_scores_kernel(const SxAttnParams q) {
    const ExAttnParams& p = q.a;
    constexpr int PK = 16 * KS, ROW = PK * 2 + 16;               // bytes per staged row (conflict-free 16-byte fragment reads: (PK / 8 + 1) odd)
    constexpr int CPR = PK / 4;                                  // 4-float chunks per row
    extern __shared__ __attribute__((aligned(16))) char sm[];
    // A workgroup owns 64 grouped query rows of one (utterance, head) and walks ALL key tiles: the query fragments are built once, and of
    // the 127-row positional band of a (query tile, key tile) pair only 64 rows are new per key tile (two 64-row halves that swap roles).
    char* sKh = sm;                                              // [64][ROW]
    char* sKl = sKh + 64 * ROW;
    char* sE = sKl + 64 * ROW;                                   // 2 halves x (hi [64][ROW] | lo [64][ROW]): band rows 0..63 / 64..127 of the current key tile
    float* sPE = reinterpret_cast<float*>(sE + 4 * 64 * ROW);    // [64][PE_LD]
    float* suv = sPE + 64 * PE_LD;                               // [2][PK]: u | v of this head's columns
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wq = wave >> 1, wk = wave & 1, lr = lane & 31, kh = lane >> 5;
    const int i0 = blockIdx.x * 64, b = blockIdx.y / p.H, h = blockIdx.y % p.H;
    const int d = p.d, Tg = p.Tg;
    const size_t hb = (size_t)h * d, gd = (size_t)p.G * p.D;
    const float* qbase = p.q + (size_t)b * p.Tp * p.D + hb;
    const float* kbase = p.k + (size_t)b * p.Tp * p.D + hb;
    const float* ebase = p.e + hb;
    for (int x = tid; x < PK; x += 256) {                        // u, v broadcast over the un-grouped feature axis: column (h d + x) mod D
        int n = (int)((hb + x) % p.D);
        suv[x] = x < d ? p.u[n] : 0.f;
        suv[PK + x] = x < d ? p.vb[n] : 0.f;
    }
    __syncthreads();
    // ---- this lane's query row as MFMA A fragments: (Q + u) and (Q + v), split
    f16x8 quh[KS], qul[KS], qvh[KS], qvl[KS];
    {
        int i = i0 + 32 * wq + lr;
        i = i < Tg ? i : Tg - 1;
        const float* qr = qbase + gd * i;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int x = 16 * s + 8 * kh;
            const float4 a = ld_span4(qr, x, d), c = ld_span4(qr, x + 4, d);
            const float4 u0 = *reinterpret_cast<const float4*>(suv + x), u1 = *reinterpret_cast<const float4*>(suv + x + 4);
            const float4 v0 = *reinterpret_cast<const float4*>(suv + PK + x), v1 = *reinterpret_cast<const float4*>(suv + PK + x + 4);
            // pad columns (x >= d): q = 0 (ld_span4) and u = v = 0 (suv)
            uint32_t uh[4], ul[4], vh[4], vl[4];
            split2(a.x + u0.x, a.y + u0.y, uh[0], ul[0]); split2(a.z + u0.z, a.w + u0.w, uh[1], ul[1]);
            split2(c.x + u1.x, c.y + u1.y, uh[2], ul[2]); split2(c.z + u1.z, c.w + u1.w, uh[3], ul[3]);
            split2(a.x + v0.x, a.y + v0.y, vh[0], vl[0]); split2(a.z + v0.z, a.w + v0.w, vh[1], vl[1]);
            split2(c.x + v1.x, c.y + v1.y, vh[2], vl[2]); split2(c.z + v1.z, c.w + v1.w, vh[3], vl[3]);
            quh[s] = as_f16x8(make_uint4(uh[0], uh[1], uh[2], uh[3])); qul[s] = as_f16x8(make_uint4(ul[0], ul[1], ul[2], ul[3]));
            qvh[s] = as_f16x8(make_uint4(vh[0], vh[1], vh[2], vh[3])); qvl[s] = as_f16x8(make_uint4(vl[0], vl[1], vl[2], vl[3]));
        }
    }
    auto stage_rows = [&](char* dh, char* dl, int nrows, auto rowptr) __attribute__((always_inline)) {     // fp32 rows -> split fp16 rows, zero padded to PK
        for (int c = tid; c < nrows * CPR; c += 256) {
            const int r = c / CPR, x = (c - r * CPR) * 4;
            const float4 v = ld_span4(rowptr(r), x, d);
            uint32_t h0, l0, h1, l1;
            split2(v.x, v.y, h0, l0); split2(v.z, v.w, h1, l1);
            *reinterpret_cast<uint2*>(dh + r * ROW + x * 2) = make_uint2(h0, h1);
            *reinterpret_cast<uint2*>(dl + r * ROW + x * 2) = make_uint2(l0, l1);
        }
    };
    auto erow = [&](int w, int j0) {                             // band row w of key tile j0: E row Tg - 1 + j0 - i0 - 63 + w, clamped (rows no valid pair touches)
        int rel = Tg - 1 + j0 - i0 - 63 + w;
        rel = rel < 0 ? 0 : (rel > 2 * Tg - 2 ? 2 * Tg - 2 : rel);
        return ebase + gd * rel;
    };
    const float irs = 1.0f / sqrtf((float)d);
    const int len = p.lens[b];
    float* srow = q.scores + ((size_t)blockIdx.y * Tg) * q.TgP;
    // the lower band half of the first key tile; afterwards every key tile brings its upper half and inherits the previous upper half as its lower one
    stage_rows(sE, sE + 64 * ROW, 64, [&](int r) { return erow(r, 0); });
    int cur = 0;                                                 // half holding band rows 0..63 of the current key tile
    for (int j0 = 0; j0 < Tg; j0 += 64, cur ^= 1) {
        char* eLo = sE + cur * 2 * 64 * ROW;
        char* eHi = sE + (cur ^ 1) * 2 * 64 * ROW;
        stage_rows(sKh, sKl, 64, [&](int r) { const int j = j0 + r < Tg ? j0 + r : Tg - 1; return kbase + gd * j; });
        stage_rows(eHi, eHi + 64 * ROW, 64, [&](int r) { return erow(64 + r, j0); });
        __syncthreads();
        // ---- S1 = (Q + u) K^T on this wave's 32 x 32 tile, PE = (Q + v) E_band^T on its 32 queries x the band half wk
        f32x16 s1h, s1x, peh[2], pex[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) { s1h[r] = 0.f; s1x[r] = 0.f; peh[0][r] = 0.f; pex[0][r] = 0.f; peh[1][r] = 0.f; pex[1][r] = 0.f; }
        const char* eh_ = wk ? eHi : eLo;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int ko = (16 * s + 8 * kh) * 2;
            const f16x8 bh = *reinterpret_cast<const f16x8*>(sKh + (32 * wk + lr) * ROW + ko), bl = *reinterpret_cast<const f16x8*>(sKl + (32 * wk + lr) * ROW + ko);
            s1h = __builtin_amdgcn_mfma_f32_32x32x16_f16(quh[s], bh, s1h, 0, 0, 0);
            s1x = __builtin_amdgcn_mfma_f32_32x32x16_f16(quh[s], bl, s1x, 0, 0, 0);
            s1x = __builtin_amdgcn_mfma_f32_32x32x16_f16(qul[s], bh, s1x, 0, 0, 0);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int er = 32 * t + lr;
                const f16x8 eh = *reinterpret_cast<const f16x8*>(eh_ + er * ROW + ko), el = *reinterpret_cast<const f16x8*>(eh_ + 64 * ROW + er * ROW + ko);
                peh[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(qvh[s], eh, peh[t], 0, 0, 0);
                pex[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(qvh[s], el, pex[t], 0, 0, 0);
                pex[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(qvl[s], eh, pex[t], 0, 0, 0);
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = 32 * wq + (r & 3) + 8 * (r >> 2) + 4 * kh;
                sPE[m * PE_LD + 64 * wk + 32 * t + lr] = fmaf(pex[t][r], LO_INV, peh[t][r]);
            }
        __syncthreads();                                         // band product complete; every wave is done with the K tile and the lower band half
        // ---- rel_to_abs: S2[i][j] = PE[i][j - i + 63]; scale, additive key mask (attentions.py:692-701), store
        const int jj = 32 * wk + lr, j = j0 + jj;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = 32 * wq + (r & 3) + 8 * (r >> 2) + 4 * kh, i = i0 + m;
            if (i >= Tg || j >= Tg) continue;
            float sv = (fmaf(s1x[r], LO_INV, s1h[r]) + sPE[m * PE_LD + jj - m + 63]) * irs;
            if (p.G * j >= len || j - i > p.band_r || i - j > p.band_l) sv += -1e9f;      // ONE additive mask = max(padding mask, streaming mask) (attentions.py:698-701, 1377-1403)
            srow[(size_t)i * q.TgP + j] = sv;
        }
        // the next iteration's staging overwrites sK and the (now free) lower half, not sPE: the skewed reads above are ordered against the
        // next band product by the barrier after the next staging
    }
}

// ------------------------------------------------------------------------------------------------ softmax + P V
constexpr int PROW = 64 * 2 + 16;                                // bytes per row of the P / V^T tiles (64 keys)

template <int NT>                                                // 32-column output tiles: 32 NT >= d
__global__ __launch_bounds__(256) void sx_pv_kernel(const SxAttnParams q) {
    const ExAttnParams& p = q.a;
    extern __shared__ __attribute__((aligned(16))) char sm[];
    float* smax = reinterpret_cast<float*>(sm);                  // [64]
    float* ssum = smax + 64;                                     // [64]
    char* sPh = sm + 512;                                        // [64 queries][PROW]
    char* sPl = sPh + 64 * PROW;
    char* sVh = sPl + 64 * PROW;                                 // [32 NT columns][PROW]: V transposed, k = key
    char* sVl = sVh + 32 * NT * PROW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wq = wave >> 1, wc = wave & 1, lr = lane & 31, kh = lane >> 5;
    const int i0 = blockIdx.x * 64, b = blockIdx.y / p.H, h = blockIdx.y % p.H;
    const int d = p.d, Tg = p.Tg;
    const size_t hb = (size_t)h * d, gd = (size_t)p.G * p.D;
    const float* srow = q.scores + ((size_t)blockIdx.y * Tg) * q.TgP;
    const float* vbase = p.v + (size_t)b * p.Tp * p.D + hb;
    // ---- row statistics: max, then e = exp(s - max) written back IN PLACE over the scores (this workgroup owns its 64 rows) and summed -
    //      the softmax of attentions.py:704 in the reference's order, with one exponential per score; rows past the last query: nothing
    //      The wave's 16 rows advance TOGETHER, 64 keys at a time: 16 independent loads in flight per step (row after row, every step was one
    //      dependent L2 round trip: ~10 per row, 130 us per workgroup - the whole kernel).  Per row the sums still run lane-wise over j = lane,
    //      lane + 64, ... and then through the same butterfly.
    {
        float* rowp[16];
        bool rok[16];
        float mx[16], sum[16];
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
            const int i = i0 + 16 * wave + rr;
            rok[rr] = i < Tg;
            rowp[rr] = q.scores + ((size_t)blockIdx.y * Tg + (rok[rr] ? i : Tg - 1)) * q.TgP;
            mx[rr] = -INFINITY; sum[rr] = 0.f;
        }
        for (int jb = 0; jb < Tg; jb += 64) {
            const int j = jb + lane;
            float v[16];
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) v[rr] = j < Tg ? rowp[rr][j] : -INFINITY;
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) mx[rr] = fmaxf(mx[rr], v[rr]);
        }
#pragma unroll
        for (int rr = 0; rr < 16; ++rr)
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) mx[rr] = fmaxf(mx[rr], __shfl_xor(mx[rr], o));
        for (int jb = 0; jb < Tg; jb += 64) {
            const int j = jb + lane;
            float v[16];
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) v[rr] = j < Tg ? rowp[rr][j] : 0.f;
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
                const float e = sx_expf(v[rr] - mx[rr]);
                if (j < Tg && rok[rr]) { rowp[rr][j] = e; sum[rr] += e; }
            }
        }
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) sum[rr] += __shfl_xor(sum[rr], o);
            if (lane == 0) ssum[16 * wave + rr] = rok[rr] ? sx_rcp(sum[rr]) : 0.f;
        }
    }
    __threadfence_block();
    f32x16 oh[(NT + 1) / 2], ox[(NT + 1) / 2];
#pragma unroll
    for (int t = 0; t < (NT + 1) / 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) { oh[t][r] = 0.f; ox[t][r] = 0.f; }
    float* att = p.att ? p.att + ((size_t)blockIdx.y * Tg) * Tg : nullptr;
    for (int j0 = 0; j0 < Tg; j0 += 64) {
        __syncthreads();                                         // the statistics are published / the previous tile's fragment reads are done
        // probabilities of the 64 x 64 tile, split (keys >= Tg: zero)
        for (int c = tid; c < 64 * 16; c += 256) {
            const int m = c >> 4, jx = (c & 15) * 4;
            const int i = i0 + m < Tg ? i0 + m : Tg - 1;
            const float isum = ssum[m];                          // 0 for rows past the last query (their clamped row holds another row's e)
            float4 sv = make_float4(0.f, 0.f, 0.f, 0.f);                                      // TgP is a multiple of 4: a quad is inside the row or past it
            if (j0 + jx < q.TgP) sv = *reinterpret_cast<const float4*>(srow + (size_t)i * q.TgP + j0 + jx);
            const float sc4[4] = {sv.x, sv.y, sv.z, sv.w};
            float pv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int j = j0 + jx + e;
                pv[e] = j < Tg ? sc4[e] * isum : 0.f;
                if (att && j < Tg && i0 + m < Tg) att[(size_t)i * Tg + j] = pv[e];
            }
            uint32_t h0, l0, h1, l1;
            split2(pv[0], pv[1], h0, l0); split2(pv[2], pv[3], h1, l1);
            *reinterpret_cast<uint2*>(sPh + m * PROW + jx * 2) = make_uint2(h0, h1);
            *reinterpret_cast<uint2*>(sPl + m * PROW + jx * 2) = make_uint2(l0, l1);
        }
        // V rows of the key tile, transposed: Vt[x][key]
        for (int c = tid; c < 64 * 8 * NT; c += 256) {
            // consecutive lanes <-> consecutive KEYS of one column quad: the transposing 2-byte stores of a wave then fall on 32 consecutive
            // dwords (two lanes per dword) instead of 2 banks (the first version, lanes <-> columns: 16-way conflicts, 8 ms per step)
            const int r = c & 63, x = (c >> 6) * 4;
            const int j = j0 + r < Tg ? j0 + r : Tg - 1;         // rows past the last key group: finite data times zero probabilities
            const float4 v = ld_span4(vbase + gd * j, x, d);
            const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 4; e += 2) {
                uint32_t hh, ll;
                split2(vv[e], vv[e + 1], hh, ll);
                *reinterpret_cast<uint16_t*>(sVh + (x + e) * PROW + r * 2) = (uint16_t)(hh & 0xFFFFu);
                *reinterpret_cast<uint16_t*>(sVh + (x + e + 1) * PROW + r * 2) = (uint16_t)(hh >> 16);
                *reinterpret_cast<uint16_t*>(sVl + (x + e) * PROW + r * 2) = (uint16_t)(ll & 0xFFFFu);
                *reinterpret_cast<uint16_t*>(sVl + (x + e + 1) * PROW + r * 2) = (uint16_t)(ll >> 16);
            }
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int ko = (16 * ks + 8 * kh) * 2;
            const f16x8 ph = *reinterpret_cast<const f16x8*>(sPh + (32 * wq + lr) * PROW + ko), pl = *reinterpret_cast<const f16x8*>(sPl + (32 * wq + lr) * PROW + ko);
#pragma unroll
            for (int t = 0; t < (NT + 1) / 2; ++t) {
                const int ct = wc + 2 * t;
                if (ct < NT) {
                    const f16x8 vh = *reinterpret_cast<const f16x8*>(sVh + (32 * ct + lr) * PROW + ko), vl = *reinterpret_cast<const f16x8*>(sVl + (32 * ct + lr) * PROW + ko);
                    oh[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph, vh, oh[t], 0, 0, 0);
                    ox[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph, vl, ox[t], 0, 0, 0);
                    ox[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pl, vh, ox[t], 0, 0, 0);
                }
            }
        }
    }
    // ---- un-group: grouped row i, head column x -> natural layout (b * Tp + G i) * D + h d + x   (attentions.py:707-712)
#pragma unroll
    for (int t = 0; t < (NT + 1) / 2; ++t) {
        const int x = 32 * (wc + 2 * t) + lr;
        if (wc + 2 * t >= NT || x >= d) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = i0 + 32 * wq + (r & 3) + 8 * (r >> 2) + 4 * kh;
            if (i >= Tg) continue;
            p.out[((size_t)b * p.Tp + (size_t)p.G * i) * p.D + hb + x] = oh[t][r] + ox[t][r] * LO_INV;
        }
    }
}

template <int KS>
int launch_scores(const SxAttnParams& q, hipStream_t s) {
    constexpr int ROW = 16 * KS * 2 + 16;
    const int lds = 6 * 64 * ROW + 64 * PE_LD * 4 + 2 * 16 * KS * 4;
    if (lds > 160 * 1024) return -2;
    static LdsAttr attr;
    ensure_dynamic_lds(reinterpret_cast<const void*>(&sx_scores_kernel<KS>), lds, attr);
    hipLaunchKernelGGL((sx_scores_kernel<KS>), dim3((q.a.Tg + 63) / 64, q.a.B * q.a.H), dim3(256), lds, s, q);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

template <int NT>
int launch_pv(const SxAttnParams& q, hipStream_t s) {
    const int lds = 512 + 2 * 64 * PROW + 2 * 32 * NT * PROW;
    static LdsAttr attr;
    ensure_dynamic_lds(reinterpret_cast<const void*>(&sx_pv_kernel<NT>), lds, attr);
    hipLaunchKernelGGL((sx_pv_kernel<NT>), dim3((q.a.Tg + 63) / 64, q.a.B * q.a.H), dim3(256), lds, s, q);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace

int launch_sx_gemm(const SxGemmParams& q, hipStream_t s) {
    const ExGemmParams& p = q.g;
    if (p.M <= 0 || p.N <= 0) return 0;
    if (p.K % 4 || p.lda % 4 || !q.Whi || !q.Wlo || q.ldh % SBK || q.ldh < p.K) return -2;
    {   // 32-bit element offsets inside the kernel
        const long long rows = p.a_rows ? (long long)((p.M + p.a_rows - 1) / p.a_rows) * p.a_pitch : p.M;
        if (rows * p.lda >= (1ll << 32) || (long long)p.N * q.ldh >= (1ll << 32)) return -2;
    }
    if (p.N % 4 || p.ldc % 4 || (p.epi == 2 && p.ldr % 4) || (p.split_cols > 0 && p.split_cols % 4)) return -2;      // 16-byte row pieces in the epilogue
    static LdsAttr attr;
    ensure_dynamic_lds(reinterpret_cast<const void*>(&sx_gemm_kernel), 2 * SX_STAGE, attr);
    if ((p.M + SBM - 1) / SBM > 65535) return -2;
    hipLaunchKernelGGL(sx_gemm_kernel, dim3((p.N + SBN - 1) / SBN, (p.M + SBM - 1) / SBM), dim3(256), 2 * SX_STAGE, s, q);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

bool sx_attention_supported(int d) { return d >= 1 && d <= 144; }      // the scores kernel's LDS image (K tile + two band halves + band product)
size_t sx_attention_scores_bytes(int B, int H, int Tg) { return (size_t)B * H * Tg * ((Tg + 3) / 4 * 4) * 4; }

// scores: B * H * Tg * round_up(Tg, 4) floats of scratch
int launch_sx_attention(const ExAttnParams& a, float* scores, hipStream_t s) {
    if (a.B <= 0 || a.Tg <= 0) return 0;
    if (!sx_attention_supported(a.d) || !scores || (long long)a.B * a.H > 65535) return -2;
    SxAttnParams q{};
    q.a = a; q.scores = scores; q.TgP = (a.Tg + 3) / 4 * 4;
    int rc;
    switch ((a.d + 15) / 16) {
        case 1: rc = launch_scores<1>(q, s); break;
        case 2: rc = launch_scores<2>(q, s); break;
        case 3: rc = launch_scores<3>(q, s); break;
        case 4: rc = launch_scores<4>(q, s); break;
        case 5: rc = launch_scores<5>(q, s); break;
        case 6: rc = launch_scores<6>(q, s); break;
        case 7: case 8: rc = launch_scores<8>(q, s); break;
        default: rc = launch_scores<9>(q, s); break;
    }
    if (rc) return rc;
    switch ((a.d + 31) / 32) {
        case 1: return launch_pv<1>(q, s);
        case 2: return launch_pv<2>(q, s);
        case 3: return launch_pv<3>(q, s);
        case 4: return launch_pv<4>(q, s);
        case 5: return launch_pv<5>(q, s);
        default: return launch_pv<6>(q, s);
    }
}
