// relpos_attention2_kernel: the second generation of the fused (grouped) relative-position attention (same math and the same
// buffers as attention.hip: S[b,h,i,j] = (Qu_i . K_j + Qv_i . E[Tg-1+j-i]) / sqrt(d), key mask from lens[b], online softmax, P V;
// reference models/attentions.py:549-718).
//
// Round 1's kernel (attention.hip) is instruction-issue bound: a wave owns 16 queries, so every K / E / V^T fragment it reads from
// LDS feeds ONE 16x16x32 MFMA, V is transposed on the way into LDS (8 v_perm + 8 ds_write_b32 per 16-byte chunk), and its phase
// profile shows the MFMA pipe 6 % busy (profiles/r1_14_phase_profiles.txt).  This kernel halves the instructions per query:
//   * a wave owns 32 queries (two 16-query column tiles): every K fragment feeds two MFMAs, the positional band of the pair is 96
//     rows instead of 2 x 80 (6 fragment rows for 10 MFMA row tiles), every V^T fragment feeds two MFMAs;
//   * V is staged row-major with plain 16-byte stores and its MFMA fragments come from ds_read_b64_tr_b16 (hardware 4x4
//     transpose): lane (c, g) supplies the address of keys 4g + (c >> 2), columns 4 (c & 3) .. +3 and receives column c of the
//     4-key x 16-column block of its 16-lane group; row pitch DP * 2 + 32 bytes keeps the 8 rows of a 32-lane half on disjoint banks;
//   * one wave per SIMD (two 2-wave workgroups per CU, or one 4-wave workgroup): the whole register file per wave, both tiles'
//     accumulators, scores and positional tiles stay in registers, two skew buffers per wave (one per tile) remove the
//     write-after-read hand-off.
// Operand order, skew realignment (write PE[r'][i], read at r' = j - i + 15), contraction-slot permutation of P and the online
// softmax are those of attention.hip, so the two kernels are interchangeable (option "attention_v2", tests compare both).
#include "kernels.h"

#include <cstdlib>
#include <type_traits>

namespace {

// Timing-only builds (tools/build_ablate.py, -DEFFCONF_ABLATE, a separate library): parts of the kernel switched off by a run-time mask.  The product kernel
// compiles the mask as the constant 0 - no branch around any load (they kept the prologue's loads in separate basic blocks)
#ifdef EFFCONF_ABLATE
#define ABLATE(p) ((p).ablate)
#else
#define ABLATE(p) 0
#endif
constexpr int BJ = 64;          // keys per block
constexpr int SKEW_LD = 68;     // floats per query row of a skew buffer: the 64 key columns of a block + 4 (16-byte rows on rotating banks).  Round 5: the skew is
                                // applied by the WRITER (band position -> key column), rows are 64 instead of 80 + 4 floats: 52 KB of LDS at head width 64 - three
                                // workgroups per CU fit (56 KB: two)
constexpr float RESCALE_T = 4.0f;   // deferred accumulator rescale: threshold on the growth of a row maximum, log2 units
constexpr float NEG_BIG = -1.0e30f;  // masked score / initial row maximum: finite, so that (max - max) never becomes inf - inf; whatever a row
                                     // accumulates while its maximum still is NEG_BIG is wiped by the first real score (alpha = exp2(-huge) = 0)

__device__ __forceinline__ uint4 ld16(const bf16_t* p) {       // 16-byte global load from a 2-byte aligned address (natural layout, odd d)
    typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(2)));
    const u32x4_a4 v = *reinterpret_cast<const u32x4_a4*>(p);
    return make_uint4(v[0], v[1], v[2], v[3]);
}

__device__ __forceinline__ uint4 ld16b(const char* p) { return ld16(reinterpret_cast<const bf16_t*>(p)); }

__device__ __forceinline__ void wave_sync() {                   // per-wave LDS hand-off (LDS operations of one wave execute in order)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint2 lds_tr16(const char* p) {      // ds_read_b64_tr_b16
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p));
    union { s16x4 s; uint2 u; } c;
    c.s = v;
    return c.u;
}

template <int DP, int NWV, int QT>
struct Attn2Smem {
    static constexpr int BI = NWV * 16 * QT;                     // QT = 16-query tiles per wave
    static constexpr int ERING = 2 * BI;                         // power of two >= BI + BJ - 1
    static constexpr bool SWZ = (DP == 64);
    static constexpr int KROW = SWZ ? DP * 2 : DP * 2 + 16;      // bytes per K / E row (layouts of attention.hip)
    static __device__ __forceinline__ int koff(int row, int chunk) { return row * KROW + ((SWZ ? (chunk ^ (2 * ((row >> 1) & 3))) : chunk) << 4); }
    static constexpr int VP = DP * 2 + 32;                       // bytes per V row (key-major)
    static constexpr int K_BYTES = BJ * KROW, V_BYTES = BJ * VP, E_BYTES = ERING * KROW;
    static constexpr int S_BYTES = NWV * QT * 16 * SKEW_LD * 4;
    static constexpr int TOTAL = K_BYTES + V_BYTES + E_BYTES + S_BYTES;
};

template <int DP, int NWV, int QT, int SETS = 0>      // SETS: staging register sets (0 = by shape, 1 = loads one key block ahead, 2 = two ahead)
__global__ __launch_bounds__(NWV * 64, (QT == 1 && NWV == 4 && DP <= 96) ? ((DP <= 64 && SETS == 1) ? 3 : 2) : 1) void relpos_attention2_kernel(const AttnParams p) {
    using SM = Attn2Smem<DP, NWV, QT>;
    constexpr int KS = DP / 32, DT = DP / 16, BI = SM::BI, NTHR = NWV * 64, CPR = DP / 8, ERING = SM::ERING;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sK = smem;
    char* sV = sK + SM::K_BYTES;
    char* sE = sV + SM::V_BYTES;
    float* sS = reinterpret_cast<float*>(sE + SM::E_BYTES);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, g = lane >> 4;
    int qtiles = (p.Tg + BI - 1) / BI;
    // utterance b on XCD b % 8: all heads / query tiles of an utterance share one L2, and a length-sorted batch spreads evenly over
    // the XCDs.  Any B: the utterances in the order (0, 8, 16, .. | 1, 9, .. | ..) form a list that is cut into the 8 contiguous,
    // equally long chunks of logical ids xcd_remap hands the XCDs (with B % 8 != 0 a chunk border falls inside an utterance: 7
    // utterances are shared by two L2s).  Round 2 applied this only for B % 8 == 0: B = 85 ran 16 % slower per utterance than 80.
    int id = xcd_remap(blockIdx.x, gridDim.x);
    // Ragged batch (p.rag_off != null, natural layout only): utterance b has its own frame count T = lens[b] (every frame valid), its
    // own grouped length Tg and number of query tiles; the workgroups of the launch are laid out utterance by utterance in the SAME
    // permuted order as below (p.rag_wg: prefix sums of heads x query tiles in that order), its rows start at row rag_off[b] of the
    // concatenated Q / K / V / output row space, and its positional rows are the LAST 2 Tg - 1 rows of the table built for the longest
    // utterance (R[m] = sinusoid(Tp - 1 - G/2 - m): a shorter sequence's table is the longer one's, shifted by Tg_max - Tg grouped rows).
    int Tg = p.Tg, T = p.T, qt_wg, h, b;
    {
        const int per_b = p.H * qtiles, q8 = p.B >> 3, r8 = p.B & 7;
        const int u = p.rag_off ? ((ABLATE(p) & 8) ? (id * p.B) / (int)gridDim.x : ragged_find_wave(p.rag_wg, p.B, id)) : id / per_b;
        int x, j;
        if (u < r8 * (q8 + 1)) { x = u / (q8 + 1); j = u - x * (q8 + 1); }
        else { const int u2 = u - r8 * (q8 + 1); x = u2 / q8; j = u2 - x * q8; x += r8; }
        b = x + 8 * j;
        int local = id - (p.rag_off ? ((ABLATE(p) & 8) ? id : p.rag_wg[u]) : u * per_b);
        if (p.rag_off) { T = p.lens[b]; Tg = (T + p.G - 1) / p.G; qtiles = (Tg + BI - 1) / BI; }
        qt_wg = local % qtiles; h = local / qtiles;
    }
    if (ABLATE(p) & 128) return;                                  // timing-only: dispatch + utterance lookup alone
    const int i0 = qt_wg * BI, iw0 = i0 + wave * 16 * QT;
    const size_t orow0 = p.rag_off ? (size_t)p.rag_off[b] : (size_t)b * p.T;          // first row of the utterance in the un-grouped output
    const size_t qoff = (p.rag_off ? orow0 * p.D : (size_t)b * p.q_bstride) + (size_t)h * p.q_hstride;
    const bf16_t* Qu = p.qu + qoff;
    const bf16_t* Kh = p.kh + qoff;
    const bf16_t* Vh = p.vt + qoff;
    const int RS = p.q_rowstride, ERS = p.e_rowstride;
    const bf16_t* Eh = p.eh + (size_t)h * p.e_hstride + (p.rag_off ? (size_t)(p.rag_tgmax - Tg) * ERS : 0);
    const int erows = p.causal ? Tg : 2 * Tg - 1;                  // causal tables hold the Tg rows of the non-negative distances only
    const int dceil = (p.d + 7) & ~7;
    const bool ragged_d = dceil != p.d;

    // Visible keys of query i (reference attentions.py:1377-1403, sliced ::s and ::G on the way here): j < nkv (key-padding mask) and
    // -band_l <= j - i <= band_r (streaming contexts in grouped positions of this stage; INT_MAX / 2 = unlimited).  A query whose band lies
    // entirely in the padding (i - band_l >= nkv; every query of an empty utterance) sees -1e9 on EVERY key in the reference: a uniform
    // softmax over all Tg key groups ("dead" rows: scores forced to 0).  A workgroup visits the key blocks its 64 queries can see - all
    // of them when it holds a dead row.
    int nkv = (p.lens[b] + p.G - 1) / p.G;
    nkv = nkv < Tg ? nkv : Tg;
    const int band_l = p.band_l < Tg ? p.band_l : Tg, band_r = p.band_r < Tg ? p.band_r : Tg;
    const bool banded = band_l < Tg || band_r < Tg;
    const bool tile_dead = nkv < 1 || i0 + BI - 1 - band_l >= nkv;    // empty utterance: every row
    int kbeg = 0, nkeys = nkv;
    if (tile_dead) nkeys = Tg;
    else if (banded) {
        kbeg = i0 - band_l; kbeg = kbeg < 0 ? 0 : kbeg & ~(BJ - 1);
        const int ke = i0 + BI + band_r;
        nkeys = ke < nkv ? ke : nkv;
    }

    // ---- first positional band of the workgroup: absolute E rows R0 .. R0 + BI + 62.  Round 5: the first key block's K / V loads and the band's loads are
    //      ISSUED before the query rows are waited for and converted (one memory round trip at the head of the workgroup instead of three)
    const int R0 = Tg - 1 - i0 - (BI - 1) + kbeg;                 // absolute E row of band row 0 of the FIRST visited key block
    constexpr int NB = ((BI + 63) * CPR + NTHR - 1) / NTHR;
    uint4 fb[NB];
    // ---- K / V / new-E staging registers, two sets (loads run two key blocks ahead)
    constexpr int NK = (BJ * CPR + NTHR - 1) / NTHR;
    struct Stage { uint4 lk[NK], lv[NK], le[NK]; };
    Stage sa, sb;
#pragma unroll
    for (int n = 0; n < NK; ++n) { sa.lk[n] = make_uint4(0, 0, 0, 0); sa.lv[n] = sa.lk[n]; sa.le[n] = sa.lk[n]; sb.lk[n] = sa.lk[n]; sb.lv[n] = sa.lk[n]; sb.le[n] = sa.lk[n]; }

    // per-thread element offsets of its K / V chunks inside a key block and of its E chunks inside a 64-row batch: computed once;
    // per block only the block's row offset is added.  Rows past the last key group / outside the table are clamped to valid rows
    // (wave-uniform tail path): their scores are masked, their probabilities are zero, so any FINITE value serves - no zero fill.
    // Column masks (elements >= d of a chunk: the next head's data, finite, times the queries' zero pad columns) are only applied on
    // the tail paths, i.e. to the loads that can run past the END of a buffer (last rows of the last head), where the bytes are not
    // the library's own and may be NaN patterns.
    uint32_t koffs[NK];                                          // K, V and E share the row stride (launch check), so one offset serves all three
    auto chunk_row = [&](int n) { int q = tid + NTHR * n; q = q < BJ * CPR ? q : BJ * CPR - 1; return q / CPR; };
    auto tail_mask = [&](uint4 a, int n) {                      // tail paths only: recomputed there instead of living in registers
        int q = tid + NTHR * n; q = q < BJ * CPR ? q : BJ * CPR - 1;
        return mask_chunk(a, p.d - (q - (q / CPR) * CPR) * 8);
    };
#pragma unroll
    for (int n = 0; n < NK; ++n) {
        int q = tid + NTHR * n;
        q = q < BJ * CPR ? q : BJ * CPR - 1;
        const int r = q / CPR, x = (q - r * CPR) * 8, xc = x < dceil ? x : 0;
        koffs[n] = (uint32_t)(r * RS + xc) * 2u;                // BYTE offsets: wave-uniform base + 32-bit lane offset
    }
    auto issue_loads = [&](Stage& st_, int jn) __attribute__((always_inline)) {
        if (((ABLATE(p) & 2) && jn > kbeg) || (ABLATE(p) & 64)) return;
        // unmasked fast path: every 16-byte chunk of the block stays inside the library's own (finite) data.  With a head width that
        // is not a multiple of 8 (d = 90 / 135 / 42) the chunk that closes a head span reads dceil - d elements of the NEXT span; behind
        // the last key row of the last head of the last utterance that is the never-written slack of the buffer (NaN patterns times the
        // queries' zero pad columns = NaN), so the block holding row Tg - 1 takes the masked tail path then.
        if (jn + BJ <= Tg - (ragged_d ? 1 : 0)) {
            const char* kb = reinterpret_cast<const char*>(Kh + (size_t)jn * RS);
            const char* vb = reinterpret_cast<const char*>(Vh + (size_t)jn * RS);
#pragma unroll
            for (int n = 0; n < NK; ++n) { st_.lk[n] = ld16b(kb + koffs[n]); st_.lv[n] = ld16b(vb + koffs[n]); }
        } else {
#pragma unroll
            for (int n = 0; n < NK; ++n) {
                const int kr = chunk_row(n), j = jn + kr;
                const size_t o = (size_t)(j < Tg ? j : Tg - 1) * RS * 2 + (koffs[n] - (uint32_t)(kr * RS) * 2u);
                st_.lk[n] = tail_mask(ld16b(reinterpret_cast<const char*>(Kh) + o), n); st_.lv[n] = tail_mask(ld16b(reinterpret_cast<const char*>(Vh) + o), n);
            }
        }
        if (jn > kbeg) {
            const int rnew = R0 + (jn - kbeg) + BI - 1;          // first new absolute E row of the block
            if (rnew >= 0 && rnew + 63 < erows - (ragged_d ? 1 : 0)) {      // the batch holding the table's last row: masked path (see above)
                const char* eb = reinterpret_cast<const char*>(Eh + (size_t)rnew * ERS);
#pragma unroll
                for (int n = 0; n < NK; ++n) st_.le[n] = ld16b(eb + koffs[n]);
            } else {
#pragma unroll
                for (int n = 0; n < NK; ++n) {
                    const int er = chunk_row(n);
                    int r = rnew + er;
                    r = r < 0 ? 0 : (r >= erows ? erows - 1 : r);
                    st_.le[n] = tail_mask(ld16b(reinterpret_cast<const char*>(Eh) + (size_t)r * ERS * 2 + (koffs[n] - (uint32_t)(er * RS) * 2u)), n);
                }
            }
        }
    };
    issue_loads(sa, kbeg);
#pragma unroll
    for (int n = 0; n < NB; ++n) {
        const int q = tid + NTHR * n;
        const int rr = q / CPR, x = (q - rr * CPR) * 8;
        const int r = R0 + (rr < BI + 63 ? rr : BI + 62);
        const int rc = r < 0 ? 0 : (r >= erows ? erows - 1 : r);
        fb[n] = (ABLATE(p) & 16) ? make_uint4(0, 0, 0, 0) : ld16(Eh + (size_t)rc * ERS + (x < dceil ? x : 0));
    }
    // ---- the wave's two query tiles: B operands of S^T = K Q^T (Q + u) and of the positional product (Q + v = (Q + u) + (v - u))
    bf16x8 qu[QT][KS], qv[QT][KS];
    {
        const float* dv = p.dvu + (size_t)h * p.dvu_ld;
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            const int i = iw0 + 16 * t + c, ic = i < Tg ? i : Tg - 1;
            uint4 ra[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int x = ks * 32 + g * 8;
                ra[ks] = (ABLATE(p) & 32) ? make_uint4(0, 0, 0, 0) : ld16(Qu + (size_t)ic * RS + (x < dceil ? x : 0));
            }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int x = ks * 32 + g * 8;
                const float4 da = *reinterpret_cast<const float4*>(dv + x), db = *reinterpret_cast<const float4*>(dv + x + 4);
                const int valid = i < Tg ? p.d - x : 0;
                const uint4 m = mask_chunk(ra[ks], valid);
                qu[t][ks] = as_bf16x8(m);
                const uint4 w = make_uint4(pack_bf2(__uint_as_float(m.x << 16) + da.x, __uint_as_float(m.x & 0xFFFF0000u) + da.y),
                                           pack_bf2(__uint_as_float(m.y << 16) + da.z, __uint_as_float(m.y & 0xFFFF0000u) + da.w),
                                           pack_bf2(__uint_as_float(m.z << 16) + db.x, __uint_as_float(m.z & 0xFFFF0000u) + db.y),
                                           pack_bf2(__uint_as_float(m.w << 16) + db.z, __uint_as_float(m.w & 0xFFFF0000u) + db.w));
                qv[t][ks] = as_bf16x8(mask_chunk(w, valid));
            }
        }
    }
#pragma unroll
    for (int n = 0; n < NB; ++n) {
        const int q = tid + NTHR * n;
        const int rr = q / CPR, x = (q - rr * CPR) * 8;
        if (q < (BI + 63) * CPR)      // ring row = band row + key offset (mod 2 BI): block 0's band sits at rows 0 .. BI + 62
            *reinterpret_cast<uint4*>(sE + SM::koff(rr, x >> 3)) = mask_chunk(fb[n], p.d - x);
    }

    f32x4 acc[QT][DT];
    float m_run[QT], l_run[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        m_run[t] = NEG_BIG; l_run[t] = 0.f;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) acc[t][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float scale2 = p.scale * 1.44269504088896340736f;
    float* skew0 = sS + (wave * QT) * 16 * SKEW_LD + c * SKEW_LD;          // tile t: skew0 + t * 16 * SKEW_LD
    const int wcol = c - 15 + 4 * g;                             // key column of the lane's band element r = 0 of band tile 0
    const int woff1 = BI - 16 * QT - 16 * QT * wave;             // first band row (workgroup band) of the wave's (64 + 16 QT)-row band: the LAST tile starts here, tile t 16 (QT - 1 - t) rows later
    const char* vbase = sV + (4 * g + (c >> 2)) * SM::VP + (c & 3) * 8;

    // LDS byte offsets of the thread's chunks: K / V tile rows, and the ring rows of a new-row batch for even / odd key blocks
    // (ring row = band row + key offset mod 2 BI = 128: the batch starts at row BI - 1 + 64 * parity)
    static_assert(BI == 64, "ring phases: two (BI == BJ)");
    uint32_t ldk[NK], ldv[NK], lde[2][NK];
#pragma unroll
    for (int n = 0; n < NK; ++n) {
        int q = tid + NTHR * n;
        q = q < BJ * CPR ? q : BJ * CPR - 1;
        const int r = q / CPR, ch = q - r * CPR;
        ldk[n] = (uint32_t)SM::koff(r, ch);
        ldv[n] = (uint32_t)(r * SM::VP + ch * 16);
        lde[0][n] = (uint32_t)SM::koff((BI - 1 + r) & (ERING - 1), ch);
        lde[1][n] = (uint32_t)SM::koff((BI - 1 + r + 64) & (ERING - 1), ch);
    }
    constexpr bool FULL = (BJ * CPR) % NTHR == 0;                // every thread owns NK real chunks
    auto publish = [&](const Stage& st_, int j0, int par) __attribute__((always_inline)) {
        __syncthreads();                                         // the previous block's LDS reads are done
#pragma unroll
        for (int n = 0; n < NK; ++n) {
            if (FULL || tid + NTHR * n < BJ * CPR) {
                *reinterpret_cast<uint4*>(sK + ldk[n]) = st_.lk[n];
                *reinterpret_cast<uint4*>(sV + ldv[n]) = st_.lv[n];
                if (j0 > kbeg) *reinterpret_cast<uint4*>(sE + lde[par][n]) = st_.le[n];
            }
        }
        __syncthreads();
    };

    // JTV = 16-key tiles of the block that are computed: 4, or 2 for an utterance's LAST block when at most 32 of its keys exist (the other tiles' scores
    // are masked, their probabilities exactly zero: S, positional band, softmax and P V of 32 keys instead of 64 - an utterance of 142 frames has 14
    // keys in its third block).  Same operations on the computed tiles, in the same order: bit-identical to the full block
    auto compute_block = [&](int j0, int par, auto jtv) __attribute__((always_inline)) {
        constexpr int JTV = decltype(jtv)::value;
        // ---- S^T tiles of both query tiles: every K fragment feeds two MFMAs
        f32x4 st[QT][JTV];
#pragma unroll
        for (int jt = 0; jt < JTV; ++jt) {
#pragma unroll
            for (int t = 0; t < QT; ++t) st[t][jt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(sK + SM::koff(jt * 16 + c, ks * 4 + g));
#pragma unroll
                for (int t = 0; t < QT; ++t) st[t][jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, qu[t][ks], st[t][jt], 0, 0, 0);
            }
        }
        // ---- positional band of the pair: 96 rows = 6 fragment rows; tile 1 uses rows 0 .. 79, tile 0 rows 16 .. 95
        const int rw1 = woff1 + 64 * par;                       // ring row of the wave's band row 0 (ring row = band row + key offset mod 128)
        constexpr int PU = JTV + 1;                             // band tiles per query tile: 16 queries + 16 JTV keys - 1 rows
        f32x4 pe[QT][PU];
#pragma unroll
        for (int t = 0; t < QT; ++t)
#pragma unroll
            for (int u = 0; u < PU; ++u) pe[t][u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < JTV + QT; ++u) {
            const int er = ((rw1 + u * 16) & (ERING - 1)) + c;      // (multiples of 16) + c: no carry into the wrap
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(sE + SM::koff(er, ks * 4 + g));
#pragma unroll
                for (int t = 0; t < QT; ++t) {                 // tile t covers fragment rows QT - 1 - t .. QT + 3 - t
                    const int ut = u - (QT - 1 - t);
                    if (ut >= 0 && ut < PU) pe[t][ut] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, qv[t][ks], pe[t][ut], 0, 0, 0);
                }
            }
        }
        // realignment on the way INTO the buffer: band position q of query c is key column q - 15 + c.  The first and the last band tile hold the columns
        // outside [0, 16 JTV) in complementary elements (q - 15 + c < 0 in tile 0 exactly where it is >= 16 JTV in tile JTV): ONE store serves both
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            float* row = skew0 + t * 16 * SKEW_LD + wcol;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool lo = wcol + r >= 0;
                row[r + (lo ? 0 : 16 * JTV)] = lo ? pe[t][0][r] : pe[t][JTV][r];
            }
#pragma unroll
            for (int u = 1; u < JTV; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) row[u * 16 + r] = pe[t][u][r];
        }
        wave_sync();

        // ---- realign, mask, online softmax, per tile.  Raw scores s = S + PE; the running maximum is tracked in raw units and
        //      p = exp2(s * scale2 - m * scale2) is one fma + one v_exp_f32 per score (scale2 = scale * log2 e > 0).  The accumulators
        //      are rescaled only when some row's maximum grew by more than RESCALE_T (log2 units) since the last rescale (always in the
        //      first block): between rescales p <= 2^RESCALE_T, harmless for bf16 P and fp32 sums; the choice is wave-uniform.
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            const float* skew = skew0 + t * 16 * SKEW_LD;
            float mloc = -INFINITY;
#pragma unroll
            for (int jt = 0; jt < JTV; ++jt) {
                const f32x4 s4 = *reinterpret_cast<const f32x4*>(skew + jt * 16 + g * 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) st[t][jt][r] += s4[r];
            }
            // masks only where the (query tile, key block) pair is not fully visible (wave-uniform test): padding tail or band edge
            const int iq0 = iw0 + 16 * t;
            if (j0 + BJ > nkv || (banded && (j0 - (iq0 + 15) < -band_l || j0 + BJ - 1 - iq0 > band_r))) {
                const int iq = iq0 + c;
#pragma unroll
                for (int jt = 0; jt < JTV; ++jt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int j = j0 + jt * 16 + g * 4 + r, dj = j - iq;
                        st[t][jt][r] = (j < nkv && dj <= band_r && dj >= -band_l) ? st[t][jt][r] : NEG_BIG;
                    }
            }
            if (tile_dead) {                                   // rows that see no key at all: every score equal -> uniform softmax over all key groups
                const bool dead = nkv < 1 || iq0 + c - band_l >= nkv;
#pragma unroll
                for (int jt = 0; jt < JTV; ++jt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) st[t][jt][r] = dead ? (j0 + jt * 16 + g * 4 + r < Tg ? 0.f : NEG_BIG) : st[t][jt][r];
            }
#pragma unroll
            for (int jt = 0; jt < JTV; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r) mloc = fmaxf(mloc, st[t][jt][r]);
            mloc = fmaxf(mloc, __shfl_xor(mloc, 16));
            mloc = fmaxf(mloc, __shfl_xor(mloc, 32));             // finite (masked scores are NEG_BIG, never -inf)
            if (!__all((mloc - m_run[t]) * scale2 <= RESCALE_T)) {   // m_run = NEG_BIG until the row's first visible key: taken there
                const float m_new = fmaxf(m_run[t], mloc);
                const float alpha = __builtin_amdgcn_exp2f((m_run[t] - m_new) * scale2);
                l_run[t] *= alpha;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) { acc[t][dt][0] *= alpha; acc[t][dt][1] *= alpha; acc[t][dt][2] *= alpha; acc[t][dt][3] *= alpha; }
                m_run[t] = m_new;
            }
            const float nm = -m_run[t] * scale2;
            float lsum = 0.f;
#pragma unroll
            for (int jt = 0; jt < JTV; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = __builtin_amdgcn_exp2f(fmaf(st[t][jt][r], scale2, nm));
                    st[t][jt][r] = e;
                    lsum += e;
                }
            l_run[t] += lsum;
        }
        // ---- O^T += V^T P^T for both tiles; contraction slot (g, e) <-> key (2*c2 + (e>>2))*16 + g*4 + (e&3) on both operands;
        //      the V^T fragment is two transposing reads of the key-major V tile
#pragma unroll
        for (int c2 = 0; c2 < JTV / 2; ++c2) {
            bf16x8 pf[QT];
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                uint4 pb;
                pb.x = pack_bf2(st[t][2 * c2][0], st[t][2 * c2][1]);
                pb.y = pack_bf2(st[t][2 * c2][2], st[t][2 * c2][3]);
                pb.z = pack_bf2(st[t][2 * c2 + 1][0], st[t][2 * c2 + 1][1]);
                pb.w = pack_bf2(st[t][2 * c2 + 1][2], st[t][2 * c2 + 1][3]);
                pf[t] = as_bf16x8(pb);
            }
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const uint2 lo = lds_tr16(vbase + (2 * c2) * 16 * SM::VP + dt * 32);
                const uint2 hi = lds_tr16(vbase + (2 * c2 + 1) * 16 * SM::VP + dt * 32);
                const bf16x8 a = as_bf16x8(make_uint4(lo.x, lo.y, hi.x, hi.y));
#pragma unroll
                for (int t = 0; t < QT; ++t) acc[t][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, pf[t], acc[t][dt], 0, 0, 0);
            }
        }
    };

    // waves whose 16 queries all lie behind the utterance's last grouped row (the last query tile of an utterance of 142 rows: three of its four waves)
    // stage and publish with the workgroup but compute nothing: their SIMD's issue slots go to the other workgroup of the CU
    const bool wave_idle = iw0 >= Tg;
    const bool short_ok = !tile_dead;                           // dead rows spread a uniform softmax over ALL key groups: full blocks
    auto run_block = [&](int j0, int par) __attribute__((always_inline)) {
        if (wave_idle || (ABLATE(p) & 1)) return;
        if (short_ok && nkeys - j0 <= 32) compute_block(j0, par, std::integral_constant<int, 2>{});
        else compute_block(j0, par, std::integral_constant<int, 4>{});
    };
    // two staging sets (loads two key blocks ahead) while they fit the register file; one set (one block ahead) for the widest heads
    // of the 2-wave workgroup, whose threads stage twice as many chunks
    constexpr bool TWO_SETS = SETS == 2 || (SETS == 0 && !(NWV == 2 && DP >= 96));
    if constexpr (TWO_SETS) {
        if (kbeg + BJ < nkeys) issue_loads(sb, kbeg + BJ);
        for (int jb = kbeg; jb < nkeys; jb += 2 * BJ)
#pragma unroll
        for (int half2 = 0; half2 < 2; ++half2) {
            const int j0 = jb + half2 * BJ;
            if (j0 >= nkeys) break;
            if (half2 == 0) publish(sa, j0, 0); else publish(sb, j0, 1);
            if (j0 + 2 * BJ < nkeys) { if (half2 == 0) issue_loads(sa, j0 + 2 * BJ); else issue_loads(sb, j0 + 2 * BJ); }
            run_block(j0, half2);
        }
    } else {
        for (int jb = kbeg; jb < nkeys; jb += 2 * BJ)
#pragma unroll
        for (int half2 = 0; half2 < 2; ++half2) {
            const int j0 = jb + half2 * BJ;
            if (j0 >= nkeys) break;
            publish(sa, j0, half2);
            if (j0 + BJ < nkeys) issue_loads(sa, j0 + BJ);
            run_block(j0, half2);
        }
    }

    // ---- normalise and scatter back to the un-grouped (B*T, D) layout (attention.hip)
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        float l_tot = l_run[t] + __shfl_xor(l_run[t], 16);
        l_tot += __shfl_xor(l_tot, 32);
        const float inv = 1.0f / l_tot;
        const int i = iw0 + 16 * t + c;
        if (i >= Tg || (ABLATE(p) & 4)) continue;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const int x0 = dt * 16 + g * 4;
            if (x0 >= p.d) continue;
            int n0 = h * p.d + x0, toff = 0;
            while (n0 >= p.D) { n0 -= p.D; ++toff; }
            const int t0 = i * p.G + toff;
            if (x0 + 3 < p.d && n0 + 3 < p.D && (n0 & 1) == 0) {
                if (t0 < T || p.rag_off) {          // ragged rows: the group-padding rows T .. Tp - 1 exist in the output and are kept zero
                    typedef uint32_t u32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));
                    u32x2_a4 w;
                    w[0] = t0 < T ? pack_bf2(acc[t][dt][0] * inv, acc[t][dt][1] * inv) : 0u;
                    w[1] = t0 < T ? pack_bf2(acc[t][dt][2] * inv, acc[t][dt][3] * inv) : 0u;
                    *reinterpret_cast<u32x2_a4*>(p.out + (orow0 + t0) * p.ldo + n0) = w;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int x = x0 + r;
                    if (x >= p.d) continue;
                    int n = h * p.d + x, tf = 0;
                    while (n >= p.D) { n -= p.D; ++tf; }
                    const int tt = i * p.G + tf;
                    if (tt < T || p.rag_off) p.out[(orow0 + tt) * p.ldo + n] = tt < T ? f2bf(acc[t][dt][r] * inv) : (bf16_t)0;
                }
            }
        }
    }
}

template <int DP, int NWV, int QT, int SETS = 0>
int launch2(const AttnParams& p, hipStream_t s) {
    using SM = Attn2Smem<DP, NWV, QT>;
    static_assert(SM::TOTAL <= 160 * 1024, "LDS");
    static LdsAttr attr;
    ensure_dynamic_lds(reinterpret_cast<const void*>(&relpos_attention2_kernel<DP, NWV, QT, SETS>), SM::TOTAL, attr);
    const int qtiles = (p.Tg + SM::BI - 1) / SM::BI;
    const int nwg = p.rag_off ? p.rag_nwg : p.B * p.H * qtiles;       // ragged: sum over the utterances of heads x query tiles (host total)
    if (nwg <= 0) return 0;
    hipLaunchKernelGGL((relpos_attention2_kernel<DP, NWV, QT, SETS>), dim3(nwg), dim3(NWV * 64), SM::TOTAL, s, p);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace

bool relpos_attention2_supported(int dpad) { return dpad == 32 || dpad == 64 || dpad == 96 || dpad == 128 || dpad == 160; }   // 192: attention.hip

// variant: 1 = 16 queries per wave, 4-wave 64-query workgroups, two per CU (attention.hip's shape: the default);
//          2 = 32 queries per wave, 2-wave 64-query workgroups (one wave per SIMD: measured slower, kept for experiments)
int launch_relpos_attention2(const AttnParams& p0, int waves, hipStream_t s) {
    if (p0.B <= 0 || p0.Tg <= 0) return 0;
    AttnParams p = p0;
#ifdef EFFCONF_ABLATE      // the timing-only library (tools/build_ablate.py); the product reads no environment variable here (advisor, round 5)
    static const int abl = getenv("EFFCONF_ATTN_ABLATE") ? atoi(getenv("EFFCONF_ATTN_ABLATE")) : 0;
    p.ablate = abl;
#endif
    if (p.dpad < p.d || p.q_rowstride != p.e_rowstride) return -2;
    if (p.rag_off && (waves != 1 || !p.rag_wg || p.q_rowstride != p.G * p.D)) return -2;      // ragged: natural layout, 64-query workgroups
#define ATT2_CASE(DPV) case DPV: return waves == 1 ? launch2<DPV, 4, 1>(p, s) : launch2<DPV, 2, 2>(p, s);
    // head width 96 (EfficientConformer Small stage 1): with two staging sets the kernel needs 9 registers more than two waves per SIMD allow
    // and reloads loop-invariant addresses from scratch inside the key loop; ONE set (loads one key block ahead) fits: attention class
    // 1.625 -> 1.555 ms per step (option attn_waves = 2 restores the two sets for comparison)
    if (waves == 1 && p.dpad == 96 && p.force_waves != 2) return launch2<96, 4, 1, 1>(p, s);
    // head width 64 (stages 2 / 3 of EfficientConformer Small): since the skew rows hold 64 floats the workgroup needs 52 KB of LDS, and with ONE staging set
    // 168 registers - THREE workgroups per CU instead of two (the kernel is bound by per-workgroup latency): step 5.235 -> 5.14 ms (profiles/r5_27_*)
    if (waves == 1 && p.dpad == 64 && p.force_waves != 2) return launch2<64, 4, 1, 1>(p, s);
    switch (p.dpad) {
        ATT2_CASE(32) ATT2_CASE(64) ATT2_CASE(96) ATT2_CASE(128)
        case 160: return launch2<160, 4, 1>(p, s);       // d = 135 (Medium / Large stage 1): one wave per SIMD either way; the 32-query variant spills
    }
#undef ATT2_CASE
    return -3;
}
