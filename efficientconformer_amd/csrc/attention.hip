// Fused (grouped) relative-position multi-head self-attention for gfx950.
//
// Reference: RelPosMultiHeadSelfAttention.forward (models/attentions.py:549-620) and
// GroupedRelPosMultiHeadSelfAttention.forward (attentions.py:645-718), closed form:
//
//     S[b,h,i,j] = (Qu[b,h,i] . K[b,h,j] + Qv[b,h,i] . E[h, Tg-1+j-i]) / sqrt(d)
//     key group j masked iff G*j >= lens[b];  P = softmax_j(S);  O = P V
//
// The reference materialises Q E^T as (T x 2T-1), then realigns it with a pad/reshape/slice
// ("rel_to_abs", attentions.py:483-547) and builds a (B,1,T,T) float mask every forward
// (attentions.py:1377-1403).  Here one workgroup owns a 64-query tile of one (b, h):
//   * keys are streamed in blocks of 64 with an online (flash-style) softmax, fp32 statistics;
//   * the positional term is computed only on the band of 64+64-1 relative rows a tile touches and
//     realigned through a per-wave LDS skew buffer (write PE[r'][i], read at r' = j - i + 15);
//   * MFMA operands are swapped (S^T = K Q^T, O^T = V^T P^T) so that every lane owns ONE query
//     column: running max / sum / rescale are lane-local and the exponentiated scores feed the
//     second MFMA straight from registers (the 32-key contraction order is permuted consistently
//     on both operands instead of shuffling P);
//   * the key-padding mask is evaluated from lens[b]; fully masked key blocks are skipped;
//   * group-reshape / head split are index arithmetic (inputs are head-major, output is (B*T, D)).
// All MFMA are v_mfma_f32_16x16x32_bf16, fp32 accumulation.
#include "kernels.h"
#include <cstdio>
#include <cstdlib>

namespace {

constexpr int BI = 64;          // queries per workgroup (16 per wave)
constexpr int BJ = 64;          // keys per block
constexpr int SKEW_LD = 84;     // floats per query row of the skew buffer (>= 80, multiple of 4)

// 16-byte global load from a (possibly only) 2-byte aligned address (natural layout: head h starts at element h*d, d may be odd);
// still one global_load_dwordx4 - global memory accesses are alignment-free on gfx950
__device__ __forceinline__ uint4 ld16(const bf16_t* p) {
    typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(2)));
    const u32x4_a4 v = *reinterpret_cast<const u32x4_a4*>(p);
    return make_uint4(v[0], v[1], v[2], v[3]);
}

// per-wave LDS hand-off: LDS operations of one wave execute in order, only the compiler must not reorder across it
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}


template <int DP, int NWV>
struct AttnSmem {
    static constexpr int ERING = NWV * 32;           // rows of the rolling relative-position ring (power of two >= BI + BJ - 1)
    // DP == 64: unpadded 128-byte rows with the 16-byte chunk index XOR-ed by 2*((row>>1)&3).  gfx950 services ds_read_b128 in the
    // lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}, ...: under them every padded-pitch layout has 2-way conflicts on the
    // K / E fragment reads (brute-forced; 43 % of the LDS-active cycles in profiles/r1_10_sq_counters.txt), this one has none for
    // any ring offset.  Other head widths keep the padded rows.
    static constexpr bool SWZ = (DP == 64);
    static constexpr int KROW = SWZ ? DP * 2 : DP * 2 + 16;         // bytes per K / E row
    static __device__ __forceinline__ int koff(int row, int chunk) { return row * KROW + ((SWZ ? (chunk ^ (2 * ((row >> 1) & 3))) : chunk) << 4); }
    static constexpr int VROW = BJ * 2 + 16;         // bytes per V^T row
    static constexpr int K_BYTES = BJ * KROW;
    static constexpr int V_BYTES = DP * VROW;
    static constexpr int E_BYTES = ERING * KROW;
    static constexpr int S_BYTES = NWV * 16 * SKEW_LD * 4;
    static constexpr int TOTAL = K_BYTES + V_BYTES + E_BYTES + S_BYTES;
};

// NWV waves = NWV*16 queries per workgroup.  K and V blocks are staged once per 16*NWV queries; the positional band
// (BI + 63 rows per key block, shifting by 64 rows per block) lives in a ring indexed by the absolute E row, so only the
// 64 NEW rows are staged per key block.  Staging is software pipelined through registers (loads of block j+1 are in
// flight during block j's MFMA / softmax work) and all loads are unconditional at clamped addresses.
template <int DP, int NWV, bool PROF = false>
// second launch bound: two 4-wave workgroups per CU share each SIMD's 512 registers, so VGPRs + AGPRs must stay <= 256 - without it
// a 272-register DP = 96 build silently ran one workgroup per CU (+46 %).  Head widths above 96 only fit one workgroup per CU in LDS
// anyway and keep the full register file (bounded to 256 they spill ~200 registers)
__global__ __launch_bounds__(NWV * 64, (NWV == 4 && DP <= 96) ? 2 : 1) void relpos_attention_kernel(const AttnParams p, unsigned long long* prof = nullptr) {
    unsigned long long ph[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, t0 = 0;
    if constexpr (PROF) t0 = __builtin_readcyclecounter();
#define AT_TICK(i) do { if constexpr (PROF) { asm volatile("" ::: "memory"); const unsigned long long t1_ = __builtin_readcyclecounter(); ph[i] += t1_ - t0; t0 = t1_; } } while (0)
    using SM = AttnSmem<DP, NWV>;
    constexpr int KS = DP / 32;     // k-steps over the head dim
    constexpr int DT = DP / 16;     // 16-wide output column tiles
    constexpr int BI = NWV * 16;
    constexpr int NTHR = NWV * 64;
    constexpr int CPR = DP / 8;     // 16-byte chunks per row
    constexpr int ERING = SM::ERING;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sK = smem;
    char* sV = sK + SM::K_BYTES;
    char* sE = sV + SM::V_BYTES;
    float* sS = reinterpret_cast<float*>(sE + SM::E_BYTES);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int qtiles = (p.Tg + BI - 1) / BI;
    // XCD-aware order: hardware places block i on XCD i % 8.  All heads and query tiles of one utterance re-read the same K / V
    // rows, so they should share an L2: utterance b goes to XCD b % 8 (utterances are sorted by length, so dealing them round-robin
    // also keeps the XCDs balanced; a contiguous range per XCD was 16 % slower).  Un-remapped, K / V / E were fetched from HBM
    // ~3x (profiles/r1_10_pmc_hbm_traffic.txt).  Any B: the utterances in the order (0, 8, 16, .. | 1, 9, .. | ..) form a list
    // that is cut into the 8 contiguous, equally long chunks of logical ids xcd_remap hands the XCDs (with B % 8 != 0 a chunk
    // border falls inside an utterance: 7 utterances are shared by two L2s).
    int id = xcd_remap(blockIdx.x, gridDim.x);
    {
        const int per_b = p.H * qtiles, u = id / per_b, q8 = p.B >> 3, r8 = p.B & 7;
        int x, j;
        if (u < r8 * (q8 + 1)) { x = u / (q8 + 1); j = u - x * (q8 + 1); }
        else { const int u2 = u - r8 * (q8 + 1); x = u2 / q8; j = u2 - x * q8; x += r8; }
        id = (x + 8 * j) * per_b + (id - u * per_b);
    }
    const int qt = id % qtiles; id /= qtiles;
    const int h = id % p.H; const int b = id / p.H;
    const int i0 = qt * BI, iw0 = i0 + wave * 16;
    const size_t qoff = (size_t)b * p.q_bstride + (size_t)h * p.q_hstride;
    const bf16_t* Qu = p.qu + qoff;
    const bf16_t* Kh = p.kh + qoff;
    const bf16_t* Vh = p.vt + qoff;                    // V, same layout as K (transposed at LDS fill)
    const bf16_t* Eh = p.eh + (size_t)h * p.e_hstride;
    const int RS = p.q_rowstride, ERS = p.e_rowstride; // row strides (elements); rows may be only 4-byte aligned (natural layout)
    const int erows = 2 * p.Tg - 1;
    // 16-byte chunks entirely beyond the head width d (DP pads d up to a multiple of 32) are masked to zero anyway: point their
    // loads at chunk 0 of the same row (same cache line as a neighbouring lane's request) instead of fetching the next head's data
    const int dceil = (p.d + 7) & ~7;

    int nkeys = (p.lens[b] + p.G - 1) / p.G;          // unmasked key groups: G*j < lens[b]
    nkeys = nkeys < p.Tg ? nkeys : p.Tg;
    // an empty utterance (lens[b] = 0) has EVERY key masked: the reference adds -1e9 to all scores (attentions.py:698-701), which makes them
    // equal in fp32, so its softmax is uniform over ALL Tg key groups - reproduced as scores * 0 over the full key range
    const bool all_masked = nkeys < 1;
    nkeys = all_masked ? p.Tg : nkeys;
    if constexpr (PROF) { asm volatile("s_nop 0" :: "s"(nkeys)); }
    AT_TICK(8);                                        // prologue a: arguments, tile indices, utterance length

    // ---- this lane's query (column c of the wave's 16): B operands of S^T = K Q^T, kept in registers
    // Only Q + u is stored (attentions.py:674): Q + v = (Q + u) + (v - u), with (v - u) per head column from a small fp32 table
    // (zero beyond d).  One query tensor less to write (the producers' Q/K/V write-out is HBM-write bound) and to read.
    bf16x8 qu[KS], qv[KS];
    {
        const int i = iw0 + c;
        const int ic = i < p.Tg ? i : p.Tg - 1;
        uint4 ra[KS];
        float4 da[KS], db[KS];
        const float* dv = p.dvu + (size_t)h * p.dvu_ld;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int x = ks * 32 + g * 8, xq = x < dceil ? x : 0;
            ra[ks] = ld16(Qu + (size_t)ic * RS + xq);
            da[ks] = *reinterpret_cast<const float4*>(dv + x);
            db[ks] = *reinterpret_cast<const float4*>(dv + x + 4);
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int valid = i < p.Tg ? p.d - (ks * 32 + g * 8) : 0;
            const uint4 m = mask_chunk(ra[ks], valid);
            qu[ks] = as_bf16x8(m);
            const uint4 w = make_uint4(pack_bf2(__uint_as_float(m.x << 16) + da[ks].x, __uint_as_float(m.x & 0xFFFF0000u) + da[ks].y),
                                       pack_bf2(__uint_as_float(m.y << 16) + da[ks].z, __uint_as_float(m.y & 0xFFFF0000u) + da[ks].w),
                                       pack_bf2(__uint_as_float(m.z << 16) + db[ks].x, __uint_as_float(m.z & 0xFFFF0000u) + db[ks].y),
                                       pack_bf2(__uint_as_float(m.w << 16) + db[ks].z, __uint_as_float(m.w & 0xFFFF0000u) + db[ks].w));
            qv[ks] = as_bf16x8(mask_chunk(w, valid));
        }
    }
    if constexpr (PROF) { asm volatile("s_nop 0" :: "v"(qu[0]), "v"(qv[KS - 1])); }
    AT_TICK(9);                                        // prologue b: query loads
    // ---- first positional band: rows R0 .. R0 + BI + 62 (absolute E rows), staged directly
    const int R0 = p.Tg - 1 - i0 - (BI - 1);           // E row of band row 0 for key block 0
    {   // all loads first (one latency), then the LDS writes
        constexpr int NB = ((BI + 63) * CPR + NTHR - 1) / NTHR;
        uint4 fb[NB];
#pragma unroll
        for (int n = 0; n < NB; ++n) {
            const int q = tid + NTHR * n;
            const int rr = q / CPR, x = (q - rr * CPR) * 8;
            const int r = R0 + (rr < BI + 63 ? rr : BI + 62);
            const int rc = r < 0 ? 0 : (r >= erows ? erows - 1 : r);
            fb[n] = ld16(Eh + (size_t)rc * ERS + (x < dceil ? x : 0));
        }
#pragma unroll
        for (int n = 0; n < NB; ++n) {
            const int q = tid + NTHR * n;
            const int rr = q / CPR, x = (q - rr * CPR) * 8;
            const int r = R0 + rr;
            if (q < (BI + 63) * CPR)
                *reinterpret_cast<uint4*>(sE + SM::koff((r + 8192) & (ERING - 1), x >> 3)) = mask_chunk(fb[n], (r >= 0 && r < erows) ? p.d - x : 0);
        }
    }

    AT_TICK(10);                                       // prologue c: first positional band -> LDS
    f32x4 acc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) acc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;
    const float scale2 = all_masked ? 0.f : p.scale * 1.44269504088896340736f;
    float* skew = sS + wave * 16 * SKEW_LD + c * SKEW_LD;
    const int woff = BI - 16 - 16 * wave;             // first band row of this wave inside the workgroup band

    // ---- K / V / new-E staging registers, TWO sets: the loads of key block n+2 are issued when block n is published, so they
    // have a whole block's publish + compute time to arrive.  (With one set the s_memtime phase profile showed the MFMA / softmax
    // phases at 9 % of a wave's life and 46 % waiting for the next block's loads at the publish point: EFFCONF_ATTN_PHASES.)
    constexpr int NK = (BJ * CPR + NTHR - 1) / NTHR, NV = ((BJ / 2) * CPR + NTHR - 1) / NTHR, NE = (64 * CPR + NTHR - 1) / NTHR;
    struct Stage { uint4 lk[NK], lv0[NV], lv1[NV], le[NE]; };
    Stage sa, sb;
#pragma unroll
    for (int n = 0; n < NK; ++n) { sa.lk[n] = make_uint4(0, 0, 0, 0); sb.lk[n] = sa.lk[n]; }
#pragma unroll
    for (int n = 0; n < NV; ++n) { sa.lv0[n] = make_uint4(0, 0, 0, 0); sa.lv1[n] = sa.lv0[n]; sb.lv0[n] = sa.lv0[n]; sb.lv1[n] = sa.lv0[n]; }
#pragma unroll
    for (int n = 0; n < NE; ++n) { sa.le[n] = make_uint4(0, 0, 0, 0); sb.le[n] = sa.le[n]; }

    auto issue_loads = [&](Stage& st_, int jn) __attribute__((always_inline)) {
#pragma unroll
        for (int n = 0; n < NK; ++n) {
            const int q = tid + NTHR * n, r = q / CPR, x = (q - r * CPR) * 8;
            const int j = jn + r;
            st_.lk[n] = ld16(Kh + (size_t)(j < p.Tg ? j : p.Tg - 1) * RS + (x < dceil ? x : 0));
        }
#pragma unroll
        for (int n = 0; n < NV; ++n) {
            const int q = tid + NTHR * n, pr = q & (BJ / 2 - 1), x = (q / (BJ / 2)) * 8;
            const int j = jn + 2 * pr, xc = x < dceil ? x : 0;
            st_.lv0[n] = ld16(Vh + (size_t)(j < p.Tg ? j : p.Tg - 1) * RS + xc);
            st_.lv1[n] = ld16(Vh + (size_t)(j + 1 < p.Tg ? j + 1 : p.Tg - 1) * RS + xc);
        }
        if (jn > 0) {
            const int rnew = R0 + jn + BI - 1;
#pragma unroll
            for (int n = 0; n < NE; ++n) {
                const int q = tid + NTHR * n, rr = q / CPR, x = (q - rr * CPR) * 8;
                int r = rnew + rr;
                r = r < 0 ? 0 : (r >= erows ? erows - 1 : r);
                st_.le[n] = ld16(Eh + (size_t)r * ERS + (x < dceil ? x : 0));
            }
        }
    };
    // Column masks of this thread's K / E chunks (elements >= d of the padded head width), loop invariant: the per-block
    // mask_chunk (two compares + two selects per dword, six chunks per thread and key block) was a fifth of the loop's VALU work.
    // Row validity only matters in an utterance's last key block / the band's tail: wave-uniform branches.
    uint4 cm[NK];
    static_assert(NE == NK, "K and E chunks share the column mapping");
#pragma unroll
    for (int n = 0; n < NK; ++n) {
        const int q = tid + NTHR * n, r = q / CPR, x = (q - r * CPR) * 8;
        cm[n] = mask_chunk(make_uint4(~0u, ~0u, ~0u, ~0u), p.d - x);
    }
    auto and4 = [](uint4 a, uint4 m) { return make_uint4(a.x & m.x, a.y & m.y, a.z & m.z, a.w & m.w); };
    auto publish = [&](const Stage& st_, int j0) __attribute__((always_inline)) {
        __syncthreads();                          // previous block's LDS reads are done
        // ---- publish block j0: K rows, transposed V, and (for j0 > 0) the 64 new band rows
        const bool rows_ok = j0 + BJ <= p.Tg;     // every key row of the block exists
#pragma unroll
        for (int n = 0; n < NK; ++n) {
            const int q = tid + NTHR * n, r = q / CPR, x = (q - r * CPR) * 8;
            uint4 v = and4(st_.lk[n], cm[n]);
            if (!rows_ok) v = (j0 + r < p.Tg) ? v : make_uint4(0, 0, 0, 0);
            if (q < BJ * CPR) *reinterpret_cast<uint4*>(sK + SM::koff(r, x >> 3)) = v;
        }
#pragma unroll
        for (int n = 0; n < NV; ++n) {
            const int q = tid + NTHR * n, pr = q & (BJ / 2 - 1), x = (q / (BJ / 2)) * 8;
            const int j = j0 + 2 * pr;
            if (q < (BJ / 2) * CPR) {
                uint4 v0 = st_.lv0[n], v1 = st_.lv1[n];
                if (!rows_ok) { v0 = mask_chunk(v0, (j < p.Tg) ? 8 : 0); v1 = mask_chunk(v1, (j + 1 < p.Tg) ? 8 : 0); }
                const uint32_t a[4] = {v0.x, v0.y, v0.z, v0.w}, bq[4] = {v1.x, v1.y, v1.z, v1.w};
#pragma unroll
                for (int e = 0; e < 8; ++e)            // element e of both keys as one dword: a byte permute (v_perm_b32)
                    *reinterpret_cast<uint32_t*>(sV + (x + e) * SM::VROW + pr * 4) =
                        __builtin_amdgcn_perm(bq[e >> 1], a[e >> 1], (e & 1) ? 0x07060302u : 0x05040100u);
            }
        }
        if (j0 > 0) {
            const int rnew = R0 + j0 + BI - 1;    // first new absolute E row of this block
            const bool band_ok = rnew >= 0 && rnew + 63 < erows;
#pragma unroll
            for (int n = 0; n < NE; ++n) {
                const int q = tid + NTHR * n, rr = q / CPR, x = (q - rr * CPR) * 8;
                const int r = rnew + rr;
                uint4 v = and4(st_.le[n], cm[n]);
                if (!band_ok) v = (r >= 0 && r < erows) ? v : make_uint4(0, 0, 0, 0);
                if (q < 64 * CPR)
                    *reinterpret_cast<uint4*>(sE + SM::koff((r + 8192) & (ERING - 1), x >> 3)) = v;
            }
        }
        __syncthreads();
    };

    issue_loads(sa, 0);
    if (BJ < nkeys) issue_loads(sb, BJ);
    AT_TICK(0);
    for (int jb = 0; jb < nkeys; jb += 2 * BJ)
#pragma unroll
    for (int half2 = 0; half2 < 2; ++half2) {
        const int j0 = jb + half2 * BJ;
        if (j0 >= nkeys) break;
        if (half2 == 0) publish(sa, j0); else publish(sb, j0);
        AT_TICK(1);
        if (j0 + 2 * BJ < nkeys) { if (half2 == 0) issue_loads(sa, j0 + 2 * BJ); else issue_loads(sb, j0 + 2 * BJ); }
        AT_TICK(2);

        // ---- S^T tiles: rows = keys (g*4+reg within tile jt), cols = queries (c)
        f32x4 st[4];
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            st[jt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(sK + SM::koff(jt * 16 + c, ks * 4 + g));
                st[jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, qu[ks], st[jt], 0, 0, 0);
            }
        }
        if constexpr (PROF) { asm volatile("s_nop 0" :: "v"(st[0]), "v"(st[3])); }
        AT_TICK(3);
        // ---- positional band: PE^T[r'][i] = E[rw0 + r'] . Qv[i], r' in [0, 80); rw0 = absolute E row of this wave's band row 0
        const int rw0 = R0 + j0 + woff + 8192;
#pragma unroll
        for (int rt = 0; rt < 5; ++rt) {
            f32x4 pe = f32x4{0.f, 0.f, 0.f, 0.f};
            const int er = (rw0 + rt * 16 + c) & (ERING - 1);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(sE + SM::koff(er, ks * 4 + g));
                pe = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, qv[ks], pe, 0, 0, 0);
            }
            *reinterpret_cast<f32x4*>(skew + rt * 16 + g * 4) = pe;
        }
        wave_sync();                                  // skew buffer is per wave: LDS ops of one wave execute in order
        AT_TICK(4);

        // ---- realign (r' = j_local - i_local + 15), scale, mask, online softmax
        // scores are kept in log2 units (scale2 = scale * log2 e), so every exponential is one v_exp_f32; the key mask is only
        // applied in an utterance's last block and the running sums are only rescaled when some row's maximum moved (both
        // wave-uniform branches): the softmax was 18 % of a wave's life, all of it VALU
        float mloc = -INFINITY;
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int jl = jt * 16 + g * 4 + r;
                st[jt][r] = (st[jt][r] + skew[jl + 15 - c]) * scale2;
            }
        if (j0 + BJ > nkeys) {
#pragma unroll
            for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                for (int r = 0; r < 4; ++r) st[jt][r] = (j0 + jt * 16 + g * 4 + r < nkeys) ? st[jt][r] : -INFINITY;
        }
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) mloc = fmaxf(mloc, st[jt][r]);
        mloc = fmaxf(mloc, __shfl_xor(mloc, 16));
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32));
        const float m_new = fmaxf(m_run, mloc);       // finite: key j0 of every visited block is unmasked
        {
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            l_run *= alpha;
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                acc[dt][0] *= alpha; acc[dt][1] *= alpha; acc[dt][2] *= alpha; acc[dt][3] *= alpha;
            }
            m_run = m_new;
        }
        float lsum = 0.f;
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = __builtin_amdgcn_exp2f(st[jt][r] - m_run);
                st[jt][r] = e;
                lsum += e;
            }
        l_run += lsum;
        if constexpr (PROF) { asm volatile("s_nop 0" :: "v"(st[0]), "v"(acc[0])); }
        AT_TICK(5);
        // ---- O^T += V^T P^T ; contraction slot (g, e) <-> key (2*c2 + (e>>2))*16 + g*4 + (e&3) on both operands
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2) {
            uint4 pb;
            pb.x = pack_bf2(st[2 * c2][0], st[2 * c2][1]);
            pb.y = pack_bf2(st[2 * c2][2], st[2 * c2][3]);
            pb.z = pack_bf2(st[2 * c2 + 1][0], st[2 * c2 + 1][1]);
            pb.w = pack_bf2(st[2 * c2 + 1][2], st[2 * c2 + 1][3]);
            const bf16x8 pfrag = as_bf16x8(pb);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const char* vrow = sV + (dt * 16 + c) * SM::VROW;
                const uint2 lo = *reinterpret_cast<const uint2*>(vrow + ((2 * c2) * 16 + g * 4) * 2);
                const uint2 hi = *reinterpret_cast<const uint2*>(vrow + ((2 * c2 + 1) * 16 + g * 4) * 2);
                const bf16x8 a = as_bf16x8(make_uint4(lo.x, lo.y, hi.x, hi.y));
                acc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, pfrag, acc[dt], 0, 0, 0);
            }
        }
    }

    if constexpr (PROF) { asm volatile("s_nop 0" :: "v"(acc[0]), "v"(acc[DT - 1])); }
    AT_TICK(6);
    // ---- normalise and scatter back to the un-grouped (B*T, D) layout: 4 consecutive head columns per store when they
    // stay inside one original frame (8-byte stores), element-wise otherwise
    float l_tot = l_run + __shfl_xor(l_run, 16);
    l_tot += __shfl_xor(l_tot, 32);
    const float inv = 1.0f / l_tot;
    const int i = iw0 + c;
    if (i < p.Tg) {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const int x0 = dt * 16 + g * 4;
            if (x0 >= p.d) continue;
            int n0 = h * p.d + x0, toff = 0;
            while (n0 >= p.D) { n0 -= p.D; ++toff; }
            const int t0 = i * p.G + toff;
            if (x0 + 3 < p.d && n0 + 3 < p.D && (n0 & 1) == 0) {
                if (t0 < p.T) {
                    typedef uint32_t u32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));
                    u32x2_a4 w;
                    w[0] = pack_bf2(acc[dt][0] * inv, acc[dt][1] * inv);
                    w[1] = pack_bf2(acc[dt][2] * inv, acc[dt][3] * inv);
                    *reinterpret_cast<u32x2_a4*>(p.out + ((size_t)b * p.T + t0) * p.ldo + n0) = w;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int x = x0 + r;
                    if (x >= p.d) continue;
                    int n = h * p.d + x, tf = 0;
                    while (n >= p.D) { n -= p.D; ++tf; }
                    const int t = i * p.G + tf;
                    if (t < p.T) p.out[((size_t)b * p.T + t) * p.ldo + n] = f2bf(acc[dt][r] * inv);
                }
            }
        }
    }
    if constexpr (PROF) {
        AT_TICK(7);
        // a 1/32 sample of the workgroups reports: same-address atomics from every wave congest the memory system the phases measure
        if ((threadIdx.x & 63) == 0 && (blockIdx.x & 31) == 0) {
            for (int i = 0; i < 11; ++i) atomicAdd(prof + i, ph[i]);
            atomicAdd(prof + 15, 1ull);
        }
    }
#undef AT_TICK
}

unsigned long long* g_attn_prof = nullptr;
void attn_prof_dump() {
    unsigned long long all[4 * 16];
    if (!g_attn_prof || hipMemcpy(all, g_attn_prof, sizeof(all), hipMemcpyDeviceToHost) != hipSuccess) return;
    static const char* names[11] = {"prologue d (issue)", "publish+barriers", "issue loads", "S = K Q^T", "PE band + skew", "softmax", "PV", "epilogue",
                                    "prologue a (args)", "prologue b (Q)", "prologue c (band)"};
    for (int c = 0; c < 4; ++c) {
        const unsigned long long* h = all + 16 * c;
        if (!h[15]) continue;
        unsigned long long tot = 0;
        for (int i = 0; i < 11; ++i) tot += h[i];
        fprintf(stderr, "[attn phases] DP=%d: waves %llu, cycles/wave %.0f\n", 32 * (c + 1), h[15], (double)tot / h[15]);
        for (int i = 0; i < 11; ++i) fprintf(stderr, "[attn phases]   %-20s %10.0f cyc/wave  %5.1f%%\n", names[i], (double)h[i] / h[15], 100.0 * h[i] / tot);
    }
}

__global__ void attn_pad_rows_kernel(GemmParams p, int B) {
    // rows t in [T, Tp): Q = 0 -> Qu = u (and Qv = u + (v - u) = v in the attention kernel);  K = V = 0
    const int Tp = p.Tg * p.G;
    const int npad = Tp - p.T;
    const int total = B * npad * p.D;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int nn = idx % p.D;
        const int rest = idx / p.D;
        const int t = p.T + rest % npad, b = rest / npad;
        const int tq = t / p.G, toff = t - tq * p.G;
        const int flat = toff * p.D + nn;
        const int h = flat / p.d, x = flat - h * p.d;
        const size_t i1 = ((size_t)(b * p.H + h) * p.Tg + tq) * p.dpad + x;
        p.qu[i1] = f2bf(p.u[nn]);
        p.kh[i1] = 0;
        p.vt[i1] = 0;
    }
}

__global__ void attn_pad_rows_nat_kernel(GemmParams p, int B) {
    const int Tp = p.Tg * p.G, npad = Tp - p.T;
    const int total = B * npad * p.D;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int nn = idx % p.D, rest = idx / p.D;
        const int t = p.T + rest % npad, b = rest / npad;
        const size_t i1 = ((size_t)b * Tp + t) * p.D + nn;
        p.qu[i1] = f2bf(p.u[nn]); p.kh[i1] = 0; p.vt[i1] = 0;
    }
}

// ragged batch, natural layout: utterance b owns rows [off[b], off[b + 1]) (its frames padded to a multiple of G); rows len[b] .. are the
// chunk padding of attentions.py:107-138, 671-675: Q = 0 -> Q + u = u, K = V = 0
__global__ void attn_pad_rows_ragged_kernel(bf16_t* qu, bf16_t* kh, bf16_t* vt, const float* __restrict__ u, int D, int G,
                                            const int* __restrict__ off, const int* __restrict__ len, int n) {
    const int total = n * (G - 1) * D;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int nn = idx % D, rest = idx / D;
        const int j = rest % (G - 1), b = rest / (G - 1);
        const int row = off[b] + len[b] + j;
        if (row < off[b + 1]) {
            const size_t i1 = (size_t)row * D + nn;
            qu[i1] = f2bf(u[nn]); kh[i1] = 0; vt[i1] = 0;
        }
    }
}

template <int DP, int NWV>
int launch_dp_w(const AttnParams& p, hipStream_t s) {
    using SM = AttnSmem<DP, NWV>;
    static LdsAttr attr;
    ensure_dynamic_lds(reinterpret_cast<const void*>(&relpos_attention_kernel<DP, NWV, false>), SM::TOTAL, attr);
    const int qtiles = (p.Tg + NWV * 16 - 1) / (NWV * 16);
    if constexpr (NWV == 4 && DP <= 128) {
        static const bool prof = getenv("EFFCONF_ATTN_PHASES") != nullptr;
        if (prof) {
            if (!g_attn_prof) {
                if (hipMalloc(&g_attn_prof, 512) != hipSuccess || hipMemset(g_attn_prof, 0, 512) != hipSuccess) return -1;
                atexit(attn_prof_dump);
            }
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&relpos_attention_kernel<DP, NWV, true>), hipFuncAttributeMaxDynamicSharedMemorySize, SM::TOTAL);
            hipLaunchKernelGGL((relpos_attention_kernel<DP, NWV, true>), dim3(p.B * p.H * qtiles), dim3(NWV * 64), SM::TOTAL, s, p, g_attn_prof + 16 * (DP / 32 - 1));
            return hipGetLastError() == hipSuccess ? 0 : -1;
        }
    }
    hipLaunchKernelGGL((relpos_attention_kernel<DP, NWV, false>), dim3(p.B * p.H * qtiles), dim3(NWV * 64), SM::TOTAL, s, p, nullptr);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

template <int DP>
int launch_dp(const AttnParams& p, hipStream_t s) {
    // 64-query (4-wave) workgroups, two per CU: the kernel is latency-bound on its K / V / E loads (phase profile above), and two
    // independent workgroups per CU hide more of it than one 128-query workgroup that stages K / V half as often (1.31 -> 1.23 ms
    // per step once the loads run two key blocks ahead).  Option "attn_waves" = 8 selects the 128-query variant.
    const int force = p.force_waves;
    constexpr bool fits8 = AttnSmem<DP, 8>::TOTAL <= 160 * 1024;
    if constexpr (fits8)
        if (force == 8) return launch_dp_w<DP, 8>(p, s);
    return launch_dp_w<DP, 4>(p, s);
}

}  // namespace

int launch_relpos_attention(const AttnParams& p, hipStream_t s) {
    if (p.B <= 0 || p.Tg <= 0) return 0;
    if (p.dpad < p.d) return -2;
    switch (p.dpad) {
        case 32: return launch_dp<32>(p, s);
        case 64: return launch_dp<64>(p, s);
        case 96: return launch_dp<96>(p, s);
        case 128: return launch_dp<128>(p, s);
        case 160: return launch_dp<160>(p, s);
        case 192: return launch_dp<192>(p, s);
    }
    return -3;
}

int launch_attn_pad_rows_nat(const GemmParams& p, int B, hipStream_t s) {
    const int npad = p.Tg * p.G - p.T;
    if (npad <= 0 || B <= 0) return 0;
    const int total = B * npad * p.D;
    hipLaunchKernelGGL(attn_pad_rows_nat_kernel, dim3((total + 255) / 256), dim3(256), 0, s, p, B);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_attn_pad_rows_ragged(bf16_t* qu, bf16_t* kh, bf16_t* vt, const float* u, int D, int G, const RaggedRows& rg, hipStream_t s) {
    if (G <= 1 || rg.n <= 0) return 0;
    const int total = rg.n * (G - 1) * D;
    hipLaunchKernelGGL(attn_pad_rows_ragged_kernel, dim3((total + 255) / 256), dim3(256), 0, s, qu, kh, vt, u, D, G, rg.off, rg.len, rg.n);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_attn_pad_rows(const GemmParams& p, int B, hipStream_t s) {
    const int npad = p.Tg * p.G - p.T;
    if (npad <= 0 || B <= 0) return 0;
    const int total = B * npad * p.D;
    hipLaunchKernelGGL(attn_pad_rows_kernel, dim3((total + 255) / 256), dim3(256), 0, s, p, B);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ---- attention maps: the third return value of ConformerEncoder.forward (reference encoders.py:126-142, attentions.py:620 / 718: att_w, the
// softmax of every block, (B, H, Tg, Tg)).  No caller of the hot path consumes them, so they are an opt-in side output
// (effconf_encoder_set_attention_outputs): one wave per (query row, head, utterance) recomputes the scores in fp32 from the bf16 Q + u, K, E the
// attention kernel reads, with the reference's additive -1e9 masks (a fully masked row is the uniform 1 / Tg, as there), and writes the
// normalised row.  Either Q / K / V layout (the strides of AttnParams); ragged batches (natural layout) since round 4.
namespace {
__global__ __launch_bounds__(64) void attention_probs_kernel(const AttnParams p, float* __restrict__ att) {
    extern __shared__ float pm[];
    float* qu = pm;
    float* qv = pm + p.dpad;
    float* sc = pm + 2 * p.dpad;
    const int i = blockIdx.x, h = blockIdx.y, b = blockIdx.z, lane = threadIdx.x;
    const int d = p.d;
    // Ragged batch (round 4; natural layout): utterance b has its own grouped length, its rows start at rag_off[b], its positional rows are the
    // last 2 Tg - 1 of the table built for the longest utterance (attention2.hip).  The maps keep the (B, H, Tgmax, Tgmax) rectangle of the
    // reference's padded batch; an utterance's Tg x Tg block is the map of that utterance run alone, everything outside it is written as zero.
    const int TgM = p.Tg;
    int Tg = TgM;
    size_t base = (size_t)b * p.q_bstride;
    size_t eshift = 0;
    if (p.rag_off) {
        Tg = (p.lens[b] + p.G - 1) / p.G;
        base = (size_t)p.rag_off[b] * p.D;
        eshift = (size_t)(p.rag_tgmax - Tg) * p.e_rowstride;
    }
    float* row = att + (((size_t)b * p.H + h) * TgM + i) * TgM;
    if (i >= Tg) {                                                  // rows behind this utterance's own grouped length (ragged only)
        for (int j = lane; j < TgM; j += 64) row[j] = 0.f;
        return;
    }
    const bf16_t* qrow = p.qu + base + (size_t)h * p.q_hstride + (size_t)i * p.q_rowstride;
    for (int x = lane; x < d; x += 64) {
        const float q = bf2f(qrow[x]);
        qu[x] = q;
        qv[x] = q + p.dvu[(size_t)h * p.dvu_ld + x];
    }
    __syncthreads();
    const int len = p.lens[b];
    const int erows = p.causal ? Tg : 2 * Tg - 1;
    float mx = -INFINITY;
    for (int j = lane; j < Tg; j += 64) {
        const bf16_t* kr = p.kh + base + (size_t)h * p.q_hstride + (size_t)j * p.q_rowstride;
        int r = Tg - 1 + j - i;
        r = r < 0 ? 0 : (r >= erows ? erows - 1 : r);                 // causal, j > i: outside the table and masked below
        const bf16_t* er = p.eh + eshift + (size_t)h * p.e_hstride + (size_t)r * p.e_rowstride;
        float s1 = 0.f, s2 = 0.f;
        for (int x = 0; x < d; ++x) { s1 = fmaf(qu[x], bf2f(kr[x]), s1); s2 = fmaf(qv[x], bf2f(er[x]), s2); }
        float s = (s1 + s2) * p.scale;
        if (p.G * j >= len || j - i > p.band_r || i - j > p.band_l || (p.causal && j > i)) s += -1e9f;   // attentions.py:698-701, 1377-1403
        sc[j] = s;
        mx = fmaxf(mx, s);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float sum = 0.f;
    for (int j = lane; j < Tg; j += 64) { const float e = expf(sc[j] - mx); sc[j] = e; sum += e; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    for (int j = lane; j < TgM; j += 64) row[j] = j < Tg ? sc[j] / sum : 0.f;
}
}  // namespace

int launch_attention_probs(const AttnParams& p, float* att, hipStream_t s) {
    if (!att || p.B <= 0 || p.Tg <= 0) return 0;
    if (p.rag_off && p.q_rowstride != p.G * p.D) return -2;          // ragged batches: natural layout
    const size_t lds = (size_t)(2 * p.dpad + p.Tg) * 4;
    if (lds > 64 * 1024) return -2;
    hipLaunchKernelGGL(attention_probs_kernel, dim3(p.Tg, p.H, p.B), dim3(64), lds, s, p, att);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
