// RNN-T greedy decode (gfx950): prediction network (Embedding + 1-layer LSTM), joint network and the greedy loop.
//
// Reference: Transducer.gready_search_decoding (models/transducer.py:139-186), RnnDecoder.forward
// (models/decoders.py:41-70: Embedding -> nn.LSTM, batch_first, 1 layer), JointNetwork.forward in decoding mode
// (models/joint_networks.py:80-104: linear_encoder(f) + linear_decoder(g) -> tanh -> linear_joint), and
// max_consec_dec_step (transducer.py:83, 173).
//
// The reference decodes one utterance at a time with one joint_network call (three nn.Linear, a softmax and an
// .argmax() host sync) per (frame, token) decision.  Here:
//   * everything input-independent is hoisted into tables at finalize():  Gin[y] = W_ih emb[y] + b_ih + b_hh for every
//     token id y (the LSTM only ever sees embedding rows), so a decoder step is one 4H x H mat-vec;
//   * linear_encoder(f) is one fp32 GEMM over all (b, t) per batch (the reference recomputes it per decision,
//     joint_networks.py:82);
//   * the sequential part runs as ONE persistent workgroup per utterance: state (h, c, g_dec, z) lives in LDS, weights
//     stream from L2 in a k-major float4 layout (W4[k/4][n] = W[n][4k/4 .. +3]) so that a wave's loads are contiguous
//     1 KB rows, and RNNT_KF consecutive encoder frames are evaluated speculatively per pass over the joint weight
//     (between two emitted tokens the decoder output does not change, so the extra frames are exactly the frames the
//     reference would evaluate next as long as it keeps reading blanks).
// All arithmetic is fp32 (the decisions feed back into the recurrence: one flipped argmax changes every later token,
// so the head is kept at the reference's precision; softmax().log().argmax() of transducer.py:164 is argmax(logits)).
// Bound: L2 -> CU bandwidth on the weights (4H*H*4 + J*H*4 bytes per token, V*J*4 per joint pass).
#include "kernels.h"
#include "../../include/effconf.h"

#include <cmath>
#include <map>
#include <string>
#include <vector>

int ec_fail(const char* msg);

namespace {

constexpr int NT = 512;        // threads per utterance
constexpr int KF = 4;          // encoder frames per joint pass
constexpr int JMC = 2;         // joint columns per thread and pass

// ------------------------------------------------------------------------------------------------------------------
// fp32 GEMM  C[m][n] = sum_k A[m][k] * B[n][k] + bias[n]   (nn.Linear semantics), K % 4 == 0.  64x64x16 tiles.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sgemm_nt_kernel(const float* __restrict__ A, int lda, const float* __restrict__ Bw, int ldb,
                                                       const float* __restrict__ bias, float* __restrict__ C, int ldc, int M, int N, int K) {
    __shared__ float As[16][68], Bs[16][68];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int lr = tid >> 2, lk = (tid & 3) * 4;
    const int am = m0 + lr < M ? m0 + lr : M - 1, bn = n0 + lr < N ? n0 + lr : N - 1;
    float acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
    for (int k0 = 0; k0 < K; k0 += 16) {
        const int kk = k0 + lk < K ? k0 + lk : K - 4;                       // clamped, masked below
        float4 a = *reinterpret_cast<const float4*>(A + (size_t)am * lda + kk);
        float4 b = *reinterpret_cast<const float4*>(Bw + (size_t)bn * ldb + kk);
        if (k0 + lk >= K) { a = make_float4(0.f, 0.f, 0.f, 0.f); b = a; }
        __syncthreads();
        As[lk + 0][lr] = a.x; As[lk + 1][lr] = a.y; As[lk + 2][lr] = a.z; As[lk + 3][lr] = a.w;
        Bs[lk + 0][lr] = b.x; Bs[lk + 1][lr] = b.y; Bs[lk + 2][lr] = b.z; Bs[lk + 3][lr] = b.w;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const float4 av = *reinterpret_cast<const float4*>(&As[k][ty * 4]);
            const float4 bv = *reinterpret_cast<const float4*>(&Bs[k][tx * 4]);
            const float ar[4] = {av.x, av.y, av.z, av.w}, br[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(ar[i], br[j], acc[i][j]);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + ty * 4 + i;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + tx * 4 + j;
            if (n < N) C[(size_t)m * ldc + n] = acc[i][j] + (bias ? bias[n] : 0.f);
        }
    }
}

int launch_sgemm_nt(const float* A, int lda, const float* Bw, int ldb, const float* bias, float* C, int ldc, int M, int N, int K, hipStream_t s) {
    if (M <= 0 || N <= 0) return 0;
    if (K % 4 || K < 4) return -2;
    hipLaunchKernelGGL(sgemm_nt_kernel, dim3((N + 63) / 64, (M + 63) / 64), dim3(256), 0, s, A, lda, Bw, ldb, bias, C, ldc, M, N, K);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ------------------------------------------------------------------------------------------------------------------
// mat-vec against a k-major float4 weight image: y[n] = sum_k W[n][k] x[k], W4[(k/4)*N + n] = W[n][k..k+3].
// Thread owns columns n = n0 + tid + NT*m (m < MC); x is NX vectors in LDS (pitch xp floats), read as broadcasts.
// ------------------------------------------------------------------------------------------------------------------
template <int MC, int NX>
__device__ __forceinline__ void matvec(const float4* __restrict__ W4, int N, int K4, int n0, const float* xs, int xp, float (&acc)[MC][NX]) {
    int col[MC];
#pragma unroll
    for (int m = 0; m < MC; ++m) {
        const int n = n0 + (int)threadIdx.x + NT * m;
        col[m] = n < N ? n : N - 1;                                          // clamped: garbage columns are never used
#pragma unroll
        for (int x = 0; x < NX; ++x) acc[m][x] = 0.f;
    }
#pragma unroll 4
    for (int k4 = 0; k4 < K4; ++k4) {
        float4 w[MC];
#pragma unroll
        for (int m = 0; m < MC; ++m) w[m] = W4[(size_t)k4 * N + col[m]];
#pragma unroll
        for (int x = 0; x < NX; ++x) {
            const float4 xv = *reinterpret_cast<const float4*>(xs + x * xp + 4 * k4);
#pragma unroll
            for (int m = 0; m < MC; ++m)
                acc[m][x] = fmaf(w[m].w, xv.w, fmaf(w[m].z, xv.z, fmaf(w[m].y, xv.y, fmaf(w[m].x, xv.x, acc[m][x]))));
        }
    }
}

struct RnntDev {
    const float* gin;        // [V][4H]   W_ih emb[y] + b_ih + b_hh
    const float4* whh4;      // [H/4][4H]
    const float4* wd4;       // [H/4][J]
    const float* bd;         // [J]
    const float4* wj4;       // [J/4][V]
    const float* bj;         // [V]
    int H, J, V, max_consec;
};

__device__ __forceinline__ float sigmoid_precise(float x) { return 1.0f / (1.0f + expf(-x)); }

// One workgroup = one utterance.  fe: [B][T][J] = linear_encoder(f) incl. bias.
__global__ __launch_bounds__(NT) void rnnt_greedy_kernel(RnntDev w, const float* __restrict__ fe, const int64_t* __restrict__ lens,
                                                         int T, int* __restrict__ tokens, int* __restrict__ counts, int max_tok) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int H = w.H, J = w.J, V = w.V;
    float* sh = lds;                       // [H]   h
    float* sc = sh + H;                    // [H]   c
    float* sg = sc + H;                    // [4H]  gate pre-activations
    float* sgd = sg + 4 * H;               // [J]   linear_decoder(h)
    float* sz = sgd + J;                   // [KF][J] tanh(fe + gd)
    float* sbv = sz + KF * J;              // [KF][NT/64] wave-best values
    int* sbi = reinterpret_cast<int*>(sbv + KF * (NT / 64));   // [KF][NT/64] wave-best indices
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x;
    int Tb = (int)lens[b];
    Tb = Tb < 0 ? 0 : (Tb > T ? T : Tb);
    for (int i = tid; i < 2 * H; i += NT) sh[i] = 0.f;   // hidden = None -> zeros (decoders.py:59-60)
    __syncthreads();
    int y = 0, enc_step = 0, consec = 0, ntok = 0;       // y = x.new_zeros(1,1) (transducer.py:151)
    const float* feb = fe + (size_t)b * T * J;

    while (enc_step < Tb) {
        // ---- decoder step: gates = Gin[y] + W_hh h   (torch LSTM gate order i, f, g, o)
        for (int n0 = 0; n0 < 4 * H; n0 += NT * 5) {
            float acc[5][1];
            matvec<5, 1>(w.whh4, 4 * H, H / 4, n0, sh, H, acc);
#pragma unroll
            for (int m = 0; m < 5; ++m) {
                const int n = n0 + tid + NT * m;
                if (n < 4 * H) sg[n] = acc[m][0] + w.gin[(size_t)y * 4 * H + n];
            }
        }
        __syncthreads();
        for (int j = tid; j < H; j += NT) {
            const float ig = sigmoid_precise(sg[j]), fg = sigmoid_precise(sg[H + j]), gg = tanhf(sg[2 * H + j]), og = sigmoid_precise(sg[3 * H + j]);
            const float c = fg * sc[j] + ig * gg;
            sc[j] = c;
            sh[j] = og * tanhf(c);
        }
        __syncthreads();
        // ---- gd = linear_decoder(h)
        for (int n0 = 0; n0 < J; n0 += NT * 2) {
            float acc[2][1];
            matvec<2, 1>(w.wd4, J, H / 4, n0, sh, H, acc);
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const int n = n0 + tid + NT * m;
                if (n < J) sgd[n] = acc[m][0] + w.bd[n];
            }
        }
        __syncthreads();
        // ---- joint loop: KF frames per pass, until a token is emitted or the utterance ends
        bool emitted = false;
        while (!emitted && enc_step < Tb) {
            for (int i = tid; i < KF * J; i += NT) {
                const int kf = i / J, j = i - kf * J;
                const int t = enc_step + kf < Tb ? enc_step + kf : Tb - 1;    // clamped: frames past the end are never consulted
                sz[i] = tanhf(feb[(size_t)t * J + j] + sgd[j]);
            }
            __syncthreads();
            float bv[KF]; int bi[KF];
#pragma unroll
            for (int kf = 0; kf < KF; ++kf) { bv[kf] = -INFINITY; bi[kf] = 0x7fffffff; }
            for (int n0 = 0; n0 < V; n0 += NT * JMC) {
                float acc[JMC][KF];
                matvec<JMC, KF>(w.wj4, V, J / 4, n0, sz, J, acc);
#pragma unroll
                for (int m = 0; m < JMC; ++m) {
                    const int n = n0 + tid + NT * m;
                    const bool ok = n < V;
                    const float bz = w.bj[ok ? n : V - 1];
#pragma unroll
                    for (int kf = 0; kf < KF; ++kf) {
                        const float v = acc[m][kf] + bz;
                        if (ok && (v > bv[kf] || (v == bv[kf] && n < bi[kf]))) { bv[kf] = v; bi[kf] = n; }
                    }
                }
            }
#pragma unroll
            for (int kf = 0; kf < KF; ++kf) {
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) {
                    const float ov = __shfl_xor(bv[kf], o);
                    const int oi = __shfl_xor(bi[kf], o);
                    if (ov > bv[kf] || (ov == bv[kf] && oi < bi[kf])) { bv[kf] = ov; bi[kf] = oi; }
                }
                if (lane == 0) { sbv[kf * (NT / 64) + wave] = bv[kf]; sbi[kf * (NT / 64) + wave] = bi[kf]; }
            }
            __syncthreads();
            // ---- every thread walks the KF decisions identically (transducer.py:158-176)
#pragma unroll
            for (int kf = 0; kf < KF; ++kf) {
                if (emitted || enc_step >= Tb) break;
                float v = sbv[kf * (NT / 64)]; int pred = sbi[kf * (NT / 64)];
#pragma unroll
                for (int q = 1; q < NT / 64; ++q) {
                    const float ov = sbv[kf * (NT / 64) + q]; const int oi = sbi[kf * (NT / 64) + q];
                    if (ov > v || (ov == v && oi < pred)) { v = ov; pred = oi; }
                }
                if (pred == 0 || consec == w.max_consec) { consec = 0; ++enc_step; }
                else {
                    ++consec;
                    if (tid == 0 && ntok < max_tok) tokens[(size_t)b * max_tok + ntok] = pred;
                    ++ntok;
                    y = pred;
                    emitted = true;
                }
            }
            __syncthreads();     // sz / sbv are rewritten by the next pass
        }
    }
    if (tid == 0) counts[b] = ntok < max_tok ? ntok : max_tok;
    for (int i = ntok + tid; i < max_tok; i += NT) tokens[(size_t)b * max_tok + i] = 0;
}

struct HostT { std::vector<int64_t> shape; std::vector<float> data; };

}  // namespace

struct EcRnnt {
    EcRnntConfig cfg;
    std::map<std::string, HostT> host;
    std::vector<void*> allocs;
    RnntDev dev{};
    float* we = nullptr;     // linear_encoder.weight [J][De]
    float* be = nullptr;
    bool finalized = false;
};

namespace {

void* upload(EcRnnt* r, const void* src, size_t bytes) {
    void* d = nullptr;
    if (hipMalloc(&d, bytes) != hipSuccess) return nullptr;
    r->allocs.push_back(d);
    if (src && hipMemcpy(d, src, bytes, hipMemcpyHostToDevice) != hipSuccess) return nullptr;
    return d;
}

// W [N][K] row-major -> float4 image [K/4][N]
std::vector<float> kmajor4(const std::vector<float>& W, int N, int K) {
    std::vector<float> o((size_t)N * K);
    for (int k4 = 0; k4 < K / 4; ++k4)
        for (int n = 0; n < N; ++n)
            for (int e = 0; e < 4; ++e) o[((size_t)k4 * N + n) * 4 + e] = W[(size_t)n * K + 4 * k4 + e];
    return o;
}

const HostT* need(EcRnnt* r, const char* key, std::initializer_list<int64_t> shape, std::string& err) {
    auto it = r->host.find(key);
    if (it == r->host.end()) { err = std::string("missing tensor ") + key; return nullptr; }
    if (it->second.shape != std::vector<int64_t>(shape)) { err = std::string("bad shape for ") + key; return nullptr; }
    return &it->second;
}

}  // namespace

extern "C" {

EcRnnt* effconf_rnnt_create(const EcRnntConfig* c) {
    if (!c) { ec_fail("null config"); return nullptr; }
    if (c->num_layers != 1) { ec_fail("RnnDecoder with num_layers == 1 is native (every shipped Transducer config)"); return nullptr; }
    if (c->joint_mode != 0 || c->joint_act != 0) { ec_fail("joint network: 'sum' + 'tanh' is native (every shipped Transducer config)"); return nullptr; }
    if (c->dim_encoder % 4 || c->dim_decoder % 4 || c->dim_joint % 4 || c->dim_encoder < 4 || c->dim_decoder < 4 || c->dim_joint < 4 ||
        c->vocab_size < 2 || c->max_consec_dec_step < 0) { ec_fail("dims must be positive multiples of 4"); return nullptr; }
    EcRnnt* r = new EcRnnt();
    r->cfg = *c;
    return r;
}

void effconf_rnnt_destroy(EcRnnt* r) {
    if (!r) return;
    for (void* p : r->allocs) (void)hipFree(p);
    delete r;
}

int effconf_rnnt_load_tensor(EcRnnt* r, const char* key, const float* host, const int64_t* shape, int32_t ndim) {
    if (!r || !key || !host) return ec_fail("null argument");
    HostT t;
    int64_t n = 1;
    for (int i = 0; i < ndim; ++i) { t.shape.push_back(shape[i]); n *= shape[i]; }
    t.data.assign(host, host + n);
    r->host[key] = std::move(t);
    r->finalized = false;
    return 0;
}

int effconf_rnnt_finalize(EcRnnt* r) {
    if (!r) return ec_fail("null handle");
    for (void* p : r->allocs) (void)hipFree(p);
    r->allocs.clear();
    const int H = r->cfg.dim_decoder, J = r->cfg.dim_joint, V = r->cfg.vocab_size, De = r->cfg.dim_encoder;
    std::string err;
    const HostT* emb = need(r, "decoder.embedding.weight", {V, H}, err);
    const HostT* wih = need(r, "decoder.rnn.weight_ih_l0", {4 * H, H}, err);
    const HostT* whh = need(r, "decoder.rnn.weight_hh_l0", {4 * H, H}, err);
    const HostT* bih = need(r, "decoder.rnn.bias_ih_l0", {4 * H}, err);
    const HostT* bhh = need(r, "decoder.rnn.bias_hh_l0", {4 * H}, err);
    const HostT* we = need(r, "joint_network.linear_encoder.weight", {J, De}, err);
    const HostT* be = need(r, "joint_network.linear_encoder.bias", {J}, err);
    const HostT* wd = need(r, "joint_network.linear_decoder.weight", {J, H}, err);
    const HostT* bd = need(r, "joint_network.linear_decoder.bias", {J}, err);
    const HostT* wj = need(r, "joint_network.linear_joint.weight", {V, J}, err);
    const HostT* bj = need(r, "joint_network.linear_joint.bias", {V}, err);
    if (!emb || !wih || !whh || !bih || !bhh || !we || !be || !wd || !bd || !wj || !bj) return ec_fail(err.c_str());
    std::vector<float> bsum(4 * H);
    for (int i = 0; i < 4 * H; ++i) bsum[i] = bih->data[i] + bhh->data[i];
    const std::vector<float> whh4 = kmajor4(whh->data, 4 * H, H), wd4 = kmajor4(wd->data, J, H), wj4 = kmajor4(wj->data, V, J);
    float* d_emb = (float*)upload(r, emb->data.data(), emb->data.size() * 4);
    float* d_wih = (float*)upload(r, wih->data.data(), wih->data.size() * 4);
    float* d_bsum = (float*)upload(r, bsum.data(), bsum.size() * 4);
    float* d_gin = (float*)upload(r, nullptr, (size_t)V * 4 * H * 4);
    r->dev.whh4 = (const float4*)upload(r, whh4.data(), whh4.size() * 4);
    r->dev.wd4 = (const float4*)upload(r, wd4.data(), wd4.size() * 4);
    r->dev.bd = (const float*)upload(r, bd->data.data(), bd->data.size() * 4);
    r->dev.wj4 = (const float4*)upload(r, wj4.data(), wj4.size() * 4);
    r->dev.bj = (const float*)upload(r, bj->data.data(), bj->data.size() * 4);
    r->we = (float*)upload(r, we->data.data(), we->data.size() * 4);
    r->be = (float*)upload(r, be->data.data(), be->data.size() * 4);
    if (!d_emb || !d_wih || !d_bsum || !d_gin || !r->dev.whh4 || !r->dev.wd4 || !r->dev.bd || !r->dev.wj4 || !r->dev.bj || !r->we || !r->be)
        return ec_fail("device allocation / upload failed");
    // Gin[y] = W_ih emb[y] + (b_ih + b_hh)   for every token id (Embedding rows are the only LSTM inputs, decoders.py:55)
    if (launch_sgemm_nt(d_emb, H, d_wih, H, d_bsum, d_gin, 4 * H, V, 4 * H, H, nullptr) != 0) return ec_fail("Gin GEMM launch failed");
    if (hipDeviceSynchronize() != hipSuccess) return ec_fail("Gin GEMM failed");
    r->dev.gin = d_gin;
    r->dev.H = H; r->dev.J = J; r->dev.V = V; r->dev.max_consec = r->cfg.max_consec_dec_step;
    r->finalized = true;
    return 0;
}

size_t effconf_rnnt_workspace_bytes(const EcRnnt* r, int32_t batch, int32_t t_out) {
    if (!r || batch < 0 || t_out < 0) return 0;
    return (size_t)batch * t_out * r->cfg.dim_joint * 4 + 256;
}

int32_t effconf_rnnt_max_tokens(const EcRnnt* r, int32_t t_out) {
    if (!r || t_out < 0) return 0;
    const int m = r->cfg.max_consec_dec_step * t_out;
    return m > 1 ? m : 1;
}

int effconf_rnnt_greedy(EcRnnt* r, const float* enc_out, const int64_t* out_len, int32_t batch, int32_t t_out,
                        int32_t* tokens, int32_t* token_len, int32_t max_tokens, void* workspace, size_t workspace_bytes, void* stream) {
    if (!r || !r->finalized) return ec_fail("rnnt handle not finalized");
    if (batch == 0) return 0;
    if (!enc_out || !out_len || !tokens || !token_len || !workspace) return ec_fail("null argument");
    if (batch < 0 || t_out <= 0 || max_tokens < effconf_rnnt_max_tokens(r, t_out)) return ec_fail("bad shape / token buffer smaller than max_consec_dec_step * T_out");
    if (workspace_bytes < effconf_rnnt_workspace_bytes(r, batch, t_out)) return ec_fail("workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const int H = r->cfg.dim_decoder, J = r->cfg.dim_joint, De = r->cfg.dim_encoder;
    float* fe = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(workspace) + 255) & ~(uintptr_t)255);
    // linear_encoder(f) for every frame of the batch, once (joint_networks.py:82 recomputes it per decision)
    if (launch_sgemm_nt(enc_out, De, r->we, De, r->be, fe, J, batch * t_out, J, De, s) != 0) return ec_fail("linear_encoder GEMM launch failed");
    const size_t lds = (size_t)(2 * H + 4 * H + J + KF * J + 2 * KF * (NT / 64)) * 4;
    hipLaunchKernelGGL(rnnt_greedy_kernel, dim3(batch), dim3(NT), lds, s, r->dev, fe, out_len, t_out, tokens, token_len, max_tokens);
    return hipGetLastError() == hipSuccess ? 0 : ec_fail("rnnt_greedy launch failed");
}

}  // extern "C"
