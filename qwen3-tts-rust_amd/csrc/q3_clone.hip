// q3_clone.hip — the two encoders of the voice-clone front-end, gfx950.
//
// Replaces AudioEncoder::encode / SpeakerEncoder::encode (/root/reference/src/models/onnx.rs:82-160: two onnxruntime CPU
// sessions) behind TtsEngine::create_voice_file (/root/reference/src/tts/engine.rs:324-387). The reference holds the graphs
// only as ONNX files that are not in its repository, so the structure is the model family's:
//   speaker encoder  log-mel [T][128] -> TDNN k5 -> 3 x SE-Res2Net -> aggregation -> attentive statistics pooling -> [2048]
//   audio encoder    PCM -> causal SEANet conv stack (x960) -> 8-layer sliding-window transformer -> stride-2 conv ->
//                    split residual VQ (1 semantic + 15 acoustic codebooks, nearest neighbour) -> [frames][16] codes
// The output of the second is integer, so everything here is exact: every convolution is an im2col (k_im2col, zero /
// reflect / replicate padding and the preceding ELU folded in) followed by the canonical exact GEMM of DESIGN.md §4.1
// (q3_launch_gemm: bf16 weights, f32 activations, v_mfma_f32_16x16x4_f32 in the fixed K order), K zero-padded to a multiple
// of 512; reductions are sequential fmaf chains in ascending index; exp / tanh / sigmoid / GELU are built on q3_expf. The
// oracle (oracle/q3_oracle_clone.c) follows the same order, so codes and floats are compared bit for bit.
// This path runs once per cloned voice (a 3 s clip: 72 000 samples, 282 mel frames, 75 transformer rows, 38 code frames);
// it is latency-bound, K padding of the narrow 24 kHz layers included, and no roofline is claimed for it.
#include <cmath>
#include <cstring>
#include <vector>

#include "q3_engine.h"

namespace {

#define Q3G_CLONE 5
enum { SC_TDNN0 = 0, SC_BLOCK = 1, SC_MFA = 8, SC_ASP_TDNN = 9, SC_ASP_CONV = 10, SC_FC = 11, AC_CONV0 = 32, AC_STAGE = 33,
       AC_LAST = 60, AC_TFM = 64, AC_DOWN = 100, AC_SEM_PROJ = 101, AC_AC_PROJ = 102, AC_CODEBOOK = 110 };
enum { SW_TDNN1 = 0, SW_TDNN2 = 2, SW_SE1 = 4, SW_SE2 = 6, SW_RES2 = 16 };
enum { TW_LN1_W = 0, TW_LN1_B, TW_QKV, TW_O, TW_LS1, TW_LN2_W, TW_LN2_B, TW_FC1, TW_FC2, TW_LS2 };
enum { PAD_ZERO = 0, PAD_REFLECT = 1, PAD_REPLICATE = 2 };
enum { ACT_NONE = 0, ACT_RELU = 1, ACT_ELU = 2, ACT_TANH = 3, ACT_SIGMOID = 4, ACT_GELU = 5, ACT_RELU_TANH = 6 };
#define ROPE_MAX_T 16384

Q3_HD float c_clamp80(float x) { return fminf(fmaxf(x, -80.0f), 80.0f); }
Q3_HD float c_tanh(float x) {
    const float a = fminf(fabsf(x), 40.0f);
    const float e = q3_expf(-2.0f * a);
    const float t = (1.0f - e) / (1.0f + e);
    return x < 0.0f ? -t : t;
}
Q3_HD float c_act(float x, int kind) {
    switch (kind) {
    case ACT_RELU: return x > 0.0f ? x : 0.0f;
    case ACT_ELU: return x > 0.0f ? x : q3_expf(c_clamp80(x)) - 1.0f;
    case ACT_TANH: return c_tanh(x);
    case ACT_SIGMOID: return 1.0f / (1.0f + q3_expf(-c_clamp80(x)));
    case ACT_GELU: {
        float u = x * x; u = u * x;
        const float inner = 0.7978845608f * fmaf(0.044715f, u, x);
        return (0.5f * x) * (1.0f + c_tanh(inner));
    }
    case ACT_RELU_TANH: return c_tanh(x > 0.0f ? x : 0.0f);
    default: return x;
    }
}

// A[t][j*cin + c] = act(src[pad(t*stride + j*dil - padl)][c] (+ src2[..][c])), zero in the K padding. 4 columns per thread.
__global__ void k_im2col(const float* __restrict__ src, const float* __restrict__ src2, int lds, int lds2, int T_in, int cin, int k,
                         int stride, int dil, int padl, int mode, int pre_act, float* __restrict__ A, int kp, int T_out) {
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int per_row = kp >> 2;
    if (gid >= (size_t)T_out * per_row) return;
    const int t = (int)(gid / per_row), kk0 = (int)(gid % per_row) * 4;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int kk = kk0 + e;
        float x = 0.0f;
        if (kk < k * cin) {
            const int j = kk / cin, ch = kk - j * cin;
            int i = t * stride + j * dil - padl;
            if (i < 0 || i >= T_in) {
                if (mode == PAD_ZERO) i = -1;
                else {
                    if (mode == PAD_REFLECT) i = i < 0 ? -i : 2 * (T_in - 1) - i;
                    i = i < 0 ? 0 : (i > T_in - 1 ? T_in - 1 : i);
                }
            }
            if (i >= 0) {
                x = src[(size_t)i * lds + ch];
                if (src2) x = x + src2[(size_t)i * lds2 + ch];
                x = c_act(x, pre_act);
            }
        }
        v[e] = x;
    }
    *(float4*)(A + (size_t)t * kp + kk0) = make_float4(v[0], v[1], v[2], v[3]);
}

__global__ void k_act(float* x, int ld, int n, int T, int kind) {
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (size_t)T * n) return;
    const size_t t = gid / n, i = gid % n;
    x[t * ld + i] = c_act(x[t * ld + i], kind);
}
// mode 0: dst = a + b; mode 1: dst = a * s[c] + b (squeeze-excitation + residual); mode 2: dst = a + s[c] * b (LayerScale)
__global__ void k_combine(float* dst, int ldd, const float* a, int lda, const float* b, int ldb, const float* s, int n, int T, int mode) {
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (size_t)T * n) return;
    const size_t t = gid / n, i = gid % n;
    const float av = a[t * lda + i], bv = b[t * ldb + i];
    float r;
    if (mode == 0) r = av + bv;
    else if (mode == 1) r = av * s[i] + bv;
    else r = av + s[i] * bv;
    dst[t * ldd + i] = r;
}
#define CS_B 16  // rows loaded per batch: the loads of a batch are in flight together, the chain consumes them in order
// per channel: mean = sum_t w_t x_t (w_t = 1/T when w == nullptr), std = sqrt(max(sum_t w_t (x_t - mean)^2, 1e-12)); ascending t
__global__ void k_colstats(const float* __restrict__ x, int ld, int T, int C, const float* __restrict__ w, int ldw, float* mean, float* sd) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float u = 1.0f / (float)T;
    float m = 0.0f;
    for (int t0 = 0; t0 < T; t0 += CS_B) {
        float xv[CS_B], wv[CS_B];
#pragma unroll
        for (int e = 0; e < CS_B; ++e) { const int t = min(t0 + e, T - 1); xv[e] = x[(size_t)t * ld + c]; wv[e] = w ? w[(size_t)t * ldw + c] : u; }
#pragma unroll
        for (int e = 0; e < CS_B; ++e) if (t0 + e < T) m = fmaf(wv[e], xv[e], m);
    }
    float v = 0.0f;
    for (int t0 = 0; t0 < T; t0 += CS_B) {
        float xv[CS_B], wv[CS_B];
#pragma unroll
        for (int e = 0; e < CS_B; ++e) { const int t = min(t0 + e, T - 1); xv[e] = x[(size_t)t * ld + c]; wv[e] = w ? w[(size_t)t * ldw + c] : u; }
#pragma unroll
        for (int e = 0; e < CS_B; ++e) if (t0 + e < T) { const float d = xv[e] - m; v = fmaf(wv[e], d * d, v); }
    }
    mean[c] = m; sd[c] = sqrtf(fmaxf(v, 1e-12f));
}
// softmax over time of every channel, in place (max, exp, sequential sum, divide)
__global__ void k_softmax_time(float* e, int ld, int T, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float mx = e[c];
    for (int t0 = 0; t0 < T; t0 += CS_B) {
        float v[CS_B];
#pragma unroll
        for (int k = 0; k < CS_B; ++k) v[k] = e[(size_t)min(t0 + k, T - 1) * ld + c];
#pragma unroll
        for (int k = 0; k < CS_B; ++k) mx = fmaxf(mx, v[k]);
    }
    float l = 0.0f;
    for (int t0 = 0; t0 < T; t0 += CS_B) {
        float p[CS_B];
#pragma unroll
        for (int k = 0; k < CS_B; ++k) p[k] = q3_expf(e[(size_t)min(t0 + k, T - 1) * ld + c] - mx);
#pragma unroll
        for (int k = 0; k < CS_B; ++k) if (t0 + k < T) { e[(size_t)(t0 + k) * ld + c] = p[k]; l += p[k]; }
    }
    for (int t0 = 0; t0 < T; t0 += CS_B) {
        float p[CS_B];
#pragma unroll
        for (int k = 0; k < CS_B; ++k) p[k] = e[(size_t)min(t0 + k, T - 1) * ld + c];
#pragma unroll
        for (int k = 0; k < CS_B; ++k) if (t0 + k < T) e[(size_t)(t0 + k) * ld + c] = p[k] / l;
    }
}
// att_in[t] = [x[t] | mean | std]
__global__ void k_att_in(const float* x, const float* mean, const float* sd, float* out, int T, int C) {
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (size_t)T * C) return;
    const size_t t = gid / C, c = gid % C;
    float* o = out + t * 3 * C;
    o[c] = x[t * C + c]; o[C + c] = mean[c]; o[2 * C + c] = sd[c];
}
// LayerNorm of one row per workgroup (64 threads): the row is staged in LDS, lane 0 runs the two sequential chains
__global__ void k_layernorm(const float* __restrict__ x, int d, const float* __restrict__ w, const float* __restrict__ b, float eps,
                            float* __restrict__ y) {
    extern __shared__ float row[];
    __shared__ float st[2];
    const float* r = x + (size_t)blockIdx.x * d;
    for (int i = threadIdx.x; i < d; i += 64) row[i] = r[i];
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.0f;
        for (int i = 0; i < d; i += 16) {  // d % 16 == 0; the 16 LDS reads are issued together, the chain consumes them in order
            float rv[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) rv[k] = row[i + k];
#pragma unroll
            for (int k = 0; k < 16; ++k) s += rv[k];
        }
        const float mean = s / (float)d;
        float v = 0.0f;
        for (int i = 0; i < d; i += 16) {
            float rv[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) rv[k] = row[i + k];
#pragma unroll
            for (int k = 0; k < 16; ++k) { const float dx = rv[k] - mean; v = fmaf(dx, dx, v); }
        }
        st[0] = mean; st[1] = 1.0f / sqrtf(v / (float)d + eps);
    }
    __syncthreads();
    const float mean = st[0], rinv = st[1];
    for (int i = threadIdx.x; i < d; i += 64) y[(size_t)blockIdx.x * d + i] = ((row[i] - mean) * rinv) * w[i] + b[i];
}
// RoPE on the q and k thirds of qkv [T][3*dq], pairs (i, i + hd/2)
__global__ void k_enc_rope(float* qkv, int T, int nh, int hd, const float* cs, const float* sn) {
    const int half = hd >> 1, dq = nh * hd;
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (size_t)T * 2 * nh * half) return;
    const int i = (int)(gid % half); size_t r = gid / half;
    const int hh = (int)(r % nh); r /= nh;
    const int part = (int)(r % 2); const int t = (int)(r / 2);
    float* v = qkv + (size_t)t * 3 * dq + part * dq + hh * hd;
    const float cc = cs[(size_t)t * half + i], ss = sn[(size_t)t * half + i], a = v[i], bq = v[i + half];
    v[i] = a * cc - bq * ss; v[i + half] = bq * cc + a * ss;
}
// causal sliding-window attention, one wave per (row, head): lanes own keys for the scores (each score an ascending-d fmaf
// chain), the probability sum is sequential over keys, lanes own output dims for the PV chains (ascending key)
__global__ void __launch_bounds__(64) k_enc_attn(const float* __restrict__ qkv, int T, int nh, int hd, int W, float qscale,
                                                 float* __restrict__ att) {
    extern __shared__ float sm[];  // [W] scores, [hd] q
    float* sc = sm; float* qs = sm + W;
    const int t = blockIdx.x / nh, hh = blockIdx.x % nh, lane = threadIdx.x, dq = nh * hd;
    const int j0 = t - W + 1 > 0 ? t - W + 1 : 0, n = t - j0 + 1;
    const float* q = qkv + (size_t)t * 3 * dq + hh * hd;
    for (int i = lane; i < hd; i += 64) qs[i] = q[i];
    __syncthreads();
    float mx = -3.0e38f;
    for (int jj = lane; jj < n; jj += 64) {
        const float* kk = qkv + (size_t)(j0 + jj) * 3 * dq + dq + hh * hd;
        float s = 0.0f;
        for (int i = 0; i < hd; i += 4) {
            const float4 k4 = *(const float4*)(kk + i);
            s = fmaf(qs[i], k4.x, s); s = fmaf(qs[i + 1], k4.y, s); s = fmaf(qs[i + 2], k4.z, s); s = fmaf(qs[i + 3], k4.w, s);
        }
        s = s * qscale; sc[jj] = s; mx = fmaxf(mx, s);
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) mx = fmaxf(mx, __shfl_xor(mx, m));
    __syncthreads();
    for (int jj = lane; jj < n; jj += 64) sc[jj] = q3_expf(sc[jj] - mx);
    __syncthreads();
    float lsum = 0.0f;
    for (int jj = 0; jj < n; ++jj) lsum += sc[jj];
    for (int i = lane; i < hd; i += 64) {
        const float* vv = qkv + (size_t)j0 * 3 * dq + 2 * dq + hh * hd + i;
        float acc = 0.0f;
        for (int jj = 0; jj < n; ++jj) acc = fmaf(sc[jj], vv[(size_t)jj * 3 * dq], acc);
        att[(size_t)t * dq + hh * hd + i] = acc / lsum;
    }
}
// codebook [CS][D] -> [D/4][CS][4]: the codewords of neighbouring lanes are neighbours in memory
__global__ void k_cb_transpose(const float* __restrict__ cb, float* __restrict__ out, int CS, int D) {
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (size_t)CS * D) return;
    const int j = (int)(gid / D), i = (int)(gid % D);
    out[((size_t)(i >> 2) * CS + j) * 4 + (i & 3)] = cb[gid];
}
// split residual VQ of one frame per workgroup: nearest codeword in squared Euclidean distance (ascending-d fmaf chain per
// codeword; ties -> smaller index), codebook 0 on the semantic projection, 1.. on the acoustic residual. 1024 lanes, two
// codewords per lane at a time, 16 dims of both loaded per batch from the transposed codebook (coalesced, 8 loads in flight
// per lane, 16 waves per CU): the launch is bound by load latency, not by the chains.
#define RVQ_T 1024
#define RVQ_CW 2
__global__ void __launch_bounds__(RVQ_T) k_rvq(const float* __restrict__ ps, const float* __restrict__ pa, int D,
                                               const float* const* __restrict__ cbT, int ncb, int CS, long long* __restrict__ codes) {
    extern __shared__ float r[];  // [D]
    __shared__ float rd[RVQ_T]; __shared__ int ri[RVQ_T];
    const int t = blockIdx.x, tid = threadIdx.x;
    for (int q = 0; q < ncb; ++q) {
        if (q < 2) { const float* p = (q == 0 ? ps : pa) + (size_t)t * D; for (int i = tid; i < D; i += RVQ_T) r[i] = p[i]; }
        __syncthreads();
        const float4* book = (const float4*)cbT[q];
        float bd = 3.0e38f; int bi = 0x7fffffff;
        for (int j0 = tid; j0 < CS; j0 += RVQ_T * RVQ_CW) {
            float dist[RVQ_CW]; int jj[RVQ_CW];
#pragma unroll
            for (int u = 0; u < RVQ_CW; ++u) { dist[u] = 0.0f; jj[u] = min(j0 + u * RVQ_T, CS - 1); }
            for (int i4 = 0; i4 < (D >> 2); i4 += 4) {  // D % 16 == 0
                float4 c4[4][RVQ_CW];
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int u = 0; u < RVQ_CW; ++u) c4[g][u] = book[(size_t)(i4 + g) * CS + jj[u]];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 rv = *(const float4*)(r + 4 * (i4 + g));
#pragma unroll
                    for (int u = 0; u < RVQ_CW; ++u) {
                        float e = rv.x - c4[g][u].x; dist[u] = fmaf(e, e, dist[u]);
                        e = rv.y - c4[g][u].y; dist[u] = fmaf(e, e, dist[u]);
                        e = rv.z - c4[g][u].z; dist[u] = fmaf(e, e, dist[u]);
                        e = rv.w - c4[g][u].w; dist[u] = fmaf(e, e, dist[u]);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < RVQ_CW; ++u)
                if (j0 + u * RVQ_T < CS && dist[u] < bd) { bd = dist[u]; bi = j0 + u * RVQ_T; }
        }
        rd[tid] = bd; ri[tid] = bi;
        __syncthreads();
        for (int m = RVQ_T / 2; m >= 1; m >>= 1) {
            if (tid < m) {
                const float od = rd[tid + m]; const int oi = ri[tid + m];
                if (od < rd[tid] || (od == rd[tid] && oi < ri[tid])) { rd[tid] = od; ri[tid] = oi; }
            }
            __syncthreads();
        }
        const int best = ri[0] == 0x7fffffff ? 0 : ri[0];
        if (tid == 0) codes[(size_t)t * ncb + q] = best;
        if (q > 0) for (int i = tid; i < D; i += RVQ_T) r[i] = r[i] - cbT[q][((size_t)(i >> 2) * CS + best) * 4 + (i & 3)];
        __syncthreads();
    }
}

struct CConv { uint4* w = nullptr; float* b = nullptr; int cin = 0, n = 0, k = 0, stride = 1, dil = 1, padl = 0, mode = 0, kp = 0; };
struct Arena {
    char* base = nullptr; size_t cap = 0, top = 0, need = 0; bool dry = true;
    template <typename T> T* get(size_t count) {
        const size_t bytes = (count * sizeof(T) + 255) & ~(size_t)255;
        T* p = dry ? nullptr : (T*)(base + top);
        top += bytes; if (top > need) need = top;
        return p;
    }
};

}  // namespace

struct Q3Clone {
    q3tts_clone_config cfg{};
    std::vector<void*> allocs;
    CConv tdnn0, mfa, asp_t, asp_c, fc;
    struct Blk { CConv t1, t2, se1, se2; std::vector<CConv> r2; } blk[3];
    CConv conv0, last, down, semp, acp;
    struct Stage { CConv ra, rb, dn; } st[4];
    struct Lay { float *ln1w, *ln1b, *ln2w, *ln2b, *ls1, *ls2; CConv qkv, o, fc1, fc2; };
    std::vector<Lay> lay;
    std::vector<float*> cb; const float** cb_dev = nullptr;
    float *cs = nullptr, *sn = nullptr;
    Arena arena;
    float* pcm = nullptr; size_t pcm_cap = 0;
    float* mel_in = nullptr; size_t mel_cap = 0;
};

namespace {

struct Ctx { q3tts_engine* e; Q3Clone* c; hipStream_t s; Arena* A; };
inline int round512(int k) { return (k + 511) / 512 * 512; }
inline unsigned nblk(size_t n) { return (unsigned)((n + 255) / 256); }

int dev_alloc(q3tts_engine* e, Q3Clone* c, void** p, size_t bytes) {
    Q3_HIP(e, hipMalloc(p, bytes));
    c->allocs.push_back(*p);
    return Q3TTS_OK;
}
int mk_conv(q3tts_engine* e, Q3Clone* c, CConv& cv, int comp, int ww, int wb, int cin, int n, int k, int stride, int dil, int padl, int mode) {
    cv.cin = cin; cv.n = n; cv.k = k; cv.stride = stride; cv.dil = dil; cv.padl = padl; cv.mode = mode; cv.kp = round512(k * cin);
    int rc;
    if ((rc = dev_alloc(e, c, (void**)&cv.w, (size_t)n * cv.kp * 2))) return rc;
    Q3Fill f{}; f.dst = cv.w; f.N = n; f.K = cv.kp; f.mode = 0; f.row0 = 0; f.rows = n; f.tid_a = Q3_TID(Q3G_CLONE, comp, ww);
    f.src_a = nullptr; f.src_b = nullptr; f.seed = e->cfg.synth_seed; f.scale = (1.0f / sqrtf((float)(k * cin))) / Q3_IH4_STD;
    q3_launch_fill_tiled(f, e->stream);
    cv.b = nullptr;
    if (wb >= 0) {
        if ((rc = dev_alloc(e, c, (void**)&cv.b, (size_t)n * 4))) return rc;
        q3_launch_fill_f32(cv.b, (size_t)n, e->cfg.synth_seed, Q3_TID(Q3G_CLONE, comp, wb), 0.0f, 0.02f / Q3_IH4_STD, 0, e->stream);
    }
    return Q3TTS_OK;
}
int mk_tdnn(q3tts_engine* e, Q3Clone* c, CConv& cv, int comp, int ww, int cin, int n, int k, int dil) {
    return mk_conv(e, c, cv, comp, ww, ww + 1, cin, n, k, 1, dil, dil * (k - 1) / 2, PAD_REFLECT);
}
int mk_causal(q3tts_engine* e, Q3Clone* c, CConv& cv, int comp, int ww, int wb, int cin, int n, int k, int stride, int dil, int mode) {
    return mk_conv(e, c, cv, comp, ww, wb, cin, n, k, stride, dil, (k - 1) * dil + 1 - stride, mode);
}
int mk_vec(q3tts_engine* e, Q3Clone* c, float** p, int comp, int which, size_t n, float base, float std) {
    int rc = dev_alloc(e, c, (void**)p, n * 4);
    if (rc) return rc;
    q3_launch_fill_f32(*p, n, e->cfg.synth_seed, Q3_TID(Q3G_CLONE, comp, which), base, std / Q3_IH4_STD, 0, e->stream);
    return Q3TTS_OK;
}

// dst[t][0..n) (leading dimension ldd) = conv(src (+ src2)) through im2col + the exact GEMM
void conv_run(Ctx& X, const CConv& c, const float* src, int lds, int T_in, int pre_act, float* dst, int ldd, int T_out,
              const float* src2 = nullptr, int lds2 = 0) {
    const size_t mark = X.A->top;
    const bool direct = c.k == 1 && c.stride == 1 && c.cin == c.kp && lds == c.kp && pre_act == ACT_NONE && !src2 && T_in == T_out;
    float* Am = direct ? nullptr : X.A->get<float>((size_t)T_out * c.kp);
    if (!X.A->dry) {
        if (!direct)
            hipLaunchKernelGGL(k_im2col, dim3(nblk((size_t)T_out * (c.kp >> 2))), dim3(256), 0, X.s, src, src2, lds, lds2, T_in, c.cin,
                               c.k, c.stride, c.dil, c.padl, c.mode, pre_act, Am, c.kp, T_out);
        Q3Gemm g{};
        g.x = direct ? src : Am; g.ldx = c.kp; g.B = T_out; g.w = c.w; g.K = c.kp; g.N = c.n; g.norm_w = nullptr; g.eps = 0.0f;
        g.bias = c.b; g.y = dst; g.ldy = ldd; g.keys = nullptr; g.key_stride = 0; g.epi = Q3_EPI_STORE;
        q3_launch_gemm(g, X.s);
    }
    X.A->top = mark;
}
void act_run(Ctx& X, float* x, int ld, int n, int T, int kind) {
    if (X.A->dry) return;
    hipLaunchKernelGGL(k_act, dim3(nblk((size_t)T * n)), dim3(256), 0, X.s, x, ld, n, T, kind);
}
void combine(Ctx& X, float* dst, int ldd, const float* a, int lda, const float* b, int ldb, const float* s, int n, int T, int mode) {
    if (X.A->dry) return;
    hipLaunchKernelGGL(k_combine, dim3(nblk((size_t)T * n)), dim3(256), 0, X.s, dst, ldd, a, lda, b, ldb, s, n, T, mode);
}
void colstats(Ctx& X, const float* x, int ld, int T, int C, const float* w, int ldw, float* mean, float* sd) {
    if (X.A->dry) return;
    hipLaunchKernelGGL(k_colstats, dim3((C + 63) / 64), dim3(64), 0, X.s, x, ld, T, C, w, ldw, mean, sd);
}

// speaker encoder on a device log-mel [T][mel_dim]; out_dev [se_dim]
void speaker_forward(Ctx& X, const float* mel, int T, float* out_dev) {
    const q3tts_clone_config& c = X.c->cfg;
    const int C = c.se_channels[0], C4 = c.se_channels[4], S = c.se_res2net_scale, wp = C / S, SE = c.se_se_channels, AC = c.se_attn_channels;
    Arena& A = *X.A;
    float* h0 = A.get<float>((size_t)T * C);
    float* cat = A.get<float>((size_t)T * 3 * C);  // the three block outputs side by side (the aggregation layer's input)
    float* y = A.get<float>((size_t)T * C); float* r2 = A.get<float>((size_t)T * C); float* z = A.get<float>((size_t)T * C);
    float* m = A.get<float>(C); float* sdv = A.get<float>(C); float* s1 = A.get<float>(SE); float* s2 = A.get<float>(C);
    conv_run(X, X.c->tdnn0, mel, c.mel_dim, T, ACT_NONE, h0, C, T);
    act_run(X, h0, C, C, T, ACT_RELU);
    for (int i = 1; i <= 3; ++i) {
        const Q3Clone::Blk& b = X.c->blk[i - 1];
        const float* hin = i == 1 ? h0 : cat + (size_t)(i - 2) * C; const int ldin = i == 1 ? C : 3 * C;
        float* hout = cat + (size_t)(i - 1) * C;
        conv_run(X, b.t1, hin, ldin, T, ACT_NONE, y, C, T); act_run(X, y, C, C, T, ACT_RELU);
        if (!A.dry) q3_launch_copy_rows(r2, C, y, C, T, wp, X.s);
        for (int p = 1; p < S; ++p) {
            conv_run(X, b.r2[p - 1], y + p * wp, C, T, ACT_NONE, r2 + p * wp, C, T, p == 1 ? nullptr : r2 + (p - 1) * wp, C);
            act_run(X, r2 + p * wp, C, wp, T, ACT_RELU);
        }
        conv_run(X, b.t2, r2, C, T, ACT_NONE, z, C, T); act_run(X, z, C, C, T, ACT_RELU);
        colstats(X, z, C, T, C, nullptr, 0, m, sdv);
        conv_run(X, b.se1, m, C, 1, ACT_NONE, s1, SE, 1); act_run(X, s1, SE, SE, 1, ACT_RELU);
        conv_run(X, b.se2, s1, SE, 1, ACT_NONE, s2, C, 1); act_run(X, s2, C, C, 1, ACT_SIGMOID);
        combine(X, hout, 3 * C, z, C, hin, ldin, s2, C, T, 1);
    }
    float* x = A.get<float>((size_t)T * C4);
    conv_run(X, X.c->mfa, cat, 3 * C, T, ACT_NONE, x, C4, T); act_run(X, x, C4, C4, T, ACT_RELU);
    float* mean = A.get<float>(C4); float* sd = A.get<float>(C4);
    float* att_in = A.get<float>((size_t)T * 3 * C4); float* a1 = A.get<float>((size_t)T * AC); float* ev = A.get<float>((size_t)T * C4);
    float* pooled = A.get<float>((size_t)2 * C4);
    colstats(X, x, C4, T, C4, nullptr, 0, mean, sd);
    if (!A.dry) hipLaunchKernelGGL(k_att_in, dim3(nblk((size_t)T * C4)), dim3(256), 0, X.s, x, mean, sd, att_in, T, C4);
    conv_run(X, X.c->asp_t, att_in, 3 * C4, T, ACT_NONE, a1, AC, T); act_run(X, a1, AC, AC, T, ACT_RELU_TANH);
    conv_run(X, X.c->asp_c, a1, AC, T, ACT_NONE, ev, C4, T);
    if (!A.dry) hipLaunchKernelGGL(k_softmax_time, dim3((C4 + 63) / 64), dim3(64), 0, X.s, ev, C4, T, C4);
    colstats(X, x, C4, T, C4, ev, C4, pooled, pooled + C4);
    conv_run(X, X.c->fc, pooled, 2 * C4, 1, ACT_NONE, out_dev, c.se_dim, 1);
}

inline int ceil_div(int64_t a, int b) { return (int)((a + b - 1) / b); }
int audio_frames(const q3tts_clone_config& c, int64_t n) {
    if (n < 1) return 0;
    int64_t T = n;
    for (int i = 0; i < c.ae_n_ratios; ++i) T = ceil_div(T, c.ae_ratios[i]);
    return ceil_div(T, c.ae_down_stride);
}
// audio encoder on device PCM [n]; lat_dev [frames][H] (pre-quantiser rows), codes_dev [frames][ncb]
void audio_forward(Ctx& X, const float* pcm, int64_t n, float* lat_dev, long long* codes_dev) {
    const q3tts_clone_config& c = X.c->cfg;
    Arena& A = *X.A;
    int T = (int)n, C = c.ae_filters;
    float* x = A.get<float>((size_t)T * C);
    conv_run(X, X.c->conv0, pcm, 1, T, ACT_NONE, x, C, T);
    for (int i = 0; i < c.ae_n_ratios; ++i) {
        const Q3Clone::Stage& st = X.c->st[i];
        const size_t mark = A.top;
        float* y = A.get<float>((size_t)T * (C / 2)); float* z = A.get<float>((size_t)T * C);
        conv_run(X, st.ra, x, C, T, ACT_ELU, y, C / 2, T);
        conv_run(X, st.rb, y, C / 2, T, ACT_ELU, z, C, T);
        combine(X, x, C, x, C, z, C, nullptr, C, T, 0);
        A.top = mark;
        const int T2 = ceil_div(T, c.ae_ratios[i]);
        float* d = A.get<float>((size_t)T2 * 2 * C);
        conv_run(X, st.dn, x, C, T, ACT_ELU, d, 2 * C, T2);
        x = d; T = T2; C *= 2;
    }
    const int H = c.ae_hidden, nh = c.ae_n_head, hd = c.ae_head_dim, dq = nh * hd, F = c.ae_d_ffn, W = c.ae_window;
    float* h = A.get<float>((size_t)T * H);
    conv_run(X, X.c->last, x, C, T, ACT_ELU, h, H, T);
    float* nrm = A.get<float>((size_t)T * std::max(H, dq)); float* qkv = A.get<float>((size_t)T * 3 * dq); float* att = A.get<float>((size_t)T * dq);
    float* o = A.get<float>((size_t)T * H); float* f1 = A.get<float>((size_t)T * F);
    const float qscale = 1.0f / sqrtf((float)hd);
    for (int l = 0; l < c.ae_n_layer; ++l) {
        const Q3Clone::Lay& L = X.c->lay[l];
        if (!A.dry) hipLaunchKernelGGL(k_layernorm, dim3(T), dim3(64), (size_t)H * 4, X.s, h, H, L.ln1w, L.ln1b, c.ae_ln_eps, nrm);
        conv_run(X, L.qkv, nrm, H, T, ACT_NONE, qkv, 3 * dq, T);
        if (!A.dry) {
            hipLaunchKernelGGL(k_enc_rope, dim3(nblk((size_t)T * nh * hd)), dim3(256), 0, X.s, qkv, T, nh, hd, X.c->cs, X.c->sn);
            hipLaunchKernelGGL(k_enc_attn, dim3(T * nh), dim3(64), (size_t)(W + hd) * 4, X.s, qkv, T, nh, hd, W, qscale, att);
        }
        conv_run(X, L.o, att, dq, T, ACT_NONE, o, H, T);
        combine(X, h, H, h, H, o, H, L.ls1, H, T, 2);
        if (!A.dry) hipLaunchKernelGGL(k_layernorm, dim3(T), dim3(64), (size_t)H * 4, X.s, h, H, L.ln2w, L.ln2b, c.ae_ln_eps, nrm);
        conv_run(X, L.fc1, nrm, H, T, ACT_NONE, f1, F, T); act_run(X, f1, F, F, T, ACT_GELU);
        conv_run(X, L.fc2, f1, F, T, ACT_NONE, o, H, T);
        combine(X, h, H, h, H, o, H, L.ls2, H, T, 2);
    }
    const int Tf = ceil_div(T, c.ae_down_stride), D = c.ae_vq_dim;
    conv_run(X, X.c->down, h, H, T, ACT_NONE, lat_dev, H, Tf);
    float* ps = A.get<float>((size_t)Tf * D); float* pa = A.get<float>((size_t)Tf * D);
    conv_run(X, X.c->semp, lat_dev, H, Tf, ACT_NONE, ps, D, Tf);
    conv_run(X, X.c->acp, lat_dev, H, Tf, ACT_NONE, pa, D, Tf);
    if (!A.dry)
        hipLaunchKernelGGL(k_rvq, dim3(Tf), dim3(RVQ_T), (size_t)D * 4, X.s, ps, pa, D, X.c->cb_dev, c.ae_n_codebooks, c.ae_codebook_size, codes_dev);
}

int validate(q3tts_engine* e, const q3tts_clone_config& c) {
    auto bad = [&](const char* m) { return q3_set_err(e, Q3TTS_ERR_INVALID, std::string("clone config: ") + m); };
    if (c.mel_dim != 128) return bad("mel_dim must be 128 (the log-mel front-end produces 128 bands)");
    for (int i = 0; i < 5; ++i) {
        if (c.se_channels[i] < 16 || c.se_channels[i] % 16) return bad("se_channels must be positive multiples of 16");
        if (c.se_kernels[i] < 1 || c.se_kernels[i] % 2 == 0 || c.se_dilations[i] < 1) return bad("se_kernels must be odd, se_dilations >= 1");
    }
    if (c.se_channels[1] != c.se_channels[0] || c.se_channels[2] != c.se_channels[0] || c.se_channels[3] != c.se_channels[0])
        return bad("the TDNN and the three SE-Res2Net blocks share one width");
    if (c.se_channels[4] != 3 * c.se_channels[0]) return bad("se_channels[4] must be 3 x the block width (aggregation of the three block outputs)");
    if (c.se_res2net_scale < 2 || c.se_channels[0] % c.se_res2net_scale || (c.se_channels[0] / c.se_res2net_scale) % 16)
        return bad("block width / se_res2net_scale must be a multiple of 16");
    if (c.se_attn_channels % 16 || c.se_se_channels % 16 || c.se_dim % 16 || c.se_attn_channels < 16 || c.se_se_channels < 16 || c.se_dim < 16)
        return bad("se_attn_channels, se_se_channels, se_dim must be positive multiples of 16");
    if (round512(3 * c.se_channels[4]) > 8192 || round512(c.se_kernels[0] * c.mel_dim) > 8192) return bad("speaker encoder GEMM depth exceeds 8192");
    if (c.ae_filters < 32 || c.ae_filters % 32) return bad("ae_filters must be a positive multiple of 32");
    if (c.ae_n_ratios < 1 || c.ae_n_ratios > 4) return bad("ae_n_ratios must be 1..4");
    int C = c.ae_filters;
    for (int i = 0; i < c.ae_n_ratios; ++i) {
        if (c.ae_ratios[i] < 1) return bad("ae_ratios must be >= 1");
        if (round512(2 * c.ae_ratios[i] * C) > 8192) return bad("a strided conv's GEMM depth (2 x ratio x channels) exceeds 8192");
        C *= 2;
    }
    if (c.ae_kernel < 1 || c.ae_res_kernel < 1 || c.ae_last_kernel < 1 || round512(c.ae_last_kernel * C) > 8192) return bad("bad conv kernel sizes");
    if (c.ae_hidden % 16 || c.ae_hidden < 16 || c.ae_hidden > 8192 || c.ae_d_ffn % 16 || c.ae_d_ffn < 16 || c.ae_d_ffn > 8192) return bad("ae_hidden / ae_d_ffn must be multiples of 16, <= 8192");
    if (c.ae_n_layer < 0 || c.ae_n_layer > 64 || c.ae_n_head < 1 || c.ae_head_dim < 4 || c.ae_head_dim % 4 || c.ae_head_dim > 256 || (c.ae_n_head * c.ae_head_dim) % 16)
        return bad("bad transformer shape");
    if (c.ae_window < 1 || c.ae_window > 8192) return bad("ae_window must be 1..8192");
    if (c.ae_down_stride < 1 || round512(2 * c.ae_down_stride * c.ae_hidden) > 8192) return bad("bad ae_down_stride");
    if (c.ae_vq_dim % 16 || c.ae_vq_dim < 16 || c.ae_vq_dim > 4096) return bad("ae_vq_dim must be a multiple of 16");
    if (c.ae_n_codebooks < 1 || c.ae_n_codebooks > 64 || c.ae_codebook_size < 1) return bad("bad codebook shape");
    return Q3TTS_OK;
}

int ensure_arena(q3tts_engine* e, Q3Clone* c) {
    Arena& A = c->arena;
    if (A.need > A.cap) {
        if (A.base) { Q3_HIP(e, hipStreamSynchronize(e->stream)); hipFree(A.base); A.base = nullptr; A.cap = 0; }
        Q3_HIP(e, hipMalloc((void**)&A.base, A.need));
        A.cap = A.need;
    }
    return Q3TTS_OK;
}
int not_loaded(q3tts_engine* e, const char* which) {
    return q3_set_err(e, Q3TTS_ERR_STATE, std::string(which) + " not loaded (q3tts_clone_init was not called)");
}

}  // namespace

void q3_clone_destroy(q3tts_engine* e) {
    Q3Clone* c = e->clone;
    if (!c) return;
    for (void* p : c->allocs) hipFree(p);
    hipFree(c->arena.base); hipFree(c->pcm); hipFree(c->mel_in);
    delete c;
    e->clone = nullptr;
}

extern "C" void q3tts_clone_default_config(q3tts_clone_config* c) {
    if (!c) return;
    memset(c, 0, sizeof(*c));
    c->mel_dim = 128;
    const int ch[5] = {512, 512, 512, 512, 1536}, ks[5] = {5, 3, 3, 3, 1}, dl[5] = {1, 2, 3, 4, 1};
    for (int i = 0; i < 5; ++i) { c->se_channels[i] = ch[i]; c->se_kernels[i] = ks[i]; c->se_dilations[i] = dl[i]; }
    c->se_attn_channels = 128; c->se_res2net_scale = 8; c->se_se_channels = 128; c->se_dim = 2048;
    c->ae_filters = 64; c->ae_kernel = 7; c->ae_res_kernel = 3; c->ae_last_kernel = 3;
    c->ae_n_ratios = 4; c->ae_ratios[0] = 4; c->ae_ratios[1] = 5; c->ae_ratios[2] = 6; c->ae_ratios[3] = 8;
    c->ae_hidden = 512; c->ae_n_layer = 8; c->ae_n_head = 8; c->ae_head_dim = 64; c->ae_d_ffn = 2048; c->ae_window = 250;
    c->ae_rope_theta = 10000.0f; c->ae_ln_eps = 1e-5f; c->ae_layer_scale = 0.01f;
    c->ae_down_stride = 2; c->ae_vq_dim = 256; c->ae_n_codebooks = 16; c->ae_codebook_size = 2048;
}

extern "C" int q3tts_clone_init(q3tts_engine* e, const q3tts_clone_config* cfg) {
    if (!e || !cfg) return q3_set_err(e, Q3TTS_ERR_INVALID, "null argument");
    int rc = validate(e, *cfg);
    if (rc) return rc;
    Q3_HIP(e, hipSetDevice(e->cfg.device));
    q3_clone_destroy(e);
    Q3Clone* c = new Q3Clone();
    e->clone = c;
    c->cfg = *cfg;
    const q3tts_clone_config& g = c->cfg;
#define CK(x) do { if ((rc = (x))) { q3_clone_destroy(e); return rc; } } while (0)
    const int C = g.se_channels[0], C4 = g.se_channels[4], wp = C / g.se_res2net_scale;
    CK(mk_tdnn(e, c, c->tdnn0, SC_TDNN0, 0, g.mel_dim, C, g.se_kernels[0], g.se_dilations[0]));
    for (int i = 1; i <= 3; ++i) {
        Q3Clone::Blk& b = c->blk[i - 1]; const int comp = SC_BLOCK + i - 1;
        CK(mk_tdnn(e, c, b.t1, comp, SW_TDNN1, C, C, 1, 1));
        b.r2.resize(g.se_res2net_scale - 1);
        for (int p = 1; p < g.se_res2net_scale; ++p) CK(mk_tdnn(e, c, b.r2[p - 1], comp, SW_RES2 + 2 * p, wp, wp, g.se_kernels[i], g.se_dilations[i]));
        CK(mk_tdnn(e, c, b.t2, comp, SW_TDNN2, C, C, 1, 1));
        CK(mk_tdnn(e, c, b.se1, comp, SW_SE1, C, g.se_se_channels, 1, 1));
        CK(mk_tdnn(e, c, b.se2, comp, SW_SE2, g.se_se_channels, C, 1, 1));
    }
    CK(mk_tdnn(e, c, c->mfa, SC_MFA, 0, 3 * C, C4, g.se_kernels[4], g.se_dilations[4]));
    CK(mk_tdnn(e, c, c->asp_t, SC_ASP_TDNN, 0, 3 * C4, g.se_attn_channels, 1, 1));
    CK(mk_tdnn(e, c, c->asp_c, SC_ASP_CONV, 0, g.se_attn_channels, C4, 1, 1));
    CK(mk_tdnn(e, c, c->fc, SC_FC, 0, 2 * C4, g.se_dim, 1, 1));
    int Ca = g.ae_filters;
    CK(mk_causal(e, c, c->conv0, AC_CONV0, 0, 1, 1, Ca, g.ae_kernel, 1, 1, PAD_ZERO));
    for (int i = 0; i < g.ae_n_ratios; ++i) {
        const int comp = AC_STAGE + 4 * i, r = g.ae_ratios[i];
        CK(mk_causal(e, c, c->st[i].ra, comp, 0, 1, Ca, Ca / 2, g.ae_res_kernel, 1, 1, PAD_ZERO));
        CK(mk_causal(e, c, c->st[i].rb, comp + 1, 0, 1, Ca / 2, Ca, 1, 1, 1, PAD_ZERO));
        CK(mk_causal(e, c, c->st[i].dn, comp + 2, 0, 1, Ca, 2 * Ca, 2 * r, r, 1, PAD_ZERO));
        Ca *= 2;
    }
    const int H = g.ae_hidden, dq = g.ae_n_head * g.ae_head_dim;
    CK(mk_causal(e, c, c->last, AC_LAST, 0, 1, Ca, H, g.ae_last_kernel, 1, 1, PAD_ZERO));
    c->lay.resize(g.ae_n_layer);
    for (int l = 0; l < g.ae_n_layer; ++l) {
        Q3Clone::Lay& L = c->lay[l]; const int comp = AC_TFM + l;
        CK(mk_vec(e, c, &L.ln1w, comp, TW_LN1_W, H, 1.0f, 0.05f)); CK(mk_vec(e, c, &L.ln1b, comp, TW_LN1_B, H, 0.0f, 0.02f));
        CK(mk_vec(e, c, &L.ln2w, comp, TW_LN2_W, H, 1.0f, 0.05f)); CK(mk_vec(e, c, &L.ln2b, comp, TW_LN2_B, H, 0.0f, 0.02f));
        CK(mk_vec(e, c, &L.ls1, comp, TW_LS1, H, g.ae_layer_scale, 0.1f * g.ae_layer_scale));
        CK(mk_vec(e, c, &L.ls2, comp, TW_LS2, H, g.ae_layer_scale, 0.1f * g.ae_layer_scale));
        CK(mk_conv(e, c, L.qkv, comp, TW_QKV, -1, H, 3 * dq, 1, 1, 1, 0, PAD_ZERO));
        CK(mk_conv(e, c, L.o, comp, TW_O, -1, dq, H, 1, 1, 1, 0, PAD_ZERO));
        CK(mk_conv(e, c, L.fc1, comp, TW_FC1, -1, H, g.ae_d_ffn, 1, 1, 1, 0, PAD_ZERO));
        CK(mk_conv(e, c, L.fc2, comp, TW_FC2, -1, g.ae_d_ffn, H, 1, 1, 1, 0, PAD_ZERO));
    }
    CK(mk_causal(e, c, c->down, AC_DOWN, 0, -1, H, H, 2 * g.ae_down_stride, g.ae_down_stride, 1, PAD_REPLICATE));
    CK(mk_conv(e, c, c->semp, AC_SEM_PROJ, 0, -1, H, g.ae_vq_dim, 1, 1, 1, 0, PAD_ZERO));
    CK(mk_conv(e, c, c->acp, AC_AC_PROJ, 0, -1, H, g.ae_vq_dim, 1, 1, 1, 0, PAD_ZERO));
    c->cb.resize(g.ae_n_codebooks);
    for (int q = 0; q < g.ae_n_codebooks; ++q) {  // generated row-major [CS][D] (the logical tensor), kept transposed for k_rvq
        const size_t ne = (size_t)g.ae_codebook_size * g.ae_vq_dim;
        float* rowmajor = nullptr;
        Q3_HIP(e, hipMalloc((void**)&rowmajor, ne * 4));
        q3_launch_fill_f32(rowmajor, ne, e->cfg.synth_seed, Q3_TID(Q3G_CLONE, AC_CODEBOOK + q, 0), 0.0f, (1.0f / sqrtf((float)g.ae_vq_dim)) / Q3_IH4_STD, 0, e->stream);
        rc = dev_alloc(e, c, (void**)&c->cb[q], ne * 4);
        if (!rc) hipLaunchKernelGGL(k_cb_transpose, dim3(nblk(ne)), dim3(256), 0, e->stream, rowmajor, c->cb[q], g.ae_codebook_size, g.ae_vq_dim);
        hipStreamSynchronize(e->stream);
        hipFree(rowmajor);
        CK(rc);
    }
    CK(dev_alloc(e, c, (void**)&c->cb_dev, sizeof(float*) * g.ae_n_codebooks));
    Q3_HIP(e, hipMemcpyAsync((void*)c->cb_dev, c->cb.data(), sizeof(float*) * g.ae_n_codebooks, hipMemcpyHostToDevice, e->stream));
    // RoPE tables (evaluated in double on the host, like the decoder's)
    const int half = g.ae_head_dim / 2;
    std::vector<float> cs((size_t)ROPE_MAX_T * half), sn((size_t)ROPE_MAX_T * half);
    for (int t = 0; t < ROPE_MAX_T; ++t)
        for (int i = 0; i < half; ++i) {
            const double a = (double)t * pow((double)g.ae_rope_theta, -2.0 * (double)i / (double)g.ae_head_dim);
            cs[(size_t)t * half + i] = (float)cos(a); sn[(size_t)t * half + i] = (float)sin(a);
        }
    CK(dev_alloc(e, c, (void**)&c->cs, cs.size() * 4)); CK(dev_alloc(e, c, (void**)&c->sn, sn.size() * 4));
    Q3_HIP(e, hipMemcpyAsync(c->cs, cs.data(), cs.size() * 4, hipMemcpyHostToDevice, e->stream));
    Q3_HIP(e, hipMemcpyAsync(c->sn, sn.data(), sn.size() * 4, hipMemcpyHostToDevice, e->stream));
    Q3_HIP(e, hipStreamSynchronize(e->stream));
    Q3_HIP(e, hipGetLastError());
#undef CK
    return Q3TTS_OK;
}

extern "C" int32_t q3tts_clone_audio_frames(const q3tts_engine* e, int64_t n_samples) {
    if (!e || !e->clone) return 0;
    return audio_frames(e->clone->cfg, n_samples);
}

namespace {
// runs `fwd` twice: once dry to size the arena, once for real
template <typename F> int run_sized(q3tts_engine* e, Q3Clone* c, F&& fwd) {
    Arena& A = c->arena;
    A.dry = true; A.top = 0; A.need = 0;
    { Ctx X{e, c, e->stream, &A}; fwd(X); }
    int rc = ensure_arena(e, c);
    if (rc) return rc;
    A.dry = false; A.top = 0;
    { Ctx X{e, c, e->stream, &A}; fwd(X); }
    Q3_HIP(e, hipGetLastError());
    return Q3TTS_OK;
}
int speaker_run(q3tts_engine* e, const float* mel_dev, int T, float* out_host) {
    Q3Clone* c = e->clone;
    float* out_dev = nullptr;
    int rc = run_sized(e, c, [&](Ctx& X) { out_dev = X.A->get<float>(c->cfg.se_dim); speaker_forward(X, mel_dev, T, out_dev); });
    if (rc) return rc;
    Q3_HIP(e, hipMemcpyAsync(out_host, out_dev, (size_t)c->cfg.se_dim * 4, hipMemcpyDeviceToHost, e->stream));
    Q3_HIP(e, hipStreamSynchronize(e->stream));
    return Q3TTS_OK;
}
int audio_run(q3tts_engine* e, const float* audio, int64_t n, int64_t* codes, float* latent, int32_t cap, int32_t* n_frames) {
    Q3Clone* c = e->clone;
    if (!c) return not_loaded(e, "AudioEncoder");
    if (!n_frames || n < 0 || (n > 0 && !audio)) return q3_set_err(e, Q3TTS_ERR_INVALID, "null argument");
    if (n > (int64_t)1 << 26) return q3_set_err(e, Q3TTS_ERR_INVALID, "reference clip longer than 2^26 samples");
    Q3_HIP(e, hipSetDevice(e->cfg.device));
    const int nf = audio_frames(c->cfg, n);
    *n_frames = nf;
    if (nf == 0) return Q3TTS_OK;
    if (nf > cap) return q3_set_err(e, Q3TTS_ERR_INVALID, "audio code buffer too small");
    int T25 = (int)n;
    for (int i = 0; i < c->cfg.ae_n_ratios; ++i) T25 = ceil_div(T25, c->cfg.ae_ratios[i]);
    if (T25 > ROPE_MAX_T) return q3_set_err(e, Q3TTS_ERR_INVALID, "reference clip too long for the encoder's position table");
    if (c->pcm_cap < (size_t)n) { hipFree(c->pcm); c->pcm = nullptr; c->pcm_cap = 0; Q3_HIP(e, hipMalloc((void**)&c->pcm, (size_t)n * 4)); c->pcm_cap = (size_t)n; }
    Q3_HIP(e, hipMemcpyAsync(c->pcm, audio, (size_t)n * 4, hipMemcpyHostToDevice, e->stream));
    float* lat = nullptr; long long* cd = nullptr;
    int rc = run_sized(e, c, [&](Ctx& X) {
        lat = X.A->get<float>((size_t)nf * c->cfg.ae_hidden); cd = X.A->get<long long>((size_t)nf * c->cfg.ae_n_codebooks);
        audio_forward(X, c->pcm, n, lat, cd);
    });
    if (rc) return rc;
    if (codes) Q3_HIP(e, hipMemcpyAsync(codes, cd, (size_t)nf * c->cfg.ae_n_codebooks * 8, hipMemcpyDeviceToHost, e->stream));
    if (latent) Q3_HIP(e, hipMemcpyAsync(latent, lat, (size_t)nf * c->cfg.ae_hidden * 4, hipMemcpyDeviceToHost, e->stream));
    Q3_HIP(e, hipStreamSynchronize(e->stream));
    return Q3TTS_OK;
}
}  // namespace

extern "C" int q3tts_clone_audio_encode(q3tts_engine* e, const float* audio, int64_t n_samples, int64_t* codes, int32_t cap_frames,
                                        int32_t* n_frames) {
    if (!e) return Q3TTS_ERR_INVALID;
    if (!codes) return q3_set_err(e, Q3TTS_ERR_INVALID, "null argument");
    return audio_run(e, audio, n_samples, codes, nullptr, cap_frames, n_frames);
}
extern "C" int q3tts_k_audio_latent(q3tts_engine* e, const float* audio, int64_t n_samples, float* latent, int32_t cap_frames,
                                    int32_t* n_frames) {
    if (!e) return Q3TTS_ERR_INVALID;
    if (!latent) return q3_set_err(e, Q3TTS_ERR_INVALID, "null argument");
    return audio_run(e, audio, n_samples, nullptr, latent, cap_frames, n_frames);
}

extern "C" int q3tts_clone_speaker_encode(q3tts_engine* e, const float* audio, int64_t n_samples, float* spk_emb) {
    if (!e) return Q3TTS_ERR_INVALID;
    if (!e->clone) return not_loaded(e, "SpeakerEncoder");
    if (!spk_emb || n_samples < 0 || (n_samples > 0 && !audio)) return q3_set_err(e, Q3TTS_ERR_INVALID, "null argument");
    Q3_HIP(e, hipSetDevice(e->cfg.device));
    int32_t nf = 0; float* mel_dev = nullptr;
    int rc = q3_mel_run(e, audio, n_samples, &nf, &mel_dev);
    if (rc) return rc;
    if (nf < 1) return q3_set_err(e, Q3TTS_ERR_INVALID, "reference clip shorter than one mel frame (256 samples)");
    return speaker_run(e, mel_dev, nf, spk_emb);
}
extern "C" int q3tts_k_speaker_from_mel(q3tts_engine* e, const float* mel, int32_t n_frames, float* spk_emb) {
    if (!e) return Q3TTS_ERR_INVALID;
    Q3Clone* c = e->clone;
    if (!c) return not_loaded(e, "SpeakerEncoder");
    if (!mel || !spk_emb || n_frames < 1) return q3_set_err(e, Q3TTS_ERR_INVALID, "null argument");
    Q3_HIP(e, hipSetDevice(e->cfg.device));
    const size_t nb = (size_t)n_frames * c->cfg.mel_dim;
    if (c->mel_cap < nb) { hipFree(c->mel_in); c->mel_in = nullptr; c->mel_cap = 0; Q3_HIP(e, hipMalloc((void**)&c->mel_in, nb * 4)); c->mel_cap = nb; }
    Q3_HIP(e, hipMemcpyAsync(c->mel_in, mel, nb * 4, hipMemcpyHostToDevice, e->stream));
    return speaker_run(e, c->mel_in, n_frames, spk_emb);
}
