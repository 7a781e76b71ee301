/*
 * q3_oracle.c — CPU restatement of the Qwen3-TTS autoregressive codec-token decoder
 * (prompt builder, Talker, Predictor, sampler, feedback loop, chunker).
 *
 * TEST INFRASTRUCTURE ONLY (see q3_oracle.h). PARITY UNPINNED against the reference
 * binaries (llama.cpp b7885 / onnxruntime 1.23.2 are not in /root/reference and the
 * reference holds no tests or golden vectors); host logic follows the cited lines of
 * /root/reference exactly, transformer math follows the standard Qwen3 decoder in the
 * canonical order of DESIGN.md §4 (bf16 GEMM operands on the restated MFMA: q3_oracle_bf16.c).
 *
 * Build: see oracle/Makefile (gcc -O3 -mavx2 -mfma -ffp-contract=off -fopenmp).
 * -ffp-contract=off matters: every fused multiply-add in the spec is an explicit fmaf().
 */
#include "q3_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------ */
/* spec primitives                                                                             */
/* ------------------------------------------------------------------------------------------ */

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* exp with a fixed operation sequence: identical results on x86 and gfx950 */
float q3o_expf(float x) {
    if (x < -87.0f) return 0.0f;
    float n = rintf(x * 1.44269504088896341f);
    float r = fmaf(n, -0.693145751953125f, x);
    r = fmaf(n, -1.42860682030941723212e-6f, r);
    float p = 1.0f / 720.0f;
    p = fmaf(p, r, 1.0f / 120.0f);
    p = fmaf(p, r, 1.0f / 24.0f);
    p = fmaf(p, r, 1.0f / 6.0f);
    p = fmaf(p, r, 0.5f);
    p = fmaf(p, r, 1.0f);
    p = fmaf(p, r, 1.0f);
    int e = (int)n + 127;
    return p * u2f((uint32_t)e << 23);
}

uint16_t q3o_bf16(float x) { /* round to nearest even; inputs are finite */
    uint32_t u = f2u(x);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static inline float bf16f(uint16_t h) { return u2f((uint32_t)h << 16); }
static inline float round_bf16(float x) { return bf16f(q3o_bf16(x)); }

static inline uint64_t mix64(uint64_t z) {
    z ^= z >> 30; z *= 0xbf58476d1ce4e5b9ULL;
    z ^= z >> 27; z *= 0x94d049bb133111ebULL;
    z ^= z >> 31; return z;
}
/* counter-based synthetic value: Irwin-Hall(4) of 16-bit uniforms, exact integer -> one f32 multiply */
float q3o_synth(uint64_t seed, uint32_t tensor, uint64_t idx, float scale) {
    uint64_t h = mix64(mix64(seed ^ ((uint64_t)tensor * 0x9E3779B97F4A7C15ULL)) + idx);
    int32_t s = (int32_t)(h & 0xffff) + (int32_t)((h >> 16) & 0xffff) + (int32_t)((h >> 32) & 0xffff) +
                (int32_t)(h >> 48) - 131070;
    return (float)s * scale;
}
#define IH4_STD 37837.227f
#define TID(g, l, w) (((uint32_t)(g) << 16) | ((uint32_t)(l) << 8) | (uint32_t)(w))
enum { G_TALKER = 1, G_PRED = 2, G_ASSET = 3, G_VOC = 4 };
enum { W_ATTN_NORM = 0, W_Q, W_K, W_V, W_QNORM, W_KNORM, W_O, W_FFN_NORM, W_GATE, W_UP, W_DOWN };
enum { WM_OUT_NORM = 0, WM_HEAD = 1 };
enum { WA_TEXT = 0, WA_PROJ_W = 1, WA_PROJ_B = 2 };
#define L_MODEL 255

/* out[i] = base + synth(seed, tensor, i, std / IH4_STD), optionally rounded to bf16 — the generator behind every synthetic
 * tensor, exported so that tests can write the same model into GGUF / NPY files (loader parity) */
void q3o_synth_fill(uint64_t seed, uint32_t tensor, uint64_t n, float base, float std, int32_t round_to_bf16, float* out) {
    const float scale = std / IH4_STD;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)n; ++i) {
        float v = base + q3o_synth(seed, tensor, (uint64_t)i, scale);
        if (round_to_bf16) { uint32_t u = (uint32_t)q3o_bf16(v) << 16; memcpy(&v, &u, 4); }
        out[i] = v;
    }
}

static float butterfly64(float* v) {
    float t[64];
    for (int m = 32; m >= 1; m >>= 1) {
        for (int j = 0; j < 64; ++j) t[j] = v[j] + v[j ^ m];
        memcpy(v, t, sizeof(t));
    }
    return v[0];
}

/* canonical sum of squares + rinv */
static float rms_rinv(const float* x, int d, float eps) {
    float part[64];
    for (int j = 0; j < 64; ++j) part[j] = 0.0f;
    for (int c = 0; c < d / 4; ++c) {
        int j = c & 63;
        for (int e = 0; e < 4; ++e) part[j] = fmaf(x[4 * c + e], x[4 * c + e], part[j]);
    }
    float ss = butterfly64(part);
    float mean = ss / (float)d;
    return 1.0f / sqrtf(mean + eps);
}
void q3o_rmsnorm(const float* x, int32_t d, const float* w, float eps, float* y) {
    float rinv = rms_rinv(x, d, eps);
    for (int i = 0; i < d; ++i) y[i] = (x[i] * rinv) * w[i];
}

/* ------------------------------------------------------------------------------------------ */
/* exact GEMM, canonical order                                                */
/*   16 contiguous K-slices (K % 512 == 0). Inside a slice: for 32-wide block kb, for t in 0..7, for kq in 0..3:  */
/*   k = kb*32 + (t/4)*16 + kq*4 + (t%4), acc = fmaf(x[k], w[k], acc) from +0.                                     */
/*   Slice partials p0..p15 combine as q_w = p_{2w} + p_{2w+1}, y = ((((((q0+q1)+q2)+q3)+q4)+q5)+q6)+q7.           */
/*   (This is the order v_mfma_f32_16x16x4_f32 produces when lane group kq holds two runs of four consecutive k.)  */
/* ------------------------------------------------------------------------------------------ */
#define NB 32
#define RB 8
#define NSLICE 16
static int g_threads = 0;

/* Wt: transposed weights [K][ldw] bf16; computes raw sums for columns [col0, col0+ncols) */
static void gemm_t(const float* xh, int B, int ldx, int K, const uint16_t* Wt, int ldw, int col0, int ncols, float* y,
                   int ldy) {
    const int bps = K / (32 * NSLICE); /* 32-blocks per slice */
    const int nblk = (ncols + NB - 1) / NB;
#pragma omp parallel for schedule(dynamic, 1) num_threads(g_threads > 0 ? g_threads : 1)
    for (int blk = 0; blk < nblk; ++blk) {
        const int n0 = blk * NB;
        const int nn = (ncols - n0 < NB) ? ncols - n0 : NB;
        for (int r0 = 0; r0 < B; r0 += RB) {
            const int rr = (B - r0 < RB) ? B - r0 : RB;
            float tot[RB][NB], pair[RB][NB];
            for (int s = 0; s < NSLICE; ++s) {
                float acc[RB][NB];
                for (int r = 0; r < RB; ++r)
                    for (int n = 0; n < NB; ++n) acc[r][n] = 0.0f;
                for (int kb = s * bps; kb < (s + 1) * bps; ++kb)
                    for (int t = 0; t < 8; ++t)
                        for (int kq = 0; kq < 4; ++kq) {
                            const int k = kb * 32 + (t >> 2) * 16 + kq * 4 + (t & 3);
                            const uint16_t* wr = Wt + (size_t)k * ldw + col0 + n0;
                            float wf[NB];
                            for (int n = 0; n < NB; ++n) wf[n] = (n < nn) ? bf16f(wr[n]) : 0.0f;
                            for (int r = 0; r < rr; ++r) {
                                const float xk = xh[(size_t)(r0 + r) * ldx + k];
                                for (int n = 0; n < NB; ++n) acc[r][n] = __builtin_fmaf(xk, wf[n], acc[r][n]);
                            }
                        }
                for (int r = 0; r < rr; ++r)
                    for (int n = 0; n < NB; ++n) {
                        if ((s & 1) == 0) pair[r][n] = acc[r][n];
                        else {
                            const float q = pair[r][n] + acc[r][n];
                            tot[r][n] = (s == 1) ? q : tot[r][n] + q;
                        }
                    }
            }
            for (int r = 0; r < rr; ++r)
                for (int n = 0; n < nn; ++n) y[(size_t)(r0 + r) * ldy + n0 + n] = tot[r][n];
        }
    }
}

/* Fused RMSNorm of a NORM GEMM (q3_gemm.hip): the row scale commutes out of the K-sum,
 *   y[r][n] = s_r * SUM_canonical((x[r][k] * nw[k]) * W[n][k]),  s_r = 1 / sqrtf(ss_r / K + eps),
 * with ss_r summed in the GEMM's own order: per (slice, kq) an fmaf chain over ascending k,
 * S_slice = (c0 + c1) + (c2 + c3), Q_w = S_2w + S_2w+1, ss = ((Q_0 + Q_1) + ...) + Q_7.
 * Writes xh = x * nw and the row scale. */
static float norm_gemm_row(const float* x, int K, const float* nw, float eps, float* xh) {
    const int bps = K / (32 * NSLICE);
    float tot = 0.0f, pair = 0.0f;
    for (int s = 0; s < NSLICE; ++s) {
        float c[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int kb = s * bps; kb < (s + 1) * bps; ++kb)
            for (int h = 0; h < 2; ++h)
                for (int kq = 0; kq < 4; ++kq)
                    for (int e = 0; e < 4; ++e) {
                        const int k = kb * 32 + h * 16 + kq * 4 + e;
                        c[kq] = __builtin_fmaf(x[k], x[k], c[kq]);
                    }
        const float S = (c[0] + c[1]) + (c[2] + c[3]);
        if ((s & 1) == 0) pair = S;
        else {
            const float q = pair + S;
            tot = (s == 1) ? q : tot + q;
        }
    }
    for (int k = 0; k < K; ++k) xh[k] = x[k] * nw[k];
    return 1.0f / sqrtf(tot / (float)K + eps);
}
static void scale_rows(float* y, int n, int ncols, int ldy, const float* sc) {
    for (int r = 0; r < n; ++r)
        for (int c = 0; c < ncols; ++c) y[(size_t)r * ldy + c] = y[(size_t)r * ldy + c] * sc[r];
}

static inline uint64_t argmax_key(float v, uint32_t n) {
    if (v != v) return 0; /* NaN never wins (reference: `val > max_val` is false) */
    if (v == 0.0f) v = 0.0f;
    uint32_t u = f2u(v);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ((uint64_t)u << 32) | (uint64_t)(0xFFFFFFFFu - n);
}
static inline int32_t argmax_idx(uint64_t key) { return key == 0 ? 0 : (int32_t)(0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFu)); }

static inline float swiglu(float g, float u) { return (g / (1.0f + q3o_expf(-g))) * u; }

static uint16_t* transpose_bf16(const uint16_t* w, int N, int K) {
    uint16_t* t = (uint16_t*)malloc((size_t)N * K * 2);
    for (int n = 0; n < N; ++n)
        for (int k = 0; k < K; ++k) t[(size_t)k * N + n] = w[(size_t)n * K + k];
    return t;
}

void q3o_gemm_exact(const float* x, int32_t B, int32_t K, const uint16_t* w, int32_t N, const float* norm_w, float eps,
                    const float* bias, int32_t epilogue, float* y, uint64_t* keys) {
    uint16_t* wt = transpose_bf16(w, N, K);
    float* xh = (float*)malloc((size_t)B * K * 4);
    float* sc = (float*)malloc((size_t)B * 4);
    for (int b = 0; b < B; ++b) {
        if (norm_w) sc[b] = norm_gemm_row(x + (size_t)b * K, K, norm_w, eps, xh + (size_t)b * K);
        else memcpy(xh + (size_t)b * K, x + (size_t)b * K, (size_t)K * 4);
    }
    float* raw = (float*)malloc((size_t)B * N * 4);
    int saved = g_threads;
    if (g_threads <= 0) g_threads = 4;
    gemm_t(xh, B, K, K, wt, N, 0, N, raw, N);
    g_threads = saved;
    if (norm_w) scale_rows(raw, B, N, N, sc);
    free(sc);
    if (epilogue == 0) {
        for (int b = 0; b < B; ++b)
            for (int n = 0; n < N; ++n) y[(size_t)b * N + n] = bias ? raw[(size_t)b * N + n] + bias[n] : raw[(size_t)b * N + n];
    } else if (epilogue == 1) {
        for (size_t i = 0; i < (size_t)B * N; ++i) y[i] = y[i] + raw[i];
    } else if (epilogue == 2) {
        int F = N / 2;
        for (int b = 0; b < B; ++b)
            for (int j = 0; j < F; ++j) y[(size_t)b * F + j] = swiglu(raw[(size_t)b * N + j], raw[(size_t)b * N + F + j]);
    } else {
        for (int b = 0; b < B; ++b) {
            uint64_t best = 0;
            for (int n = 0; n < N; ++n) {
                uint64_t kk = argmax_key(raw[(size_t)b * N + n], (uint32_t)n);
                if (kk > best) best = kk;
                y[(size_t)b * N + n] = raw[(size_t)b * N + n];
            }
            keys[b] = best;
        }
    }
    free(raw); free(xh); free(wt);
}

/* ------------------------------------------------------------------------------------------ */
/* RoPE tables and attention                                            */
/* ------------------------------------------------------------------------------------------ */
/* M-RoPE with t = h = w = pos (the only layout the reference produces: src/tts/engine.rs:306-314);
 * pairs in section 3 ("channel", always position 0) get angle 0. NeoX pairing (i, i + hd/2). */
static void rope_tables(int n_pos, int hd, float theta, const int* sections, float* cs, float* sn) {
    const int half = hd / 2;
    int s3_begin = half;
    if (sections) s3_begin = sections[0] + sections[1] + sections[2];
    for (int p = 0; p < n_pos; ++p)
        for (int i = 0; i < half; ++i) {
            double inv = pow((double)theta, -2.0 * (double)i / (double)hd);
            double ang = (i < s3_begin) ? (double)p * inv : 0.0;
            cs[(size_t)p * half + i] = (float)cos(ang);
            sn[(size_t)p * half + i] = (float)sin(ang);
        }
}
static void rope_apply(float* x, int hd, const float* cs, const float* sn) {
    const int half = hd / 2;
    for (int i = 0; i < half; ++i) {
        float a = x[i], b = x[i + half], c = cs[i], s = sn[i];
        x[i] = fmaf(-b, s, a * c);
        x[i + half] = fmaf(a, s, b * c);
    }
}

/* one query head against T cached keys (bf16-rounded floats) */
static void attend_one(const float* qh, const float* Kc, const float* Vc, int T, int hd, float* out) {
    float* sc = (float*)malloc((size_t)T * 4);
    const float scale = 1.0f / sqrtf((float)hd);
    float m = -INFINITY;
    for (int t = 0; t < T; ++t) {
        float s = 0.0f;
        const float* kr = Kc + (size_t)t * hd;
        for (int d = 0; d < hd; ++d) s = fmaf(qh[d], kr[d], s);
        s = s * scale;
        sc[t] = s;
        if (s > m) m = s;
    }
    float part[256];
    for (int j = 0; j < 256; ++j) part[j] = 0.0f;
    for (int t = 0; t < T; ++t) {
        sc[t] = q3o_expf(sc[t] - m);
        part[t & 255] += sc[t];
    }
    float lw[4];
    for (int w = 0; w < 4; ++w) lw[w] = butterfly64(part + 64 * w);
    const float l = ((lw[0] + lw[1]) + lw[2]) + lw[3];
    float* ou = (float*)calloc((size_t)16 * hd, 4);
    for (int t = 0; t < T; ++t) {
        float* o = ou + (size_t)(t & 15) * hd;
        const float* vr = Vc + (size_t)t * hd;
        for (int d = 0; d < hd; ++d) o[d] = fmaf(sc[t], vr[d], o[d]);
    }
    for (int d = 0; d < hd; ++d) {
        float r[4];
        for (int w = 0; w < 4; ++w)
            r[w] = (ou[(size_t)(4 * w) * hd + d] + ou[(size_t)(4 * w + 1) * hd + d]) +
                   (ou[(size_t)(4 * w + 2) * hd + d] + ou[(size_t)(4 * w + 3) * hd + d]);
        float o = ((r[0] + r[1]) + r[2]) + r[3];
        out[d] = o / l;
    }
    free(ou); free(sc);
}

/* rows of one sequence in order: q/k RMSNorm -> RoPE -> append bf16 K/V -> attention */
static void attn_rows(const float* qkv, int n_rows, int pos0, int Hq, int Hkv, int hd, const float* qnw, const float* knw,
                      float eps, const float* cs, const float* sn, float* kc, float* vc, int n_ctx, float* out) {
    const int ld = (Hq + 2 * Hkv) * hd, R = Hq / Hkv, half = hd / 2;
    float* tmp = (float*)malloc((size_t)hd * 4);
    for (int r = 0; r < n_rows; ++r) {
        const int pos = pos0 + r;
        const float* row = qkv + (size_t)r * ld;
        for (int g = 0; g < Hkv; ++g) {
            q3o_rmsnorm(row + (size_t)(Hq + g) * hd, hd, knw, eps, tmp);
            rope_apply(tmp, hd, cs + (size_t)pos * half, sn + (size_t)pos * half);
            float* kd = kc + ((size_t)g * n_ctx + pos) * hd;
            float* vd = vc + ((size_t)g * n_ctx + pos) * hd;
            const float* vs = row + (size_t)(Hq + Hkv + g) * hd;
            for (int d = 0; d < hd; ++d) { kd[d] = round_bf16(tmp[d]); vd[d] = round_bf16(vs[d]); }
        }
        for (int h = 0; h < Hq; ++h) {
            const int g = h / R;
            q3o_rmsnorm(row + (size_t)h * hd, hd, qnw, eps, tmp);
            rope_apply(tmp, hd, cs + (size_t)pos * half, sn + (size_t)pos * half);
            attend_one(tmp, kc + (size_t)g * n_ctx * hd, vc + (size_t)g * n_ctx * hd, pos + 1, hd,
                       out + (size_t)r * Hq * hd + (size_t)h * hd);
        }
    }
    free(tmp);
}

void q3o_attention(const float* qkv, int32_t n_rows, int32_t pos0, int32_t Hq, int32_t Hkv, int32_t hd, const float* qnw,
                   const float* knw, float eps, float theta, const int32_t* sections, float* out) {
    const int n_ctx = pos0 + n_rows, half = hd / 2;
    float* cs = (float*)malloc((size_t)n_ctx * half * 4);
    float* sn = (float*)malloc((size_t)n_ctx * half * 4);
    rope_tables(n_ctx, hd, theta, sections, cs, sn);
    float* kc = (float*)calloc((size_t)Hkv * n_ctx * hd, 4);
    float* vc = (float*)calloc((size_t)Hkv * n_ctx * hd, 4);
    attn_rows(qkv, n_rows, pos0, Hq, Hkv, hd, qnw, knw, eps, cs, sn, kc, vc, n_ctx, out);
    free(kc); free(vc); free(cs); free(sn);
}

/* ------------------------------------------------------------------------------------------ */
/* rand 0.8 StdRng = ChaCha12 (rand_chacha 0.3, crate not vendored in /root/reference:          */
/* Cargo.toml:18 `rand = "0.8"`, Cargo.lock git-ignored). seed_from_u64 = rand_core 0.6 PCG32   */
/* expansion; gen::<f32>() = (next_u32() >> 8) * 2^-24. Call sites: src/models/llama/mod.rs:    */
/* 648 (seed_from_u64) and :757 (rng.gen()).                                                    */
/* ------------------------------------------------------------------------------------------ */
#define ROTL(v, n) (((v) << (n)) | ((v) >> (32 - (n))))
#define QR(a, b, c, d) a += b; d ^= a; d = ROTL(d, 16); c += d; b ^= c; b = ROTL(b, 12); a += b; d ^= a; d = ROTL(d, 8); c += d; b ^= c; b = ROTL(b, 7);
void q3o_chacha_block(const uint32_t in[16], int32_t rounds, uint32_t out[16]) {
    uint32_t x[16];
    memcpy(x, in, 64);
    for (int i = 0; i < rounds; i += 2) {
        QR(x[0], x[4], x[8], x[12]) QR(x[1], x[5], x[9], x[13]) QR(x[2], x[6], x[10], x[14]) QR(x[3], x[7], x[11], x[15])
        QR(x[0], x[5], x[10], x[15]) QR(x[1], x[6], x[11], x[12]) QR(x[2], x[7], x[8], x[13]) QR(x[3], x[4], x[9], x[14])
    }
    for (int i = 0; i < 16; ++i) out[i] = x[i] + in[i];
}
typedef struct { uint32_t st[16]; uint32_t buf[64]; int idx; } stdrng;
static void stdrng_seed_u64(stdrng* r, uint64_t state) {
    const uint64_t MUL = 6364136223846793005ULL, INC = 11634580027462260723ULL;
    r->st[0] = 0x61707865u; r->st[1] = 0x3320646eu; r->st[2] = 0x79622d32u; r->st[3] = 0x6b206574u;
    for (int i = 0; i < 8; ++i) {
        state = state * MUL + INC;
        uint32_t xs = (uint32_t)(((state >> 18) ^ state) >> 27);
        uint32_t rot = (uint32_t)(state >> 59);
        r->st[4 + i] = (xs >> rot) | (xs << ((32 - rot) & 31));
    }
    r->st[12] = r->st[13] = r->st[14] = r->st[15] = 0;
    r->idx = 64;
}
static uint32_t stdrng_next_u32(stdrng* r) {
    if (r->idx >= 64) {
        for (int b = 0; b < 4; ++b) {
            q3o_chacha_block(r->st, 12, r->buf + 16 * b);
            if (++r->st[12] == 0) ++r->st[13];
        }
        r->idx = 0;
    }
    return r->buf[r->idx++];
}
static float stdrng_f32(stdrng* r) { return (float)(stdrng_next_u32(r) >> 8) * (1.0f / 16777216.0f); }
void q3o_rng_f32(uint64_t seed, int32_t n, float* out) {
    stdrng r; stdrng_seed_u64(&r, seed);
    for (int i = 0; i < n; ++i) out[i] = stdrng_f32(&r);
}

/* ------------------------------------------------------------------------------------------ */
/* H4 sampler — src/models/llama/mod.rs:666-772                                                */
/* ------------------------------------------------------------------------------------------ */
typedef struct { int32_t idx; float v; } cand;
static void stable_sort_desc(cand* a, cand* tmp, int n) { /* merge sort; ties (and NaN = "Equal") keep order */
    for (int w = 1; w < n; w *= 2) {
        for (int i = 0; i < n; i += 2 * w) {
            int l = i, m = (i + w < n) ? i + w : n, r = (i + 2 * w < n) ? i + 2 * w : n, p = l, q = m, o = l;
            while (p < m && q < r) tmp[o++] = (a[q].v > a[p].v) ? a[q++] : a[p++];
            while (p < m) tmp[o++] = a[p++];
            while (q < r) tmp[o++] = a[q++];
        }
        memcpy(a, tmp, (size_t)n * sizeof(cand));
    }
}
int32_t q3o_sample(const float* logits, int32_t limit, float temperature, int32_t top_k_i, float top_p, float r) {
    const int start = 0, end = limit;
    if (temperature <= 0.0f) { /* :690-701 first max, strict > */
        float mv = -INFINITY; int mi = start;
        for (int i = start; i < end; ++i) if (logits[i] > mv) { mv = logits[i]; mi = i; }
        return mi;
    }
    int n = end - start;
    cand* c = (cand*)malloc((size_t)n * sizeof(cand));
    cand* tmp = (cand*)malloc((size_t)n * sizeof(cand));
    for (int i = 0; i < n; ++i) { c[i].idx = start + i; c[i].v = logits[start + i]; }  /* :705 */
    stable_sort_desc(c, tmp, n);                                                        /* :708 */
    size_t top_k = (size_t)(int64_t)top_k_i; /* `top_k as usize` :646 — negatives become huge */
    if (top_k > 0 && top_k < (size_t)n) n = (int)top_k;                                 /* :711-713 */
    const float max_logit = n > 0 ? c[0].v : 0.0f;                                      /* :716 */
    for (int i = 0; i < n; ++i) c[i].v = q3o_expf((c[i].v - max_logit) / temperature);  /* :717-723 */
    float sum = 0.0f;
    for (int i = 0; i < n; ++i) sum += c[i].v;                                          /* :726 */
    if (sum > 0.0f) for (int i = 0; i < n; ++i) c[i].v /= sum;
    if (top_p < 1.0f) {                                                                 /* :734-753 */
        float cum = 0.0f; int cutoff = n;
        for (int i = 0; i < n; ++i) { cum += c[i].v; if (cum >= top_p) { cutoff = i + 1; break; } }
        n = cutoff;
        float ns = 0.0f;
        for (int i = 0; i < n; ++i) ns += c[i].v;
        if (ns > 0.0f) for (int i = 0; i < n; ++i) c[i].v /= ns;
    }
    float cum = 0.0f; int32_t res = -1;                                                 /* :756-764 */
    for (int i = 0; i < n; ++i) { cum += c[i].v; if (r < cum) { res = c[i].idx; break; } }
    if (res < 0) res = n > 0 ? c[0].idx : start;                                        /* :767-770 */
    free(c); free(tmp);
    return res;
}

/* H2 — src/tts/engine.rs:306-314 */
void q3o_qwen3_position(int32_t start, int32_t len, int32_t* out) {
    for (int j = 0; j < 3; ++j) for (int i = 0; i < len; ++i) out[j * len + i] = start + i;
    for (int i = 0; i < len; ++i) out[3 * len + i] = 0;
}

/* H8 — src/tts/engine.rs:507-541 */
int32_t q3o_chunk_plan(int32_t n_frames, int32_t* cf, int32_t* cl, int32_t max_calls) {
    int calls = 0, buffer = 0;
    for (int msg = 0; msg <= n_frames; ++msg) {
        const int is_final = (msg == n_frames);
        if (!is_final) buffer += 16;
        if (buffer >= 64 || is_final) {
            int valid = (buffer / 16) * 16;
            if (valid > 0) {
                if (calls < max_calls) { cf[calls] = valid / 16; cl[calls] = is_final; }
                ++calls;
                int remaining = buffer - valid;
                buffer = (remaining > 0 && !is_final) ? remaining : 0;
            } else buffer = 0;
        }
        if (is_final) break;
    }
    return calls;
}

/* ------------------------------------------------------------------------------------------ */
/* model                                                                                        */
/* ------------------------------------------------------------------------------------------ */
/* q3_oracle_bf16.c: the canonical bf16-MFMA arithmetic both transformers run in */
void q3o_permute_rows_bf16(const uint16_t* src, int32_t rows, int32_t K, uint16_t* dst);
void q3o_bgemm_raw_p(const uint16_t* xp, int32_t rows, int32_t K, const uint16_t* wp, int32_t N, float* out, int32_t ldo, int32_t threads);
void q3o_bgemm_q8_raw_p(const uint16_t* xp, int32_t rows, int32_t K, const uint16_t* qp, const float* dsc, int32_t N, float* out, int32_t ldo, int32_t threads);
void q3o_quantize_q8_0(const float* x, int64_t n, int8_t* q, uint16_t* d_f16);
void q3o_permute_rows_q8(const int8_t* src, int32_t rows, int32_t K, uint16_t* dst);
void q3o_bgemm_q8a8_raw(const int8_t* qa, const float* da, int32_t rows, int32_t K, const uint16_t* qp, const float* dsc, int32_t N, float* out,
                        int32_t ldo, int32_t threads);
float q3o_f16_to_f32(uint16_t h);

typedef struct {
    int L, d, Hq, Hkv, hd, F;
    /* matrices: bf16, row-major [N][K], the k of every 32-block permuted to the MFMA operand order (q3_oracle_bf16.c) */
    float** attn_norm; uint16_t** wqkv; float** qn; float** kn; uint16_t** wo; float** ffn_norm;
    uint16_t** wg; uint16_t** wu; uint16_t** wd;
    float* out_norm; uint16_t* head; int head_n;
    /* Q8_0 mode (q3o_set_talker_q8): the matrices above then hold the block quants as (exact) bf16 in operand order and these the
     * block scales f32(f16 d) [N][K/32]; NULL = bf16 weights */
    float** s_qkv; float** s_o; float** s_g; float** s_u; float** s_d; float* s_head;
    int a8;  /* 1: the ACTIVATIONS of every GEMM are Q8_0 blocks too (W8A8, ggml's vec_dot_q8_0_q8_0: q3o_bgemm_q8a8_raw); needs s_* */
    float *kc, *vc; int n_ctx; /* [L][Hkv][n_ctx][hd] */
    float *cs, *sn;
} tfm;

struct q3o_model {
    q3o_model_config c; uint64_t seed; int n_ctx;
    tfm T, P;
    float* proj_w; float* proj_b;  /* f32 [p_d_model][d_embed], as the reference keeps them (src/assets_manager.rs:212-241) */
    int arith;                     /* 0: canonical bf16-MFMA order (what the device computes); 1: plain f32 (structure pinning, tests) */
};

static inline int kperm32(int p) { const int g = p >> 3, e = p & 7; return e < 4 ? 4 * g + e : 16 + 4 * g + (e - 4); }
static inline int kpos32(int k) { return 8 * ((k & 15) >> 2) + (k & 3) + ((k & 16) ? 4 : 0); }  /* inverse of kperm32 */

static float* gen_f32(uint64_t seed, uint32_t tid, size_t n, float base, float std) {
    float* p = (float*)malloc(n * 4);
    const float scale = std / IH4_STD;
    for (size_t i = 0; i < n; ++i) p[i] = base + q3o_synth(seed, tid, i, scale);
    return p;
}
/* logical [N][K] tensor -> bf16 rows in operand order */
static uint16_t* gen_mat_p(uint64_t seed, uint32_t tid, int N, int K, float std) {
    uint16_t* dst = (uint16_t*)malloc((size_t)N * K * 2);
    const float scale = std / IH4_STD;
#pragma omp parallel for schedule(static) num_threads(g_threads > 0 ? g_threads : 1)
    for (int n = 0; n < N; ++n)
        for (int k0 = 0; k0 < K; k0 += 32)
            for (int p = 0; p < 32; ++p)
                dst[(size_t)n * K + k0 + p] = q3o_bf16(q3o_synth(seed, tid, (size_t)n * K + k0 + kperm32(p), scale));
    return dst;
}
static inline float w_nat(const uint16_t* wp, int K, int n, int k) { return bf16f(wp[(size_t)n * K + (k & ~31) + kpos32(k & 31)]); }

static void tfm_init(tfm* t, uint64_t seed, int grp, int L, int d, int Hq, int Hkv, int hd, int F, int head_n, float theta,
                     const int* sections, int n_ctx) {
    t->L = L; t->d = d; t->Hq = Hq; t->Hkv = Hkv; t->hd = hd; t->F = F; t->head_n = head_n; t->n_ctx = n_ctx;
    t->attn_norm = calloc(L, sizeof(void*)); t->wqkv = calloc(L, sizeof(void*)); t->qn = calloc(L, sizeof(void*));
    t->kn = calloc(L, sizeof(void*)); t->wo = calloc(L, sizeof(void*)); t->ffn_norm = calloc(L, sizeof(void*));
    t->wg = calloc(L, sizeof(void*)); t->wu = calloc(L, sizeof(void*)); t->wd = calloc(L, sizeof(void*));
    const int nq = Hq * hd, nkv = Hkv * hd, nqkv = nq + 2 * nkv;
    for (int l = 0; l < L; ++l) {
        t->attn_norm[l] = gen_f32(seed, TID(grp, l, W_ATTN_NORM), d, 1.0f, 0.05f);
        t->ffn_norm[l] = gen_f32(seed, TID(grp, l, W_FFN_NORM), d, 1.0f, 0.05f);
        t->qn[l] = gen_f32(seed, TID(grp, l, W_QNORM), hd, 1.0f, 0.05f);
        t->kn[l] = gen_f32(seed, TID(grp, l, W_KNORM), hd, 1.0f, 0.05f);
        t->wqkv[l] = (uint16_t*)malloc((size_t)nqkv * d * 2);
        uint16_t* q = gen_mat_p(seed, TID(grp, l, W_Q), nq, d, 0.02f);
        uint16_t* k = gen_mat_p(seed, TID(grp, l, W_K), nkv, d, 0.02f);
        uint16_t* v = gen_mat_p(seed, TID(grp, l, W_V), nkv, d, 0.02f);
        memcpy(t->wqkv[l], q, (size_t)nq * d * 2);
        memcpy(t->wqkv[l] + (size_t)nq * d, k, (size_t)nkv * d * 2);
        memcpy(t->wqkv[l] + (size_t)(nq + nkv) * d, v, (size_t)nkv * d * 2);
        free(q); free(k); free(v);
        t->wo[l] = gen_mat_p(seed, TID(grp, l, W_O), d, nq, 0.02f);
        t->wg[l] = gen_mat_p(seed, TID(grp, l, W_GATE), F, d, 0.02f);
        t->wu[l] = gen_mat_p(seed, TID(grp, l, W_UP), F, d, 0.02f);
        t->wd[l] = gen_mat_p(seed, TID(grp, l, W_DOWN), d, F, 0.02f);
    }
    t->out_norm = gen_f32(seed, TID(grp, L_MODEL, WM_OUT_NORM), d, 1.0f, 0.05f);
    t->head = gen_mat_p(seed, TID(grp, L_MODEL, WM_HEAD), head_n, d, 0.02f);
    t->kc = calloc((size_t)L * Hkv * n_ctx * hd, 4);
    t->vc = calloc((size_t)L * Hkv * n_ctx * hd, 4);
    t->cs = malloc((size_t)n_ctx * (hd / 2) * 4); t->sn = malloc((size_t)n_ctx * (hd / 2) * 4);
    rope_tables(n_ctx, hd, theta, sections, t->cs, t->sn);
}
static void tfm_free(tfm* t) {
    for (int l = 0; l < t->L; ++l) {
        free(t->attn_norm[l]); free(t->wqkv[l]); free(t->qn[l]); free(t->kn[l]); free(t->wo[l]); free(t->ffn_norm[l]);
        free(t->wg[l]); free(t->wu[l]); free(t->wd[l]);
    }
    free(t->attn_norm); free(t->wqkv); free(t->qn); free(t->kn); free(t->wo); free(t->ffn_norm);
    free(t->wg); free(t->wu); free(t->wd); free(t->out_norm); free(t->head); free(t->kc); free(t->vc);
    free(t->cs); free(t->sn);
    if (t->s_qkv) {
        for (int l = 0; l < t->L; ++l) { free(t->s_qkv[l]); free(t->s_o[l]); free(t->s_g[l]); free(t->s_u[l]); free(t->s_d[l]); }
        free(t->s_qkv); free(t->s_o); free(t->s_g); free(t->s_u); free(t->s_d); free(t->s_head);
    }
}

/* one matrix (bf16, operand order) -> its ggml Q8_0 blocks, in place: quants as bf16 in operand order + the returned f32 scales. The
 * quantiser sees the weights in their NATURAL order (a block = 32 consecutive input columns of one output row), as a GGUF file holds them */
static float* matrix_to_q8(uint16_t* wp, int N, int K) {
    float* dsc = (float*)malloc((size_t)N * (K / 32) * 4);
#pragma omp parallel for schedule(static) num_threads(g_threads > 0 ? g_threads : 1)
    for (int n = 0; n < N; ++n) {
        float row[32]; int8_t q[32]; uint16_t d16;
        for (int k0 = 0; k0 < K; k0 += 32) {
            for (int k = 0; k < 32; ++k) row[k] = w_nat(wp, K, n, k0 + k);
            q3o_quantize_q8_0(row, 32, q, &d16);
            dsc[(size_t)n * (K / 32) + k0 / 32] = q3o_f16_to_f32(d16);
            q3o_permute_rows_q8(q, 1, 32, wp + (size_t)n * K + k0);
        }
    }
    return dsc;
}
static void tfm_to_q8(tfm* t) {
    if (t->s_qkv) return;
    const int L = t->L, d = t->d, nq = t->Hq * t->hd, nkv = t->Hkv * t->hd, nqkv = nq + 2 * nkv, F = t->F;
    t->s_qkv = calloc(L, sizeof(void*)); t->s_o = calloc(L, sizeof(void*)); t->s_g = calloc(L, sizeof(void*)); t->s_u = calloc(L, sizeof(void*)); t->s_d = calloc(L, sizeof(void*));
    for (int l = 0; l < L; ++l) {
        t->s_qkv[l] = matrix_to_q8(t->wqkv[l], nqkv, d); t->s_o[l] = matrix_to_q8(t->wo[l], d, nq);
        t->s_g[l] = matrix_to_q8(t->wg[l], F, d); t->s_u[l] = matrix_to_q8(t->wu[l], F, d); t->s_d[l] = matrix_to_q8(t->wd[l], d, F);
    }
    t->s_head = matrix_to_q8(t->head, t->head_n, d);
}

/* RAW of natural-order bf16 rows against one of the model's matrices */
static void bgemm_rows(const uint16_t* ab, int n, int K, const uint16_t* wp, const float* dsc, int N, float* out, int ldo) {
    uint16_t* ap = (uint16_t*)malloc((size_t)n * K * 2);
    q3o_permute_rows_bf16(ab, n, K, ap);
    if (dsc) q3o_bgemm_q8_raw_p(ap, n, K, wp, dsc, N, out, ldo, g_threads > 0 ? g_threads : 1);  /* Q8_0 weights (the Talker in Q8 mode) */
    else q3o_bgemm_raw_p(ap, n, K, wp, N, out, ldo, g_threads > 0 ? g_threads : 1);
    free(ap);
}

/* W8A8: f32 activation rows v [n][K] -> ggml Q8_0 blocks -> RAW against a Q8_0 matrix (q3_oracle_bf16.c) */
static void bgemm_rows_a8(const float* v, int n, int K, const uint16_t* wp, const float* dsc, int N, float* out, int ldo) {
    int8_t* qa = (int8_t*)malloc((size_t)n * K);
    uint16_t* d16 = (uint16_t*)malloc((size_t)n * (K / 32) * 2);
    float* da = (float*)malloc((size_t)n * (K / 32) * 4);
    for (int r = 0; r < n; ++r) q3o_quantize_q8_0(v + (size_t)r * K, K, qa + (size_t)r * K, d16 + (size_t)r * (K / 32));
    for (size_t i = 0; i < (size_t)n * (K / 32); ++i) da[i] = q3o_f16_to_f32(d16[i]);
    q3o_bgemm_q8a8_raw(qa, da, n, K, wp, dsc, N, out, ldo, g_threads > 0 ? g_threads : 1);
    free(qa); free(d16); free(da);
}

/* One transformer over n rows (positions pos0.. of one sequence), canonical arithmetic. x [n][d] is the f32 residual stream,
 * updated in place. On entry xb / ssp are the norm inputs of x for attn_norm[0] (q3o_norm_inputs); on exit for out_norm.
 * Device: run_layers in q3_engine.hip — bgemm(QKV) -> k_attend -> bgemm(O, residual + norm outputs) -> bgemm(gate/up, SwiGLU)
 * -> bgemm(down, residual + norm outputs). */
static void tfm_layers(tfm* t, float* x, uint16_t* xb, float* ssp, int n, int pos0, float eps) {
    const int d = t->d, nq = t->Hq * t->hd, nkv = t->Hkv * t->hd, nqkv = nq + 2 * nkv, F = t->F, nt = d / 16;
    float* qkv = malloc((size_t)n * nqkv * 4);
    float* att = malloc((size_t)n * nq * 4);
    uint16_t* ab = malloc((size_t)n * (nq > F ? nq : F) * 2);
    float* g = malloc((size_t)n * F * 4);
    float* u = malloc((size_t)n * F * 4);
    float* y = malloc((size_t)n * d * 4);
    float* sc = malloc((size_t)n * 4);
    float* vf = t->a8 ? malloc((size_t)n * d * 4) : NULL;
    for (int l = 0; l < t->L; ++l) {
        for (int r = 0; r < n; ++r) sc[r] = q3o_row_scale(ssp + (size_t)r * nt, nt, d, eps);
        if (t->a8) {  /* the quantiser sees v = x * nw in f32 (what the producer multiplies before it rounds to bf16 in the other modes) */
            const float* nw = t->attn_norm[l];
            for (int r = 0; r < n; ++r) for (int k = 0; k < d; ++k) vf[(size_t)r * d + k] = x[(size_t)r * d + k] * nw[k];
            bgemm_rows_a8(vf, n, d, t->wqkv[l], t->s_qkv[l], nqkv, qkv, nqkv);
        } else
        bgemm_rows(xb, n, d, t->wqkv[l], t->s_qkv ? t->s_qkv[l] : NULL, nqkv, qkv, nqkv);
        scale_rows(qkv, n, nqkv, nqkv, sc);
        size_t co = (size_t)l * t->Hkv * t->n_ctx * t->hd;
        attn_rows(qkv, n, pos0, t->Hq, t->Hkv, t->hd, t->qn[l], t->kn[l], eps, t->cs, t->sn, t->kc + co, t->vc + co, t->n_ctx, att);
        if (t->a8) bgemm_rows_a8(att, n, nq, t->wo[l], t->s_o[l], d, y, d);
        else {
        for (size_t i = 0; i < (size_t)n * nq; ++i) ab[i] = q3o_bf16(att[i]);
        bgemm_rows(ab, n, nq, t->wo[l], t->s_o ? t->s_o[l] : NULL, d, y, d);
        }
        for (size_t i = 0; i < (size_t)n * d; ++i) x[i] = x[i] + y[i];
        for (int r = 0; r < n; ++r) q3o_norm_inputs(x + (size_t)r * d, d, t->ffn_norm[l], xb + (size_t)r * d, ssp + (size_t)r * nt);
        for (int r = 0; r < n; ++r) sc[r] = q3o_row_scale(ssp + (size_t)r * nt, nt, d, eps);
        if (t->a8) {
            const float* nw = t->ffn_norm[l];
            for (int r = 0; r < n; ++r) for (int k = 0; k < d; ++k) vf[(size_t)r * d + k] = x[(size_t)r * d + k] * nw[k];
            bgemm_rows_a8(vf, n, d, t->wg[l], t->s_g[l], F, g, F);
            bgemm_rows_a8(vf, n, d, t->wu[l], t->s_u[l], F, u, F);
        } else {
        bgemm_rows(xb, n, d, t->wg[l], t->s_g ? t->s_g[l] : NULL, F, g, F);
        bgemm_rows(xb, n, d, t->wu[l], t->s_u ? t->s_u[l] : NULL, F, u, F);
        }
        scale_rows(g, n, F, F, sc);
        scale_rows(u, n, F, F, sc);
        if (t->a8) {
            for (size_t i = 0; i < (size_t)n * F; ++i) g[i] = swiglu(g[i], u[i]);   /* h stays f32: the quantiser's input */
            bgemm_rows_a8(g, n, F, t->wd[l], t->s_d[l], d, y, d);
        } else {
        for (size_t i = 0; i < (size_t)n * F; ++i) ab[i] = q3o_bf16(swiglu(g[i], u[i]));
        bgemm_rows(ab, n, F, t->wd[l], t->s_d ? t->s_d[l] : NULL, d, y, d);
        }
        for (size_t i = 0; i < (size_t)n * d; ++i) x[i] = x[i] + y[i];
        const float* nxt = l + 1 < t->L ? t->attn_norm[l + 1] : t->out_norm;
        for (int r = 0; r < n; ++r) q3o_norm_inputs(x + (size_t)r * d, d, nxt, xb + (size_t)r * d, ssp + (size_t)r * nt);
    }
    free(qkv); free(att); free(ab); free(g); free(u); free(y); free(sc); free(vf);
}

/* ---- plain f32 arithmetic of the same structure (arith == 1): what the family code computes up to summation order; used by
 * tests/test_decoder_family_cpu.py to pin the STRUCTURE (norm placement, NeoX pairing, GQA mapping, QK-norm) against
 * transformers' Qwen3, and to state the distance between the canonical bf16 order and f32 ---------------------------------- */
static void rmsnorm_plain(const float* x, int d, const float* w, float eps, float* y) {
    double ss = 0.0;
    for (int i = 0; i < d; ++i) ss += (double)x[i] * (double)x[i];
    const float rinv = (float)(1.0 / sqrt(ss / (double)d + (double)eps));
    for (int i = 0; i < d; ++i) y[i] = (x[i] * rinv) * w[i];
}
static void gemm_plain(const float* x, int n, int K, const uint16_t* wp, int N, float* out, int ldo) {
#pragma omp parallel for schedule(static) num_threads(g_threads > 0 ? g_threads : 1)
    for (int c = 0; c < N; ++c)
        for (int r = 0; r < n; ++r) {
            double acc = 0.0;
            for (int k = 0; k < K; ++k) acc += (double)x[(size_t)r * K + k] * (double)w_nat(wp, K, c, k);
            out[(size_t)r * ldo + c] = (float)acc;
        }
}
static void attn_rows_plain(const float* qkv, int n_rows, int pos0, int Hq, int Hkv, int hd, const float* qnw, const float* knw,
                            float eps, const float* cs, const float* sn, float* kc, float* vc, int n_ctx, float* out) {
    const int ld = (Hq + 2 * Hkv) * hd, R = Hq / Hkv, half = hd / 2;
    float* tmp = (float*)malloc((size_t)hd * 4);
    for (int r = 0; r < n_rows; ++r) {
        const int pos = pos0 + r;
        const float* row = qkv + (size_t)r * ld;
        for (int g = 0; g < Hkv; ++g) {
            rmsnorm_plain(row + (size_t)(Hq + g) * hd, hd, knw, eps, tmp);
            rope_apply(tmp, hd, cs + (size_t)pos * half, sn + (size_t)pos * half);
            memcpy(kc + ((size_t)g * n_ctx + pos) * hd, tmp, (size_t)hd * 4);
            memcpy(vc + ((size_t)g * n_ctx + pos) * hd, row + (size_t)(Hq + Hkv + g) * hd, (size_t)hd * 4);
        }
        for (int h = 0; h < Hq; ++h) {
            const int g = h / R, T = pos + 1;
            rmsnorm_plain(row + (size_t)h * hd, hd, qnw, eps, tmp);
            rope_apply(tmp, hd, cs + (size_t)pos * half, sn + (size_t)pos * half);
            double* p = (double*)malloc((size_t)T * 8);
            double m = -1e300, l = 0.0;
            for (int t = 0; t < T; ++t) {
                const float* kr = kc + ((size_t)g * n_ctx + t) * hd;
                double s = 0.0;
                for (int dd = 0; dd < hd; ++dd) s += (double)tmp[dd] * (double)kr[dd];
                p[t] = s / sqrt((double)hd);
                if (p[t] > m) m = p[t];
            }
            for (int t = 0; t < T; ++t) { p[t] = exp(p[t] - m); l += p[t]; }
            float* o = out + (size_t)r * Hq * hd + (size_t)h * hd;
            for (int dd = 0; dd < hd; ++dd) {
                double a = 0.0;
                for (int t = 0; t < T; ++t) a += p[t] * (double)vc[((size_t)g * n_ctx + t) * hd + dd];
                o[dd] = (float)(a / l);
            }
            free(p);
        }
    }
    free(tmp);
}
static void tfm_layers_plain(tfm* t, float* x, int n, int pos0, float eps) {
    const int d = t->d, nq = t->Hq * t->hd, nkv = t->Hkv * t->hd, nqkv = nq + 2 * nkv, F = t->F;
    float* xn = malloc((size_t)n * d * 4);
    float* qkv = malloc((size_t)n * nqkv * 4);
    float* att = malloc((size_t)n * nq * 4);
    float* g = malloc((size_t)n * F * 4);
    float* u = malloc((size_t)n * F * 4);
    float* y = malloc((size_t)n * d * 4);
    for (int l = 0; l < t->L; ++l) {
        for (int r = 0; r < n; ++r) rmsnorm_plain(x + (size_t)r * d, d, t->attn_norm[l], eps, xn + (size_t)r * d);
        gemm_plain(xn, n, d, t->wqkv[l], nqkv, qkv, nqkv);
        size_t co = (size_t)l * t->Hkv * t->n_ctx * t->hd;
        attn_rows_plain(qkv, n, pos0, t->Hq, t->Hkv, t->hd, t->qn[l], t->kn[l], eps, t->cs, t->sn, t->kc + co, t->vc + co, t->n_ctx, att);
        gemm_plain(att, n, nq, t->wo[l], d, y, d);
        for (size_t i = 0; i < (size_t)n * d; ++i) x[i] += y[i];
        for (int r = 0; r < n; ++r) rmsnorm_plain(x + (size_t)r * d, d, t->ffn_norm[l], eps, xn + (size_t)r * d);
        gemm_plain(xn, n, d, t->wg[l], F, g, F);
        gemm_plain(xn, n, d, t->wu[l], F, u, F);
        for (size_t i = 0; i < (size_t)n * F; ++i) g[i] = (float)((double)g[i] / (1.0 + exp(-(double)g[i]))) * u[i];
        gemm_plain(g, n, F, t->wd[l], d, y, d);
        for (size_t i = 0; i < (size_t)n * d; ++i) x[i] += y[i];
    }
    free(xn); free(qkv); free(att); free(g); free(u); free(y);
}

q3o_model* q3o_create(const q3o_model_config* c, uint64_t seed, int32_t n_ctx, int32_t n_threads) {
    q3o_model* m = calloc(1, sizeof(*m));
    m->c = *c; m->seed = seed; m->n_ctx = n_ctx;
    g_threads = n_threads;
    tfm_init(&m->T, seed, G_TALKER, c->t_n_layer, c->t_d_model, c->t_n_head, c->t_n_kv_head, c->t_head_dim, c->t_d_ffn,
             c->t_vocab, c->t_rope_theta, c->t_mrope_sections, n_ctx);
    tfm_init(&m->P, seed, G_PRED, c->p_n_layer, c->p_d_model, c->p_n_head, c->p_n_kv_head, c->p_head_dim, c->p_d_ffn,
             (c->n_codebooks - 1) * c->codebook_size, c->p_rope_theta, NULL, 64);
    {   /* proj.weight / proj.bias: f32 tensors (the synthetic values are bf16-representable, like every synthetic matrix) */
        const size_t n = (size_t)c->p_d_model * c->d_embed;
        m->proj_w = (float*)malloc(n * 4);
        const float scale = 0.02f / IH4_STD;
        for (size_t i = 0; i < n; ++i) m->proj_w[i] = round_bf16(q3o_synth(seed, TID(G_ASSET, 0, WA_PROJ_W), i, scale));
    }
    m->proj_b = gen_f32(seed, TID(G_ASSET, 0, WA_PROJ_B), c->p_d_model, 0.0f, 0.02f);
    return m;
}
void q3o_destroy(q3o_model* m) {
    if (!m) return;
    tfm_free(&m->T); tfm_free(&m->P); free(m->proj_w); free(m->proj_b); free(m);
}
void q3o_set_arith(q3o_model* m, int32_t arith) { m->arith = arith; }
/* The Talker's matrices (and lm_head) as ggml Q8_0 blocks, multiplied in the canonical Q8 order of q3_oracle_bf16.c — what the device
 * computes with q3tts_engine_config.talker_q8_0 = 1. One-way (the bf16 weights are replaced). */
void q3o_set_talker_q8(q3o_model* m) { tfm_to_q8(&m->T); }
/* ... and every GEMM's activations as Q8_0 blocks as well: W8A8, what llama.cpp computes on a gguf_q8_0 model (talker_q8_0 = 2) */
void q3o_set_talker_q8a8(q3o_model* m) { tfm_to_q8(&m->T); m->T.a8 = 1; }
void q3o_set_threads(int32_t n) { g_threads = n > 0 ? n : 1; }  /* OpenMP threads of the GEMMs (bench.py: the 4-thread and all-cores legs) */
/* natural-order f32 copies of the synthetic tensors, for loading the same model into the family code (tests) */
const float* q3o_norm_weight(const q3o_model* m, int32_t talker, int32_t layer, int32_t which) {
    const tfm* t = talker ? &m->T : &m->P;
    if (layer < 0) return t->out_norm;
    return which == 0 ? t->attn_norm[layer] : which == 1 ? t->ffn_norm[layer] : which == 2 ? t->qn[layer] : t->kn[layer];
}
/* which: 0 qkv (fused rows q | k | v), 1 o, 2 gate, 3 up, 4 down, 5 head (layer ignored); out [N][K] f32 */
int32_t q3o_matrix(const q3o_model* m, int32_t talker, int32_t layer, int32_t which, float* out) {
    const tfm* t = talker ? &m->T : &m->P;
    const int nq = t->Hq * t->hd, nqkv = nq + 2 * t->Hkv * t->hd;
    const uint16_t* w; int N, K;
    switch (which) {
        case 0: w = t->wqkv[layer]; N = nqkv; K = t->d; break;
        case 1: w = t->wo[layer]; N = t->d; K = nq; break;
        case 2: w = t->wg[layer]; N = t->F; K = t->d; break;
        case 3: w = t->wu[layer]; N = t->F; K = t->d; break;
        case 4: w = t->wd[layer]; N = t->d; K = t->F; break;
        default: w = t->head; N = t->head_n; K = t->d; break;
    }
    if (out) for (int n = 0; n < N; ++n) for (int k = 0; k < K; ++k) out[(size_t)n * K + k] = w_nat(w, K, n, k);
    return N;
}

/* embedding tables are generated on demand (bf16-representable f32, like the shipped presets) */
void q3o_text_embedding(const q3o_model* m, int64_t id, float* out) { /* src/assets_manager.rs:444-460 */
    const int d = m->c.d_embed;
    if (id >= 0 && id < m->c.text_vocab) {
        const float scale = 0.05f / IH4_STD;
        for (int i = 0; i < d; ++i) out[i] = round_bf16(q3o_synth(m->seed, TID(G_ASSET, 0, WA_TEXT), (uint64_t)id * d + i, scale));
    } else {
        for (int i = 0; i < d; ++i) out[i] = fmodf((float)((uint64_t)id * 17 + (uint64_t)i), 2.0f) - 1.0f;
    }
}
void q3o_codec_embedding(const q3o_model* m, int32_t q, int32_t code, float* out) { /* src/assets_manager.rs:419-437 */
    const int d = m->c.d_embed;
    const int rows = (q == 0) ? m->c.codec0_rows : m->c.codecq_rows;
    if (code < 0) code = 0;
    if (q >= 0 && q < m->c.n_codebooks && code < rows) {
        const float scale = 0.05f / IH4_STD;
        for (int i = 0; i < d; ++i) out[i] = round_bf16(q3o_synth(m->seed, TID(G_ASSET, 1 + q, 0), (uint64_t)code * d + i, scale));
    } else {
        for (int i = 0; i < d; ++i) out[i] = 0.0f;
    }
}
/* H6 — Assets::project, src/assets_manager.rs:383-399, in the reference's OWN arithmetic: the accumulator starts from the bias
 * and takes `sum += h * w` (one f32 multiply, one f32 add) over the inputs in ascending order. This is one of the few
 * floating-point sequences the crate itself spells out, so here the oracle (and the device kernel k_project) is pinned. */
void q3o_project_rows(const float* w, const float* bias, int32_t n_in, int32_t n_out, const float* x, int32_t rows, float* y) {
#pragma omp parallel for schedule(static) collapse(2) num_threads(g_threads > 0 ? g_threads : 1)
    for (int r = 0; r < rows; ++r)
        for (int o = 0; o < n_out; ++o) {
            float sum = bias[o];
            const float* wr = w + (size_t)o * n_in;
            const float* xr = x + (size_t)r * n_in;
            for (int i = 0; i < n_in; ++i) sum += xr[i] * wr[i];
            y[(size_t)r * n_out + o] = sum;
        }
}
void q3o_project(const q3o_model* m, const float* x, float* y) {
    q3o_project_rows(m->proj_w, m->proj_b, m->c.d_embed, m->c.p_d_model, x, 1, y);
}

/* H1 — src/tts/prompt.rs */
enum { PAD = 2148, BOS = 2149, THINK = 2154, NOTHINK = 2155, THINK_BOS = 2156, THINK_EOS = 2157, CODEC_BOS_ICL = 2160 };
enum { BOS_TOKEN = 151672, EOS_TOKEN = 151673 };
static void add_rows(float* dst, const float* a, const float* b, int d) { for (int i = 0; i < d; ++i) dst[i] = a[i] + b[i]; }

int32_t q3o_build_prompt(const q3o_model* m, const q3o_prompt_desc* p, float* out, int32_t max_tok) {
    const int d = m->c.d_embed;
    const int marker_id = m->c.tts_pad_id; /* TEXT_AUDIO_MARKER 151671, src/tts/prompt.rs:16 */
    int n = 0;
    float* a = malloc((size_t)d * 4); float* b = malloc((size_t)d * 4); float* marker = malloc((size_t)d * 4);
    float* pad0 = malloc((size_t)d * 4);
#define EMIT(src) do { if (out && n < max_tok) memcpy(out + (size_t)n * d, (src), (size_t)d * 4); ++n; } while (0)
    q3o_text_embedding(m, marker_id, marker);
    q3o_codec_embedding(m, 0, PAD, pad0);
    if (p->instruct_ids) { /* :153-169 */
        const int64_t pre[3] = {151644, 872, 198}, suf[2] = {151645, 198};
        for (int i = 0; i < 3; ++i) { q3o_text_embedding(m, pre[i], a); EMIT(a); }
        for (int i = 0; i < p->n_instruct; ++i) { q3o_text_embedding(m, p->instruct_ids[i], a); EMIT(a); }
        for (int i = 0; i < 2; ++i) { q3o_text_embedding(m, suf[i], a); EMIT(a); }
    }
    { const int64_t role[3] = {151644, 77091, 198}; /* :171-175 */
      for (int i = 0; i < 3; ++i) { q3o_text_embedding(m, role[i], a); EMIT(a); } }
    if (p->lang_id >= 0) { /* :180-191 */
        const int ids[4] = {THINK, THINK_BOS, p->lang_id, THINK_EOS};
        for (int i = 0; i < 4; ++i) { q3o_codec_embedding(m, 0, ids[i], b); add_rows(a, marker, b, d); EMIT(a); }
    } else { /* :192-204 */
        const int ids[3] = {NOTHINK, THINK_BOS, THINK_EOS};
        for (int i = 0; i < 3; ++i) { q3o_codec_embedding(m, 0, ids[i], b); add_rows(a, marker, b, d); EMIT(a); }
    }
    if (p->spk_id >= 0) { q3o_codec_embedding(m, 0, p->spk_id, b); add_rows(a, marker, b, d); EMIT(a); } /* :207-214 */
    else if (p->spk_emb) { add_rows(a, marker, p->spk_emb, d); EMIT(a); }                                 /* :215-222 */
    if (p->ref_codes) { /* mid embeds of build_clone_prompt :38-106 */
        q3o_text_embedding(m, BOS_TOKEN, b); add_rows(a, b, pad0, d); EMIT(a);
        for (int i = 0; i < p->n_ref_text; ++i) { q3o_text_embedding(m, p->ref_text_ids[i], b); add_rows(a, b, pad0, d); EMIT(a); }
        q3o_text_embedding(m, EOS_TOKEN, b); add_rows(a, b, pad0, d); EMIT(a);
        q3o_codec_embedding(m, 0, CODEC_BOS_ICL, b); add_rows(a, marker, b, d); EMIT(a); /* :67-74 */
        float* sum = malloc((size_t)d * 4);
        for (int s = 0; s < p->n_ref_frames; ++s) { /* :79-96 */
            for (int i = 0; i < d; ++i) sum[i] = 0.0f;
            for (int q = 0; q < 16; ++q) {
                q3o_codec_embedding(m, q, p->ref_codes[s * 16 + q], b);
                for (int i = 0; i < d; ++i) sum[i] += b[i];
            }
            add_rows(a, marker, sum, d); EMIT(a);
        }
        free(sum);
        add_rows(a, marker, pad0, d); EMIT(a); /* :98-106 */
    }
    q3o_text_embedding(m, BOS_TOKEN, b); add_rows(a, b, pad0, d); EMIT(a); /* :229-239 */
    for (int i = 0; i < p->n_text; ++i) { q3o_text_embedding(m, p->text_ids[i], b); add_rows(a, b, pad0, d); EMIT(a); } /* :241-245 */
    q3o_text_embedding(m, EOS_TOKEN, b); add_rows(a, b, pad0, d); EMIT(a); /* :247-254 */
    q3o_codec_embedding(m, 0, BOS, b); add_rows(a, marker, b, d); EMIT(a); /* :256-264 */
#undef EMIT
    free(a); free(b); free(marker); free(pad0);
    return n;
}

/* final norm + lm_head (optionally a column slice) of ONE row whose norm inputs for out_norm are xb / ssp */
static void head_row(const q3o_model* m, tfm* t, const float* xrow, const uint16_t* xb, const float* ssp, float eps, int col0, int ncols,
                     float* hidden_out, float* logits) {
    if (m->arith == 1) {
        float* xn = malloc((size_t)t->d * 4);
        rmsnorm_plain(xrow, t->d, t->out_norm, eps, xn);
        if (hidden_out) memcpy(hidden_out, xn, (size_t)t->d * 4);
        gemm_plain(xn, 1, t->d, t->head + (size_t)col0 * t->d, ncols, logits, ncols);
        free(xn);
        return;
    }
    if (hidden_out) q3o_rmsnorm(xrow, t->d, t->out_norm, eps, hidden_out); /* standalone canonical RMSNorm (§4.2): the projection's input */
    const float sc = q3o_row_scale(ssp, t->d / 16, t->d, eps);
    if (t->a8) {
        float* v = malloc((size_t)t->d * 4);
        for (int k = 0; k < t->d; ++k) v[k] = xrow[k] * t->out_norm[k];
        bgemm_rows_a8(v, 1, t->d, t->head + (size_t)col0 * t->d, t->s_head + (size_t)col0 * (t->d / 32), ncols, logits, ncols);
        free(v);
    } else
    bgemm_rows(xb, 1, t->d, t->head + (size_t)col0 * t->d, t->s_head ? t->s_head + (size_t)col0 * (t->d / 32) : NULL, ncols, logits, ncols);
    scale_rows(logits, 1, ncols, ncols, &sc);
}

/* rows through one transformer in the model's arithmetic; xb / ssp as in tfm_layers (ignored by the plain form) */
static void run_tfm(const q3o_model* m, tfm* t, float* x, uint16_t* xb, float* ssp, int n, int pos0) {
    if (m->arith == 1) { tfm_layers_plain(t, x, n, pos0, m->c.rms_eps); return; }
    for (int r = 0; r < n; ++r) q3o_norm_inputs(x + (size_t)r * t->d, t->d, t->attn_norm[0], xb + (size_t)r * t->d, ssp + (size_t)r * (t->d / 16));
    tfm_layers(t, x, xb, ssp, n, pos0, m->c.rms_eps);
}

void q3o_talker_prefill(q3o_model* m, const float* embd, int32_t n_tok, float* hidden_out, float* logits_out) {
    const int d = m->T.d;
    float* x = malloc((size_t)n_tok * d * 4);
    uint16_t* xb = malloc((size_t)n_tok * d * 2);
    float* ssp = malloc((size_t)n_tok * (d / 16) * 4);
    memcpy(x, embd, (size_t)n_tok * d * 4);
    run_tfm(m, &m->T, x, xb, ssp, n_tok, 0);
    const size_t last = (size_t)(n_tok - 1);
    head_row(m, &m->T, x + last * d, xb + last * d, ssp + last * (d / 16), m->c.rms_eps, 0, m->c.t_vocab, hidden_out, logits_out);
    free(x); free(xb); free(ssp);
}

/* run_inference_stream — src/tts/engine.rs:445-656 (ids only; the vocoder side is q3o_vocoder_*) */
int32_t q3o_generate(q3o_model* m, const float* prompt, int32_t n_tok, float temperature, int32_t top_k, float top_p,
                     uint64_t seed, int32_t max_steps, int32_t min_frames, int32_t force_eos_at, int32_t* codes,
                     int32_t* hit_eos) {
    const q3o_model_config* c = &m->c;
    const int d = c->t_d_model, dp = c->p_d_model, de = c->d_embed, ncb = c->n_codebooks, cbs = c->codebook_size;
    float* hidden = malloc((size_t)d * 4);
    float* logits = malloc((size_t)c->t_vocab * 4);
    float* pl = malloc((size_t)cbs * 4);
    float* emb = malloc((size_t)de * 4);
    float* fb = malloc((size_t)de * 4);
    float* pad = malloc((size_t)de * 4);
    float* pin = malloc((size_t)2 * dp * 4);
    float* px = malloc((size_t)2 * dp * 4);
    uint16_t* xb = malloc((size_t)2 * (d > dp ? d : dp) * 2);
    float* ssp = malloc((size_t)2 * ((d > dp ? d : dp) / 16) * 4);
    stdrng rng; stdrng_seed_u64(&rng, seed); /* :473-485 */
    /* tts_pad: row 151671 of the text table when the table is that large, else zeros (src/assets_manager.rs:244-249) */
    if (c->tts_pad_id < c->text_vocab) q3o_text_embedding(m, c->tts_pad_id, pad);
    else memset(pad, 0, (size_t)de * 4);
    q3o_talker_prefill(m, prompt, n_tok, hidden, logits); /* :455-462 */
    int cur_pos = n_tok, n_frames = 0;
    *hit_eos = 0;
    for (int step = 0; step < max_steps; ++step) { /* :545 */
        int code0;
        if (force_eos_at >= 0 && step == force_eos_at) code0 = c->eos_code;
        else {
            if (step < min_frames && c->eos_code < c->sample_limit) logits[c->eos_code] = -INFINITY;
            float r = (temperature > 0.0f) ? stdrng_f32(&rng) : 0.0f;
            code0 = q3o_sample(logits, c->sample_limit, temperature, top_k, top_p, r); /* :555 */
        }
        if (code0 == c->eos_code) { *hit_eos = 1; break; } /* :558-561 */
        codes[n_frames * ncb + 0] = code0;
        q3o_project(m, hidden, pin);                 /* :568 */
        q3o_codec_embedding(m, 0, code0, emb);       /* :569 */
        q3o_project(m, emb, pin + dp);
        for (int i = 0; i < de; ++i) fb[i] = 0.0f + emb[i]; /* :584-585, :622-627 */
        memcpy(px, pin, (size_t)2 * dp * 4);
        run_tfm(m, &m->P, px, xb, ssp, 2, 0);        /* :575-582 (cache cleared == positions restart at 0) */
        head_row(m, &m->P, px + dp, xb + dp, ssp + dp / 16, c->rms_eps, 0, cbs, NULL, pl);
        for (int q = 1; q < ncb; ++q) {              /* :587 */
            float mv = -INFINITY; int mi = 0;        /* greedy :590-596 via :690-701 */
            for (int i = 0; i < cbs; ++i) if (pl[i] > mv) { mv = pl[i]; mi = i; }
            codes[n_frames * ncb + q] = mi;
            q3o_codec_embedding(m, q, mi, emb);      /* :599 */
            for (int i = 0; i < de; ++i) fb[i] += emb[i];
            if (q < ncb - 1) {                       /* :602-610 */
                q3o_project(m, emb, px);
                run_tfm(m, &m->P, px, xb, ssp, 1, q + 1);
                head_row(m, &m->P, px, xb, ssp, c->rms_eps, q * cbs, cbs, NULL, pl);
            }
        }
        ++n_frames;
        for (int i = 0; i < de; ++i) fb[i] += pad[i]; /* :628-630 */
        run_tfm(m, &m->T, fb, xb, ssp, 1, cur_pos);   /* :633-639 (fb is consumed as the residual stream) */
        head_row(m, &m->T, fb, xb, ssp, c->rms_eps, 0, c->t_vocab, hidden, logits);
        ++cur_pos;                                   /* :641 */
    }
    free(hidden); free(logits); free(pl); free(emb); free(fb); free(pad); free(pin); free(px); free(xb); free(ssp);
    return n_frames;
}
