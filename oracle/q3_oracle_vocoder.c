/*
 * q3_oracle_vocoder.c — CPU restatement of the streaming codec vocoder (V1-V6 of SURVEY.md §8a).
 *
 * TEST INFRASTRUCTURE ONLY (see q3_oracle.h). PARITY UNPINNED: the reference's vocoder is an ONNX graph
 * (qwen3_tts_decoder.onnx) executed by onnxruntime; neither is in /root/reference. What the reference pins is
 * the I/O contract (src/models/onnx.rs:342-459: codes [1,N,16] + is_last -> final_wav[:valid_samples]) and the
 * state shapes (src/models/onnx.rs:474-495: pre-conv 512 ch, latent 1024, 8 layers x 16 heads x 64). The layer
 * structure follows the same model family as readable in transformers' qwen3_omni_moe Code2Wav
 * (modeling_qwen3_omni_moe.py:3180-3696): codebook sum -> causal pre-conv -> sliding-window transformer with
 * LayerScale -> (ConvTranspose k=r,s=r + ConvNeXt) x n_upsample -> Conv k7 -> n_dec_blocks x {SnakeBeta,
 * ConvTranspose k=2r s=r, 3 residual units (SnakeBeta, Conv k7 dil 1/3/9, SnakeBeta, Conv k1)} -> SnakeBeta ->
 * Conv k7 -> clamp. Every convolution is strictly causal (right-trimmed transposed convs), so streaming in chunks
 * is bit-identical to one call; this restatement therefore recomputes from the first frame on every call and
 * returns only the new samples.
 *
 * Compute dtype: bf16 weights, GEMM/conv inputs rounded to bf16, f32 accumulation, f32 elsewhere.
 */
#include "q3_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

extern float q3o_synth(uint64_t seed, uint32_t tensor, uint64_t idx, float scale);
extern uint16_t q3o_bf16(float x);

static inline float rb(float x) { /* round to bf16 and back */
    uint32_t u = (uint32_t)q3o_bf16(x) << 16; float f; memcpy(&f, &u, 4); return f;
}
#define IH4_STD 37837.227f
#define VTID(l, w) (((uint32_t)4 << 16) | ((uint32_t)(l) << 8) | (uint32_t)(w))
/* tensor ids of group 4 (vocoder): layer field = component, which = tensor */
enum { VC_CODEBOOK = 0 /* +q */, VC_PRE = 32, VC_TFM = 40 /* +layer */, VC_FINAL_NORM = 60, VC_UP = 64 /* +u */, VC_DEC_IN = 72,
       VC_BLK = 80 /* +4*b + {0: convT, 1..3: res unit} */, VC_OUT = 120 };
enum { VW_W = 0, VW_B = 1, VW_IN_NORM = 2, VW_Q = 3, VW_K = 4, VW_V = 5, VW_O = 6, VW_LS_ATTN = 7, VW_POST_NORM = 8, VW_GATE = 9,
       VW_UP = 10, VW_DOWN = 11, VW_LS_MLP = 12, VW_DW_W = 13, VW_DW_B = 14, VW_LN_W = 15, VW_LN_B = 16, VW_PW1 = 17, VW_PW1_B = 18,
       VW_PW2 = 19, VW_PW2_B = 20, VW_GAMMA = 21, VW_ALPHA = 22, VW_BETA = 23, VW_W2 = 24, VW_B2 = 25, VW_ALPHA2 = 26, VW_BETA2 = 27 };

static int g_vthreads = 1;
static unsigned g_vf32_mask = 0; static int g_vgroup = 0;  /* per-stage-group override (q3o_vocoder_set_arith_mask): bit g set = group g keeps f32 inputs */
static int g_vf32 = 0;  /* 1: GEMM / conv inputs stay f32 (q3o_vocoder_set_arith): the f32 arithmetic of the reference's ORT CPU path, up to summation order */

/* bf16-representable matrix [rows][cols], std = gain / sqrt(fan_in) */
static float* gen_mat(uint64_t seed, uint32_t tid, size_t rows, size_t cols, int fan_in, float gain) {
    float* p = malloc(rows * cols * 4);
    const float scale = (gain / sqrtf((float)fan_in)) / IH4_STD;
    for (size_t i = 0; i < rows * cols; ++i) p[i] = rb(q3o_synth(seed, tid, i, scale));
    return p;
}
static float* gen_vec(uint64_t seed, uint32_t tid, size_t n, float base, float std) {
    float* p = malloc(n * 4);
    const float scale = std / IH4_STD;
    for (size_t i = 0; i < n; ++i) p[i] = base + q3o_synth(seed, tid, i, scale);
    return p;
}
/* SnakeBeta parameters are stored in log scale; both sides evaluate exp() in double on the host */
static void snake_params(uint64_t seed, uint32_t ta, uint32_t tb, int C, float** ea, float** ib) {
    float* a = gen_vec(seed, ta, C, 0.0f, 0.1f); float* b = gen_vec(seed, tb, C, 0.0f, 0.1f);
    *ea = malloc((size_t)C * 4); *ib = malloc((size_t)C * 4);
    for (int i = 0; i < C; ++i) { (*ea)[i] = (float)exp((double)a[i]); (*ib)[i] = (float)(1.0 / (exp((double)b[i]) + 1e-9)); }
    free(a); free(b);
}

typedef struct { int ntap, dil, cin, nout; float* w; /* [ntap][nout][cin] */ float* b; /* [bias_n] */ int bias_n; } conv_t;
static conv_t gen_conv(uint64_t seed, int comp, int ww, int wb, int ntap, int dil, int cin, int nout, int bias_n, float gain) {
    conv_t c; c.ntap = ntap; c.dil = dil; c.cin = cin; c.nout = nout; c.bias_n = bias_n;
    c.w = gen_mat(seed, VTID(comp, ww), (size_t)ntap * nout, cin, ntap * cin, gain);
    c.b = bias_n ? gen_vec(seed, VTID(comp, wb), bias_n, 0.0f, 0.02f) : NULL;
    return c;
}
/* out[t][n] = bias[n % bias_n] + sum_tap sum_ci rb(X[t - (ntap-1-tap)*dil][ci]) * W[tap][n][ci]; X rows < 0 are zero */
static void conv_fwd(const conv_t* c, const float* x, int T, float* out) {
    float* xr = malloc((size_t)T * c->cin * 4);
    for (size_t i = 0; i < (size_t)T * c->cin; ++i) xr[i] = (g_vf32 || ((g_vf32_mask >> g_vgroup) & 1u)) ? x[i] : rb(x[i]);
#pragma omp parallel for schedule(static) num_threads(g_vthreads)
    for (int t = 0; t < T; ++t)
        for (int n = 0; n < c->nout; ++n) {
            float acc = 0.0f;
            for (int tap = 0; tap < c->ntap; ++tap) {
                const int ts = t - (c->ntap - 1 - tap) * c->dil;
                if (ts < 0) continue;
                const float* xp = xr + (size_t)ts * c->cin;
                const float* wp = c->w + ((size_t)tap * c->nout + n) * c->cin;
                float s = 0.0f;
                for (int ci = 0; ci < c->cin; ++ci) s += xp[ci] * wp[ci];
                acc += s;
            }
            out[(size_t)t * c->nout + n] = acc + (c->b ? c->b[n % c->bias_n] : 0.0f);
        }
    free(xr);
}
static void snake(float* x, int T, int C, const float* ea, const float* ib) {
    for (int t = 0; t < T; ++t)
        for (int c = 0; c < C; ++c) { float v = x[(size_t)t * C + c]; float s = sinf(v * ea[c]); x[(size_t)t * C + c] = v + ib[c] * (s * s); }
}
static void rmsnorm_rows(const float* x, int T, int d, const float* w, float eps, float* y) {
    for (int t = 0; t < T; ++t) {
        float ss = 0.0f;
        for (int i = 0; i < d; ++i) ss += x[(size_t)t * d + i] * x[(size_t)t * d + i];
        const float rinv = 1.0f / sqrtf(ss / (float)d + eps);
        for (int i = 0; i < d; ++i) y[(size_t)t * d + i] = (x[(size_t)t * d + i] * rinv) * w[i];
    }
}

typedef struct {
    float *in_norm, *post_norm, *ls_attn, *ls_mlp; conv_t q, k, v, o, gate, up, down;
} vlayer;
typedef struct { conv_t ct; float *dw_w, *dw_b, *ln_w, *ln_b, *gamma; conv_t pw1, pw2; int r; } vup;
typedef struct { float *ea, *ib, *ea2, *ib2; conv_t c1, c2; } vres;
typedef struct { float *ea, *ib; conv_t ct; vres res[3]; int r, cin, cout; } vblk;

struct q3o_vocoder {
    q3o_vocoder_config c; uint64_t seed;
    float** cb; conv_t pre; vlayer* L; float* final_norm; vup* U; conv_t dec_in; vblk* B; float *oea, *oib; conv_t out;
    int32_t* codes; int n_frames, cap, emitted_frames; int spf;
};

q3o_vocoder* q3o_vocoder_create(const q3o_vocoder_config* c, uint64_t seed, int32_t n_threads) {
    q3o_vocoder* v = calloc(1, sizeof(*v));
    v->c = *c; v->seed = seed; g_vthreads = n_threads > 0 ? n_threads : 1;
    const int d = c->latent_dim, H = c->n_head * c->head_dim;
    v->cb = calloc(c->n_codebooks, sizeof(float*));
    for (int q = 0; q < c->n_codebooks; ++q) v->cb[q] = gen_mat(seed, VTID(VC_CODEBOOK + q, VW_W), c->codebook_size, c->codebook_dim, 16, 1.0f);
    v->pre = gen_conv(seed, VC_PRE, VW_W, VW_B, c->pre_conv_kernel, 1, c->codebook_dim, d, d, 1.0f);
    v->L = calloc(c->n_layer, sizeof(vlayer));
    for (int l = 0; l < c->n_layer; ++l) {
        vlayer* y = &v->L[l]; const int comp = VC_TFM + l;
        y->in_norm = gen_vec(seed, VTID(comp, VW_IN_NORM), d, 1.0f, 0.05f);
        y->post_norm = gen_vec(seed, VTID(comp, VW_POST_NORM), d, 1.0f, 0.05f);
        y->ls_attn = gen_vec(seed, VTID(comp, VW_LS_ATTN), d, c->layer_scale_init, 0.1f * c->layer_scale_init);
        y->ls_mlp = gen_vec(seed, VTID(comp, VW_LS_MLP), d, c->layer_scale_init, 0.1f * c->layer_scale_init);
        y->q = gen_conv(seed, comp, VW_Q, 0, 1, 1, d, H, 0, 1.0f); y->k = gen_conv(seed, comp, VW_K, 0, 1, 1, d, H, 0, 1.0f);
        y->v = gen_conv(seed, comp, VW_V, 0, 1, 1, d, H, 0, 1.0f); y->o = gen_conv(seed, comp, VW_O, 0, 1, 1, H, d, 0, 1.0f);
        y->gate = gen_conv(seed, comp, VW_GATE, 0, 1, 1, d, c->d_ffn, 0, 1.0f); y->up = gen_conv(seed, comp, VW_UP, 0, 1, 1, d, c->d_ffn, 0, 1.0f);
        y->down = gen_conv(seed, comp, VW_DOWN, 0, 1, 1, c->d_ffn, d, 0, 1.0f);
    }
    v->final_norm = gen_vec(seed, VTID(VC_FINAL_NORM, VW_W), d, 1.0f, 0.05f);
    v->U = calloc(c->n_upsample ? c->n_upsample : 1, sizeof(vup));
    v->spf = 1;
    for (int u = 0; u < c->n_upsample; ++u) {
        vup* p = &v->U[u]; const int comp = VC_UP + u, r = c->upsample_ratios[u]; p->r = r; v->spf *= r;
        p->ct = gen_conv(seed, comp, VW_W, VW_B, 1, 1, d, r * d, d, 1.0f);
        p->dw_w = gen_vec(seed, VTID(comp, VW_DW_W), (size_t)7 * d, 0.0f, 0.3f); p->dw_b = gen_vec(seed, VTID(comp, VW_DW_B), d, 0.0f, 0.02f);
        p->ln_w = gen_vec(seed, VTID(comp, VW_LN_W), d, 1.0f, 0.05f); p->ln_b = gen_vec(seed, VTID(comp, VW_LN_B), d, 0.0f, 0.02f);
        p->pw1 = gen_conv(seed, comp, VW_PW1, VW_PW1_B, 1, 1, d, 4 * d, 4 * d, 1.0f);
        p->pw2 = gen_conv(seed, comp, VW_PW2, VW_PW2_B, 1, 1, 4 * d, d, d, 1.0f);
        p->gamma = gen_vec(seed, VTID(comp, VW_GAMMA), d, 0.1f, 0.01f);
    }
    v->dec_in = gen_conv(seed, VC_DEC_IN, VW_W, VW_B, 7, 1, d, c->decoder_dim, c->decoder_dim, 1.0f);
    v->B = calloc(c->n_dec_blocks ? c->n_dec_blocks : 1, sizeof(vblk));
    int ch = c->decoder_dim;
    for (int b = 0; b < c->n_dec_blocks; ++b) {
        vblk* k = &v->B[b]; const int comp = VC_BLK + 4 * b, r = c->dec_rates[b]; k->r = r; k->cin = ch; k->cout = ch / 2; v->spf *= r;
        snake_params(seed, VTID(comp, VW_ALPHA), VTID(comp, VW_BETA), ch, &k->ea, &k->ib);
        k->ct = gen_conv(seed, comp, VW_W, VW_B, 2, 1, ch, r * k->cout, k->cout, 1.0f);
        const int dil[3] = {1, 3, 9};
        for (int u = 0; u < 3; ++u) {
            vres* s = &k->res[u]; const int rc = comp + 1 + u;
            snake_params(seed, VTID(rc, VW_ALPHA), VTID(rc, VW_BETA), k->cout, &s->ea, &s->ib);
            s->c1 = gen_conv(seed, rc, VW_W, VW_B, 7, dil[u], k->cout, k->cout, k->cout, 0.5f);
            snake_params(seed, VTID(rc, VW_ALPHA2), VTID(rc, VW_BETA2), k->cout, &s->ea2, &s->ib2);
            s->c2 = gen_conv(seed, rc, VW_W2, VW_B2, 1, 1, k->cout, k->cout, k->cout, 0.5f);
        }
        ch = k->cout;
    }
    snake_params(seed, VTID(VC_OUT, VW_ALPHA), VTID(VC_OUT, VW_BETA), ch, &v->oea, &v->oib);
    v->out = gen_conv(seed, VC_OUT, VW_W, VW_B, 7, 1, ch, 1, 1, 0.1f);
    return v;
}
static void free_conv(conv_t* c) { free(c->w); free(c->b); }
void q3o_vocoder_destroy(q3o_vocoder* v) {
    if (!v) return;
    const q3o_vocoder_config* c = &v->c;
    for (int q = 0; q < c->n_codebooks; ++q) free(v->cb[q]);
    free(v->cb); free_conv(&v->pre);
    for (int l = 0; l < c->n_layer; ++l) { vlayer* y = &v->L[l]; free(y->in_norm); free(y->post_norm); free(y->ls_attn); free(y->ls_mlp);
        free_conv(&y->q); free_conv(&y->k); free_conv(&y->v); free_conv(&y->o); free_conv(&y->gate); free_conv(&y->up); free_conv(&y->down); }
    free(v->L); free(v->final_norm);
    for (int u = 0; u < c->n_upsample; ++u) { vup* p = &v->U[u]; free_conv(&p->ct); free(p->dw_w); free(p->dw_b); free(p->ln_w); free(p->ln_b);
        free(p->gamma); free_conv(&p->pw1); free_conv(&p->pw2); }
    free(v->U); free_conv(&v->dec_in);
    for (int b = 0; b < c->n_dec_blocks; ++b) { vblk* k = &v->B[b]; free(k->ea); free(k->ib); free_conv(&k->ct);
        for (int u = 0; u < 3; ++u) { vres* s = &k->res[u]; free(s->ea); free(s->ib); free(s->ea2); free(s->ib2); free_conv(&s->c1); free_conv(&s->c2); } }
    free(v->B); free(v->oea); free(v->oib); free_conv(&v->out); free(v->codes); free(v);
}
void q3o_vocoder_reset(q3o_vocoder* v) { v->n_frames = 0; v->emitted_frames = 0; }
/* synthetic tensors by (component, which) id for the family test: the same generator calls as q3o_vocoder_create */
void q3o_vocoder_mat(const q3o_vocoder* v, int32_t comp, int32_t which, int64_t rows, int64_t cols, int32_t fan_in, float gain, float* out) {
    float* p = gen_mat(v->seed, VTID(comp, which), (size_t)rows, (size_t)cols, fan_in, gain); memcpy(out, p, (size_t)rows * cols * 4); free(p);
}
void q3o_vocoder_vec(const q3o_vocoder* v, int32_t comp, int32_t which, int64_t n, float base, float std, float* out) {
    float* p = gen_vec(v->seed, VTID(comp, which), (size_t)n, base, std); memcpy(out, p, (size_t)n * 4); free(p);
}
/* 0 (default): GEMM / conv inputs rounded to bf16 like the device path; 1: plain f32 — used to state how far the bf16 vocoder is from
 * f32 arithmetic (the reference runs the graph in f32 on the ORT CPU provider, src/models/onnx.rs:47-62) and by the family test */
void q3o_vocoder_set_arith(q3o_vocoder* v, int32_t f32_inputs) { (void)v; g_vf32 = f32_inputs ? 1 : 0; }
/* Error budget of the bf16 operand rounding, stage group by stage group: bit g of `mask` set = the convolutions / GEMMs of group g take f32
 * inputs while the others round theirs to bf16. Groups: 0 pre-conv + transformer, 1 up-sampling stages, 2 decoder input convolution,
 * 3 + b decoder block b (transposed convolution + three residual units), 3 + n_dec_blocks output convolution. */
void q3o_vocoder_set_arith_mask(q3o_vocoder* v, uint32_t mask) { (void)v; g_vf32_mask = mask; }

/* whole-utterance decode of frames [0, T): returns malloc'd pcm of T*spf samples */
/* stage / stage_out (tests): 1 the transformer's input [T][d], 2 its output after the final norm [T][d], 3 the up-sampled latent
 * [T * prod(upsample_ratios)][d], 4 the PCM before the clamp */
static int g_stage = 0; static float* g_stage_out = NULL;
#define STAGE(k, ptr, n) do { if (g_stage == (k) && g_stage_out) memcpy(g_stage_out, (ptr), (size_t)(n) * 4); } while (0)
static float* decode_all(q3o_vocoder* v, int T) {
    const q3o_vocoder_config* c = &v->c;
    const int d = c->latent_dim, H = c->n_head, hd = c->head_dim, HH = H * hd, F = c->d_ffn, W = c->sliding_window;
    /* V1: sum of the 16 codebook rows (q ascending) */
    float* e = calloc((size_t)T * c->codebook_dim, 4);
    for (int t = 0; t < T; ++t)
        for (int q = 0; q < c->n_codebooks; ++q) {
            int code = v->codes[t * c->n_codebooks + q];
            if (code < 0) code = 0; if (code >= c->codebook_size) code = c->codebook_size - 1;
            const float* row = v->cb[q] + (size_t)code * c->codebook_dim;
            for (int i = 0; i < c->codebook_dim; ++i) e[(size_t)t * c->codebook_dim + i] += row[i];
        }
    /* V2: causal pre-conv */
    g_vgroup = 0;
    float* x = malloc((size_t)T * d * 4);
    conv_fwd(&v->pre, e, T, x); free(e);
    STAGE(1, x, (size_t)T * d);
    /* V3: sliding-window transformer */
    float* xn = malloc((size_t)T * d * 4); float* q = malloc((size_t)T * HH * 4); float* k = malloc((size_t)T * HH * 4);
    float* vv = malloc((size_t)T * HH * 4); float* att = malloc((size_t)T * HH * 4); float* y = malloc((size_t)T * d * 4);
    float* g = malloc((size_t)T * F * 4); float* u = malloc((size_t)T * F * 4);
    const int half = hd / 2;
    for (int l = 0; l < c->n_layer; ++l) {
        vlayer* L = &v->L[l];
        rmsnorm_rows(x, T, d, L->in_norm, c->rms_eps, xn);
        conv_fwd(&L->q, xn, T, q); conv_fwd(&L->k, xn, T, k); conv_fwd(&L->v, xn, T, vv);
        for (int t = 0; t < T; ++t)
            for (int h = 0; h < H; ++h)
                for (int i = 0; i < half; ++i) {
                    const double inv = pow((double)c->rope_theta, -2.0 * (double)i / (double)hd), ang = (double)t * inv;
                    const float cs = (float)cos(ang), sn = (float)sin(ang);
                    float* qp = q + (size_t)t * HH + h * hd; float* kp = k + (size_t)t * HH + h * hd;
                    float a = qp[i], b = qp[i + half]; qp[i] = a * cs - b * sn; qp[i + half] = b * cs + a * sn;
                    a = kp[i]; b = kp[i + half]; kp[i] = a * cs - b * sn; kp[i + half] = b * cs + a * sn;
                }
        const float scale = 1.0f / sqrtf((float)hd);
        for (int t = 0; t < T; ++t)
            for (int h = 0; h < H; ++h) {
                const int j0 = t - W + 1 > 0 ? t - W + 1 : 0;
                float sc[512]; float m = -INFINITY;
                for (int j = j0; j <= t; ++j) {
                    float s = 0.0f;
                    for (int i = 0; i < hd; ++i) s += q[(size_t)t * HH + h * hd + i] * k[(size_t)j * HH + h * hd + i];
                    sc[j - j0] = s * scale; if (sc[j - j0] > m) m = sc[j - j0];
                }
                float l = 0.0f;
                for (int j = j0; j <= t; ++j) { sc[j - j0] = expf(sc[j - j0] - m); l += sc[j - j0]; }
                for (int i = 0; i < hd; ++i) {
                    float o = 0.0f;
                    for (int j = j0; j <= t; ++j) o += sc[j - j0] * vv[(size_t)j * HH + h * hd + i];
                    att[(size_t)t * HH + h * hd + i] = o / l;
                }
            }
        conv_fwd(&L->o, att, T, y);
        for (int t = 0; t < T; ++t) for (int i = 0; i < d; ++i) x[(size_t)t * d + i] += L->ls_attn[i] * y[(size_t)t * d + i];
        rmsnorm_rows(x, T, d, L->post_norm, c->rms_eps, xn);
        conv_fwd(&L->gate, xn, T, g); conv_fwd(&L->up, xn, T, u);
        for (size_t i = 0; i < (size_t)T * F; ++i) g[i] = (g[i] / (1.0f + expf(-g[i]))) * u[i];
        conv_fwd(&L->down, g, T, y);
        for (int t = 0; t < T; ++t) for (int i = 0; i < d; ++i) x[(size_t)t * d + i] += L->ls_mlp[i] * y[(size_t)t * d + i];
    }
    rmsnorm_rows(x, T, d, v->final_norm, c->rms_eps, xn);
    STAGE(2, xn, (size_t)T * d);
    free(q); free(k); free(vv); free(att); free(y); free(g); free(u); free(x);
    /* V5a: upsample stages */
    g_vgroup = 1;
    float* cur = xn; int Tc = T;
    for (int s = 0; s < c->n_upsample; ++s) {
        vup* p = &v->U[s]; const int r = p->r;
        float* up = malloc((size_t)Tc * r * d * 4);
        conv_fwd(&p->ct, cur, Tc, up); /* [Tc][r*d] == [Tc*r][d] */
        free(cur); Tc *= r;
        float* dw = malloc((size_t)Tc * d * 4);
        for (int t = 0; t < Tc; ++t)
            for (int i = 0; i < d; ++i) {
                float a = p->dw_b[i];
                for (int tap = 0; tap < 7; ++tap) { const int ts = t - (6 - tap); if (ts >= 0) a += up[(size_t)ts * d + i] * p->dw_w[(size_t)tap * d + i]; }
                dw[(size_t)t * d + i] = a;
            }
        for (int t = 0; t < Tc; ++t) { /* LayerNorm eps 1e-6 */
            float mean = 0.0f, var = 0.0f;
            for (int i = 0; i < d; ++i) mean += dw[(size_t)t * d + i];
            mean /= (float)d;
            for (int i = 0; i < d; ++i) { const float z = dw[(size_t)t * d + i] - mean; var += z * z; }
            var /= (float)d;
            const float rinv = 1.0f / sqrtf(var + 1e-6f);
            for (int i = 0; i < d; ++i) dw[(size_t)t * d + i] = ((dw[(size_t)t * d + i] - mean) * rinv) * p->ln_w[i] + p->ln_b[i];
        }
        float* h1 = malloc((size_t)Tc * 4 * d * 4);
        conv_fwd(&p->pw1, dw, Tc, h1);
        for (size_t i = 0; i < (size_t)Tc * 4 * d; ++i) h1[i] = 0.5f * h1[i] * (1.0f + erff(h1[i] * 0.70710678118654752f));
        conv_fwd(&p->pw2, h1, Tc, dw); free(h1);
        for (int t = 0; t < Tc; ++t) for (int i = 0; i < d; ++i) up[(size_t)t * d + i] += p->gamma[i] * dw[(size_t)t * d + i];
        free(dw); cur = up;
    }
    STAGE(3, cur, (size_t)Tc * d);
    /* V5b: decoder */
    int ch = c->decoder_dim;
    float* z = malloc((size_t)Tc * ch * 4);
    g_vgroup = 2;
    conv_fwd(&v->dec_in, cur, Tc, z); free(cur);
    for (int b = 0; b < c->n_dec_blocks; ++b) {
        vblk* k2 = &v->B[b];
        g_vgroup = 3 + b;
        snake(z, Tc, k2->cin, k2->ea, k2->ib);
        float* o = malloc((size_t)Tc * k2->r * k2->cout * 4);
        conv_fwd(&k2->ct, z, Tc, o); free(z); Tc *= k2->r; ch = k2->cout;
        float* t1 = malloc((size_t)Tc * ch * 4); float* t2 = malloc((size_t)Tc * ch * 4);
        for (int s = 0; s < 3; ++s) {
            vres* rs = &k2->res[s];
            memcpy(t1, o, (size_t)Tc * ch * 4);
            snake(t1, Tc, ch, rs->ea, rs->ib);
            conv_fwd(&rs->c1, t1, Tc, t2);
            snake(t2, Tc, ch, rs->ea2, rs->ib2);
            conv_fwd(&rs->c2, t2, Tc, t1);
            for (size_t i = 0; i < (size_t)Tc * ch; ++i) o[i] += t1[i];
        }
        free(t1); free(t2); z = o;
    }
    /* V6 */
    g_vgroup = 3 + c->n_dec_blocks;
    snake(z, Tc, ch, v->oea, v->oib);
    float* pcm = malloc((size_t)Tc * 4);
    conv_fwd(&v->out, z, Tc, pcm); free(z);
    STAGE(4, pcm, Tc);
    for (int t = 0; t < Tc; ++t) { if (pcm[t] > 1.0f) pcm[t] = 1.0f; if (pcm[t] < -1.0f) pcm[t] = -1.0f; }
    return pcm;
}

/* one whole-utterance decode of `codes` with an intermediate tensor copied out (see decode_all); returns the PCM sample count */
int32_t q3o_vocoder_stage(q3o_vocoder* v, const int32_t* codes, int32_t n_frames, int32_t stage, float* out) {
    const int ncb = v->c.n_codebooks;
    int32_t* keep = v->codes; const int kn = v->n_frames;
    v->codes = (int32_t*)malloc((size_t)n_frames * ncb * sizeof(int32_t));
    memcpy(v->codes, codes, (size_t)n_frames * ncb * sizeof(int32_t));
    g_stage = stage; g_stage_out = out;
    float* pcm = decode_all(v, n_frames);
    g_stage = 0; g_stage_out = NULL;
    free(pcm); free(v->codes); v->codes = keep; v->n_frames = kn;
    return n_frames * v->spf;
}

int32_t q3o_vocoder_decode(q3o_vocoder* v, const int32_t* codes, int32_t n_frames, int32_t is_last, float* pcm_out, int32_t max_samples) {
    const int ncb = v->c.n_codebooks;
    if (v->n_frames + n_frames > v->cap) {
        v->cap = (v->n_frames + n_frames) * 2 + 16;
        v->codes = realloc(v->codes, (size_t)v->cap * ncb * sizeof(int32_t));
    }
    memcpy(v->codes + (size_t)v->n_frames * ncb, codes, (size_t)n_frames * ncb * sizeof(int32_t));
    v->n_frames += n_frames;
    int upto = is_last ? v->n_frames : v->n_frames - v->c.lookahead_frames; /* V4: frames withheld until flushed */
    if (upto <= v->emitted_frames) return 0;
    float* pcm = decode_all(v, upto);
    const int s0 = v->emitted_frames * v->spf, s1 = upto * v->spf;
    int n = s1 - s0;
    if (n > max_samples) n = max_samples;
    memcpy(pcm_out, pcm + s0, (size_t)n * 4);
    free(pcm);
    v->emitted_frames = upto;
    return n;
}
