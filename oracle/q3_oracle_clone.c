/*
 * q3_oracle_clone.c — CPU restatement of the two encoders of the voice-clone front-end.
 *
 * TEST INFRASTRUCTURE ONLY (see q3_oracle.h). PARITY UNPINNED: the reference runs qwen3_tts_speaker_encoder.onnx and
 * qwen3_tts_codec_encoder.onnx through onnxruntime (src/models/onnx.rs:82-165); neither graph is in /root/reference.
 * What the reference pins is the I/O contract — "mels" [1,T,128] -> "spk_emb" [1,2048] (src/models/onnx.rs:141-157),
 * "input_values" [1,N] 24 kHz -> "audio_codes" [1,frames,16] i64 (:104-121) — and the caller (src/tts/engine.rs:324-387).
 * The layer structure follows the model families readable in this container's transformers package (structure only):
 *   speaker encoder = ECAPA-TDNN (qwen2_5_omni/modeling_qwen2_5_omni.py:2412-2700: TDNN k5 -> 3 x SE-Res2Net -> MFA ->
 *     attentive statistics pooling -> 1x1 conv), reflect "same" padding, ReLU;
 *   audio encoder = Mimi (mimi/modeling_mimi.py:210-494, 964-1140: causal SEANet encoder with ELU and strided convs,
 *     LayerNorm/GELU transformer with LayerScale, RoPE and a causal sliding window, stride-2 replicate-padded conv,
 *     split residual VQ = 1 semantic + (ncb-1) acoustic codebooks, nearest neighbour in Euclidean distance).
 * Every dimension is a config field. Numerics: every convolution is an im2col followed by the canonical
 * exact GEMM of §4.1 (bf16 weights, f32 activations, K zero-padded to a multiple of 512); all reductions are sequential
 * fmaf chains in ascending index; transcendental functions are built on q3o_expf. The HIP kernels follow the same order,
 * so codes AND floats are compared bit for bit.
 */
#include "q3_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define IH4_STD 37837.227f
#define CTID(l, w) (((uint32_t)5 << 16) | ((uint32_t)(l) << 8) | (uint32_t)(w))
/* component ids (tensor-id "layer" field) */
enum { SC_TDNN0 = 0, SC_BLOCK = 1 /* +i, i = 0..2 */, SC_MFA = 8, SC_ASP_TDNN = 9, SC_ASP_CONV = 10, SC_FC = 11,
       AC_CONV0 = 32, AC_STAGE = 33 /* +4*i: res a, res b, down */, AC_LAST = 60, AC_TFM = 64 /* +layer */, AC_DOWN = 100,
       AC_SEM_PROJ = 101, AC_AC_PROJ = 102, AC_CODEBOOK = 110 /* +q */ };
enum { SW_TDNN1 = 0, SW_TDNN2 = 2, SW_SE1 = 4, SW_SE2 = 6, SW_RES2 = 16 /* +2*p */ };
enum { TW_LN1_W = 0, TW_LN1_B, TW_QKV, TW_O, TW_LS1, TW_LN2_W, TW_LN2_B, TW_FC1, TW_FC2, TW_LS2 };
enum { PAD_ZERO = 0, PAD_REFLECT = 1, PAD_REPLICATE = 2 };
enum { ACT_NONE = 0, ACT_RELU = 1, ACT_ELU = 2, ACT_TANH = 3, ACT_SIGMOID = 4, ACT_GELU = 5 };

static inline float clamp80(float x) { return fminf(fmaxf(x, -80.0f), 80.0f); }
static float c_tanh(float x) {
    const float a = fminf(fabsf(x), 40.0f);
    const float e = q3o_expf(-2.0f * a);
    const float t = (1.0f - e) / (1.0f + e);
    return x < 0.0f ? -t : t;
}
static float c_act(float x, int kind) {
    switch (kind) {
    case ACT_RELU: return x > 0.0f ? x : 0.0f;
    case ACT_ELU: return x > 0.0f ? x : q3o_expf(clamp80(x)) - 1.0f;
    case ACT_TANH: return c_tanh(x);
    case ACT_SIGMOID: return 1.0f / (1.0f + q3o_expf(-clamp80(x)));
    case ACT_GELU: {
        float u = x * x; u = u * x;
        const float inner = 0.7978845608f * fmaf(0.044715f, u, x);
        return (0.5f * x) * (1.0f + c_tanh(inner));
    }
    default: return x;
    }
}

typedef struct { uint16_t* w; float* b; int cin, n, k, stride, dil, padl, mode, kp; } cconv;
static int round512(int k) { return (k + 511) / 512 * 512; }
/* W [n][kp] bf16, column j*cin + c = tap j, channel c; std = 1/sqrt(k*cin); bias N(0, 0.02) or none (wb < 0) */
static cconv mk_conv(uint64_t seed, int comp, int ww, int wb, int cin, int n, int k, int stride, int dil, int padl, int mode) {
    cconv c; c.cin = cin; c.n = n; c.k = k; c.stride = stride; c.dil = dil; c.padl = padl; c.mode = mode; c.kp = round512(k * cin);
    const float scale = (1.0f / sqrtf((float)(k * cin))) / IH4_STD;
    c.w = malloc((size_t)n * c.kp * 2);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)n * c.kp; ++i) c.w[i] = q3o_bf16(q3o_synth(seed, CTID(comp, ww), (uint64_t)i, scale));
    c.b = NULL;
    if (wb >= 0) { c.b = malloc((size_t)n * 4); q3o_synth_fill(seed, CTID(comp, wb), (uint64_t)n, 0.0f, 0.02f, 0, c.b); }
    return c;
}
static void free_conv(cconv* c) { free(c->w); free(c->b); }
static int pad_index(int i, int T, int mode) {
    if (i >= 0 && i < T) return i;
    if (mode == PAD_ZERO) return -1;
    if (mode == PAD_REFLECT) i = i < 0 ? -i : 2 * (T - 1) - i;
    return i < 0 ? 0 : (i > T - 1 ? T - 1 : i);
}
/* dst[t][dcol + n] = bias[n] + sum_{j,c} act(src[pad(t*stride + j*dil - padl)][scol + c]) * W[n][j*cin + c] */
static void conv_run(const cconv* c, const float* src, int lds, int scol, int T_in, int pre_act, float* dst, int ldd, int dcol,
                     int T_out) {
    float* A = calloc((size_t)T_out * c->kp, 4);
#pragma omp parallel for schedule(static)
    for (int t = 0; t < T_out; ++t)
        for (int j = 0; j < c->k; ++j) {
            const int i = pad_index(t * c->stride + j * c->dil - c->padl, T_in, c->mode);
            if (i < 0) continue;
            for (int ch = 0; ch < c->cin; ++ch)
                A[(size_t)t * c->kp + (size_t)j * c->cin + ch] = c_act(src[(size_t)i * lds + scol + ch], pre_act);
        }
    float* Y = malloc((size_t)T_out * c->n * 4);
    q3o_gemm_exact(A, T_out, c->kp, c->w, c->n, NULL, 0.0f, c->b, 0, Y, NULL);
    for (int t = 0; t < T_out; ++t) memcpy(dst + (size_t)t * ldd + dcol, Y + (size_t)t * c->n, (size_t)c->n * 4);
    free(Y); free(A);
}
static void act_run(float* x, int ld, int col, int n, int T, int kind) {
    for (int t = 0; t < T; ++t) for (int i = 0; i < n; ++i) x[(size_t)t * ld + col + i] = c_act(x[(size_t)t * ld + col + i], kind);
}
/* same-length reflect conv of the ECAPA family: total pad dil*(k-1), left half */
static cconv mk_tdnn(uint64_t seed, int comp, int ww, int cin, int n, int k, int dil) {
    return mk_conv(seed, comp, ww, ww + 1, cin, n, k, 1, dil, dil * (k - 1) / 2, PAD_REFLECT);
}
/* weighted statistics over time of one channel: mean = sum w_t x_t, std = sqrt(max(sum w_t (x_t - mean)^2, 1e-12));
 * w == NULL: w_t = 1/T */
static void wstats(const float* x, int ld, int T, const float* w, int ldw, float* mean, float* std) {
    const float u = 1.0f / (float)T;
    float m = 0.0f;
    for (int t = 0; t < T; ++t) m = fmaf(w ? w[(size_t)t * ldw] : u, x[(size_t)t * ld], m);
    float v = 0.0f;
    for (int t = 0; t < T; ++t) { const float d = x[(size_t)t * ld] - m; v = fmaf(w ? w[(size_t)t * ldw] : u, d * d, v); }
    *mean = m; *std = sqrtf(fmaxf(v, 1e-12f));
}

/* ------------------------------------------------------------------------------------------ */
/* speaker encoder                                                                             */
/* ------------------------------------------------------------------------------------------ */
int32_t q3o_speaker_encode(const q3o_clone_config* c, uint64_t seed, const float* mel, int32_t T, float* out) {
    if (T < 1) return -1;
    const int C = c->se_channels[0], C4 = c->se_channels[4];
    float* h[4];
    for (int i = 0; i < 4; ++i) h[i] = malloc((size_t)T * C * 4);
    cconv t0 = mk_tdnn(seed, SC_TDNN0, 0, c->mel_dim, C, c->se_kernels[0], c->se_dilations[0]);
    conv_run(&t0, mel, c->mel_dim, 0, T, ACT_NONE, h[0], C, 0, T);
    act_run(h[0], C, 0, C, T, ACT_RELU);
    free_conv(&t0);
    float* y = malloc((size_t)T * C * 4); float* r2 = malloc((size_t)T * C * 4); float* tmp = malloc((size_t)T * C * 4);
    float* z = malloc((size_t)T * C * 4);
    const int S = c->se_res2net_scale, wpart = C / S, SE = c->se_se_channels;
    for (int i = 1; i <= 3; ++i) {
        const int comp = SC_BLOCK + i - 1;
        cconv a = mk_tdnn(seed, comp, SW_TDNN1, C, C, 1, 1);
        conv_run(&a, h[i - 1], C, 0, T, ACT_NONE, y, C, 0, T); act_run(y, C, 0, C, T, ACT_RELU); free_conv(&a);
        /* Res2Net: part 0 passes through; part p = relu(conv(y_p (+ out_{p-1} for p >= 2))) */
        for (int t = 0; t < T; ++t) memcpy(r2 + (size_t)t * C, y + (size_t)t * C, (size_t)wpart * 4);
        for (int p = 1; p < S; ++p) {
            for (int t = 0; t < T; ++t)
                for (int ch = 0; ch < wpart; ++ch)
                    tmp[(size_t)t * wpart + ch] = p == 1 ? y[(size_t)t * C + p * wpart + ch]
                                                         : y[(size_t)t * C + p * wpart + ch] + r2[(size_t)t * C + (p - 1) * wpart + ch];
            cconv rc = mk_tdnn(seed, comp, SW_RES2 + 2 * p, wpart, wpart, c->se_kernels[i], c->se_dilations[i]);
            conv_run(&rc, tmp, wpart, 0, T, ACT_NONE, r2, C, p * wpart, T); act_run(r2, C, p * wpart, wpart, T, ACT_RELU);
            free_conv(&rc);
        }
        cconv b = mk_tdnn(seed, comp, SW_TDNN2, C, C, 1, 1);
        conv_run(&b, r2, C, 0, T, ACT_NONE, z, C, 0, T); act_run(z, C, 0, C, T, ACT_RELU); free_conv(&b);
        /* squeeze-excitation: channel means over time -> 1x1 -> relu -> 1x1 -> sigmoid -> scale */
        float* m = malloc((size_t)C * 4); float* s1 = malloc((size_t)SE * 4); float* s2 = malloc((size_t)C * 4); float sd;
        for (int ch = 0; ch < C; ++ch) wstats(z + ch, C, T, NULL, 0, &m[ch], &sd);
        cconv e1 = mk_tdnn(seed, comp, SW_SE1, C, SE, 1, 1), e2 = mk_tdnn(seed, comp, SW_SE2, SE, C, 1, 1);
        conv_run(&e1, m, C, 0, 1, ACT_NONE, s1, SE, 0, 1); act_run(s1, SE, 0, SE, 1, ACT_RELU);
        conv_run(&e2, s1, SE, 0, 1, ACT_NONE, s2, C, 0, 1); act_run(s2, C, 0, C, 1, ACT_SIGMOID);
        free_conv(&e1); free_conv(&e2);
        for (int t = 0; t < T; ++t)
            for (int ch = 0; ch < C; ++ch) h[i][(size_t)t * C + ch] = z[(size_t)t * C + ch] * s2[ch] + h[i - 1][(size_t)t * C + ch];
        free(m); free(s1); free(s2);
    }
    /* multi-layer feature aggregation over the three block outputs */
    float* cat = malloc((size_t)T * 3 * C * 4);
    for (int t = 0; t < T; ++t) for (int i = 0; i < 3; ++i) memcpy(cat + ((size_t)t * 3 + i) * C, h[i + 1] + (size_t)t * C, (size_t)C * 4);
    float* x = malloc((size_t)T * C4 * 4);
    cconv mfa = mk_tdnn(seed, SC_MFA, 0, 3 * C, C4, c->se_kernels[4], c->se_dilations[4]);
    conv_run(&mfa, cat, 3 * C, 0, T, ACT_NONE, x, C4, 0, T); act_run(x, C4, 0, C4, T, ACT_RELU); free_conv(&mfa);
    /* attentive statistics pooling */
    const int AC = c->se_attn_channels;
    float* att_in = malloc((size_t)T * 3 * C4 * 4);
    for (int ch = 0; ch < C4; ++ch) {
        float m, sd; wstats(x + ch, C4, T, NULL, 0, &m, &sd);
        for (int t = 0; t < T; ++t) {
            att_in[(size_t)t * 3 * C4 + ch] = x[(size_t)t * C4 + ch];
            att_in[(size_t)t * 3 * C4 + C4 + ch] = m;
            att_in[(size_t)t * 3 * C4 + 2 * C4 + ch] = sd;
        }
    }
    float* a1 = malloc((size_t)T * AC * 4); float* e = malloc((size_t)T * C4 * 4);
    cconv at = mk_tdnn(seed, SC_ASP_TDNN, 0, 3 * C4, AC, 1, 1), ac = mk_tdnn(seed, SC_ASP_CONV, 0, AC, C4, 1, 1);
    conv_run(&at, att_in, 3 * C4, 0, T, ACT_NONE, a1, AC, 0, T); act_run(a1, AC, 0, AC, T, ACT_RELU); act_run(a1, AC, 0, AC, T, ACT_TANH);
    conv_run(&ac, a1, AC, 0, T, ACT_NONE, e, C4, 0, T);
    free_conv(&at); free_conv(&ac);
    float* pooled = malloc((size_t)2 * C4 * 4);
    for (int ch = 0; ch < C4; ++ch) { /* softmax over time, then weighted mean / std */
        float mx = e[ch];
        for (int t = 1; t < T; ++t) mx = fmaxf(mx, e[(size_t)t * C4 + ch]);
        float l = 0.0f;
        for (int t = 0; t < T; ++t) { const float p = q3o_expf(e[(size_t)t * C4 + ch] - mx); e[(size_t)t * C4 + ch] = p; l += p; }
        for (int t = 0; t < T; ++t) e[(size_t)t * C4 + ch] = e[(size_t)t * C4 + ch] / l;
        wstats(x + ch, C4, T, e + ch, C4, &pooled[ch], &pooled[C4 + ch]);
    }
    cconv fc = mk_tdnn(seed, SC_FC, 0, 2 * C4, c->se_dim, 1, 1);
    conv_run(&fc, pooled, 2 * C4, 0, 1, ACT_NONE, out, c->se_dim, 0, 1); free_conv(&fc);
    free(pooled); free(e); free(a1); free(att_in); free(x); free(cat); free(z); free(tmp); free(r2); free(y);
    for (int i = 0; i < 4; ++i) free(h[i]);
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* audio encoder                                                                               */
/* ------------------------------------------------------------------------------------------ */
static int ceil_div(int64_t a, int b) { return (int)((a + b - 1) / b); }
int32_t q3o_audio_frames(const q3o_clone_config* c, int64_t n) {
    if (n < 1) return 0;
    int64_t T = n;
    for (int i = 0; i < c->ae_n_ratios; ++i) T = ceil_div(T, c->ae_ratios[i]);
    return ceil_div(T, c->ae_down_stride);
}
/* causal conv of the SEANet family: left pad (k-1)*dil + 1 - stride, T_out = ceil(T/stride), right side padded by `mode` */
static cconv mk_causal(uint64_t seed, int comp, int ww, int wb, int cin, int n, int k, int stride, int dil, int mode) {
    return mk_conv(seed, comp, ww, wb, cin, n, k, stride, dil, (k - 1) * dil + 1 - stride, mode);
}
static float* gen_vecf(uint64_t seed, uint32_t tid, size_t n, float base, float std) {
    float* p = malloc(n * 4); q3o_synth_fill(seed, tid, n, base, std, 0, p); return p;
}
static void layernorm_rows(const float* x, int T, int d, const float* w, const float* b, float eps, float* y) {
    for (int t = 0; t < T; ++t) {
        const float* r = x + (size_t)t * d;
        float s = 0.0f;
        for (int i = 0; i < d; ++i) s += r[i];
        const float mean = s / (float)d;
        float v = 0.0f;
        for (int i = 0; i < d; ++i) { const float dx = r[i] - mean; v = fmaf(dx, dx, v); }
        const float rinv = 1.0f / sqrtf(v / (float)d + eps);
        for (int i = 0; i < d; ++i) y[(size_t)t * d + i] = ((r[i] - mean) * rinv) * w[i] + b[i];
    }
}
/* plain linear layer through the same exact GEMM (k = 1 "conv", no bias) */
static void linear(uint64_t seed, int comp, int ww, const float* x, int T, int din, int dout, float* y) {
    cconv l = mk_conv(seed, comp, ww, -1, din, dout, 1, 1, 1, 0, PAD_ZERO);
    conv_run(&l, x, din, 0, T, ACT_NONE, y, dout, 0, T); free_conv(&l);
}

int32_t q3o_audio_encode(const q3o_clone_config* c, uint64_t seed, const float* pcm, int64_t n, int32_t* codes, int32_t cap,
                         float* latent_out) {
    const int nf = q3o_audio_frames(c, n);
    if (nf < 1) return 0;
    if (nf > cap) return -1;
    int T = (int)n, C = c->ae_filters;
    float* x = malloc((size_t)T * C * 4);
    cconv c0 = mk_causal(seed, AC_CONV0, 0, 1, 1, C, c->ae_kernel, 1, 1, PAD_ZERO);
    conv_run(&c0, pcm, 1, 0, T, ACT_NONE, x, C, 0, T); free_conv(&c0);
    for (int i = 0; i < c->ae_n_ratios; ++i) {
        const int r = c->ae_ratios[i], comp = AC_STAGE + 4 * i;
        float* y = malloc((size_t)T * (C / 2) * 4); float* z = malloc((size_t)T * C * 4);
        cconv ra = mk_causal(seed, comp, 0, 1, C, C / 2, c->ae_res_kernel, 1, 1, PAD_ZERO);
        cconv rb = mk_causal(seed, comp + 1, 0, 1, C / 2, C, 1, 1, 1, PAD_ZERO);
        conv_run(&ra, x, C, 0, T, ACT_ELU, y, C / 2, 0, T);
        conv_run(&rb, y, C / 2, 0, T, ACT_ELU, z, C, 0, T);
        for (size_t k = 0; k < (size_t)T * C; ++k) x[k] = x[k] + z[k];
        free_conv(&ra); free_conv(&rb); free(y); free(z);
        const int T2 = ceil_div(T, r);
        float* d = malloc((size_t)T2 * 2 * C * 4);
        cconv dn = mk_causal(seed, comp + 2, 0, 1, C, 2 * C, 2 * r, r, 1, PAD_ZERO);
        conv_run(&dn, x, C, 0, T, ACT_ELU, d, 2 * C, 0, T2); free_conv(&dn);
        free(x); x = d; T = T2; C *= 2;
    }
    const int H = c->ae_hidden;
    float* hcur = malloc((size_t)T * H * 4);
    cconv cl = mk_causal(seed, AC_LAST, 0, 1, C, H, c->ae_last_kernel, 1, 1, PAD_ZERO);
    conv_run(&cl, x, C, 0, T, ACT_ELU, hcur, H, 0, T); free_conv(&cl); free(x);

    /* transformer: pre-LN, RoPE (pairs i, i + hd/2), causal sliding window, GELU MLP, LayerScale */
    const int nh = c->ae_n_head, hd = c->ae_head_dim, dq = nh * hd, F = c->ae_d_ffn, W = c->ae_window, half = hd / 2;
    float* cs = malloc((size_t)T * half * 4); float* sn = malloc((size_t)T * half * 4);
    for (int t = 0; t < T; ++t)
        for (int i = 0; i < half; ++i) {
            const double a = (double)t * pow((double)c->ae_rope_theta, -2.0 * (double)i / (double)hd);
            cs[(size_t)t * half + i] = (float)cos(a); sn[(size_t)t * half + i] = (float)sin(a);
        }
    float* nrm = malloc((size_t)T * H * 4); float* qkv = malloc((size_t)T * 3 * dq * 4); float* att = malloc((size_t)T * dq * 4);
    float* o = malloc((size_t)T * H * 4); float* f1 = malloc((size_t)T * F * 4); float* sc = malloc((size_t)(W > T ? W : T) * 4);
    const float qscale = 1.0f / sqrtf((float)hd);
    for (int l = 0; l < c->ae_n_layer; ++l) {
        const int comp = AC_TFM + l;
        float* w1 = gen_vecf(seed, CTID(comp, TW_LN1_W), H, 1.0f, 0.05f); float* b1 = gen_vecf(seed, CTID(comp, TW_LN1_B), H, 0.0f, 0.02f);
        float* w2 = gen_vecf(seed, CTID(comp, TW_LN2_W), H, 1.0f, 0.05f); float* b2 = gen_vecf(seed, CTID(comp, TW_LN2_B), H, 0.0f, 0.02f);
        float* ls1 = gen_vecf(seed, CTID(comp, TW_LS1), H, c->ae_layer_scale, 0.1f * c->ae_layer_scale);
        float* ls2 = gen_vecf(seed, CTID(comp, TW_LS2), H, c->ae_layer_scale, 0.1f * c->ae_layer_scale);
        layernorm_rows(hcur, T, H, w1, b1, c->ae_ln_eps, nrm);
        linear(seed, comp, TW_QKV, nrm, T, H, 3 * dq, qkv);
        for (int t = 0; t < T; ++t)
            for (int part = 0; part < 2; ++part)
                for (int hh = 0; hh < nh; ++hh) {
                    float* v = qkv + (size_t)t * 3 * dq + part * dq + hh * hd;
                    for (int i = 0; i < half; ++i) {
                        const float cc = cs[(size_t)t * half + i], ss = sn[(size_t)t * half + i], a = v[i], b = v[i + half];
                        v[i] = a * cc - b * ss; v[i + half] = b * cc + a * ss;
                    }
                }
        for (int t = 0; t < T; ++t)
            for (int hh = 0; hh < nh; ++hh) {
                const float* q = qkv + (size_t)t * 3 * dq + hh * hd;
                const int j0 = t - W + 1 > 0 ? t - W + 1 : 0;
                float mx = 0.0f;
                for (int j = j0; j <= t; ++j) {
                    const float* kk = qkv + (size_t)j * 3 * dq + dq + hh * hd;
                    float s = 0.0f;
                    for (int i = 0; i < hd; ++i) s = fmaf(q[i], kk[i], s);
                    s = s * qscale; sc[j - j0] = s; mx = j == j0 ? s : fmaxf(mx, s);
                }
                float lsum = 0.0f;
                for (int j = j0; j <= t; ++j) { sc[j - j0] = q3o_expf(sc[j - j0] - mx); lsum += sc[j - j0]; }
                for (int i = 0; i < hd; ++i) {
                    float acc = 0.0f;
                    for (int j = j0; j <= t; ++j) acc = fmaf(sc[j - j0], qkv[(size_t)j * 3 * dq + 2 * dq + hh * hd + i], acc);
                    att[(size_t)t * dq + hh * hd + i] = acc / lsum;
                }
            }
        linear(seed, comp, TW_O, att, T, dq, H, o);
        for (int t = 0; t < T; ++t) for (int i = 0; i < H; ++i) hcur[(size_t)t * H + i] = hcur[(size_t)t * H + i] + ls1[i] * o[(size_t)t * H + i];
        layernorm_rows(hcur, T, H, w2, b2, c->ae_ln_eps, nrm);
        linear(seed, comp, TW_FC1, nrm, T, H, F, f1); act_run(f1, F, 0, F, T, ACT_GELU);
        linear(seed, comp, TW_FC2, f1, T, F, H, o);
        for (int t = 0; t < T; ++t) for (int i = 0; i < H; ++i) hcur[(size_t)t * H + i] = hcur[(size_t)t * H + i] + ls2[i] * o[(size_t)t * H + i];
        free(w1); free(b1); free(w2); free(b2); free(ls1); free(ls2);
    }
    free(nrm); free(qkv); free(att); free(o); free(f1); free(sc); free(cs); free(sn);

    /* frame-rate conv (replicate padding, no bias) */
    const int st = c->ae_down_stride, Tf = ceil_div(T, st);
    float* lat = malloc((size_t)Tf * H * 4);
    cconv dn = mk_causal(seed, AC_DOWN, 0, -1, H, H, 2 * st, st, 1, PAD_REPLICATE);
    conv_run(&dn, hcur, H, 0, T, ACT_NONE, lat, H, 0, Tf); free_conv(&dn); free(hcur);
    if (latent_out) memcpy(latent_out, lat, (size_t)Tf * H * 4);

    /* split residual VQ: codebook 0 on the semantic projection, codebooks 1.. on the acoustic projection's residual */
    const int D = c->ae_vq_dim, ncb = c->ae_n_codebooks, CS = c->ae_codebook_size;
    float* ps = malloc((size_t)Tf * D * 4); float* pa = malloc((size_t)Tf * D * 4);
    linear(seed, AC_SEM_PROJ, 0, lat, Tf, H, D, ps);
    linear(seed, AC_AC_PROJ, 0, lat, Tf, H, D, pa);
    for (int q = 0; q < ncb; ++q) {
        float* cb = gen_vecf(seed, CTID(AC_CODEBOOK + q, 0), (size_t)CS * D, 0.0f, 1.0f / sqrtf((float)D));
        for (int t = 0; t < Tf; ++t) {
            float* r = (q == 0 ? ps : pa) + (size_t)t * D;
            int best = 0; float bd = 0.0f;
            for (int j = 0; j < CS; ++j) {
                float dist = 0.0f;
                for (int i = 0; i < D; ++i) { const float e = r[i] - cb[(size_t)j * D + i]; dist = fmaf(e, e, dist); }
                if (j == 0 || dist < bd) { bd = dist; best = j; }
            }
            codes[(size_t)t * ncb + q] = best;
            if (q > 0) for (int i = 0; i < D; ++i) r[i] = r[i] - cb[(size_t)best * D + i];
        }
        free(cb);
    }
    free(ps); free(pa); free(lat);
    return Tf;
}
