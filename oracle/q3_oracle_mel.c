/* q3_oracle_mel.c — CPU restatement of the reference's log-mel front-end (TEST INFRASTRUCTURE, see q3_oracle.h).
 *
 * Follows SpeakerEncoder::compute_mel, /root/reference/src/models/onnx.rs:166-321, statement by statement:
 * 24 kHz, n_fft 1024, hop 256, 128 Slaney mels (fmin 0, fmax 12000, Slaney norm), reflect padding of 384 samples with
 * the reference's exact index rules (:237-252), periodic Hann (:255-257), magnitude sqrt(|X|^2 + 1e-9) (:279-282),
 * mel = sum_k fb[m][k] * mag[k] in ascending k with separate multiply and add (:286-289), ln(max(mel, 1e-5)) (:291).
 * The one thing the reference does not pin is the FFT's rounding (rustfft). The spec here is a direct DFT:
 *   re_k = sum_n fmaf(xw[n],  cos_tab[(k*n) mod 1024], .),  im_k = sum_n fmaf(xw[n], -sin_tab[(k*n) mod 1024], .),
 * n ascending, tables = (float)cos/sin(2*pi*j/1024) evaluated in double. The HIP kernel (q3_mel.hip) follows the same
 * order, so kernel and oracle agree bit for bit up to the final logf.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "q3_oracle.h"

#define N_FFT 1024
#define HOP 256
#define N_MELS 128
#define N_BINS (N_FFT / 2 + 1)

static float hz_to_mel(float freq) { /* :182-193 */
    const float f_min = 0.0f, f_sp = 200.0f / 3.0f, min_log_hz = 1000.0f;
    const float min_log_mel = (min_log_hz - f_min) / f_sp, logstep = logf(6.4f) / 27.0f;
    return freq >= min_log_hz ? min_log_mel + (logf(freq / min_log_hz) / logstep) : (freq - f_min) / f_sp;
}
static float mel_to_hz(float mel) { /* :196-207 */
    const float f_min = 0.0f, f_sp = 200.0f / 3.0f, min_log_hz = 1000.0f;
    const float min_log_mel = (min_log_hz - f_min) / f_sp, logstep = logf(6.4f) / 27.0f;
    return mel >= min_log_mel ? min_log_hz * expf(logstep * (mel - min_log_mel)) : f_min + f_sp * mel;
}

/* tables shared (by value) with the device path: hann[1024], cos[1024], sin[1024], fb[128][513] */
void q3o_mel_tables(float* hann, float* cs, float* sn, float* fb) {
    const float mel_min = hz_to_mel(0.0f), mel_max = hz_to_mel(12000.0f);
    float edges[N_MELS + 2];
    for (int i = 0; i <= N_MELS + 1; ++i) edges[i] = mel_to_hz(mel_min + (mel_max - mel_min) * (float)i / (float)(N_MELS + 1)); /* :215-219 */
    for (int m = 0; m < N_MELS; ++m) { /* :227-245 */
        const float fl = edges[m], fc = edges[m + 1], fr = edges[m + 2], norm = 2.0f / (fr - fl);
        for (int k = 0; k < N_BINS; ++k) {
            const float freq = (float)k * 24000.0f / (float)N_FFT;
            float w = 0.0f;
            if (freq >= fl && freq <= fc) w = (freq - fl) / (fc - fl);
            else if (freq > fc && freq <= fr) w = (fr - freq) / (fr - fc);
            fb[m * N_BINS + k] = w * norm;
        }
    }
    for (int i = 0; i < N_FFT; ++i) {
        hann[i] = 0.5f * (1.0f - cosf(2.0f * 3.14159265358979323846f * (float)i / (float)N_FFT)); /* :255-257 */
        cs[i] = (float)cos(2.0 * 3.14159265358979323846 * (double)i / (double)N_FFT);
        sn[i] = (float)sin(2.0 * 3.14159265358979323846 * (double)i / (double)N_FFT);
    }
}

int32_t q3o_mel_frames(int64_t n_samples) { /* :248-262 */
    const int64_t padded = n_samples + 2 * ((N_FFT - HOP) / 2);
    return (int32_t)((padded >= N_FFT ? (padded - N_FFT) : 0) / HOP + 1);
}

/* sample of the padded signal at index j (:237-252) */
static float padded_at(const float* audio, int64_t n, int64_t j) {
    const int64_t pad = (N_FFT - HOP) / 2;
    if (j < pad) { const int64_t i = pad - j; return i < n ? audio[i] : 0.0f; }
    if (j < pad + n) return audio[j - pad];
    const int64_t i = j - pad - n + 1; /* 1..pad */
    const int64_t idx = n >= 1 + i ? n - 1 - i : 0; /* saturating_sub */
    return idx < n ? audio[idx] : 0.0f;
}

/* out [n_frames][128]; pre_log (optional) receives the mel sums before max/ln */
int32_t q3o_mel(const float* audio, int64_t n_samples, float* out, float* pre_log) {
    float* hann = malloc(N_FFT * 4); float* cs = malloc(N_FFT * 4); float* sn = malloc(N_FFT * 4);
    float* fb = malloc((size_t)N_MELS * N_BINS * 4);
    q3o_mel_tables(hann, cs, sn, fb);
    const int64_t padded = n_samples + 2 * ((N_FFT - HOP) / 2);
    const int32_t nf = q3o_mel_frames(n_samples);
    int32_t done = 0;
#pragma omp parallel for schedule(static)
    for (int32_t f = 0; f < nf; ++f) {
        const int64_t start = (int64_t)f * HOP;
        if (start + N_FFT > padded) continue; /* :266-269 */
        float xw[N_FFT], mag[N_BINS];
        for (int i = 0; i < N_FFT; ++i) xw[i] = padded_at(audio, n_samples, start + i) * hann[i];
        for (int k = 0; k < N_BINS; ++k) {
            float re = 0.0f, im = 0.0f;
            for (int n = 0; n < N_FFT; ++n) {
                const int j = (k * n) & (N_FFT - 1);
                re = fmaf(xw[n], cs[j], re);
                im = fmaf(xw[n], -sn[j], im);
            }
            mag[k] = sqrtf((re * re + im * im) + 1e-9f); /* norm_sqr() + 1e-9 */
        }
        for (int m = 0; m < N_MELS; ++m) {
            float acc = 0.0f;
            for (int k = 0; k < N_BINS; ++k) acc = acc + fb[m * N_BINS + k] * mag[k];
            if (pre_log) pre_log[(size_t)f * N_MELS + m] = acc;
            out[(size_t)f * N_MELS + m] = logf(fmaxf(acc, 1e-5f));
        }
    }
    for (int32_t f = 0; f < nf; ++f) if ((int64_t)f * HOP + N_FFT <= padded) done = f + 1;
    free(hann); free(cs); free(sn); free(fb);
    return done;
}
