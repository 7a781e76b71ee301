// q3_mel.hip — log-mel front-end of the voice-clone path (the part the reference pins in-repo).
//
// Replaces SpeakerEncoder::compute_mel, /root/reference/src/models/onnx.rs:166-321 (CPU, rustfft, one frame at a time):
// 24 kHz, n_fft 1024, hop 256, 128 Slaney mels, reflect padding with the reference's index rules, periodic Hann,
// sqrt(|X|^2 + 1e-9), filterbank sum in ascending k, ln(max(., 1e-5)). One workgroup per frame: the windowed frame and the
// twiddle table live in LDS, 513 bins are a direct DFT (fmaf chains over n ascending, twiddle index (k*n) mod 1024 — the
// order the oracle restates, so both agree bit for bit before the final logf), then 128 filterbank sums.
// A 3 s reference clip is 282 frames x 1.05 MFLOP: latency, not throughput, is what matters here.
#include <cmath>
#include <vector>

#include "q3_engine.h"

#define MEL_NFFT 1024
#define MEL_HOP 256
#define MEL_NMELS 128
#define MEL_NBINS 513
#define MEL_PAD ((MEL_NFFT - MEL_HOP) / 2)

namespace {

float hz_to_mel(float freq) {  // :182-193
    const float f_min = 0.0f, f_sp = 200.0f / 3.0f, min_log_hz = 1000.0f;
    const float min_log_mel = (min_log_hz - f_min) / f_sp, logstep = logf(6.4f) / 27.0f;
    return freq >= min_log_hz ? min_log_mel + (logf(freq / min_log_hz) / logstep) : (freq - f_min) / f_sp;
}
float mel_to_hz(float mel) {  // :196-207
    const float f_min = 0.0f, f_sp = 200.0f / 3.0f, min_log_hz = 1000.0f;
    const float min_log_mel = (min_log_hz - f_min) / f_sp, logstep = logf(6.4f) / 27.0f;
    return mel >= min_log_mel ? min_log_hz * expf(logstep * (mel - min_log_mel)) : f_min + f_sp * mel;
}

}  // namespace

struct Q3Mel {
    float *hann = nullptr, *cs = nullptr, *sn = nullptr, *fb = nullptr;  // device tables
    float *audio = nullptr, *out = nullptr; size_t audio_cap = 0, out_cap = 0;
};

// padded signal at index j (:237-252): reflect at the start, the reference's saturating rule at the end
__device__ __forceinline__ float mel_padded_at(const float* audio, long long n, long long j) {
    if (j < MEL_PAD) { const long long i = MEL_PAD - j; return i < n ? audio[i] : 0.0f; }
    if (j < MEL_PAD + n) return audio[j - MEL_PAD];
    const long long i = j - MEL_PAD - n + 1;
    const long long idx = n >= 1 + i ? n - 1 - i : 0;
    return idx < n ? audio[idx] : 0.0f;
}

__global__ __launch_bounds__(256) void k_mel(const float* audio, long long n, const float* hann, const float* cs, const float* sn,
                                             const float* fb, float* out) {
    __shared__ float xw[MEL_NFFT], ct[MEL_NFFT], st[MEL_NFFT], mag[MEL_NBINS + 3];
    const int f = blockIdx.x, tid = threadIdx.x;
    const long long start = (long long)f * MEL_HOP;
    for (int i = tid; i < MEL_NFFT; i += 256) { xw[i] = mel_padded_at(audio, n, start + i) * hann[i]; ct[i] = cs[i]; st[i] = sn[i]; }
    __syncthreads();
    for (int k = tid; k < MEL_NBINS; k += 256) {
        float re = 0.0f, im = 0.0f;
        for (int t = 0; t < MEL_NFFT; ++t) {
            const int j = (k * t) & (MEL_NFFT - 1);
            re = fmaf(xw[t], ct[j], re);
            im = fmaf(xw[t], -st[j], im);
        }
        mag[k] = sqrtf((re * re + im * im) + 1e-9f);
    }
    __syncthreads();
    if (tid < MEL_NMELS) {
        const float* w = fb + (size_t)tid * MEL_NBINS;
        float acc = 0.0f;
        for (int k = 0; k < MEL_NBINS; ++k) acc = acc + w[k] * mag[k];
        out[(size_t)f * MEL_NMELS + tid] = logf(fmaxf(acc, 1e-5f));
    }
}

static int mel_init(q3tts_engine* e) {
    if (e->mel) return Q3TTS_OK;
    std::vector<float> hann(MEL_NFFT), cs(MEL_NFFT), sn(MEL_NFFT), fb((size_t)MEL_NMELS * MEL_NBINS);
    const float mel_min = hz_to_mel(0.0f), mel_max = hz_to_mel(12000.0f);
    float edges[MEL_NMELS + 2];
    for (int i = 0; i <= MEL_NMELS + 1; ++i) edges[i] = mel_to_hz(mel_min + (mel_max - mel_min) * (float)i / (float)(MEL_NMELS + 1));  // :215-219
    for (int m = 0; m < MEL_NMELS; ++m) {  // :227-245
        const float fl = edges[m], fc = edges[m + 1], fr = edges[m + 2], norm = 2.0f / (fr - fl);
        for (int k = 0; k < MEL_NBINS; ++k) {
            const float freq = (float)k * 24000.0f / (float)MEL_NFFT;
            float w = 0.0f;
            if (freq >= fl && freq <= fc) w = (freq - fl) / (fc - fl);
            else if (freq > fc && freq <= fr) w = (fr - freq) / (fr - fc);
            fb[(size_t)m * MEL_NBINS + k] = w * norm;
        }
    }
    for (int i = 0; i < MEL_NFFT; ++i) {
        hann[i] = 0.5f * (1.0f - cosf(2.0f * 3.14159265358979323846f * (float)i / (float)MEL_NFFT));  // :255-257
        cs[i] = (float)cos(2.0 * 3.14159265358979323846 * (double)i / (double)MEL_NFFT);
        sn[i] = (float)sin(2.0 * 3.14159265358979323846 * (double)i / (double)MEL_NFFT);
    }
    Q3Mel* m = new Q3Mel();
    e->mel = m;
    auto up = [&](float** dst, const std::vector<float>& h) -> int {
        Q3_HIP(e, hipMalloc((void**)dst, h.size() * 4));
        Q3_HIP(e, hipMemcpy(*dst, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        return Q3TTS_OK;
    };
    int rc;
    if ((rc = up(&m->hann, hann)) || (rc = up(&m->cs, cs)) || (rc = up(&m->sn, sn)) || (rc = up(&m->fb, fb))) return rc;
    return Q3TTS_OK;
}

void q3_mel_destroy(q3tts_engine* e) {
    Q3Mel* m = e->mel;
    if (!m) return;
    hipFree(m->hann); hipFree(m->cs); hipFree(m->sn); hipFree(m->fb); hipFree(m->audio); hipFree(m->out);
    delete m;
    e->mel = nullptr;
}

extern "C" int32_t q3tts_mel_frames(int64_t n_samples) {  // :248-262
    if (n_samples < 0) return 0;
    const int64_t padded = n_samples + 2 * MEL_PAD;
    return (int32_t)((padded >= MEL_NFFT ? padded - MEL_NFFT : 0) / MEL_HOP + 1);
}

// log-mel of a host clip, left on the device (m->out, [nf][128]); no synchronisation — the caller's work on e->stream follows it
int q3_mel_run(q3tts_engine* e, const float* audio, int64_t n_samples, int32_t* n_frames, float** out_dev) {
    const int64_t padded = n_samples + 2 * MEL_PAD;
    int32_t nf = q3tts_mel_frames(n_samples);
    if ((int64_t)(nf - 1) * MEL_HOP + MEL_NFFT > padded) nf -= 1;  // the `break` of :266-269 (only for inputs shorter than one window)
    *n_frames = nf;
    *out_dev = nullptr;
    if (nf <= 0) return Q3TTS_OK;
    int rc = mel_init(e);
    if (rc != Q3TTS_OK) return rc;
    Q3Mel* m = e->mel;
    const size_t na = (size_t)std::max<int64_t>(n_samples, 1), no = (size_t)nf * MEL_NMELS;
    if (m->audio_cap < na) { hipFree(m->audio); m->audio = nullptr; m->audio_cap = 0; Q3_HIP(e, hipMalloc((void**)&m->audio, na * 4)); m->audio_cap = na; }
    if (m->out_cap < no) { hipFree(m->out); m->out = nullptr; m->out_cap = 0; Q3_HIP(e, hipMalloc((void**)&m->out, no * 4)); m->out_cap = no; }
    hipStream_t s = e->stream;
    if (n_samples > 0) Q3_HIP(e, hipMemcpyAsync(m->audio, audio, (size_t)n_samples * 4, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_mel, dim3(nf), dim3(256), 0, s, m->audio, (long long)n_samples, m->hann, m->cs, m->sn, m->fb, m->out);
    Q3_HIP(e, hipGetLastError());
    *out_dev = m->out;
    return Q3TTS_OK;
}

extern "C" int q3tts_mel(q3tts_engine* e, const float* audio, int64_t n_samples, float* out, int32_t cap_frames, int32_t* n_frames) {
    if (!e || !out || !n_frames || n_samples < 0 || (n_samples > 0 && !audio)) return q3_set_err(e, Q3TTS_ERR_INVALID, "null argument");
    Q3_HIP(e, hipSetDevice(e->cfg.device));
    int32_t nf = q3tts_mel_frames(n_samples);
    if ((int64_t)(nf - 1) * MEL_HOP + MEL_NFFT > n_samples + 2 * MEL_PAD) nf -= 1;
    *n_frames = nf;
    if (nf <= 0) return Q3TTS_OK;
    if (nf > cap_frames) return q3_set_err(e, Q3TTS_ERR_INVALID, "mel output buffer too small");
    float* dev = nullptr;
    int rc = q3_mel_run(e, audio, n_samples, &nf, &dev);
    if (rc != Q3TTS_OK) return rc;
    Q3_HIP(e, hipMemcpyAsync(out, dev, (size_t)nf * MEL_NMELS * 4, hipMemcpyDeviceToHost, e->stream));
    Q3_HIP(e, hipStreamSynchronize(e->stream));
    return Q3TTS_OK;
}
