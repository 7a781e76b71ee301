// q3_engine.hip — host side of libq3tts: weights, KV slabs, the replayable frame-step graph, the continuous
// batching loop and the C ABI of include/q3tts.h. The loop restates run_inference_stream
// (/root/reference/src/tts/engine.rs:445-656) with every per-frame decision on the device: one graph replay =
// sample -> 15 predictor passes -> feedback -> Talker step, no host round trip (the reference crosses the
// host<->backend boundary >= 33 times per frame).
#include "q3_engine.h"
#include "q3_gguf.h"

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

static thread_local std::string g_err;

int q3_set_err(q3tts_engine* e, int code, const std::string& msg) {
    if (e) e->err = msg;
    g_err = msg;
    return code;
}
extern "C" const char* q3tts_last_error(const q3tts_engine* e) { return e ? e->err.c_str() : g_err.c_str(); }

// ------------------------------------------------------------------------------------------------
// configuration
// ------------------------------------------------------------------------------------------------
extern "C" void q3tts_default_config(q3tts_engine_config* c) {
    memset(c, 0, sizeof(*c));
    q3tts_model_config& m = c->model;
    m.t_n_layer = 28; m.t_d_model = 2048; m.t_n_head = 16; m.t_n_kv_head = 8; m.t_head_dim = 128; m.t_d_ffn = 6144;
    m.t_vocab = 3072; m.t_rope_theta = 1000000.0f;
    m.t_mrope_sections[0] = 24; m.t_mrope_sections[1] = 20; m.t_mrope_sections[2] = 20; m.t_mrope_sections[3] = 0;
    m.p_n_layer = 5; m.p_d_model = 1024; m.p_n_head = 16; m.p_n_kv_head = 8; m.p_head_dim = 128; m.p_d_ffn = 3072;
    m.p_rope_theta = 1000000.0f;
    m.n_codebooks = 16; m.codebook_size = 2048; m.rms_eps = 1e-6f;
    m.d_embed = 2048; m.text_vocab = 151936; m.codec0_rows = 3072; m.codecq_rows = 2048;
    m.sample_limit = 2160; m.eos_code = 2150; m.tts_pad_id = 151671;
    q3tts_vocoder_config& v = c->vocoder;
    v.n_codebooks = 16; v.codebook_size = 2048; v.codebook_dim = 512; v.latent_dim = 1024; v.pre_conv_kernel = 3;
    v.n_layer = 8; v.n_head = 16; v.head_dim = 64; v.d_ffn = 3072; v.sliding_window = 72;
    v.rope_theta = 10000.0f; v.rms_eps = 1e-5f; v.layer_scale_init = 0.01f;
    v.n_upsample = 2; v.upsample_ratios[0] = 2; v.upsample_ratios[1] = 2;
    v.decoder_dim = 1536; v.n_dec_blocks = 4;
    v.dec_rates[0] = 8; v.dec_rates[1] = 5; v.dec_rates[2] = 4; v.dec_rates[3] = 3;
    v.lookahead_frames = 0; v.sample_rate = 24000;
    c->device = 0; c->max_batch = 1; c->n_ctx = 4096; c->max_steps_cap = 512; c->with_vocoder = 1;
    c->synth_seed = 0; c->weights_path = nullptr; c->talker_q8_0 = 0; c->vocoder_flush_tail = 0;
}

static int validate(const q3tts_engine_config& c, std::string& why) {
    const q3tts_model_config& m = c.model;
#define REQ(cond) do { if (!(cond)) { why = "config check failed: " #cond; return Q3TTS_ERR_INVALID; } } while (0)
    REQ(m.t_n_layer > 0 && m.p_n_layer > 0);
    REQ(m.t_head_dim == 128 && m.p_head_dim == 128);  // exact attention kernel: 16 lanes x 8 dims per key
    REQ(m.t_d_model % 512 == 0 && m.p_d_model % 512 == 0 && m.t_d_ffn % 512 == 0 && m.p_d_ffn % 512 == 0);
    REQ(m.t_d_model <= 8192 && m.p_d_model <= 8192);  // fused RMSNorm: a wave's share of the norm weights is one LDS strip of <= 1024 floats
    REQ((m.t_n_head * m.t_head_dim) % 512 == 0 && (m.p_n_head * m.p_head_dim) % 512 == 0);
    REQ(m.t_n_head % m.t_n_kv_head == 0 && m.p_n_head % m.p_n_kv_head == 0);
    { int r = m.t_n_head / m.t_n_kv_head; REQ(r == 1 || r == 2 || r == 4); r = m.p_n_head / m.p_n_kv_head; REQ(r == 1 || r == 2 || r == 4); }
    REQ(m.t_vocab % 16 == 0 && m.codebook_size % 16 == 0 && m.t_d_ffn % 8 == 0);
    REQ(m.d_embed == m.t_d_model);  // feedback row feeds the Talker directly (src/tts/engine.rs:631)
    REQ(m.d_embed % 512 == 0);
    REQ(m.n_codebooks >= 2 && m.n_codebooks <= 16);
    REQ(m.sample_limit > 0 && m.sample_limit <= m.t_vocab && m.sample_limit <= 4096);
    REQ(m.t_mrope_sections[0] + m.t_mrope_sections[1] + m.t_mrope_sections[2] + m.t_mrope_sections[3] == m.t_head_dim / 2);
    REQ(m.tts_pad_id >= 0 && m.tts_pad_id < m.text_vocab);
    REQ(c.max_batch >= 1 && c.max_batch <= 64);
    REQ(c.n_ctx >= 64 && c.n_ctx % 64 == 0 && c.n_ctx <= 8192);
    REQ(c.max_steps_cap >= 1 && c.max_steps_cap < c.n_ctx);
    REQ(m.n_codebooks + 1 <= 64);
    REQ(c.talker_q8_0 >= 0 && c.talker_q8_0 <= 2);
    if (c.talker_q8_0) REQ(m.t_d_model % 512 == 0 && m.t_d_ffn % 512 == 0 && (m.t_n_head * m.t_head_dim) % 512 == 0);  // Q8_0: an even number of 32-blocks per K slice
    if (c.talker_q8_0 == 2) REQ(m.t_vocab % 32 == 0 && ((m.t_n_head + 2 * m.t_n_kv_head) * m.t_head_dim) % 32 == 0 && m.t_d_ffn % 64 == 0);  // W8A8: whole 32-column blocks per workgroup
#undef REQ
    return Q3TTS_OK;
}

// Zero-filled device memory. The fill has COMPLETED when the pointer is handed out: an upload on any stream (the null stream, the
// vocoder stream, a caller's) can never be overtaken by it. (Rounds 1 and 2 filled asynchronously on e->stream, a non-blocking stream,
// and patched three call sites whose uploads were zeroed again; the allocator is the one place to close that race.)
int q3_dev_alloc_zeroed(q3tts_engine* e, void** p, size_t bytes) {
    void* q = nullptr;
    hipError_t err = hipMalloc(&q, bytes);
    if (err != hipSuccess) return q3_set_err(e, Q3TTS_ERR_OOM, std::string("hipMalloc: ") + hipGetErrorString(err));
    err = hipMemsetAsync(q, 0, bytes, e->stream);
    if (err == hipSuccess) err = hipStreamSynchronize(e->stream);
    if (err != hipSuccess) { hipFree(q); return q3_set_err(e, Q3TTS_ERR_DEVICE, std::string("hipMemset: ") + hipGetErrorString(err)); }
    *p = q;
    return Q3TTS_OK;
}
template <class T>
static int dalloc(q3tts_engine* e, T** p, size_t n) {
    void* q = nullptr;
    const int rc = q3_dev_alloc_zeroed(e, &q, n * sizeof(T) + 64);
    if (rc != Q3TTS_OK) return rc;
    *p = (T*)q;
    return Q3TTS_OK;
}
#define TRY(x) do { int rc__ = (x); if (rc__ != Q3TTS_OK) return rc__; } while (0)

// RoPE tables in double on the host (same formula the oracle restates)
static void rope_tables(int n_pos, int hd, float theta, const int* sections, std::vector<float>& cs, std::vector<float>& sn) {
    const int half = hd / 2;
    int s3 = half;
    if (sections) s3 = sections[0] + sections[1] + sections[2];
    cs.resize((size_t)n_pos * half); sn.resize((size_t)n_pos * half);
    for (int p = 0; p < n_pos; ++p)
        for (int i = 0; i < half; ++i) {
            const double inv = pow((double)theta, -2.0 * (double)i / (double)hd);
            const double ang = (i < s3) ? (double)p * inv : 0.0;
            cs[(size_t)p * half + i] = (float)cos(ang);
            sn[(size_t)p * half + i] = (float)sin(ang);
        }
}

// ---- real weights: llama.cpp's tensor names for the qwen3 architecture --------------------------
struct GgSrc {
    q3tts_engine* e; const Q3Gguf* g; const char* file;
    std::vector<uint16_t> host; uint16_t* dev[2] = {nullptr, nullptr}; size_t dev_cap[2] = {0, 0};
    ~GgSrc() { for (auto p : dev) if (p) hipFree(p); for (auto p : dev8) if (p) hipFree(p); }
    int fail(const std::string& msg) { return q3_set_err(e, Q3TTS_ERR_INVALID, std::string(file) + ": " + msg); }
    const Q3GgufTensor* need(const std::string& name, uint64_t ne0, uint64_t ne1, int* rc) {
        const Q3GgufTensor* t = g->find(name);
        if (!t) { *rc = fail("tensor '" + name + "' is missing"); return nullptr; }
        const uint64_t d1 = t->dims.size() > 1 ? t->dims[1] : 1;
        if (t->dims[0] != ne0 || d1 != ne1 || t->dims.size() > 2) {
            *rc = fail("tensor '" + name + "' has shape [" + std::to_string(d1) + "][" + std::to_string(t->dims[0]) + "], the configuration needs [" +
                       std::to_string(ne1) + "][" + std::to_string(ne0) + "]");
            return nullptr;
        }
        *rc = Q3TTS_OK;
        return t;
    }
    // f32 vector -> device
    int vec(const std::string& name, size_t n, float* dst) {
        int rc; const Q3GgufTensor* t = need(name, n, 1, &rc);
        if (!t) return rc;
        std::vector<float> h(n); std::string err;
        if (q3_gguf_to_f32(*t, h.data(), err)) return fail(err);
        Q3_HIP(e, hipMemcpy(dst, h.data(), n * 4, hipMemcpyHostToDevice));  // (h is a local: synchronous copy)
        return Q3TTS_OK;
    }
    // Q8_0 mode: a tensor stored as Q8_0 goes to the device as it is ([N][K/32] blocks of 34 bytes) into staging buffer `which`
    // (raw[which] = true); any other type is widened to bf16 as below and quantised on the device
    uint8_t* dev8[2] = {nullptr, nullptr}; size_t dev8_cap[2] = {0, 0}; bool raw[2] = {false, false};
    int mat_q8(const std::string& name, size_t N, size_t K, int which) {
        int rc; const Q3GgufTensor* t = need(name, K, N, &rc);
        if (!t) return rc;
        raw[which] = false;
        if (t->type != Q3_GGML_Q8_0) return mat(name, N, K, which);
        const size_t bytes = N * (K / 32) * 34;
        if (t->nbytes < bytes) return fail("tensor '" + name + "' is shorter than its Q8_0 shape");
        if (dev8_cap[which] < bytes) {
            if (dev8[which]) hipFree(dev8[which]);
            dev8[which] = nullptr; dev8_cap[which] = 0;
            void* p = nullptr;
            if (hipMalloc(&p, bytes) != hipSuccess) return q3_set_err(e, Q3TTS_ERR_OOM, "hipMalloc (Q8_0 staging)");
            dev8[which] = (uint8_t*)p; dev8_cap[which] = bytes;
        }
        Q3_HIP(e, hipMemcpy(dev8[which], t->data, bytes, hipMemcpyHostToDevice));
        raw[which] = true;
        return Q3TTS_OK;
    }
    // [N][K] matrix -> bf16 row-major staging buffer `which` on the device
    int mat(const std::string& name, size_t N, size_t K, int which) {
        int rc; const Q3GgufTensor* t = need(name, K, N, &rc);
        if (!t) return rc;
        host.resize(N * K); std::string err;
        if (q3_gguf_to_bf16(*t, host.data(), err)) return fail(err);
        if (dev_cap[which] < N * K) {
            if (dev[which]) hipFree(dev[which]);
            dev[which] = nullptr; dev_cap[which] = 0;
            void* p = nullptr;
            if (hipMalloc(&p, N * K * 2) != hipSuccess) return q3_set_err(e, Q3TTS_ERR_OOM, "hipMalloc (weight staging)");
            dev[which] = (uint16_t*)p; dev_cap[which] = N * K;
        }
        Q3_HIP(e, hipMemcpy(dev[which], host.data(), N * K * 2, hipMemcpyHostToDevice));
        return Q3TTS_OK;
    }
};

static int init_tfm(q3tts_engine* e, Q3Tfm& t, int grp, int L, int d, int Hq, int Hkv, int hd, int F, int head_n, float theta,
                    const int* sections, int n_ctx, int n_slots, GgSrc* gg = nullptr, int q8mode = 0) {
    const bool q8 = q8mode != 0;
    t.L = L; t.d = d; t.Hq = Hq; t.Hkv = Hkv; t.hd = hd; t.F = F; t.nq = Hq * hd; t.nkv = Hkv * hd; t.nqkv = t.nq + 2 * t.nkv;
    t.head_n = head_n; t.n_ctx = n_ctx; t.n_slots = n_slots; t.q8 = q8; t.a8 = q8mode == 2;
    const uint64_t seed = e->cfg.synth_seed;
    const float ms = 0.02f / Q3_IH4_STD, ns = 0.05f / Q3_IH4_STD;
    hipStream_t s = e->stream;
    t.attn_norm.resize(L); t.ffn_norm.resize(L); t.qn.resize(L); t.kn.resize(L);
    t.wqkv.resize(L); t.wo.resize(L); t.wgu.resize(L); t.wd.resize(L);
    if (q8) { t.sqkv.assign(L, nullptr); t.so.assign(L, nullptr); t.sgu.assign(L, nullptr); t.sd.assign(L, nullptr); }
    // a matrix [N][K]: bf16 tiles (2 bytes per weight), or in Q8_0 mode block quants (1 byte) + f16 block scales [N][K/32]
    const size_t wdiv = q8 ? 16 : 8;  // weights per uint4
    auto alloc_mat = [&](uint4** w, uint16_t** sc, size_t N, size_t K) -> int {
        TRY(dalloc(e, w, N * K / wdiv));
        if (q8) TRY(dalloc(e, sc, N * K / 32));
        return Q3TTS_OK;
    };
    auto fill = [&](Q3Fill& f, uint16_t* sc) { if (q8) { f.dst_scale = sc; q3_launch_fill_tiled_q8(f, s); } else q3_launch_fill_tiled(f, s); };
    const double bpw = q8 ? 1.0625 : 2.0;  // bytes per weight streamed by a GEMM
    for (int l = 0; l < L; ++l) {
        TRY(dalloc(e, &t.attn_norm[l], (size_t)d)); TRY(dalloc(e, &t.ffn_norm[l], (size_t)d));
        TRY(dalloc(e, &t.qn[l], (size_t)hd)); TRY(dalloc(e, &t.kn[l], (size_t)hd));
        uint16_t *sc_qkv = nullptr, *sc_o = nullptr, *sc_gu = nullptr, *sc_d = nullptr;
        TRY(alloc_mat(&t.wqkv[l], &sc_qkv, t.nqkv, d)); TRY(alloc_mat(&t.wo[l], &sc_o, d, t.nq));
        TRY(alloc_mat(&t.wgu[l], &sc_gu, (size_t)2 * F, d)); TRY(alloc_mat(&t.wd[l], &sc_d, d, F));
        if (q8) { t.sqkv[l] = sc_qkv; t.so[l] = sc_o; t.sgu[l] = sc_gu; t.sd[l] = sc_d; }
        t.weight_bytes += (size_t)(bpw * (double)((size_t)t.nqkv * d + (size_t)d * t.nq + 3ull * F * d));
        if (gg) {  // blk.N.* of a llama.cpp qwen3 GGUF (weights [out][in], NeoX RoPE: no q/k permutation)
            const std::string b = "blk." + std::to_string(l) + ".";
            Q3_HIP(e, hipStreamSynchronize(s));
            TRY(gg->vec(b + "attn_norm.weight", d, t.attn_norm[l])); TRY(gg->vec(b + "ffn_norm.weight", d, t.ffn_norm[l]));
            TRY(gg->vec(b + "attn_q_norm.weight", hd, t.qn[l])); TRY(gg->vec(b + "attn_k_norm.weight", hd, t.kn[l]));
            Q3Fill f{}; f.mode = 0;
            auto stage = [&](const std::string& name, size_t N, size_t K, int which) -> int { return q8 ? gg->mat_q8(name, N, K, which) : gg->mat(name, N, K, which); };
            auto put = [&](const std::string& name, uint4* dst, uint16_t* sc, int Ntot, int K, int row0, int rows) -> int {
                TRY(stage(name, rows, K, 0));
                f.dst = dst; f.N = Ntot; f.K = K; f.mode = 0; f.row0 = row0; f.rows = rows; f.src_b = nullptr; f.src8_b = nullptr;
                f.src_a = (q8 && gg->raw[0]) ? nullptr : gg->dev[0]; f.src8_a = (q8 && gg->raw[0]) ? gg->dev8[0] : nullptr;
                fill(f, sc);
                Q3_HIP(e, hipStreamSynchronize(s));  // the staging buffer is reused by the next tensor
                return Q3TTS_OK;
            };
            TRY(put(b + "attn_q.weight", t.wqkv[l], sc_qkv, t.nqkv, d, 0, t.nq));
            TRY(put(b + "attn_k.weight", t.wqkv[l], sc_qkv, t.nqkv, d, t.nq, t.nkv));
            TRY(put(b + "attn_v.weight", t.wqkv[l], sc_qkv, t.nqkv, d, t.nq + t.nkv, t.nkv));
            TRY(put(b + "attn_output.weight", t.wo[l], sc_o, d, t.nq, 0, d));
            TRY(stage(b + "ffn_gate.weight", F, d, 0)); TRY(stage(b + "ffn_up.weight", F, d, 1));
            f.dst = t.wgu[l]; f.N = 2 * F; f.K = d; f.mode = 1;
            f.src_a = (q8 && gg->raw[0]) ? nullptr : gg->dev[0]; f.src8_a = (q8 && gg->raw[0]) ? gg->dev8[0] : nullptr;
            f.src_b = (q8 && gg->raw[1]) ? nullptr : gg->dev[1]; f.src8_b = (q8 && gg->raw[1]) ? gg->dev8[1] : nullptr;
            if (q8 && gg->raw[0] != gg->raw[1]) return gg->fail("ffn_gate / ffn_up of block " + std::to_string(l) + " differ in type (one Q8_0, one not)");
            fill(f, sc_gu);
            Q3_HIP(e, hipStreamSynchronize(s));
            TRY(put(b + "ffn_down.weight", t.wd[l], sc_d, d, F, 0, d));
            continue;
        }
        q3_launch_fill_f32(t.attn_norm[l], d, seed, Q3_TID(grp, l, Q3W_ATTN_NORM), 1.0f, ns, 0, s);
        q3_launch_fill_f32(t.ffn_norm[l], d, seed, Q3_TID(grp, l, Q3W_FFN_NORM), 1.0f, ns, 0, s);
        q3_launch_fill_f32(t.qn[l], hd, seed, Q3_TID(grp, l, Q3W_QNORM), 1.0f, ns, 0, s);
        q3_launch_fill_f32(t.kn[l], hd, seed, Q3_TID(grp, l, Q3W_KNORM), 1.0f, ns, 0, s);
        Q3Fill f{}; f.seed = seed; f.scale = ms;
        f.dst = t.wqkv[l]; f.N = t.nqkv; f.K = d; f.mode = 0;
        f.row0 = 0; f.rows = t.nq; f.tid_a = Q3_TID(grp, l, Q3W_Q); fill(f, sc_qkv);
        f.row0 = t.nq; f.rows = t.nkv; f.tid_a = Q3_TID(grp, l, Q3W_K); fill(f, sc_qkv);
        f.row0 = t.nq + t.nkv; f.rows = t.nkv; f.tid_a = Q3_TID(grp, l, Q3W_V); fill(f, sc_qkv);
        f.dst = t.wo[l]; f.N = d; f.K = t.nq; f.row0 = 0; f.rows = d; f.tid_a = Q3_TID(grp, l, Q3W_O); fill(f, sc_o);
        f.dst = t.wgu[l]; f.N = 2 * F; f.K = d; f.mode = 1; f.tid_a = Q3_TID(grp, l, Q3W_GATE); f.tid_b = Q3_TID(grp, l, Q3W_UP);
        fill(f, sc_gu);
        f.dst = t.wd[l]; f.N = d; f.K = F; f.mode = 0; f.row0 = 0; f.rows = d; f.tid_a = Q3_TID(grp, l, Q3W_DOWN); fill(f, sc_d);
    }
    TRY(dalloc(e, &t.out_norm, (size_t)d));
    TRY(alloc_mat(&t.head, &t.shead, (size_t)head_n, d));
    t.weight_bytes += (size_t)(bpw * (double)((size_t)head_n * d));
    if (gg) {
        Q3_HIP(e, hipStreamSynchronize(s));
        TRY(gg->vec("output_norm.weight", d, t.out_norm));
        TRY(q8 ? gg->mat_q8("output.weight", head_n, d, 0) : gg->mat("output.weight", head_n, d, 0));
        Q3Fill f{}; f.dst = t.head; f.N = head_n; f.K = d; f.mode = 0; f.row0 = 0; f.rows = head_n;
        f.src_a = (q8 && gg->raw[0]) ? nullptr : gg->dev[0]; f.src8_a = (q8 && gg->raw[0]) ? gg->dev8[0] : nullptr;
        fill(f, t.shead);
        Q3_HIP(e, hipStreamSynchronize(s));
    } else {
        q3_launch_fill_f32(t.out_norm, d, seed, Q3_TID(grp, Q3_L_MODEL, Q3WM_OUT_NORM), 1.0f, ns, 0, s);
        Q3Fill f{}; f.seed = seed; f.scale = ms; f.dst = t.head; f.N = head_n; f.K = d; f.mode = 0; f.row0 = 0; f.rows = head_n;
        f.tid_a = Q3_TID(grp, Q3_L_MODEL, Q3WM_HEAD); fill(f, t.shead);
    }
    t.layer_stride = (size_t)n_slots * Hkv * n_ctx * hd;
    TRY(dalloc(e, &t.kc, t.layer_stride * L)); TRY(dalloc(e, &t.vc, t.layer_stride * L));
    std::vector<float> cs, sn;
    rope_tables(n_ctx, hd, theta, sections, cs, sn);
    TRY(dalloc(e, &t.cs, cs.size())); TRY(dalloc(e, &t.sn, sn.size()));
    Q3_HIP(e, hipMemcpyAsync(t.cs, cs.data(), cs.size() * 4, hipMemcpyHostToDevice, s));
    Q3_HIP(e, hipMemcpyAsync(t.sn, sn.data(), sn.size() * 4, hipMemcpyHostToDevice, s));
    Q3_HIP(e, hipStreamSynchronize(s));
    return Q3TTS_OK;
}
static void free_tfm(Q3Tfm& t) {
    for (auto p : t.attn_norm) hipFree(p); for (auto p : t.ffn_norm) hipFree(p); for (auto p : t.qn) hipFree(p);
    for (auto p : t.kn) hipFree(p); for (auto p : t.wqkv) hipFree(p); for (auto p : t.wo) hipFree(p);
    for (auto p : t.wgu) hipFree(p); for (auto p : t.wd) hipFree(p);
    hipFree(t.out_norm); hipFree(t.head); hipFree(t.kc); hipFree(t.vc); hipFree(t.cs); hipFree(t.sn);
    for (auto p : t.sqkv) hipFree(p); for (auto p : t.so) hipFree(p); for (auto p : t.sgu) hipFree(p); for (auto p : t.sd) hipFree(p);
    hipFree(t.shead);
}

static int alloc_scratch(q3tts_engine* e, Q3Scratch& sc, int rows, int nqkv, int nq, int F, int dmax) {
    sc.rows = rows;
    const size_t r16 = ((size_t)rows + 15) & ~(size_t)15;  // A-tiled buffers hold whole 16-row tiles
    TRY(dalloc(e, &sc.qkv, (size_t)rows * nqkv)); TRY(dalloc(e, &sc.att, r16 * nq)); TRY(dalloc(e, &sc.h, r16 * F));
    sc.rt16 = (int)(r16 / 16);
    if (e->T.a8) { TRY(dalloc(e, &sc.asc_att, r16 * (nq / 32))); TRY(dalloc(e, &sc.asc_h, r16 * (F / 32))); }  // W8A8: block scales of both operands
    return Q3TTS_OK;
}

// K1-K8 of SURVEY.md §8a: one decoder block per iteration, 5 launches — QKV GEMM (row scale from the producer's tile partials),
// attention (q/k norm + RoPE + KV append fused for decode rows), O GEMM (+ residual, + the FFN norm inputs), gate/up GEMM
// (+ SwiGLU), down GEMM (+ residual, + the next block's / the head's norm inputs). x: f32 residual rows; xb / ssp: their norm
// inputs for attn_norm[0] on entry, for out_norm on exit. Restated by oracle/q3_oracle.c tfm_layers.
// Returns the number of launches the GEMM launcher refused (a shape it cannot run: stale activations would follow silently).
static int run_layers(q3tts_engine* e, Q3Tfm& t, float* x, uint16_t* xb, float* ssp, int rows, const int* row_pos, const int* row_slot, Q3Scratch& sc,
                       hipStream_t s, bool one_row_per_slot = false, hipEvent_t* probe = nullptr, int slot_mod = 0, int pos_const = 0,
                       const int* seg = nullptr, int n_seg = 0, int seg_max_n = 0, uint16_t* xscale = nullptr, int x_rt16 = 0) {
    // W8A8 (t.a8: the Talker with talker_q8_0 = 2): xb / sc.att / sc.h hold Q8_0 blocks (int8 quants + the f16 scales xscale / sc.asc_att /
    // sc.asc_h) and every GEMM runs q3_launch_bgemm8: ggml's Q8_0 x Q8_0 arithmetic
    auto gemm = [&](Q3BGemm& g) { return t.a8 ? q3_launch_bgemm8(g, s) : q3_launch_bgemm(g, s); };
    const float eps = e->cfg.model.rms_eps;
    int bad = 0;
    const int nt = t.d / 16;
    const int once = &t == &e->T ? 1 : 0;  // the Talker's 2.8 GB stream once per frame step; the Predictor's weights are re-read 15 times (Infinity Cache)
    for (int l = 0; l < t.L; ++l) {
        Q3BGemm g{};
        g.w_once = once;
        g.a = xb; g.B = rows; g.w = t.wqkv[l]; g.wscale = t.q8 ? t.sqkv[l] : nullptr; g.K = t.d; g.N = t.nqkv; g.ssp = ssp; g.ld_ssp = nt; g.ntiles = nt; g.d_norm = t.d; g.eps = eps;
        g.epi = Q3_EPI_STORE; g.y = sc.qkv; g.ldy = t.nqkv;
        if (t.a8) { g.ascale = xscale; g.a_rt16 = x_rt16; }
        const int pk = (probe && l == 0) ? e->probe_kind : -1;  // which launch of block 0 the probe events bracket (q3tts_k_probe)
        if (pk == 1) hipEventRecord(probe[0], s);
        bad += gemm(g) != 0;
        if (pk == 1) hipEventRecord(probe[1], s);
        Q3QkPrep qp{}; qp.qkv = sc.qkv; qp.ld = t.nqkv; qp.rows = rows; qp.Hq = t.Hq; qp.Hkv = t.Hkv; qp.hd = t.hd;
        qp.qnw = t.qn[l]; qp.knw = t.kn[l]; qp.eps = eps; qp.cs = t.cs; qp.sn = t.sn;
        qp.kc = t.kc + l * t.layer_stride; qp.vc = t.vc + l * t.layer_stride; qp.n_ctx = t.n_ctx; qp.row_pos = row_pos; qp.row_slot = row_slot;
        qp.slot_mod = slot_mod; qp.pos_const = pos_const;
        const bool fused = one_row_per_slot && t.Hq / t.Hkv >= 2;
        // the Predictor's pass A: rows [0, B) at position 0 and [B, 2B) at position 1 of an empty per-frame cache: one fused launch
        const bool pair = !one_row_per_slot && slot_mod > 0 && rows == 2 * slot_mod && pos_const == 0 && t.Hq / t.Hkv == 2 && t.hd == 128;
        if (!fused && !pair) q3_launch_qk_prep(qp, s);
        Q3Attend at{}; at.qkv = sc.qkv; at.ld = t.nqkv; at.rows = rows; at.out = (float*)sc.att; at.ldo = t.nq; at.Hq = t.Hq; at.Hkv = t.Hkv; at.hd = t.hd;
        at.kc = qp.kc; at.vc = qp.vc; at.n_ctx = t.n_ctx; at.row_pos = row_pos; at.row_slot = row_slot;
        at.fused = pair ? 2 : (fused ? 1 : 0); at.prep = qp; at.out_bf16 = 1; at.slot_mod = slot_mod; at.pos_const = pos_const;
        if (t.a8) { at.out_bf16 = 2; at.out_scale = sc.asc_att; at.out_rt16 = sc.rt16; }
        if (!fused && !pair && n_seg > 0) { at.seg = seg; at.n_seg = n_seg; at.seg_max_n = seg_max_n; }  // prefill of whole prompts (admit_group): the launch's rows as per-slot runs
        if (pk == 2) hipEventRecord(probe[0], s);
        q3_launch_attend(at, s);
        if (pk == 2) hipEventRecord(probe[1], s);
        g = Q3BGemm{}; g.w_once = once; g.a = sc.att; g.B = rows; g.w = t.wo[l]; g.wscale = t.q8 ? t.so[l] : nullptr; g.K = t.nq; g.N = t.d; g.epi = Q3_EPI_RESID; g.y = x; g.ldy = t.d;
        g.yb = xb; g.nw_next = t.ffn_norm[l]; g.ssp_out = ssp; g.ld_ssp_out = nt;
        if (t.a8) { g.ascale = sc.asc_att; g.a_rt16 = sc.rt16; g.yscale = xscale; g.y_rt16 = x_rt16; }
        if (pk == 3) hipEventRecord(probe[0], s);
        bad += gemm(g) != 0;
        if (pk == 3) hipEventRecord(probe[1], s);
        g = Q3BGemm{}; g.w_once = once; g.a = xb; g.B = rows; g.w = t.wgu[l]; g.wscale = t.q8 ? t.sgu[l] : nullptr; g.K = t.d; g.N = 2 * t.F; g.ssp = ssp; g.ld_ssp = nt; g.ntiles = nt; g.d_norm = t.d;
        g.eps = eps; g.epi = Q3_EPI_SWIGLU; g.yb = sc.h;
        if (t.a8) { g.ascale = xscale; g.a_rt16 = x_rt16; g.yscale = sc.asc_h; g.y_rt16 = sc.rt16; }
        if (pk == 0) hipEventRecord(probe[0], s);
        bad += gemm(g) != 0;
        if (pk == 0) hipEventRecord(probe[1], s);
        g = Q3BGemm{}; g.w_once = once; g.a = sc.h; g.B = rows; g.w = t.wd[l]; g.wscale = t.q8 ? t.sd[l] : nullptr; g.K = t.F; g.N = t.d; g.epi = Q3_EPI_RESID; g.y = x; g.ldy = t.d;
        g.yb = xb; g.nw_next = l + 1 < t.L ? t.attn_norm[l + 1] : t.out_norm; g.ssp_out = ssp; g.ld_ssp_out = nt;
        if (t.a8) { g.ascale = sc.asc_h; g.a_rt16 = sc.rt16; g.yscale = xscale; g.y_rt16 = x_rt16; }
        if (pk == 4) hipEventRecord(probe[0], s);
        bad += gemm(g) != 0;
        if (pk == 4) hipEventRecord(probe[1], s);
    }
    return bad;
}

// one frame: src/tts/engine.rs:545-642 for the slots [b0, b0 + nb) of one lane
// Returns the number of refused launches (0 = the frame was issued completely).
static int record_frame(q3tts_engine* e, Q3Lane& L, hipStream_t s, int B) {
    int bad = 0;
    const q3tts_model_config& m = e->cfg.model;
    const int ncb = m.n_codebooks, cbs = m.codebook_size, dp = m.p_d_model, de = m.d_embed, cap = e->cfg.max_steps_cap;
    const float eps = m.rms_eps;
    Q3Slot* slots = e->slots;
    int* codes = e->codes;
    Q3Sample sa{}; sa.logits = L.logits; sa.ld = m.t_vocab; sa.limit = m.sample_limit; sa.eos = m.eos_code; sa.slots = slots; sa.B = B; sa.row_slot = L.slot_id;
    sa.rng = e->rng; sa.codes = codes; sa.max_steps_cap = cap; sa.ncb = ncb;
    Q3PredInput pi{}; pi.xT = L.xT; pi.out_norm = e->T.out_norm; pi.eps = eps; pi.d = de; pi.codec0 = e->codec[0]; pi.codec0_rows = m.codec0_rows;
    pi.slots = slots; pi.row_slot = L.slot_id; pi.X = nullptr; pi.fb = L.fb; pi.B = B; pi.pproj0 = e->pproj[0]; pi.proj_b = e->proj_b; pi.dp = dp; pi.px = L.px;
    pi.nw = e->P.attn_norm[0]; pi.xb = L.xbP; pi.ssp = L.sspP;
    {   // H6 (src/assets_manager.rs:383-399) for the hidden rows only (every code embedding arrives pre-projected), in the same launch
        // as the sampler: the tiles normalise the Talker's raw output rows themselves
        Q3Project pj{}; pj.x = L.xT; pj.ldx = de; pj.rows = B; pj.w = e->proj_w; pj.bias = e->proj_b; pj.n_in = de; pj.n_out = dp; pj.y = L.px; pj.ldy = dp;
        pj.nw = e->P.attn_norm[0]; pj.xb = L.xbP; pj.ssp = L.sspP; pj.ld_ssp = dp / 16;  // rows [0, B) of pass A
        pj.norm_w = e->T.out_norm; pj.eps = eps;
        bad += q3_launch_sample_input(sa, pi, pj, s) != 0;
    }
    const size_t head_tile_stride = (size_t)(cbs / 16) * (dp / 32) * 64;  // uint4 per predictor head
    auto pred_next = [&](int q) {
        Q3PredNext pn{}; pn.keys = L.keys; pn.n_key_parts = cbs / 16; pn.q = q; pn.ncb = ncb; pn.codec_q = e->codec[q]; pn.rows_q = m.codecq_rows; pn.d = de;
        pn.slots = slots; pn.row_slot = L.slot_id; pn.B = B; pn.codes = codes; pn.max_steps_cap = cap; pn.fb = L.fb;
        pn.tts_pad = e->tts_pad; pn.xT = L.xT; pn.row_pos_t = L.row_pos_t; pn.pproj_q = e->pproj[q]; pn.proj_b = e->proj_b; pn.dp = dp; pn.px = L.px;
        const bool last = q == ncb - 1;
        pn.nw = last ? e->T.attn_norm[0] : e->P.attn_norm[0]; pn.xb = last ? L.xbT : L.xbP; pn.ssp = last ? L.sspT : L.sspP;
        if (last && e->T.a8) { pn.xscale = L.ascT; pn.x_rt16 = L.rt16T; }  // W8A8 Talker: its first operand as Q8_0 blocks
        q3_launch_pred_next(pn, s);
    };
    for (int q = 0; q < ncb - 1; ++q) {  // pass q produces code_{q+1}
        const int rows = q == 0 ? 2 * B : B;
        if (q > 0) pred_next(q);
        hipEvent_t* pe = nullptr;
        if (e->probe == 1 && q == 1 && B == L.nb && e->probe_i + 2 <= 8) { pe = &e->probe_ev[e->probe_i]; e->probe_i += 2; }
        // the Predictor's cache lives for one frame (src/tts/engine.rs:575: cleared per frame), so it is indexed by ROW: slot = row % B,
        // position = (q == 0 ? row / B : q + 1) — known without a load, the attention kernels request their operands at once
        bad += run_layers(e, e->P, L.px, L.xbP, L.sspP, rows, nullptr, nullptr, L.sc, s, q > 0, pe, B, q == 0 ? 0 : q + 1);
        // head q on the rows that carry the newest position (pass 0: rows [B, 2B)), argmax epilogue
        Q3BGemm g{}; g.a = L.xbP; g.a_row0 = q == 0 ? B : 0; g.B = B; g.w = e->P.head + head_tile_stride * q; g.K = dp; g.N = cbs;
        g.ssp = q == 0 ? L.sspP + (size_t)B * (dp / 16) : L.sspP; g.ld_ssp = dp / 16; g.ntiles = dp / 16; g.d_norm = dp; g.eps = eps;
        g.epi = Q3_EPI_ARGMAX; g.keys = L.keys; g.key_stride = cbs / 16;  // per-tile maxima; k_pred_next(q + 1) reduces them
        bad += q3_launch_bgemm(g, s) != 0;
    }
    pred_next(ncb - 1);
    hipEvent_t* pt = nullptr;  // probe mode 2: the Talker's layer-0 gate/up GEMM (the largest GEMM of the frame step)
    if (e->probe == 2 && B == L.nb && e->probe_i + 2 <= 8) { pt = &e->probe_ev[e->probe_i]; e->probe_i += 2; }
    bad += run_layers(e, e->T, L.xT, L.xbT, L.sspT, B, L.row_pos_t, L.slot_id, L.sc, s, true, pt, 0, 0, nullptr, 0, 0, L.ascT, L.rt16T);
    Q3BGemm g{}; g.w_once = 1; g.a = L.xbT; g.B = B; g.w = e->T.head; g.wscale = e->T.q8 ? e->T.shead : nullptr; g.K = m.t_d_model; g.N = m.t_vocab;
    g.ssp = L.sspT; g.ld_ssp = m.t_d_model / 16; g.ntiles = m.t_d_model / 16; g.d_norm = m.t_d_model; g.eps = eps;
    g.epi = Q3_EPI_STORE; g.y = L.logits; g.ldy = m.t_vocab;
    if (e->T.a8) { g.ascale = L.ascT; g.a_rt16 = L.rt16T; bad += q3_launch_bgemm8(g, s) != 0; }
    else bad += q3_launch_bgemm(g, s) != 0;
    return bad;
}

static bool file_exists(const std::string& p) { FILE* f = fopen(p.c_str(), "rb"); if (f) fclose(f); return f != nullptr; }
static uint16_t host_bf16(float f) {
    uint32_t u; memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);
    return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
static int upload_table(q3tts_engine* e, float** dst, const float* host, size_t n) {
    TRY(dalloc(e, dst, n));
    Q3_HIP(e, hipMemcpy(*dst, host, n * 4, hipMemcpyHostToDevice));  // `host` may be a temporary of the caller: synchronous copy
    return Q3TTS_OK;
}
static int upload_proj(q3tts_engine* e, const float* w, const float* b) {  // proj.weight stays f32 (src/assets_manager.rs:212-241, :383-399)
    const q3tts_model_config& m = e->cfg.model;
    TRY(upload_table(e, &e->proj_w, w, (size_t)m.p_d_model * m.d_embed));
    return upload_table(e, &e->proj_b, b, (size_t)m.p_d_model);
}
// Assets::load (src/assets_manager.rs:14-26): qwen3_assets.gguf if present, else the NPY files. Table row counts come from
// the files (they define the out-of-range rules, :419-460); a missing text table means "every id is out of range".
static int load_assets_files(q3tts_engine* e, const std::string& dir) {
    q3tts_model_config& m = e->cfg.model;
    const size_t d = (size_t)m.d_embed;
    std::vector<std::vector<float>> tabs(1 + m.n_codebooks);  // text, codec 0..
    std::vector<size_t> rows(1 + m.n_codebooks, 0);
    std::vector<float> pw, pb;
    std::string err;
    const std::string gpath = dir + "/qwen3_assets.gguf";
    if (file_exists(gpath)) {
        Q3Gguf g;
        if (g.open(gpath, err)) return q3_set_err(e, Q3TTS_ERR_INVALID, err);
        auto fetch = [&](const std::string& name, uint64_t ne0, bool required, std::vector<float>& out, size_t* nrows) -> int {
            const Q3GgufTensor* t = g.find(name);
            if (!t) return required ? q3_set_err(e, Q3TTS_ERR_INVALID, gpath + ": " + name + " (tensor) missing") : Q3TTS_OK;
            if (t->dims[0] != ne0 || t->dims.size() > 2) return q3_set_err(e, Q3TTS_ERR_INVALID, gpath + ": tensor '" + name + "' has the wrong row length");
            out.resize(t->nelem);
            if (q3_gguf_to_f32(*t, out.data(), err)) return q3_set_err(e, Q3TTS_ERR_INVALID, gpath + ": " + err);
            if (nrows) *nrows = t->dims.size() > 1 ? (size_t)t->dims[1] : 1;
            return Q3TTS_OK;
        };
        size_t pr = 0;
        TRY(fetch("proj.weight", d, true, pw, &pr));
        if (pr != (size_t)m.p_d_model) return q3_set_err(e, Q3TTS_ERR_INVALID, gpath + ": proj.weight does not have p_d_model rows");
        TRY(fetch("proj.bias", (uint64_t)m.p_d_model, true, pb, nullptr));
        TRY(fetch("text_embd", d, false, tabs[0], &rows[0]));
        for (int q = 0; q < m.n_codebooks; ++q) TRY(fetch("codec_embd." + std::to_string(q), d, true, tabs[1 + q], &rows[1 + q]));
    } else {
        auto fetch = [&](const std::string& file, bool required, std::vector<float>& out, size_t* nrows, size_t row_len) -> int {
            const std::string path = dir + "/" + file;
            if (!file_exists(path)) return required ? q3_set_err(e, Q3TTS_ERR_INVALID, "neither qwen3_assets.gguf nor " + file + " in " + dir) : Q3TTS_OK;
            std::vector<size_t> shape;
            if (q3_npy_load_f32(path, out, shape, err)) return q3_set_err(e, Q3TTS_ERR_INVALID, err);
            if (out.size() % row_len) return q3_set_err(e, Q3TTS_ERR_INVALID, path + ": size is not a multiple of the row length");
            if (nrows) *nrows = out.size() / row_len;
            return Q3TTS_OK;
        };
        size_t pr = 0, br = 0;
        TRY(fetch("proj_weight.npy", true, pw, &pr, d));
        TRY(fetch("proj_bias.npy", true, pb, &br, 1));
        if (pr != (size_t)m.p_d_model || br != (size_t)m.p_d_model) return q3_set_err(e, Q3TTS_ERR_INVALID, dir + ": projection shape does not match p_d_model");
        TRY(fetch("text_embedding_projected.npy", false, tabs[0], &rows[0], d));
        for (int q = 0; q < m.n_codebooks; ++q) TRY(fetch("codec_embedding_" + std::to_string(q) + ".npy", true, tabs[1 + q], &rows[1 + q], d));
    }
    for (int q = 2; q < m.n_codebooks; ++q)
        if (rows[1 + q] != rows[2]) return q3_set_err(e, Q3TTS_ERR_INVALID, dir + ": codec tables 1.." + std::to_string(m.n_codebooks - 1) + " differ in size");
    m.text_vocab = (int32_t)rows[0]; m.codec0_rows = (int32_t)rows[1];
    if (m.n_codebooks > 1) m.codecq_rows = (int32_t)rows[2];
    if (rows[0]) TRY(upload_table(e, &e->text, tabs[0].data(), tabs[0].size()));
    e->codec.resize(m.n_codebooks);
    for (int q = 0; q < m.n_codebooks; ++q) TRY(upload_table(e, &e->codec[q], tabs[1 + q].data(), tabs[1 + q].size()));
    TRY(upload_proj(e, pw.data(), pb.data()));
    // tts_pad = row 151671 of the text table when it is that large, else zeros (src/assets_manager.rs:244-249)
    if ((size_t)m.tts_pad_id < rows[0]) e->tts_pad = e->text + (size_t)m.tts_pad_id * d;
    else { TRY(dalloc(e, &e->tts_pad_own, d)); Q3_HIP(e, hipMemset(e->tts_pad_own, 0, d * 4)); e->tts_pad = e->tts_pad_own; }
    return Q3TTS_OK;
}

extern "C" int q3tts_engine_create(const q3tts_engine_config* cfg, q3tts_engine** out) {
    if (!cfg || !out) return q3_set_err(nullptr, Q3TTS_ERR_INVALID, "null argument");
    std::string why;
    if (validate(*cfg, why) != Q3TTS_OK) return q3_set_err(nullptr, Q3TTS_ERR_INVALID, why);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return q3_set_err(nullptr, Q3TTS_ERR_DEVICE, "no HIP device: libq3tts has no CPU fallback");
    if (cfg->device < 0 || cfg->device >= ndev) return q3_set_err(nullptr, Q3TTS_ERR_INVALID, "device ordinal out of range");
    q3tts_engine* e = new q3tts_engine();
    e->cfg = *cfg; e->cfg.weights_path = nullptr;
    e->max_steps = cfg->max_steps_cap < 512 ? cfg->max_steps_cap : 512;
    auto fail = [&](int rc) { std::string m = e->err; q3tts_engine_destroy(e); g_err = m; return rc; };
#define TRYC(x) do { int rc__ = (x); if (rc__ != Q3TTS_OK) return fail(rc__); } while (0)
#define HIPC(call) do { hipError_t er__ = (call); if (er__ != hipSuccess) { q3_set_err(e, Q3TTS_ERR_DEVICE, std::string(#call) + ": " + hipGetErrorString(er__)); return fail(Q3TTS_ERR_DEVICE); } } while (0)
    HIPC(hipSetDevice(cfg->device));
    q3_bgemm_prepare();
    HIPC(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
    // Q3TTS_VOC_SERIAL=1: the vocoder shares the decoder stream (no overlap): isolates its kernels in a profile
    // (highest / lowest stream priority for the decoder / vocoder streams was measured in both rounds: no change, left out)
    // Q3TTS_VOC_CUMASK=<hex word>: experiment — the vocoder stream only runs on the CUs whose bit is set in the word (repeated over the
    // 256 CUs), so that decoder workgroups always find CUs without long-lived vocoder workgroups (profiles/README.md, r03)
    if (getenv("Q3TTS_VOC_SERIAL") && atoi(getenv("Q3TTS_VOC_SERIAL"))) e->vstream = e->stream;
    else if (getenv("Q3TTS_VOC_CUMASK")) {
        uint32_t w = (uint32_t)strtoul(getenv("Q3TTS_VOC_CUMASK"), nullptr, 16), mask[8];
        for (auto& x : mask) x = w;
        HIPC(hipExtStreamCreateWithCUMask(&e->vstream, 8, mask));
    } else HIPC(hipStreamCreateWithFlags(&e->vstream, hipStreamNonBlocking));
    HIPC(hipEventCreate(&e->ev0)); HIPC(hipEventCreate(&e->ev1)); HIPC(hipEventCreate(&e->ev2)); HIPC(hipEventCreate(&e->ev3));
    e->fin_ev.resize(cfg->max_batch, nullptr);
    for (auto& ev : e->fin_ev) HIPC(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    q3tts_model_config& m = e->cfg.model;  // (table row counts follow the files when weights_path is given)
    const int B = cfg->max_batch;
    e->B = B;
    hipStream_t s = e->stream;
    // weights_path = the reference's quant directory (src/tts/engine.rs:91-131): qwen3_tts_talker.gguf,
    // qwen3_tts_predictor.gguf, qwen3_assets.gguf (or the NPY fallback). NULL: seeded synthetic weights.
    const std::string wdir = cfg->weights_path ? cfg->weights_path : "";
    Q3Gguf gt, gp;
    GgSrc st{e, &gt, "qwen3_tts_talker.gguf"}, sp{e, &gp, "qwen3_tts_predictor.gguf"};
    if (!wdir.empty()) {
        std::string er;
        if (gt.open(wdir + "/qwen3_tts_talker.gguf", er) || gp.open(wdir + "/qwen3_tts_predictor.gguf", er)) { q3_set_err(e, Q3TTS_ERR_INVALID, er); return fail(Q3TTS_ERR_INVALID); }
    }
    TRYC(init_tfm(e, e->T, Q3G_TALKER, m.t_n_layer, m.t_d_model, m.t_n_head, m.t_n_kv_head, m.t_head_dim, m.t_d_ffn, m.t_vocab,
                  m.t_rope_theta, m.t_mrope_sections, cfg->n_ctx, B, wdir.empty() ? nullptr : &st, cfg->talker_q8_0));
    TRYC(init_tfm(e, e->P, Q3G_PRED, m.p_n_layer, m.p_d_model, m.p_n_head, m.p_n_kv_head, m.p_head_dim, m.p_d_ffn,
                  (m.n_codebooks - 1) * m.codebook_size, m.p_rope_theta, nullptr, 64, B, wdir.empty() ? nullptr : &sp));
    // assets (F32 tables like qwen3_assets.gguf: src/assets_manager.rs:212-241; values bf16-representable)
    const uint64_t seed = cfg->synth_seed;
    const float es = 0.05f / Q3_IH4_STD;
    if (!wdir.empty()) {
        TRYC(load_assets_files(e, wdir));
        { void* cd = nullptr; HIPC(hipMalloc(&cd, sizeof(float*) * 16)); e->codec_dev = (const float**)cd; }
        HIPC(hipMemcpyAsync((void*)e->codec_dev, e->codec.data(), sizeof(float*) * m.n_codebooks, hipMemcpyHostToDevice, s));
    } else {
    TRYC(dalloc(e, &e->text, (size_t)m.text_vocab * m.d_embed));
    q3_launch_fill_f32(e->text, (size_t)m.text_vocab * m.d_embed, seed, Q3_TID(Q3G_ASSET, 0, Q3WA_TEXT), 0.0f, es, 1, s);
    e->codec.resize(m.n_codebooks);
    for (int q = 0; q < m.n_codebooks; ++q) {
        const size_t rows = q == 0 ? m.codec0_rows : m.codecq_rows;
        TRYC(dalloc(e, &e->codec[q], rows * m.d_embed));
        q3_launch_fill_f32(e->codec[q], rows * m.d_embed, seed, Q3_TID(Q3G_ASSET, 1 + q, 0), 0.0f, es, 1, s);
    }
    { void* cd = nullptr; HIPC(hipMalloc(&cd, sizeof(float*) * 16)); e->codec_dev = (const float**)cd; }
    HIPC(hipMemcpyAsync((void*)e->codec_dev, e->codec.data(), sizeof(float*) * m.n_codebooks, hipMemcpyHostToDevice, s));
    TRYC(dalloc(e, &e->proj_w, (size_t)m.p_d_model * m.d_embed));  // f32 [out][in]; synthetic values are bf16-representable like every synthetic matrix
    q3_launch_fill_f32(e->proj_w, (size_t)m.p_d_model * m.d_embed, seed, Q3_TID(Q3G_ASSET, 0, Q3WA_PROJ_W), 0.0f, 0.02f / Q3_IH4_STD, 1, s);
    TRYC(dalloc(e, &e->proj_b, (size_t)m.p_d_model));
    q3_launch_fill_f32(e->proj_b, m.p_d_model, seed, Q3_TID(Q3G_ASSET, 0, Q3WA_PROJ_B), 0.0f, 0.02f / Q3_IH4_STD, 0, s);
    e->tts_pad = e->text + (size_t)m.tts_pad_id * m.d_embed;  // src/assets_manager.rs:244-249
    }
    // pre-projected codec tables: proj(codec_q[code]) for every code, computed once with the projection kernel (a row's result
    // does not depend on the other rows, so a table row equals the on-the-fly projection bit for bit): the 15 Predictor passes
    // after the first read their input with a gather instead of a projection launch each
    e->pproj.assign(m.n_codebooks, nullptr);
    for (int q = 0; q < m.n_codebooks; ++q) {
        const int rows = q == 0 ? m.codec0_rows : m.codecq_rows;
        TRYC(dalloc(e, &e->pproj[q], (size_t)rows * m.p_d_model));
        Q3Project pj{}; pj.x = e->codec[q]; pj.ldx = m.d_embed; pj.rows = rows; pj.w = e->proj_w; pj.bias = e->proj_b; pj.n_in = m.d_embed; pj.n_out = m.p_d_model;
        pj.y = e->pproj[q]; pj.ldy = m.p_d_model;
        q3_launch_project(pj, s);
    }
    HIPC(hipStreamSynchronize(s));
    // decode state
    TRYC(dalloc(e, &e->slots, (size_t)B));
    HIPC(hipHostMalloc((void**)&e->slots_host, sizeof(Q3Slot) * 2 * B, hipHostMallocDefault));
    memset(e->slots_host, 0, sizeof(Q3Slot) * 2 * B);
    TRYC(dalloc(e, &e->codes, (size_t)B * cfg->max_steps_cap * m.n_codebooks)); TRYC(dalloc(e, &e->rng, (size_t)B * cfg->max_steps_cap));
    {
        const int nb = B;
        const int nqkv_max = std::max(e->T.nqkv, e->P.nqkv), nq_max = std::max(e->T.nq, e->P.nq), F_max = std::max(e->T.F, e->P.F);
        e->lanes.resize(1);
        Q3Lane& L = e->lanes[0];
        L.nb = nb;
        HIPC(hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking));
        HIPC(hipEventCreate(&L.ev_begin)); HIPC(hipEventCreate(&L.ev_end));
        TRYC(dalloc(e, &L.xT, (size_t)nb * m.t_d_model)); TRYC(dalloc(e, &L.logits, (size_t)nb * m.t_vocab)); TRYC(dalloc(e, &L.logits_tmp, (size_t)nb * std::max(m.t_vocab, m.t_d_model)));
        TRYC(dalloc(e, &L.fb, (size_t)nb * m.d_embed));
        TRYC(dalloc(e, &L.px, (size_t)2 * nb * m.p_d_model)); TRYC(dalloc(e, &L.keys, (size_t)nb * (m.codebook_size / 16)));
        const size_t nb16 = ((size_t)nb + 15) & ~(size_t)15, nb2_16 = ((size_t)2 * nb + 15) & ~(size_t)15;  // A-tiled buffers hold whole 16-row tiles
        TRYC(dalloc(e, &L.xbT, nb16 * m.t_d_model)); TRYC(dalloc(e, &L.sspT, (size_t)nb * (m.t_d_model / 16)));
        L.rt16T = (int)(nb16 / 16);
        if (e->T.a8) TRYC(dalloc(e, &L.ascT, nb16 * (m.t_d_model / 32)));
        TRYC(dalloc(e, &L.xbP, nb2_16 * m.p_d_model)); TRYC(dalloc(e, &L.sspP, (size_t)2 * nb * (m.p_d_model / 16)));
        TRYC(dalloc(e, &L.row_pos_t, (size_t)nb)); TRYC(dalloc(e, &L.slot_id, (size_t)nb)); TRYC(dalloc(e, &L.perm, (size_t)nb));
        TRYC(dalloc(e, &L.posA, (size_t)2 * nb)); TRYC(dalloc(e, &L.slotA, (size_t)2 * nb)); TRYC(dalloc(e, &L.pos_q, (size_t)m.n_codebooks * nb));
        std::vector<int> sid(nb), pa(2 * nb), sla(2 * nb), pq((size_t)m.n_codebooks * nb), rp(nb, -1);
        // pass A of the Predictor runs 2 rows per slot: rows [0, rows) at position 0 (the projected hidden state), rows [rows, 2 rows) at
        // position 1 (the code row); the maps follow the current row bucket (plan_rows)
        for (int b = 0; b < nb; ++b) { sid[b] = b; pa[b] = 0; pa[nb + b] = 1; sla[b] = sla[nb + b] = b; }
        for (int q = 0; q < m.n_codebooks; ++q) for (int b = 0; b < nb; ++b) pq[(size_t)q * nb + b] = q + 1;  // src/tts/engine.rs:604
        HIPC(hipMemcpyAsync(L.slot_id, sid.data(), nb * 4, hipMemcpyHostToDevice, s));
        HIPC(hipMemcpyAsync(L.posA, pa.data(), 2 * nb * 4, hipMemcpyHostToDevice, s));
        HIPC(hipMemcpyAsync(L.slotA, sla.data(), 2 * nb * 4, hipMemcpyHostToDevice, s));
        HIPC(hipMemcpyAsync(L.pos_q, pq.data(), pq.size() * 4, hipMemcpyHostToDevice, s));
        HIPC(hipMemcpyAsync(L.row_pos_t, rp.data(), nb * 4, hipMemcpyHostToDevice, s));
        HIPC(hipStreamSynchronize(s));
        TRYC(alloc_scratch(e, L.sc, 2 * nb, nqkv_max, nq_max, F_max, std::max(m.t_d_model, m.p_d_model)));
        // row buckets: 1, 2, 4, 8, then the multiples of 16 (the GEMM's row tiles are 16 wide: a 48-row step costs 3/4 of a 64-row one)
        for (int r = 1; r < nb && r < 16; r *= 2) e->buckets.push_back(r);
        for (int r = 16; r < nb; r += 16) e->buckets.push_back(r);
        e->buckets.push_back(nb);
        e->cur_bucket = (int)e->buckets.size() - 1;
        e->row_of_slot = sid; e->slot_of_row = sid;
    }
    TRYC(alloc_scratch(e, e->sc_pre, cfg->n_ctx, e->T.nqkv, e->T.nq, e->T.F, m.t_d_model));
    TRYC(dalloc(e, &e->xp, (size_t)cfg->n_ctx * m.t_d_model));
    TRYC(dalloc(e, &e->xbp, (((size_t)cfg->n_ctx + 15) & ~(size_t)15) * m.t_d_model)); TRYC(dalloc(e, &e->sspp, (size_t)cfg->n_ctx * (m.t_d_model / 16)));
    e->rt16p = (cfg->n_ctx + 15) / 16;
    if (e->T.a8) TRYC(dalloc(e, &e->ascp, (size_t)e->rt16p * 16 * (m.t_d_model / 32)));
    TRYC(dalloc(e, &e->pf_pos, (size_t)cfg->n_ctx)); TRYC(dalloc(e, &e->pf_slot, (size_t)cfg->n_ctx)); TRYC(dalloc(e, &e->pf_seg, (size_t)3 * cfg->max_batch));
    { std::vector<int> pp(cfg->n_ctx); for (int i = 0; i < cfg->n_ctx; ++i) pp[i] = i;
      HIPC(hipMemcpyAsync(e->pf_pos, pp.data(), pp.size() * 4, hipMemcpyHostToDevice, s)); HIPC(hipStreamSynchronize(s)); }
    e->prow_cap = cfg->n_ctx;
    TRYC(dalloc(e, &e->prow_dev, (size_t)e->prow_cap)); TRYC(dalloc(e, &e->spk_dev, (size_t)m.d_embed));
    TRYC(dalloc(e, &e->refcodes_dev, (size_t)cfg->n_ctx * 16));
    {   // the marker row text[151671] through the table's out-of-range rule (src/assets_manager.rs:444-460): a missing or short text
        // table gives the fallback pattern, never a null / out-of-bounds read (the clone prompt adds this row to every reference frame)
        TRYC(dalloc(e, &e->marker_row, (size_t)m.d_embed));
        const Q3PromptRow mr{1, m.tts_pad_id, 0, 0};
        HIPC(hipMemcpyAsync(e->prow_dev, &mr, sizeof(mr), hipMemcpyHostToDevice, s));
        HIPC(hipStreamSynchronize(s));
        q3_launch_prompt_rows(e->prow_dev, 1, e->text, m.text_vocab, e->codec_dev, m.codec0_rows, m.codecq_rows, m.n_codebooks, e->spk_dev, m.d_embed, e->marker_row, s);
        HIPC(hipStreamSynchronize(s));
    }
    if (cfg->with_vocoder) {
        TRYC(q3_voc_create(e));
        HIPC(hipHostMalloc((void**)&e->first_chunk_host, sizeof(float) * 4 * (size_t)q3_voc_samples_per_frame(e), hipHostMallocDefault));
    }
    // capture the frame step once per row-count bucket; every later frame is a replay (Q3TTS_NO_GRAPH=1: eager launches,
    // for profilers)
    HIPC(hipStreamSynchronize(s));
    if (!(getenv("Q3TTS_NO_GRAPH") && atoi(getenv("Q3TTS_NO_GRAPH")))) {
        Q3Lane& L = e->lanes[0];
        L.graphs.resize(e->buckets.size(), nullptr); L.execs.resize(e->buckets.size(), nullptr);
        for (size_t bi = 0; bi < e->buckets.size(); ++bi) {
            HIPC(hipStreamBeginCapture(L.stream, hipStreamCaptureModeThreadLocal));
            const int refused = record_frame(e, L, L.stream, e->buckets[bi]);
            HIPC(hipStreamEndCapture(L.stream, &L.graphs[bi]));
            if (refused) { q3_set_err(e, Q3TTS_ERR_INVALID, "frame step: " + std::to_string(refused) + " kernel launch(es) refused for this model shape"); return fail(Q3TTS_ERR_INVALID); }
            HIPC(hipGraphInstantiate(&L.execs[bi], L.graphs[bi], nullptr, nullptr, 0));
            HIPC(hipStreamSynchronize(L.stream));
        }
    }
    // algorithmic bytes of one frame step, context term added per run
    e->tm.algo_bytes_per_step = 0;
#undef TRYC
#undef HIPC
    *out = e;
    return Q3TTS_OK;
}

extern "C" void q3tts_engine_destroy(q3tts_engine* e) {
    if (!e) return;
    hipSetDevice(e->cfg.device);
    if (e->stream) hipStreamSynchronize(e->stream);
    if (e->vstream) hipStreamSynchronize(e->vstream);
    if (e->voc) q3_voc_destroy(e);
    q3_mel_destroy(e);
    q3_clone_destroy(e);
    if (e->first_chunk_host) hipHostFree(e->first_chunk_host);
    for (auto& L : e->lanes) {
        if (L.stream) hipStreamSynchronize(L.stream);
        for (auto ge : L.execs) if (ge) hipGraphExecDestroy(ge);
        for (auto gr : L.graphs) if (gr) hipGraphDestroy(gr);
        hipFree(L.logits_tmp); hipFree(L.perm);
        hipFree(L.xT); hipFree(L.logits); hipFree(L.fb); hipFree(L.px); hipFree(L.keys);
        hipFree(L.xbT); hipFree(L.sspT); hipFree(L.xbP); hipFree(L.sspP); hipFree(L.ascT);
        hipFree(L.row_pos_t); hipFree(L.slot_id); hipFree(L.posA); hipFree(L.slotA); hipFree(L.pos_q);
        hipFree(L.sc.qkv); hipFree(L.sc.att); hipFree(L.sc.h); hipFree(L.sc.asc_att); hipFree(L.sc.asc_h);
        if (L.ev_begin) hipEventDestroy(L.ev_begin); if (L.ev_end) hipEventDestroy(L.ev_end);
        if (L.stream) hipStreamDestroy(L.stream);
    }
    free_tfm(e->T); free_tfm(e->P);
    hipFree(e->text); for (auto p : e->codec) hipFree(p); for (auto p : e->pproj) hipFree(p); hipFree((void*)e->codec_dev); hipFree(e->proj_w); hipFree(e->proj_b);
    hipFree(e->tts_pad_own); hipFree(e->marker_row); hipFree(e->dev_pcm);
    hipFree(e->slots); if (e->slots_host) hipHostFree(e->slots_host);
    hipFree(e->codes); hipFree(e->rng);
    hipFree(e->sc_pre.qkv); hipFree(e->sc_pre.att); hipFree(e->sc_pre.h); hipFree(e->sc_pre.asc_att); hipFree(e->sc_pre.asc_h); hipFree(e->ascp);
    hipFree(e->xp); hipFree(e->xbp); hipFree(e->sspp); hipFree(e->pf_pos); hipFree(e->pf_slot); hipFree(e->pf_seg); hipFree(e->prow_dev); hipFree(e->spk_dev); hipFree(e->refcodes_dev);
    for (auto ev : e->fin_ev) if (ev) hipEventDestroy(ev);
    for (auto ev : e->probe_ev) if (ev) hipEventDestroy(ev);
    if (e->ev0) hipEventDestroy(e->ev0); if (e->ev1) hipEventDestroy(e->ev1); if (e->ev2) hipEventDestroy(e->ev2); if (e->ev3) hipEventDestroy(e->ev3);
    if (e->stream) hipStreamDestroy(e->stream);
    if (e->vstream && e->vstream != e->stream) hipStreamDestroy(e->vstream);
    delete e;
}

extern "C" int q3tts_set_sampler(q3tts_engine* e, float temperature, int32_t top_k, float top_p, int32_t has_seed, uint64_t seed) {
    if (!e) return q3_set_err(nullptr, Q3TTS_ERR_INVALID, "null engine");
    e->temperature = temperature; e->top_k = top_k; e->top_p = top_p; e->has_seed = has_seed; e->seed = seed;
    return Q3TTS_OK;
}
extern "C" int q3tts_set_max_steps(q3tts_engine* e, int32_t n) {
    if (!e) return q3_set_err(nullptr, Q3TTS_ERR_INVALID, "null engine");
    if (n < 0 || n > e->cfg.max_steps_cap) return q3_set_err(e, Q3TTS_ERR_INVALID, "max_steps exceeds max_steps_cap");
    e->max_steps = n;
    return Q3TTS_OK;
}
extern "C" void q3tts_free(void* p) { free(p); }
// Device-resident results for multi-GPU gathers: with enable = 1 every q3tts_generate_batch call also keeps each
// request's PCM in row i of an engine-owned device buffer [n][stride] f32 (valid until the next call); requests with want_pcm = 2
// skip the host copy altogether.
extern "C" int q3tts_set_device_pcm(q3tts_engine* e, int32_t enable) {
    if (!e) return q3_set_err(nullptr, Q3TTS_ERR_INVALID, "null engine");
    e->dev_pcm_on = enable ? 1 : 0;
    return Q3TTS_OK;
}
extern "C" int q3tts_get_device_pcm(q3tts_engine* e, float** base, int64_t* stride_samples, int32_t* n_rows) {
    if (!e || !base || !stride_samples || !n_rows) return q3_set_err(e, Q3TTS_ERR_INVALID, "null argument");
    *base = e->dev_pcm; *stride_samples = (int64_t)e->dev_pcm_stride; *n_rows = e->dev_pcm_n;
    return Q3TTS_OK;
}

// ------------------------------------------------------------------------------------------------
// H1 prompt builder: row list on the host (src/tts/prompt.rs:141-277, :28-118), gathers on the device
// ------------------------------------------------------------------------------------------------
enum { PAD = 2148, BOS = 2149, THINK = 2154, NOTHINK = 2155, THINK_BOS = 2156, THINK_EOS = 2157, CODEC_BOS_ICL = 2160 };
enum { BOS_TOKEN = 151672, EOS_TOKEN = 151673 };

static int build_prompt_dev(q3tts_engine* e, const q3tts_prompt_desc* p, float* out, int max_rows, int* n_out) {
    const q3tts_model_config& m = e->cfg.model;
    if (!p || (p->n_text > 0 && !p->text_ids)) return q3_set_err(e, Q3TTS_ERR_INVALID, "prompt: text_ids missing");
    const int marker = m.tts_pad_id;
    std::vector<Q3PromptRow> rows;
    int ref_row0 = -1;
    auto T1 = [&](int id) { rows.push_back({1, id, 0, 0}); };
    auto MC = [&](int cid) { rows.push_back({1, marker, 2, cid}); };       // marker + codec0[cid]
    auto TP = [&](int tid) { rows.push_back({1, tid, 2, PAD}); };          // text[tid] + codec0[PAD]
    if (p->instruct_ids) {  // :153-169
        T1(151644); T1(872); T1(198);
        for (int i = 0; i < p->n_instruct; ++i) T1((int)p->instruct_ids[i]);
        T1(151645); T1(198);
    }
    T1(151644); T1(77091); T1(198);  // :171-175
    if (p->lang_id >= 0) { MC(THINK); MC(THINK_BOS); MC(p->lang_id); MC(THINK_EOS); }  // :180-191
    else { MC(NOTHINK); MC(THINK_BOS); MC(THINK_EOS); }                                // :192-204
    if (p->spk_id >= 0) MC(p->spk_id);                                                 // :207-214
    else if (p->spk_emb) rows.push_back({1, marker, -1, 0});                           // :215-222
    if (p->ref_codes) {  // build_clone_prompt :38-106
        TP(BOS_TOKEN);
        for (int i = 0; i < p->n_ref_text; ++i) TP((int)p->ref_text_ids[i]);
        TP(EOS_TOKEN);
        MC(CODEC_BOS_ICL);
        ref_row0 = (int)rows.size();
        for (int i = 0; i < p->n_ref_frames; ++i) rows.push_back({-2, 0, 0, 0});  // filled by the frame kernel
        MC(PAD);
    }
    TP(BOS_TOKEN);                                             // :229-239
    for (int i = 0; i < p->n_text; ++i) TP((int)p->text_ids[i]);  // :241-245
    TP(EOS_TOKEN);                                             // :247-254
    MC(BOS);                                                   // :256-264
    const int n = (int)rows.size();
    if (n > max_rows || n > e->prow_cap) return q3_set_err(e, Q3TTS_ERR_INVALID, "prompt longer than n_ctx");
    hipStream_t s = e->stream;
    Q3_HIP(e, hipMemcpyAsync(e->prow_dev, rows.data(), sizeof(Q3PromptRow) * n, hipMemcpyHostToDevice, s));
    if (p->spk_emb) Q3_HIP(e, hipMemcpyAsync(e->spk_dev, p->spk_emb, (size_t)m.d_embed * 4, hipMemcpyHostToDevice, s));
    Q3_HIP(e, hipStreamSynchronize(s));  // rows is a stack vector
    q3_launch_prompt_rows(e->prow_dev, n, e->text, m.text_vocab, e->codec_dev, m.codec0_rows, m.codecq_rows, m.n_codebooks, e->spk_dev,
                          m.d_embed, out, s);
    if (ref_row0 >= 0 && p->n_ref_frames > 0) {
        Q3_HIP(e, hipMemcpyAsync(e->refcodes_dev, p->ref_codes, (size_t)p->n_ref_frames * 16 * 4, hipMemcpyHostToDevice, s));
        Q3_HIP(e, hipStreamSynchronize(s));
        q3_launch_prompt_ref_frames(e->refcodes_dev, p->n_ref_frames, e->marker_row, e->codec_dev, m.codec0_rows,
                                    m.codecq_rows, m.n_codebooks, m.d_embed, out + (size_t)ref_row0 * m.d_embed, s);
    }
    *n_out = n;
    return Q3TTS_OK;
}

extern "C" int q3tts_build_prompt(q3tts_engine* e, const q3tts_prompt_desc* p, float** out_embd, int32_t* out_n) {
    if (!e || !p || !out_embd || !out_n) return q3_set_err(e, Q3TTS_ERR_INVALID, "null argument");
    Q3_HIP(e, hipSetDevice(e->cfg.device));
    int n = 0;
    TRY(build_prompt_dev(e, p, e->xp, e->cfg.n_ctx, &n));
    const size_t bytes = (size_t)n * e->cfg.model.d_embed * 4;
    float* h = (float*)malloc(bytes);
    if (!h) return q3_set_err(e, Q3TTS_ERR_OOM, "malloc");
    hipError_t er = hipMemcpyAsync(h, e->xp, bytes, hipMemcpyDeviceToHost, e->stream);
    if (er == hipSuccess) er = hipStreamSynchronize(e->stream);
    if (er != hipSuccess) { free(h); return q3_set_err(e, Q3TTS_ERR_DEVICE, hipGetErrorString(er)); }
    *out_embd = h; *out_n = n;
    return Q3TTS_OK;
}

// ------------------------------------------------------------------------------------------------
// generation
// ------------------------------------------------------------------------------------------------
// Map the live slots onto rows [0, n) of the smallest bucket that holds them (idle slots fill the rest: every row keeps
// a distinct, valid slot). Row-indexed state that outlives a frame (the Talker logits) moves with its slot.
static int plan_rows(q3tts_engine* e, const std::vector<int>& live_in) {
    Q3Lane& L = e->lanes[0];
    const int B = e->B;
    std::vector<int> live(live_in);
    std::sort(live.begin(), live.end());
    int bi = 0;
    while (bi + 1 < (int)e->buckets.size() && e->buckets[bi] < (int)live.size()) ++bi;
    bool ok = bi == e->cur_bucket;
    if (ok) for (int b : live) if (e->row_of_slot[b] >= e->buckets[bi]) { ok = false; break; }
    if (ok) return Q3TTS_OK;
    std::vector<int> slot_of_row(B, -1), perm(B), used(B, 0), sla(2 * (size_t)B, 0), pa(2 * (size_t)B, 0);
    int r = 0;
    for (int b : live) { slot_of_row[r++] = b; used[b] = 1; }
    for (int b = 0; b < B && r < B; ++b) if (!used[b]) slot_of_row[r++] = b;
    const int bs = e->buckets[bi];  // pass A: rows [0, bs) at position 0, rows [bs, 2 bs) at position 1
    for (r = 0; r < B; ++r) perm[r] = e->row_of_slot[slot_of_row[r]];
    for (r = 0; r < bs; ++r) { sla[r] = sla[bs + r] = slot_of_row[r]; pa[bs + r] = 1; }
    hipStream_t s = e->stream;
    Q3_HIP(e, hipMemcpyAsync(L.perm, perm.data(), (size_t)B * 4, hipMemcpyHostToDevice, s));
    Q3_HIP(e, hipMemcpyAsync(L.slot_id, slot_of_row.data(), (size_t)B * 4, hipMemcpyHostToDevice, s));
    Q3_HIP(e, hipMemcpyAsync(L.slotA, sla.data(), (size_t)2 * B * 4, hipMemcpyHostToDevice, s));
    Q3_HIP(e, hipMemcpyAsync(L.posA, pa.data(), (size_t)2 * B * 4, hipMemcpyHostToDevice, s));
    // row state that outlives a frame: the Talker logits (sampled at the next frame) and its last hidden row (the
    // Predictor's first input)
    q3_launch_gather_rows(L.logits_tmp, L.logits, L.perm, B, e->cfg.model.t_vocab, s);
    Q3_HIP(e, hipMemcpyAsync(L.logits, L.logits_tmp, (size_t)B * e->cfg.model.t_vocab * 4, hipMemcpyDeviceToDevice, s));
    q3_launch_gather_rows(L.logits_tmp, L.xT, L.perm, B, e->cfg.model.t_d_model, s);
    Q3_HIP(e, hipMemcpyAsync(L.xT, L.logits_tmp, (size_t)B * e->cfg.model.t_d_model * 4, hipMemcpyDeviceToDevice, s));
    Q3_HIP(e, hipStreamSynchronize(s));  // the uploads read locals
    for (r = 0; r < B; ++r) e->row_of_slot[slot_of_row[r]] = r;
    e->slot_of_row = slot_of_row;
    e->cur_bucket = bi;
    return Q3TTS_OK;
}

// CH frame steps over the current row bucket; afterwards the slot mirror is on the host. Returns the device time (ms).
static double now_ms();
static int run_chunk(q3tts_engine* e, int CH, float* dev_ms) {
    hipStream_t s = e->stream;
    Q3Lane& L = e->lanes[0];
    const double hp0 = now_ms();
    Q3_HIP(e, hipEventRecord(e->ev1, s));  // admissions (prefill, state uploads) precede the frames
    Q3_HIP(e, hipStreamWaitEvent(L.stream, e->ev1, 0));
    Q3_HIP(e, hipEventRecord(L.ev_begin, L.stream));
    e->probe_i = 0;
    if (e->probe) { hipEventRecord(e->probe_ev[8], L.stream); hipEventRecord(e->probe_ev[9], L.stream); }  // empty bracket
    for (int i = 0; i < CH; ++i) {
        if (!L.execs.empty() && !e->probe) { Q3_HIP(e, hipGraphLaunch(L.execs[e->cur_bucket], L.stream)); }
        else {
            if (record_frame(e, L, L.stream, e->buckets[e->cur_bucket])) return q3_set_err(e, Q3TTS_ERR_INVALID, "frame step: a kernel launch was refused for this model shape");
            Q3_HIP(e, hipGetLastError());
        }
    }
    e->row_steps += (long long)CH * e->buckets[e->cur_bucket];
    Q3_HIP(e, hipEventRecord(L.ev_end, L.stream));
    Q3_HIP(e, hipStreamWaitEvent(s, L.ev_end, 0));
    Q3_HIP(e, hipEventRecord(e->ev3, s));  // the vocoder stream waits on this
    Q3_HIP(e, hipMemcpyAsync(e->slots_host, e->slots, sizeof(Q3Slot) * e->B, hipMemcpyDeviceToHost, s));
    const double hp1 = now_ms();
    Q3_HIP(e, hipStreamSynchronize(s));
    e->hp_launch += hp1 - hp0; e->hp_sync += now_ms() - hp1;  // Q3TTS_HOST_PROF: host wall of the launch part / of the wait
    if (dev_ms) { *dev_ms = 0.0f; hipEventElapsedTime(dev_ms, L.ev_begin, L.ev_end); }
    for (int i = 0; i + 1 < e->probe_i; i += 2) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, e->probe_ev[i], e->probe_ev[i + 1]) == hipSuccess) { e->probe_ms += ms; ++e->probe_cnt; }
    }
    if (e->probe) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, e->probe_ev[8], e->probe_ev[9]) == hipSuccess) { e->probe_empty_ms += ms; ++e->probe_empty_cnt; }
    }
    return Q3TTS_OK;
}

static uint64_t wall_seed() {
    return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::system_clock::now().time_since_epoch()).count();
}

// Talker prefill (src/tts/engine.rs:455-462) of several requests at once: their prompt rows are concatenated into one
// batch (row -> (slot, position) maps), so the weights stream once for all of them; then per request the last row
// seeds the slot (logits, state, sampler draws). rc[i] receives the per-request status.
struct Adm { int b; const q3tts_request* r; int n, row0, max_steps; };

static int admit_group(q3tts_engine* e, std::vector<Adm>& grp, int total) {
    const q3tts_model_config& m = e->cfg.model;
    hipStream_t s = e->stream;
    if (grp.empty()) return Q3TTS_OK;
    std::vector<int> pos(total), slot(total);
    for (const Adm& a : grp) for (int i = 0; i < a.n; ++i) { pos[a.row0 + i] = i; slot[a.row0 + i] = a.b; }
    Q3_HIP(e, hipMemcpyAsync(e->pf_pos, pos.data(), (size_t)total * 4, hipMemcpyHostToDevice, s));
    Q3_HIP(e, hipMemcpyAsync(e->pf_slot, slot.data(), (size_t)total * 4, hipMemcpyHostToDevice, s));
    std::vector<int> seg; int seg_max = 0;  // the same rows as runs: every request's rows are consecutive, positions 0 .. n - 1
    for (const Adm& a : grp) { seg.push_back(a.row0); seg.push_back(a.n); seg.push_back(a.b); seg_max = std::max(seg_max, a.n); }
    Q3_HIP(e, hipMemcpyAsync(e->pf_seg, seg.data(), seg.size() * 4, hipMemcpyHostToDevice, s));
    Q3_HIP(e, hipStreamSynchronize(s));  // pos/slot are locals
    if (e->T.a8) q3_launch_norm_inputs_q8(e->xp, m.t_d_model, total, m.t_d_model, e->T.attn_norm[0], (int8_t*)e->xbp, e->ascp, e->rt16p, e->sspp, m.t_d_model / 16, s);
    else q3_launch_norm_inputs(e->xp, m.t_d_model, total, m.t_d_model, e->T.attn_norm[0], e->xbp, 0, e->sspp, m.t_d_model / 16, s);
    const int rl = run_layers(e, e->T, e->xp, e->xbp, e->sspp, total, e->pf_pos, e->pf_slot, e->sc_pre, s, false, nullptr, 0, 0, e->pf_seg, (int)grp.size(), seg_max, e->ascp, e->rt16p);
    if (rl) return q3_set_err(e, Q3TTS_ERR_INVALID, "prefill: a kernel launch was refused for this model shape");
    Q3_HIP(e, hipGetLastError());
    for (const Adm& a : grp) {
        const q3tts_request* r = a.r;
        const int b = a.b;
        Q3Lane& L = e->lanes[0];
        const int row = e->row_of_slot[b];
        q3_launch_copy_rows(L.xT + (size_t)row * m.t_d_model, m.t_d_model, e->xp + (size_t)(a.row0 + a.n - 1) * m.t_d_model, m.t_d_model, 1, m.t_d_model, s);
        const size_t lastr = (size_t)(a.row0 + a.n - 1);  // the last prompt row's norm inputs for out_norm came out of the last block
        Q3BGemm g{}; g.a = e->xbp; g.a_row0 = (int)lastr; g.B = 1; g.w = e->T.head; g.wscale = e->T.q8 ? e->T.shead : nullptr; g.K = m.t_d_model; g.N = m.t_vocab;
        g.ssp = e->sspp + lastr * (m.t_d_model / 16); g.ld_ssp = m.t_d_model / 16; g.ntiles = m.t_d_model / 16; g.d_norm = m.t_d_model; g.eps = m.rms_eps;
        g.epi = Q3_EPI_STORE; g.y = L.logits + (size_t)row * m.t_vocab; g.ldy = m.t_vocab;
        if (e->T.a8) { g.ascale = e->ascp; g.a_rt16 = e->rt16p; }
        if (e->T.a8 ? q3_launch_bgemm8(g, s) : q3_launch_bgemm(g, s)) return q3_set_err(e, Q3TTS_ERR_INVALID, "prefill head: launch refused for this model shape");
        // sampler stream (src/tts/engine.rs:473-485)
        float temperature = e->temperature, top_p = e->top_p; int top_k = e->top_k, has_seed = e->has_seed; uint64_t seed = e->seed;
        if (!r->use_engine_sampler) { temperature = r->temperature; top_k = r->top_k; top_p = r->top_p; has_seed = r->has_seed; seed = r->seed; }
        if (!has_seed) seed = wall_seed();
        if (temperature > 0.0f) {
            std::vector<float> draws(a.max_steps);
            q3_stdrng_f32(seed, a.max_steps, draws.data());
            Q3_HIP(e, hipMemcpyAsync(e->rng + (size_t)b * e->cfg.max_steps_cap, draws.data(), (size_t)a.max_steps * 4, hipMemcpyHostToDevice, s));
            Q3_HIP(e, hipStreamSynchronize(s));
        }
        Q3Slot* st = e->slots_host + e->B + b;  // pinned staging half
        memset(st, 0, sizeof(*st));
        st->active = 1; st->cur_pos = a.n; st->n_frames = 0; st->max_steps = a.max_steps; st->min_frames = r->min_frames;
        st->force_eos_at = r->force_eos_at; st->top_k = top_k; st->temperature = temperature; st->top_p = top_p;
        st->rng_base = b * e->cfg.max_steps_cap;
        Q3_HIP(e, hipMemcpyAsync(e->slots + b, st, sizeof(Q3Slot), hipMemcpyHostToDevice, s));
        if (e->voc) TRY(q3_voc_reset(e, b));
    }
    return Q3TTS_OK;
}

// slots[i] <- reqs[i]; rc[i] = status of request i (a failing request does not stop the others)
static int admit_many(q3tts_engine* e, const int* slots, const q3tts_request* const* reqs, int count, int* rc) {
    const q3tts_model_config& m = e->cfg.model;
    hipStream_t s = e->stream;
    std::vector<Adm> grp;
    int total = 0;
    for (int i = 0; i < count; ++i) {
        const q3tts_request* r = reqs[i];
        rc[i] = Q3TTS_OK;
        const int max_steps = r->max_steps > 0 ? r->max_steps : e->max_steps;
        if (max_steps > e->cfg.max_steps_cap) { rc[i] = q3_set_err(e, Q3TTS_ERR_INVALID, "max_steps exceeds max_steps_cap"); continue; }
        int n = 0;
        for (int attempt = 0; attempt < 2; ++attempt) {
            const int room = e->cfg.n_ctx - total;
            if (r->prompt_embd) {
                n = r->n_tok;
                if (n <= 0 || n > e->cfg.n_ctx) { rc[i] = q3_set_err(e, Q3TTS_ERR_INVALID, "n_tok out of range"); break; }
                if (n > room) { if (total == 0) { rc[i] = q3_set_err(e, Q3TTS_ERR_INVALID, "n_tok out of range"); break; } }
                else { Q3_HIP(e, hipMemcpyAsync(e->xp + (size_t)total * m.d_embed, r->prompt_embd, (size_t)n * m.d_embed * 4, hipMemcpyHostToDevice, s)); break; }
            } else if (r->prompt) {
                const int brc = build_prompt_dev(e, r->prompt, e->xp + (size_t)total * m.d_embed, room, &n);
                if (brc == Q3TTS_OK) break;
                if (total == 0) { rc[i] = brc; break; }
            } else { rc[i] = q3_set_err(e, Q3TTS_ERR_INVALID, "request has neither prompt_embd nor prompt"); break; }
            TRY(admit_group(e, grp, total));  // batch full: flush, then retry this request in an empty batch
            grp.clear(); total = 0;
        }
        if (rc[i] != Q3TTS_OK) continue;
        if (n + max_steps > e->cfg.n_ctx) { rc[i] = q3_set_err(e, Q3TTS_ERR_INVALID, "prompt + max_steps exceeds n_ctx"); continue; }
        grp.push_back(Adm{slots[i], r, n, total, max_steps});
        total += n;
    }
    return admit_group(e, grp, total);
}

static int admit(q3tts_engine* e, int b, const q3tts_request* r) {
    int rc = Q3TTS_OK;
    int st = admit_many(e, &b, &r, 1, &rc);
    return st != Q3TTS_OK ? st : rc;
}

static double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct SlotRun { int req = -1; int voc_frames = 0; double t_first = 0; };

// Pinned result buffers are recycled through a small process-wide pool: hipHostMalloc / hipHostFree cost ~0.3 ms each, and a
// batch hands out one PCM buffer per utterance. A 64-byte header in front of the payload remembers the capacity.
#include <mutex>
namespace {
struct PinHdr { size_t cap; size_t magic; };
std::mutex g_pin_mu;
std::vector<PinHdr*> g_pin_pool;
size_t g_pin_bytes = 0;
float* pin_alloc(size_t bytes) {
    {
        std::lock_guard<std::mutex> lk(g_pin_mu);
        for (size_t i = 0; i < g_pin_pool.size(); ++i)
            if (g_pin_pool[i]->cap >= bytes && g_pin_pool[i]->cap <= 2 * bytes + 4096) {
                PinHdr* h = g_pin_pool[i];
                g_pin_pool[i] = g_pin_pool.back(); g_pin_pool.pop_back(); g_pin_bytes -= h->cap;
                return (float*)((char*)h + 64);
            }
    }
    void* p = nullptr;
    const size_t cap = (bytes + 65535) & ~(size_t)65535;
    if (hipHostMalloc(&p, cap + 64, hipHostMallocDefault) != hipSuccess) return nullptr;
    PinHdr* h = (PinHdr*)p; h->cap = cap; h->magic = 0x5133505043ull;
    return (float*)((char*)p + 64);
}
void pin_free(float* q) {
    if (!q) return;
    PinHdr* h = (PinHdr*)((char*)q - 64);
    if (h->magic != 0x5133505043ull) return;  // not ours: leave it alone
    std::lock_guard<std::mutex> lk(g_pin_mu);
    if (g_pin_pool.size() < 256 && g_pin_bytes + h->cap <= ((size_t)1 << 30)) { g_pin_pool.push_back(h); g_pin_bytes += h->cap; return; }
    hipHostFree(h);
}
}  // namespace

// Results of a finished slot. The codes come back on the decoder stream at once; the PCM (pinned host buffer) is copied
// on the vocoder stream. With `defer` the call does not wait for the vocoder: the copy is enqueued behind the slot's last
// vocoder chunk, `fin_ev` is recorded after it and the caller completes the result later (complete_result), so the next
// decode chunk is launched while the vocoder is still working.
static int finalize(q3tts_engine* e, int b, const q3tts_request* r, q3tts_result* o, const SlotRun& sr, double t0, bool defer = false,
                    hipEvent_t fin_ev = nullptr, int ri = -1) {
    const int ncb = e->cfg.model.n_codebooks;
    const Q3Slot& st = e->slots_host[b];
    o->n_frames = st.n_frames; o->hit_eos = st.hit_eos;
    o->codes = (int32_t*)malloc(sizeof(int32_t) * (size_t)std::max(1, st.n_frames * ncb));
    if (!o->codes) return q3_set_err(e, Q3TTS_ERR_OOM, "malloc");
    if (st.n_frames > 0)
        Q3_HIP(e, hipMemcpyAsync(o->codes, e->codes + (size_t)b * e->cfg.max_steps_cap * ncb, sizeof(int32_t) * (size_t)st.n_frames * ncb,
                                 hipMemcpyDeviceToHost, e->stream));
    o->sample_rate = e->cfg.vocoder.sample_rate;
    if (r->want_pcm && e->voc && e->dev_pcm && ri >= 0 && ri < e->dev_pcm_n) {
        // device copy of the utterance's PCM (q3tts_set_device_pcm): row ri of the packed buffer, for collectives that read device memory
        const int ns = q3_voc_samples(e, b);
        if (ns > 0) Q3_HIP(e, hipMemcpyAsync(e->dev_pcm + (size_t)ri * e->dev_pcm_stride, q3_voc_pcm(e, b), sizeof(float) * (size_t)ns, hipMemcpyDeviceToDevice, e->vstream));
    }
    if (r->want_pcm == 2 && e->voc) {  // device only: no host copy
        o->n_samples = q3_voc_samples(e, b);
        if (defer) Q3_HIP(e, hipEventRecord(fin_ev, e->vstream));
        else Q3_HIP(e, hipStreamSynchronize(e->vstream));
    } else if (r->want_pcm && e->voc) {
        const int ns = q3_voc_samples(e, b);
        o->n_samples = ns;
        o->pcm = pin_alloc(sizeof(float) * (size_t)std::max(1, ns));
        if (!o->pcm) return q3_set_err(e, Q3TTS_ERR_OOM, "hipHostMalloc");
        if (ns > 0) Q3_HIP(e, hipMemcpyAsync(o->pcm, q3_voc_pcm(e, b), sizeof(float) * (size_t)ns, hipMemcpyDeviceToHost, e->vstream));
        if (defer) Q3_HIP(e, hipEventRecord(fin_ev, e->vstream));
        else Q3_HIP(e, hipStreamSynchronize(e->vstream));
    } else if (defer) {
        Q3_HIP(e, hipEventRecord(fin_ev, e->vstream));
    }
    Q3_HIP(e, hipStreamSynchronize(e->stream));
    o->first_chunk_ms = sr.t_first > 0 ? (float)(sr.t_first - t0) : 0.0f;
    o->total_ms = (float)(now_ms() - t0);
    o->status = defer ? Q3TTS_ERR_STATE : Q3TTS_OK;  // a deferred result becomes OK in complete_result
    return Q3TTS_OK;
}

static int complete_result(q3tts_engine* e, q3tts_result* o, hipEvent_t fin_ev, double t0) {
    Q3_HIP(e, hipEventSynchronize(fin_ev));
    o->total_ms = (float)(now_ms() - t0);
    o->status = Q3TTS_OK;
    return Q3TTS_OK;
}

extern "C" int q3tts_generate_batch(q3tts_engine* e, const q3tts_request* reqs, int32_t n, q3tts_result* outs) {
    if (!e || !reqs || !outs || n <= 0) return q3_set_err(e, Q3TTS_ERR_INVALID, "null/empty argument");
    Q3_HIP(e, hipSetDevice(e->cfg.device));
    for (int i = 0; i < n; ++i) { memset(&outs[i], 0, sizeof(outs[i])); outs[i].status = Q3TTS_ERR_STATE; }
    for (int i = 0; i < n; ++i) {
        if (reqs[i].want_pcm && !e->voc) return q3_set_err(e, Q3TTS_ERR_STATE, "want_pcm on an engine created with with_vocoder = 0");
        if (reqs[i].want_pcm == 2 && !e->dev_pcm_on) return q3_set_err(e, Q3TTS_ERR_STATE, "want_pcm = 2 (device only) needs q3tts_set_device_pcm(engine, 1)");
    }
    if (e->dev_pcm_on && e->voc) {  // one row of max_steps_cap frames per request of this call
        const size_t stride = (size_t)e->cfg.max_steps_cap * q3_voc_samples_per_frame(e);
        if (e->dev_pcm_n < n || e->dev_pcm_stride != stride) {
            if (e->dev_pcm) { Q3_HIP(e, hipStreamSynchronize(e->vstream)); hipFree(e->dev_pcm); e->dev_pcm = nullptr; e->dev_pcm_n = 0; }
            void* p = nullptr;
            if (hipMalloc(&p, (size_t)n * stride * sizeof(float)) != hipSuccess) return q3_set_err(e, Q3TTS_ERR_OOM, "hipMalloc (device PCM)");
            e->dev_pcm = (float*)p; e->dev_pcm_n = n; e->dev_pcm_stride = stride;
        }
    }
    const int B = e->B, CH = 4;  // 4-frame chunks: src/tts/engine.rs:509-512
    const int spf = e->voc ? q3_voc_samples_per_frame(e) : 0;
    std::vector<SlotRun> run(B);
    const double t0 = now_ms();
    std::vector<int> pending(B, -1);  // request whose PCM copy is still in flight on the vocoder stream, per slot
    auto drain = [&](int b) -> int {
        if (pending[b] >= 0) { TRY(complete_result(e, &outs[pending[b]], e->fin_ev[b], t0)); pending[b] = -1; }
        return Q3TTS_OK;
    };
    int next = 0, done = 0;
    double dec_ms = 0, pre_ms = 0, voc_ms = 0;
    static const bool host_prof = [] { const char* ev = getenv("Q3TTS_HOST_PROF"); return ev && atoi(ev); }();  // host wall time per phase of this loop, to stderr
    double hp_admit = 0, hp_chunk = 0, hp_voc = 0, hp_fin = 0, hp_tail = 0, hp_t = now_ms();
    auto hp_lap = [&](double& acc) { if (host_prof) { const double t = now_ms(); acc += t - hp_t; hp_t = t; } };
    long long steps = 0, ctx_tokens = 0, live_slot_steps = 0;
    e->probe_ms = 0; e->probe_cnt = 0; e->row_steps = 0; e->probe_empty_ms = 0; e->probe_empty_cnt = 0;
    hipStream_t s = e->stream;
    while (done < n) {
        bool admitted = false;
        {
            std::vector<int> as, ai; std::vector<const q3tts_request*> ar;
            for (int b = 0; b < B && next < n; ++b)
                if (run[b].req < 0) { TRY(drain(b)); as.push_back(b); ar.push_back(&reqs[next]); ai.push_back(next++); }
            {
                std::vector<int> live(as);
                for (int b = 0; b < B; ++b) if (run[b].req >= 0) live.push_back(b);
                TRY(plan_rows(e, live));
            }
            if (!as.empty()) {
                Q3_HIP(e, hipEventRecord(e->ev0, s));
                admitted = true;
                std::vector<int> rcs(as.size());
                TRY(admit_many(e, as.data(), ar.data(), (int)as.size(), rcs.data()));
                for (size_t i = 0; i < as.size(); ++i) {
                    if (rcs[i] != Q3TTS_OK) { outs[ai[i]].status = rcs[i]; ++done; }
                    else { run[as[i]] = SlotRun{}; run[as[i]].req = ai[i]; }
                }
            }
        }
        if (admitted) { Q3_HIP(e, hipEventRecord(e->ev2, s)); }
        hp_lap(hp_admit);
        bool any = false;
        for (int b = 0; b < B; ++b) if (run[b].req >= 0) any = true;
        if (!any) break;
        float ms = 0;
        TRY(run_chunk(e, CH, &ms));
        hp_lap(hp_chunk);
        dec_ms += ms; steps += CH;
        if (admitted) { hipEventElapsedTime(&ms, e->ev0, e->ev2); pre_ms += ms; }
        for (int b = 0; b < B; ++b) if (run[b].req >= 0) { ctx_tokens += (long long)e->slots_host[b].cur_pos * CH; live_slot_steps += CH; }
        // H8: the vocoder consumes 4-frame chunks (src/tts/engine.rs:507-541). It runs on its own stream behind an
        // event, batched over every slot that has a chunk ready, so chunk k's PCM overlaps the decoding of chunk k+1.
        if (e->voc) {
            const double tv0 = now_ms();
            hipStream_t vs = e->vstream;
            bool waited = false, first = false;
            auto ensure_wait = [&]() -> int { if (!waited) { Q3_HIP(e, hipStreamWaitEvent(vs, e->ev3, 0)); waited = true; } return Q3TTS_OK; };
            // ONE batched call per chunk: every slot with new frames runs nf = 4. A finished utterance whose tail is
            // shorter is padded with throw-away frames: the vocoder is causal, so they cannot change the samples already
            // due, their own samples are never reported, and the slot's vocoder state is reset at its next admission.
            int list[64], real[64];
            for (;;) {
                int ns = 0;
                for (int b = 0; b < B; ++b) {
                    if (run[b].req < 0 || !reqs[run[b].req].want_pcm) continue;
                    const int pend = e->slots_host[b].n_frames - run[b].voc_frames;
                    if (pend >= 4 || (pend > 0 && !e->slots_host[b].active)) { real[ns] = std::min(pend, 4); list[ns++] = b; }
                }
                if (!ns) break;
                TRY(ensure_wait());
                int still = 0;  // slots that go on decoding while this call runs
                for (int b = 0; b < B; ++b) if (run[b].req >= 0 && e->slots_host[b].active) ++still;
                TRY(q3_voc_decode_batch(e, list, real, ns, 4, vs, (still > 0 || next < n) ? 1 : 0));
                for (int i = 0; i < ns; ++i) { if (run[list[i]].voc_frames == 0) first = true; run[list[i]].voc_frames += real[i]; }
            }
            if (first) {  // first-chunk latency: the first chunk's PCM resident on the host
                for (int b = 0; b < B; ++b)
                    if (run[b].req >= 0 && run[b].t_first == 0 && run[b].voc_frames > 0) {
                        const int nsmp = std::min(run[b].voc_frames, 4) * spf;
                        Q3_HIP(e, hipMemcpyAsync(e->first_chunk_host, q3_voc_pcm(e, b), sizeof(float) * (size_t)nsmp, hipMemcpyDeviceToHost, vs));
                    }
                Q3_HIP(e, hipStreamSynchronize(vs));
                const double tn = now_ms();
                for (int b = 0; b < B; ++b) if (run[b].req >= 0 && run[b].t_first == 0 && run[b].voc_frames > 0) run[b].t_first = tn;
            }
            voc_ms += now_ms() - tv0;
        }
        hp_lap(hp_voc);
        // results whose PCM copy has landed are completed now, so total_ms is an utterance's own latency (to one chunk's granularity)
        for (int j = 0; j < B; ++j)
            if (pending[j] >= 0 && hipEventQuery(e->fin_ev[j]) == hipSuccess) TRY(drain(j));
        for (int b = 0; b < B; ++b) {
            if (run[b].req < 0 || e->slots_host[b].active) continue;
            // V4 flush: the reference sends is_last only when its final buffer is not empty, i.e. n_frames % 4 != 0 (src/tts/engine.rs:510-536,
            // restated by q3o_chunk_plan); with lookahead_frames > 0 an utterance of n_frames % 4 == 0 keeps its withheld tail (vocoder_flush_tail = 1: always flush)
            if (e->voc && (e->cfg.vocoder_flush_tail || e->slots_host[b].n_frames % 4 != 0)) q3_voc_mark_last(e, b);
            // hand the slot's results over without waiting for the vocoder (completed at slot reuse / at the end)
            for (int j = 0; j < B; ++j)
                if (pending[j] >= 0 && hipEventQuery(e->fin_ev[j]) == hipSuccess) TRY(drain(j));
            TRY(finalize(e, b, &reqs[run[b].req], &outs[run[b].req], run[b], t0, true, e->fin_ev[b], run[b].req));
            pending[b] = run[b].req;
            run[b].req = -1; ++done;
        }
        hp_lap(hp_fin);
    }
    for (int b = 0; b < B; ++b) TRY(drain(b));
    hp_lap(hp_tail);
    if (host_prof) {
        fprintf(stderr, "q3tts_generate_batch host wall (ms): admit + plan %.1f | chunks (launch + wait) %.1f = launch %.1f + wait %.1f, device events %.1f over %lld steps | vocoder issue %.1f | finalize %.1f | tail (last results) %.1f | total %.1f\n",
                hp_admit, hp_chunk, e->hp_launch, e->hp_sync, dec_ms, steps, hp_voc, hp_fin, hp_tail, now_ms() - t0);
        e->hp_launch = e->hp_sync = 0;
    }
    e->tm.prefill_ms = (float)pre_ms; e->tm.decode_ms = (float)dec_ms; e->tm.vocoder_ms = (float)voc_ms;
    e->tm.total_ms = (float)(now_ms() - t0); e->tm.frame_steps = steps; e->tm.frame_step_ms = steps ? (float)(dec_ms / steps) : 0.0f;
    // bytes = 2*W_T + 15*2*W_P(layers) + 15*2*h + 16*2*pj + KV bytes of the live context + gathers
    {
        const q3tts_model_config& m = e->cfg.model;
        const long long wt = (long long)e->T.weight_bytes;  // includes lm_head
        const long long wp_layers = (long long)e->P.weight_bytes - 2ll * e->P.head_n * m.p_d_model;
        const long long head1 = 2ll * m.codebook_size * m.p_d_model, pj = 2ll * m.p_d_model * m.d_embed;
        const long long kv_per_tok = 2ll * m.t_n_layer * 2 * m.t_n_kv_head * m.t_head_dim;
        const long long fixed = wt + (m.n_codebooks - 1) * (wp_layers + head1) + pj;  // one projection GEMM per frame (hidden rows); codes come pre-projected
        e->tm.algo_bytes_per_step = fixed + (steps ? kv_per_tok * (ctx_tokens / steps) : 0);
        e->tm.mean_live_slots = steps ? (float)((double)live_slot_steps / (double)steps) : 0.0f;
        e->tm.algo_flops_per_step = (long long)((double)fixed * (double)e->tm.mean_live_slots);  // 2 flop per bf16 weight (2 bytes) per live row
        e->tm.mean_rows = steps ? (float)((double)e->row_steps / (double)steps) : 0.0f;
        e->tm.mean_ctx_tokens = steps ? (float)((double)ctx_tokens / (double)steps) : 0.0f;
        e->tm.probe_kernel_ms = e->probe_cnt ? (float)(e->probe_ms / (double)e->probe_cnt) : 0.0f;
        e->tm.probe_count = e->probe_cnt;
        e->tm.probe_empty_ms = e->probe_empty_cnt ? (float)(e->probe_empty_ms / (double)e->probe_empty_cnt) : 0.0f;
    }
    return Q3TTS_OK;
}

extern "C" int q3tts_generate(q3tts_engine* e, const q3tts_request* req, q3tts_result* out) {
    int rc = q3tts_generate_batch(e, req, 1, out);
    if (rc != Q3TTS_OK) return rc;
    return out->status;
}

extern "C" void q3tts_result_free(q3tts_result* r) {
    if (!r) return;
    free(r->codes);
    pin_free(r->pcm);  // pinned (filled by an asynchronous device-to-host copy); goes back to the pool
    r->codes = nullptr; r->pcm = nullptr;
}

extern "C" int q3tts_get_timings(const q3tts_engine* e, q3tts_timings* out) {
    if (!e || !out) return Q3TTS_ERR_INVALID;
    *out = e->tm;
    return Q3TTS_OK;
}

// ------------------------------------------------------------------------------------------------
// streaming (H8): 4-frame chunks
// ------------------------------------------------------------------------------------------------
struct q3tts_stream {
    q3tts_engine* e; q3tts_request req; int voc_frames = 0; bool finished = false; bool final_sent = false;
    std::vector<float> chunk; double t0 = 0, t_first = 0;
};

extern "C" int q3tts_stream_begin(q3tts_engine* e, const q3tts_request* req, q3tts_stream** out) {
    if (!e || !req || !out) return q3_set_err(e, Q3TTS_ERR_INVALID, "null argument");
    if (!e->voc) return q3_set_err(e, Q3TTS_ERR_STATE, "streaming needs with_vocoder = 1");
    Q3_HIP(e, hipSetDevice(e->cfg.device));
    q3tts_stream* st = new q3tts_stream();
    st->e = e; st->req = *req; st->t0 = now_ms();
    int rc = plan_rows(e, std::vector<int>{0});
    if (rc == Q3TTS_OK) rc = admit(e, 0, req);
    if (rc != Q3TTS_OK) { delete st; return rc; }
    *out = st;
    return Q3TTS_OK;
}

extern "C" int q3tts_stream_poll(q3tts_stream* st, const float** chunk, int32_t* n_samples, int32_t* is_final) {
    if (!st || !chunk || !n_samples || !is_final) return q3_set_err(nullptr, Q3TTS_ERR_INVALID, "null argument");
    q3tts_engine* e = st->e;
    *chunk = nullptr; *n_samples = 0; *is_final = 0;
    if (st->final_sent) { *is_final = 1; return Q3TTS_OK; }
    hipStream_t s = e->stream;
    const int spf = q3_voc_samples_per_frame(e);
    for (;;) {
        const Q3Slot& sl = e->slots_host[0];
        if (!st->finished) {
            TRY(run_chunk(e, 4, nullptr));
            if (!sl.active) st->finished = true;
        }
        int nf = 0, last = 0;
        if (sl.n_frames - st->voc_frames >= 4) nf = 4;
        else if (st->finished && sl.n_frames > st->voc_frames) { nf = sl.n_frames - st->voc_frames; last = 1; }
        // (a last chunk of exactly 4 frames goes without is_last, as the reference's vocoder thread sends it: src/tts/engine.rs:510-536;
        // vocoder_flush_tail = 1 flushes the look-ahead tail with it)
        if (nf == 4 && st->finished && st->voc_frames + 4 >= sl.n_frames && e->cfg.vocoder_flush_tail) last = 1;
        if (nf > 0) {
            const int before = q3_voc_samples(e, 0);
            TRY(q3_voc_decode(e, 0, st->voc_frames, nf, last, s));
            st->voc_frames += nf;
            const int after = q3_voc_samples(e, 0);
            st->chunk.resize((size_t)std::max(1, after - before));
            if (after > before)
                Q3_HIP(e, hipMemcpyAsync(st->chunk.data(), q3_voc_pcm(e, 0) + before, sizeof(float) * (size_t)(after - before), hipMemcpyDeviceToHost, s));
            Q3_HIP(e, hipStreamSynchronize(s));
            if (st->t_first == 0) st->t_first = now_ms();
            *chunk = st->chunk.data(); *n_samples = after - before;
            if (st->finished && st->voc_frames >= sl.n_frames) { *is_final = 1; st->final_sent = true; }
            (void)spf;
            return Q3TTS_OK;
        }
        if (st->finished) {
            if (e->cfg.vocoder_flush_tail) {  // EOS arrived with no frames left over: the withheld look-ahead tail goes out as a last chunk of its own
                const int before = q3_voc_samples(e, 0);
                q3_voc_mark_last(e, 0);
                const int after = q3_voc_samples(e, 0);
                if (after > before) {
                    st->chunk.resize((size_t)(after - before));
                    Q3_HIP(e, hipMemcpyAsync(st->chunk.data(), q3_voc_pcm(e, 0) + before, sizeof(float) * (size_t)(after - before), hipMemcpyDeviceToHost, s));
                    Q3_HIP(e, hipStreamSynchronize(s));
                    *chunk = st->chunk.data(); *n_samples = after - before;
                }
            }
            *is_final = 1; st->final_sent = true; return Q3TTS_OK;
        }
    }
}

extern "C" int q3tts_stream_end(q3tts_stream* st, q3tts_result* out) {
    if (!st) return Q3TTS_ERR_INVALID;
    q3tts_engine* e = st->e;
    int rc = Q3TTS_OK;
    // make sure the slot is retired even if the caller stops early
    Q3Slot* stage = e->slots_host + e->B;
    memset(stage, 0, sizeof(Q3Slot));
    hipMemcpyAsync(e->slots_host, e->slots, sizeof(Q3Slot), hipMemcpyDeviceToHost, e->stream);
    hipStreamSynchronize(e->stream);
    if (out) {
        memset(out, 0, sizeof(*out));
        SlotRun sr; sr.t_first = st->t_first;
        q3tts_request r = st->req; r.want_pcm = 1;
        rc = finalize(e, 0, &r, out, sr, st->t0);
    }
    hipMemcpyAsync(e->slots, stage, sizeof(Q3Slot), hipMemcpyHostToDevice, e->stream);
    hipStreamSynchronize(e->stream);
    delete st;
    return rc;
}

// ------------------------------------------------------------------------------------------------
// kernel-level test hooks
// ------------------------------------------------------------------------------------------------
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) hipFree(p); }
    int alloc(size_t bytes) { return hipMalloc(&p, bytes + 64) == hipSuccess && hipMemset(p, 0, bytes + 64) == hipSuccess ? 0 : -1; }
};
#define HK(call) do { hipError_t er__ = (call); if (er__ != hipSuccess) return q3_set_err(nullptr, Q3TTS_ERR_DEVICE, std::string(#call) + ": " + hipGetErrorString(er__)); } while (0)

extern "C" int q3tts_k_gemm_exact(int32_t device, const float* x, int32_t B, int32_t K, const uint16_t* w, int32_t N, const float* norm_w,
                                  float eps, const float* bias, int32_t epi, float* y, uint64_t* keys, int32_t iters, float* mean_ms) {
    if (!x || !w || !y || B <= 0 || K % 512 || N % 16 || (norm_w && K > 8192)) return q3_set_err(nullptr, Q3TTS_ERR_INVALID, "gemm hook: bad shape");
    if (epi == Q3_EPI_SWIGLU && (N % 32)) return q3_set_err(nullptr, Q3TTS_ERR_INVALID, "swiglu needs N % 32 == 0");
    HK(hipSetDevice(device));
    const int F = N / 2;
    const size_t ny = epi == Q3_EPI_SWIGLU ? (size_t)B * F : (size_t)B * N;
    DevBuf dx, dw, dwt, dn, db, dy, dk;
    if (dx.alloc((size_t)B * K * 4) || dw.alloc((size_t)N * K * 2) || dwt.alloc((size_t)N * K * 2) || dn.alloc((size_t)K * 4) ||
        db.alloc((size_t)N * 4) || dy.alloc(ny * 4) || dk.alloc((size_t)B * 8))
        return q3_set_err(nullptr, Q3TTS_ERR_OOM, "hipMalloc");
    HK(hipMemcpy(dx.p, x, (size_t)B * K * 4, hipMemcpyHostToDevice));
    HK(hipMemcpy(dw.p, w, (size_t)N * K * 2, hipMemcpyHostToDevice));
    if (norm_w) HK(hipMemcpy(dn.p, norm_w, (size_t)K * 4, hipMemcpyHostToDevice));
    if (bias) HK(hipMemcpy(db.p, bias, (size_t)N * 4, hipMemcpyHostToDevice));
    if (epi == Q3_EPI_RESID) HK(hipMemcpy(dy.p, y, ny * 4, hipMemcpyHostToDevice));
    Q3Fill f{}; f.dst = (uint4*)dwt.p; f.N = N; f.K = K;
    if (epi == Q3_EPI_SWIGLU) { f.mode = 1; f.src_a = (const uint16_t*)dw.p; f.src_b = (const uint16_t*)dw.p + (size_t)F * K; }
    else { f.mode = 0; f.row0 = 0; f.rows = N; f.src_a = (const uint16_t*)dw.p; }
    q3_launch_fill_tiled(f, nullptr);
    Q3Gemm g{}; g.x = (const float*)dx.p; g.ldx = K; g.B = B; g.w = (const uint4*)dwt.p; g.K = K; g.N = N;
    g.norm_w = norm_w ? (const float*)dn.p : nullptr; g.eps = eps; g.bias = bias ? (const float*)db.p : nullptr;
    g.y = (float*)dy.p; g.ldy = epi == Q3_EPI_SWIGLU ? F : N; g.keys = (unsigned long long*)dk.p; g.key_stride = 1; g.epi = epi;
    q3_launch_gemm(g, nullptr);
    HK(hipDeviceSynchronize());
    if (epi == Q3_EPI_ARGMAX) {
        if (keys) HK(hipMemcpy(keys, dk.p, (size_t)B * 8, hipMemcpyDeviceToHost));
    } else HK(hipMemcpy(y, dy.p, ny * 4, hipMemcpyDeviceToHost));
#ifdef Q3_STAMPS
    {  // experiment builds: phase stamps of workgroup 0 / wave 0 of one warm launch (shader-clock cycles from kernel entry)
        DevBuf dd; dd.alloc(64 * 8);
        g.dbg = nullptr; q3_launch_gemm(g, nullptr); q3_launch_gemm(g, nullptr);
        g.dbg = (unsigned long long*)dd.p; hipMemset(dd.p, 0, 64 * 8);
        q3_launch_gemm(g, nullptr); hipDeviceSynchronize();
        unsigned long long st[8]; hipMemcpy(st, dd.p, 64, hipMemcpyDeviceToHost);
        fprintf(stderr, "stamps B=%d K=%d N=%d norm=%d epi=%d: entry->loop %llu | first operands %llu | loop end %llu | barrier %llu | sums %llu | stores done %llu\n",
                B, K, N, norm_w ? 1 : 0, epi, st[1] - st[0], st[2] - st[0], st[3] - st[0], st[4] - st[0], st[5] - st[0], st[6] - st[0]);
        unsigned long long ws[64]; hipMemcpy(ws, dd.p, 64 * 8, hipMemcpyDeviceToHost);
        for (int w = 0; w < 8; ++w)
            fprintf(stderr, "   wave %d: first operands %llu, mid loop %llu, loop end %llu\n", w, ws[8 + w * 4] - st[0], ws[8 + w * 4 + 1] - st[0], ws[8 + w * 4 + 2] - st[0]);
        g.dbg = nullptr;
    }
#endif
    if (iters > 0 && mean_ms) {
        g.epi = epi == Q3_EPI_RESID ? Q3_EPI_STORE : epi;
        hipEvent_t a, b; HK(hipEventCreate(&a)); HK(hipEventCreate(&b));
        q3_launch_gemm(g, nullptr);
        HK(hipEventRecord(a, nullptr));
        for (int i = 0; i < iters; ++i) q3_launch_gemm(g, nullptr);
        HK(hipEventRecord(b, nullptr)); HK(hipEventSynchronize(b));
        float ms = 0; hipEventElapsedTime(&ms, a, b); *mean_ms = ms / iters;
        hipEventDestroy(a); hipEventDestroy(b);
    }
    return Q3TTS_OK;
}

extern "C" int q3tts_k_attention(int32_t device, const float* qkv, int32_t n_rows, int32_t pos0, int32_t Hq, int32_t Hkv, int32_t hd,
                                 const float* qnw, const float* knw, float eps, float theta, const int32_t* sections, float* out) {
    if (!qkv || !out || hd != 128 || n_rows <= 0 || Hq % Hkv) return q3_set_err(nullptr, Q3TTS_ERR_INVALID, "attention hook: bad shape");
    HK(hipSetDevice(device));
    const int n_ctx = ((pos0 + n_rows + 63) / 64) * 64, ld = (Hq + 2 * Hkv) * hd;
    std::vector<float> cs, sn;
    rope_tables(n_ctx, hd, theta, sections, cs, sn);
    std::vector<int> rp(n_rows), rs(n_rows, 0);
    for (int i = 0; i < n_rows; ++i) rp[i] = pos0 + i;
    DevBuf dq, dout, dqn, dkn, dcs, dsn, dkc, dvc, drp, drs;
    if (dq.alloc((size_t)n_rows * ld * 4) || dout.alloc((size_t)n_rows * Hq * hd * 4) || dqn.alloc(hd * 4) || dkn.alloc(hd * 4) ||
        dcs.alloc(cs.size() * 4) || dsn.alloc(sn.size() * 4) || dkc.alloc((size_t)Hkv * n_ctx * hd * 2) || dvc.alloc((size_t)Hkv * n_ctx * hd * 2) ||
        drp.alloc(n_rows * 4) || drs.alloc(n_rows * 4))
        return q3_set_err(nullptr, Q3TTS_ERR_OOM, "hipMalloc");
    HK(hipMemcpy(dq.p, qkv, (size_t)n_rows * ld * 4, hipMemcpyHostToDevice));
    HK(hipMemcpy(dqn.p, qnw, hd * 4, hipMemcpyHostToDevice)); HK(hipMemcpy(dkn.p, knw, hd * 4, hipMemcpyHostToDevice));
    HK(hipMemcpy(dcs.p, cs.data(), cs.size() * 4, hipMemcpyHostToDevice)); HK(hipMemcpy(dsn.p, sn.data(), sn.size() * 4, hipMemcpyHostToDevice));
    HK(hipMemcpy(drp.p, rp.data(), n_rows * 4, hipMemcpyHostToDevice)); HK(hipMemcpy(drs.p, rs.data(), n_rows * 4, hipMemcpyHostToDevice));
    Q3QkPrep qp{}; qp.qkv = (float*)dq.p; qp.ld = ld; qp.rows = n_rows; qp.Hq = Hq; qp.Hkv = Hkv; qp.hd = hd; qp.qnw = (const float*)dqn.p;
    qp.knw = (const float*)dkn.p; qp.eps = eps; qp.cs = (const float*)dcs.p; qp.sn = (const float*)dsn.p; qp.kc = (uint16_t*)dkc.p;
    qp.vc = (uint16_t*)dvc.p; qp.n_ctx = n_ctx; qp.row_pos = (const int*)drp.p; qp.row_slot = (const int*)drs.p;
    q3_launch_qk_prep(qp, nullptr);
    Q3Attend at{}; at.qkv = (const float*)dq.p; at.ld = ld; at.rows = n_rows; at.out = (float*)dout.p; at.ldo = Hq * hd; at.Hq = Hq; at.Hkv = Hkv;
    at.hd = hd; at.kc = (const uint16_t*)dkc.p; at.vc = (const uint16_t*)dvc.p; at.n_ctx = n_ctx; at.row_pos = qp.row_pos; at.row_slot = qp.row_slot;
    q3_launch_attend(at, nullptr);
    HK(hipDeviceSynchronize());
    HK(hipMemcpy(out, dout.p, (size_t)n_rows * Hq * hd * 4, hipMemcpyDeviceToHost));
    return Q3TTS_OK;
}

extern "C" int q3tts_k_sample(int32_t device, const float* logits, int32_t n, int32_t ld, int32_t limit, float temperature, int32_t top_k,
                              float top_p, const float* r, int32_t* out) {
    if (!logits || !out || n <= 0 || limit <= 0 || limit > 4096 || limit > ld) return q3_set_err(nullptr, Q3TTS_ERR_INVALID, "sample hook: bad shape");
    HK(hipSetDevice(device));
    DevBuf dl, dr, dout;
    if (dl.alloc((size_t)n * ld * 4) || dr.alloc((size_t)n * 4) || dout.alloc((size_t)n * 4)) return q3_set_err(nullptr, Q3TTS_ERR_OOM, "hipMalloc");
    HK(hipMemcpy(dl.p, logits, (size_t)n * ld * 4, hipMemcpyHostToDevice));
    if (r) HK(hipMemcpy(dr.p, r, (size_t)n * 4, hipMemcpyHostToDevice));
    q3_launch_sample_rows((const float*)dl.p, n, ld, limit, temperature, top_k, top_p, r ? (const float*)dr.p : nullptr, (int*)dout.p, nullptr);
    HK(hipDeviceSynchronize());
    HK(hipMemcpy(out, dout.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    return Q3TTS_OK;
}

// natural row-major bf16 rows <-> the A-tiled layout of the device buffers (q3_kernels.h)
static std::vector<uint16_t> atile_host(const uint16_t* src, int rows, int K) {
    std::vector<uint16_t> out((((size_t)rows + 15) & ~(size_t)15) * K, 0);
    for (int r = 0; r < rows; ++r) for (int k = 0; k < K; ++k) out[q3_atile_off(r, k, K >> 5)] = src[(size_t)r * K + k];
    return out;
}
static void untile_host(const std::vector<uint16_t>& t, int rows, int K, uint16_t* dst) {
    for (int r = 0; r < rows; ++r) for (int k = 0; k < K; ++k) dst[(size_t)r * K + k] = t[q3_atile_off(r, k, K >> 5)];
}

// the decoder's GEMM through its launcher (q3_bgemm.hip): xb bf16 bits [B][K]; w bf16 bits row-major [N][K] (epi 2: the F gate rows,
// then the F up rows); ssp [B][ntiles] or NULL; y in/out for epi 1. Mirrors oracle/q3_oracle_bf16.c q3o_bgemm.
extern "C" int q3tts_k_bgemm(int32_t device, const uint16_t* xb, int32_t B, int32_t K, const uint16_t* w, int32_t N, const float* ssp, int32_t ntiles,
                             int32_t d_norm, float eps, int32_t epi, const float* nw_next, float* y, uint16_t* yb, float* ssp_out, uint64_t* keys,
                             int32_t iters, float* mean_ms) {
    if (!xb || !w || B <= 0 || K % 256 || K < 256 || N % 16 || epi < 0 || epi > 3) return q3_set_err(nullptr, Q3TTS_ERR_INVALID, "bgemm hook: K % 256 == 0, N % 16 == 0");
    if (epi == Q3_EPI_SWIGLU && (N % 64 || !yb)) return q3_set_err(nullptr, Q3TTS_ERR_INVALID, "bgemm hook: swiglu needs N % 64 == 0 and yb");
    if (epi == Q3_EPI_RESID && nw_next && N % 32) return q3_set_err(nullptr, Q3TTS_ERR_INVALID, "bgemm hook: norm outputs need N % 32 == 0");
    if ((epi == Q3_EPI_STORE || epi == Q3_EPI_RESID) && !y) return q3_set_err(nullptr, Q3TTS_ERR_INVALID, "bgemm hook: y missing");
    if (epi == Q3_EPI_ARGMAX && !keys) return q3_set_err(nullptr, Q3TTS_ERR_INVALID, "bgemm hook: keys missing");
    if (epi == Q3_EPI_RESID && nw_next && (!yb || !ssp_out)) return q3_set_err(nullptr, Q3TTS_ERR_INVALID, "bgemm hook: norm outputs missing");
    HK(hipSetDevice(device));
    const int F = N / 2;
    const size_t B16 = ((size_t)B + 15) & ~(size_t)15;
    DevBuf dx, dw, dwt, ds, dn, dy, dyb, dso, dk;
    if (dx.alloc(B16 * K * 2) || dw.alloc((size_t)N * K * 2) || dwt.alloc((size_t)N * K * 2) || ds.alloc((size_t)B * (ntiles > 0 ? ntiles : 1) * 4) ||
        dn.alloc((size_t)N * 4) || dy.alloc((size_t)B * N * 4) || dyb.alloc(B16 * N * 2) || dso.alloc((size_t)B * (N / 16) * 4) || dk.alloc((size_t)B * (N / 16) * 8))
        return q3_set_err(nullptr, Q3TTS_ERR_OOM, "hipMalloc");
    { const std::vector<uint16_t> xt = atile_host(xb, B, K); HK(hipMemcpy(dx.p, xt.data(), xt.size() * 2, hipMemcpyHostToDevice)); }
    HK(hipMemcpy(dw.p, w, (size_t)N * K * 2, hipMemcpyHostToDevice));
    if (ssp) HK(hipMemcpy(ds.p, ssp, (size_t)B * ntiles * 4, hipMemcpyHostToDevice));
    if (nw_next) HK(hipMemcpy(dn.p, nw_next, (size_t)N * 4, hipMemcpyHostToDevice));
    if (epi == Q3_EPI_RESID) HK(hipMemcpy(dy.p, y, (size_t)B * N * 4, hipMemcpyHostToDevice));
    Q3Fill f{}; f.dst = (uint4*)dwt.p; f.N = N; f.K = K;
    if (epi == Q3_EPI_SWIGLU) { f.mode = 1; f.src_a = (const uint16_t*)dw.p; f.src_b = (const uint16_t*)dw.p + (size_t)F * K; }
    else { f.mode = 0; f.row0 = 0; f.rows = N; f.src_a = (const uint16_t*)dw.p; }
    q3_launch_fill_tiled(f, nullptr);
    Q3BGemm g{}; g.a = (const uint16_t*)dx.p; g.a_row0 = 0; g.B = B; g.w = (const uint4*)dwt.p; g.K = K; g.N = N;
    g.ssp = ssp ? (const float*)ds.p : nullptr; g.ld_ssp = ntiles; g.ntiles = ntiles; g.d_norm = d_norm; g.eps = eps; g.epi = epi;
    g.y = (float*)dy.p; g.ldy = N; g.yb = (uint16_t*)dyb.p;
    g.nw_next = nw_next ? (const float*)dn.p : nullptr; g.ssp_out = (float*)dso.p; g.ld_ssp_out = N / 16;
    g.keys = (unsigned long long*)dk.p; g.key_stride = N / 16;
    if (q3_launch_bgemm(g, nullptr)) return q3_set_err(nullptr, Q3TTS_ERR_INVALID, "bgemm: shape");
    HK(hipDeviceSynchronize());
    if (epi == Q3_EPI_STORE || epi == Q3_EPI_RESID) HK(hipMemcpy(y, dy.p, (size_t)B * N * 4, hipMemcpyDeviceToHost));
    if (epi == Q3_EPI_SWIGLU) { std::vector<uint16_t> t(B16 * F); HK(hipMemcpy(t.data(), dyb.p, t.size() * 2, hipMemcpyDeviceToHost)); untile_host(t, B, F, yb); }
    if (epi == Q3_EPI_RESID && nw_next) {
        { std::vector<uint16_t> t(B16 * N); HK(hipMemcpy(t.data(), dyb.p, t.size() * 2, hipMemcpyDeviceToHost)); untile_host(t, B, N, yb); }
        HK(hipMemcpy(ssp_out, dso.p, (size_t)B * (N / 16) * 4, hipMemcpyDeviceToHost));
    }
    if (epi == Q3_EPI_ARGMAX) {  // the kernel leaves one maximum per (row, 16-column tile); the consumer (here: the hook) takes the row maximum
        std::vector<uint64_t> parts((size_t)B * (N / 16));
        HK(hipMemcpy(parts.data(), dk.p, parts.size() * 8, hipMemcpyDeviceToHost));
        for (int b = 0; b < B; ++b) { uint64_t m = 0; for (int t = 0; t < N / 16; ++t) m = std::max(m, parts[(size_t)b * (N / 16) + t]); keys[b] = m; }
    }
    if (iters > 0 && mean_ms) {
        if (epi == Q3_EPI_RESID) { g.epi = Q3_EPI_STORE; g.nw_next = nullptr; }
        hipEvent_t a, b; HK(hipEventCreate(&a)); HK(hipEventCreate(&b));
        q3_launch_bgemm(g, nullptr);
        HK(hipEventRecord(a, nullptr));
        for (int i = 0; i < iters; ++i) q3_launch_bgemm(g, nullptr);
        HK(hipEventRecord(b, nullptr)); HK(hipEventSynchronize(b));
        float ms = 0; hipEventElapsedTime(&ms, a, b); *mean_ms = ms / iters;
        hipEventDestroy(a); hipEventDestroy(b);
    }
    return Q3TTS_OK;
}

// the same launch with ggml Q8_0 weights kept in block form: q int8 [N][K], d_f16 [N][K/32]. Mirrors oracle q3o_bgemm_q8.
extern "C" int q3tts_k_bgemm_q8(int32_t device, const uint16_t* xb, int32_t B, int32_t K, const int8_t* q, const uint16_t* d_f16, int32_t N, const float* ssp,
                                int32_t ntiles, int32_t d_norm, float eps, int32_t epi, const float* nw_next, float* y, uint16_t* yb, float* ssp_out,
                                uint64_t* keys, int32_t iters, float* mean_ms) {
    if (!xb || !q || !d_f16 || B <= 0 || K % 512 || K < 512 || N % 16 || epi < 0 || epi > 3) return q3_set_err(nullptr, Q3TTS_ERR_INVALID, "bgemm_q8 hook: K % 512 == 0, N % 16 == 0");
    if (epi == Q3_EPI_SWIGLU && (N % 64 || !yb)) return q3_set_err(nullptr, Q3TTS_ERR_INVALID, "bgemm_q8 hook: swiglu needs N % 64 == 0 and yb");
    if (epi == Q3_EPI_RESID && nw_next && (N % 32 || !yb || !ssp_out)) return q3_set_err(nullptr, Q3TTS_ERR_INVALID, "bgemm_q8 hook: norm outputs");
    if ((epi == Q3_EPI_STORE || epi == Q3_EPI_RESID) && !y) return q3_set_err(nullptr, Q3TTS_ERR_INVALID, "bgemm_q8 hook: y missing");
    if (epi == Q3_EPI_ARGMAX && !keys) return q3_set_err(nullptr, Q3TTS_ERR_INVALID, "bgemm_q8 hook: keys missing");
    HK(hipSetDevice(device));
    const int F = N / 2, kb = K / 32;
    const size_t B16 = ((size_t)B + 15) & ~(size_t)15;
    std::vector<uint8_t> blocks((size_t)N * kb * 34);  // the rows as a GGUF file holds them: block_q8_0 = f16 d, 32 x int8
    for (size_t n = 0; n < (size_t)N; ++n)
        for (int b = 0; b < kb; ++b) {
            uint8_t* blk = &blocks[(n * kb + b) * 34];
            const uint16_t dd = d_f16[n * kb + b];
            blk[0] = (uint8_t)(dd & 0xff); blk[1] = (uint8_t)(dd >> 8);
            memcpy(blk + 2, q + n * K + (size_t)b * 32, 32);
        }
    DevBuf dx, dw, dwt, dsc, ds, dn, dy, dyb, dso, dk;
    if (dx.alloc(B16 * K * 2) || dw.alloc(blocks.size()) || dwt.alloc((size_t)N * K) || dsc.alloc((size_t)N * kb * 2) || ds.alloc((size_t)B * (ntiles > 0 ? ntiles : 1) * 4) ||
        dn.alloc((size_t)N * 4) || dy.alloc((size_t)B * N * 4) || dyb.alloc(B16 * N * 2) || dso.alloc((size_t)B * (N / 16) * 4) || dk.alloc((size_t)B * (N / 16) * 8))
        return q3_set_err(nullptr, Q3TTS_ERR_OOM, "hipMalloc");
    { const std::vector<uint16_t> xt = atile_host(xb, B, K); HK(hipMemcpy(dx.p, xt.data(), xt.size() * 2, hipMemcpyHostToDevice)); }
    HK(hipMemcpy(dw.p, blocks.data(), blocks.size(), hipMemcpyHostToDevice));
    if (ssp) HK(hipMemcpy(ds.p, ssp, (size_t)B * ntiles * 4, hipMemcpyHostToDevice));
    if (nw_next) HK(hipMemcpy(dn.p, nw_next, (size_t)N * 4, hipMemcpyHostToDevice));
    if (epi == Q3_EPI_RESID) HK(hipMemcpy(dy.p, y, (size_t)B * N * 4, hipMemcpyHostToDevice));
    Q3Fill f{}; f.dst = (uint4*)dwt.p; f.dst_scale = (uint16_t*)dsc.p; f.N = N; f.K = K;
    if (epi == Q3_EPI_SWIGLU) { f.mode = 1; f.src8_a = (const uint8_t*)dw.p; f.src8_b = (const uint8_t*)dw.p + (size_t)F * kb * 34; }
    else { f.mode = 0; f.row0 = 0; f.rows = N; f.src8_a = (const uint8_t*)dw.p; }
    q3_launch_fill_tiled_q8(f, nullptr);
    Q3BGemm g{}; g.a = (const uint16_t*)dx.p; g.a_row0 = 0; g.B = B; g.w = (const uint4*)dwt.p; g.wscale = (const uint16_t*)dsc.p; g.K = K; g.N = N;
    g.ssp = ssp ? (const float*)ds.p : nullptr; g.ld_ssp = ntiles; g.ntiles = ntiles; g.d_norm = d_norm; g.eps = eps; g.epi = epi;
    g.y = (float*)dy.p; g.ldy = N; g.yb = (uint16_t*)dyb.p;
    g.nw_next = nw_next ? (const float*)dn.p : nullptr; g.ssp_out = (float*)dso.p; g.ld_ssp_out = N / 16;
    g.keys = (unsigned long long*)dk.p; g.key_stride = N / 16;
    if (q3_launch_bgemm(g, nullptr)) return q3_set_err(nullptr, Q3TTS_ERR_INVALID, "bgemm_q8: shape");
    HK(hipDeviceSynchronize());
    if (epi == Q3_EPI_STORE || epi == Q3_EPI_RESID) HK(hipMemcpy(y, dy.p, (size_t)B * N * 4, hipMemcpyDeviceToHost));
    if (epi == Q3_EPI_SWIGLU) { std::vector<uint16_t> t(B16 * F); HK(hipMemcpy(t.data(), dyb.p, t.size() * 2, hipMemcpyDeviceToHost)); untile_host(t, B, F, yb); }
    if (epi == Q3_EPI_RESID && nw_next) {
        { std::vector<uint16_t> t(B16 * N); HK(hipMemcpy(t.data(), dyb.p, t.size() * 2, hipMemcpyDeviceToHost)); untile_host(t, B, N, yb); }
        HK(hipMemcpy(ssp_out, dso.p, (size_t)B * (N / 16) * 4, hipMemcpyDeviceToHost));
    }
    if (epi == Q3_EPI_ARGMAX) {
        std::vector<uint64_t> parts((size_t)B * (N / 16));
        HK(hipMemcpy(parts.data(), dk.p, parts.size() * 8, hipMemcpyDeviceToHost));
        for (int b = 0; b < B; ++b) { uint64_t m = 0; for (int t = 0; t < N / 16; ++t) m = std::max(m, parts[(size_t)b * (N / 16) + t]); keys[b] = m; }
    }
    if (iters > 0 && mean_ms) {
        if (epi == Q3_EPI_RESID) { g.epi = Q3_EPI_STORE; g.nw_next = nullptr; }
        hipEvent_t a, b; HK(hipEventCreate(&a)); HK(hipEventCreate(&b));
        q3_launch_bgemm(g, nullptr);
        HK(hipEventRecord(a, nullptr));
        for (int i = 0; i < iters; ++i) q3_launch_bgemm(g, nullptr);
        HK(hipEventRecord(b, nullptr)); HK(hipEventSynchronize(b));
        float ms = 0; hipEventElapsedTime(&ms, a, b); *mean_ms = ms / iters;
        hipEventDestroy(a); hipEventDestroy(b);
    }
    return Q3TTS_OK;
}

// H6 through the projection kernel: y[rows][n_out] = bias + sum x * w (reference order); nw != NULL: the rows' norm inputs too
// The vocoder's extras of the decoder GEMM (bias, GELU -> bf16, LayerScale column scale, per-slot row segments, a bf16 copy of the
// residual result) through one hook: epi 0 (store) / 1 (residual) / 4 (GELU). y0 / y are dense [B][N]; with seg_rows > 0 the kernel
// works on a buffer of B / seg_rows segments, each preceded by gap_rows sentinel rows that must come back untouched.
// W8A8 (q3_bgemm8.hip): activations and weights as ggml Q8_0 blocks in natural order in / out; the hook tiles them for the device
extern "C" int q3tts_k_bgemm_q8a8(int32_t device, const int8_t* aq, const uint16_t* ad, int32_t B, int32_t K, const int8_t* q, const uint16_t* d_f16, int32_t N,
                                  const float* ssp, int32_t ntiles, int32_t d_norm, float eps, int32_t epi, const float* nw_next, float* y, int8_t* yq, uint16_t* yd,
                                  float* ssp_out, int32_t iters, float* mean_ms) {
    if (!aq || !ad || !q || !d_f16 || B <= 0 || K % 512 || K < 512 || N % 32 || epi < 0 || epi > 2) return q3_set_err(nullptr, Q3TTS_ERR_INVALID, "bgemm_q8a8 hook: K % 512 == 0, N % 32 == 0, epilogue 0..2");
    if (epi == Q3_EPI_SWIGLU && (N % 128 || !yq || !yd)) return q3_set_err(nullptr, Q3TTS_ERR_INVALID, "bgemm_q8a8 hook: swiglu needs N % 128 == 0, yq, yd");
    if (epi == Q3_EPI_RESID && (N % 64 || !nw_next || !yq || !yd || !ssp_out || !y)) return q3_set_err(nullptr, Q3TTS_ERR_INVALID, "bgemm_q8a8 hook: residual needs N % 64 == 0, nw_next, y, yq, yd, ssp_out");
    if (epi == Q3_EPI_STORE && !y) return q3_set_err(nullptr, Q3TTS_ERR_INVALID, "bgemm_q8a8 hook: y missing");
    HK(hipSetDevice(device));
    const int F = N / 2, kb = K / 32, Nout = epi == Q3_EPI_SWIGLU ? F : N;
    const size_t B16 = ((size_t)B + 15) & ~(size_t)15; const int rt16 = (int)(B16 / 16);
    std::vector<uint8_t> blocks((size_t)N * kb * 34);
    for (size_t n = 0; n < (size_t)N; ++n)
        for (int b = 0; b < kb; ++b) {
            uint8_t* blk = &blocks[(n * kb + b) * 34];
            const uint16_t dd = d_f16[n * kb + b];
            blk[0] = (uint8_t)(dd & 0xff); blk[1] = (uint8_t)(dd >> 8);
            memcpy(blk + 2, q + n * K + (size_t)b * 32, 32);
        }
    std::vector<int8_t> at(B16 * K, 0); std::vector<uint16_t> ast((size_t)kb * B16, 0);
    for (int r = 0; r < B; ++r) {
        for (int k = 0; k < K; ++k) at[q3_q8_off(r, k, K >> 6)] = aq[(size_t)r * K + k];
        for (int b = 0; b < kb; ++b) ast[q3_q8_scale_idx(r, b, rt16)] = ad[(size_t)r * kb + b];
    }
    DevBuf dx, dxs, dw, dwt, dsc, ds, dn, dy, dyq, dys, dso;
    if (dx.alloc(at.size()) || dxs.alloc(ast.size() * 2) || dw.alloc(blocks.size()) || dwt.alloc((size_t)N * K) || dsc.alloc((size_t)N * kb * 2) ||
        ds.alloc((size_t)B * (ntiles > 0 ? ntiles : 1) * 4) || dn.alloc((size_t)N * 4) || dy.alloc((size_t)B * N * 4) || dyq.alloc(B16 * Nout) ||
        dys.alloc((size_t)(Nout / 32 + 2) * B16 * 2) || dso.alloc((size_t)B * (N / 16) * 4))
        return q3_set_err(nullptr, Q3TTS_ERR_OOM, "hipMalloc");
    HK(hipMemcpy(dx.p, at.data(), at.size(), hipMemcpyHostToDevice)); HK(hipMemcpy(dxs.p, ast.data(), ast.size() * 2, hipMemcpyHostToDevice));
    HK(hipMemcpy(dw.p, blocks.data(), blocks.size(), hipMemcpyHostToDevice));
    if (ssp) HK(hipMemcpy(ds.p, ssp, (size_t)B * ntiles * 4, hipMemcpyHostToDevice));
    if (nw_next) HK(hipMemcpy(dn.p, nw_next, (size_t)N * 4, hipMemcpyHostToDevice));
    if (epi == Q3_EPI_RESID) HK(hipMemcpy(dy.p, y, (size_t)B * N * 4, hipMemcpyHostToDevice));
    Q3Fill f{}; f.dst = (uint4*)dwt.p; f.dst_scale = (uint16_t*)dsc.p; f.N = N; f.K = K;
    if (epi == Q3_EPI_SWIGLU) { f.mode = 1; f.src8_a = (const uint8_t*)dw.p; f.src8_b = (const uint8_t*)dw.p + (size_t)F * kb * 34; }
    else { f.mode = 0; f.row0 = 0; f.rows = N; f.src8_a = (const uint8_t*)dw.p; }
    q3_launch_fill_tiled_q8(f, nullptr);
    Q3BGemm g{}; g.a = (const uint16_t*)dx.p; g.ascale = (const uint16_t*)dxs.p; g.a_rt16 = rt16; g.a_row0 = 0; g.B = B;
    g.w = (const uint4*)dwt.p; g.wscale = (const uint16_t*)dsc.p; g.K = K; g.N = N;
    g.ssp = ssp ? (const float*)ds.p : nullptr; g.ld_ssp = ntiles; g.ntiles = ntiles; g.d_norm = d_norm; g.eps = eps; g.epi = epi;
    g.y = (float*)dy.p; g.ldy = N; g.yb = (uint16_t*)dyq.p; g.yscale = (uint16_t*)dys.p; g.y_rt16 = rt16;
    g.nw_next = nw_next ? (const float*)dn.p : nullptr; g.ssp_out = (float*)dso.p; g.ld_ssp_out = N / 16;
    if (q3_launch_bgemm8(g, nullptr)) return q3_set_err(nullptr, Q3TTS_ERR_INVALID, "bgemm_q8a8: shape");
    HK(hipDeviceSynchronize());
    if (epi == Q3_EPI_STORE || epi == Q3_EPI_RESID) HK(hipMemcpy(y, dy.p, (size_t)B * N * 4, hipMemcpyDeviceToHost));
    if (epi != Q3_EPI_STORE) {
        std::vector<int8_t> qt(B16 * Nout); std::vector<uint16_t> st((size_t)(Nout / 32) * B16);
        HK(hipMemcpy(qt.data(), dyq.p, qt.size(), hipMemcpyDeviceToHost)); HK(hipMemcpy(st.data(), dys.p, st.size() * 2, hipMemcpyDeviceToHost));
        for (int r = 0; r < B; ++r) {
            for (int k = 0; k < Nout; ++k) yq[(size_t)r * Nout + k] = qt[q3_q8_off(r, k, Nout >> 6)];
            for (int b = 0; b < Nout / 32; ++b) yd[(size_t)r * (Nout / 32) + b] = st[q3_q8_scale_idx(r, b, rt16)];
        }
        if (epi == Q3_EPI_RESID) HK(hipMemcpy(ssp_out, dso.p, (size_t)B * (N / 16) * 4, hipMemcpyDeviceToHost));
    }
    if (iters > 0 && mean_ms) {
        if (epi == Q3_EPI_RESID) g.epi = Q3_EPI_STORE;
        hipEvent_t a, b; HK(hipEventCreate(&a)); HK(hipEventCreate(&b));
        q3_launch_bgemm8(g, nullptr);
        HK(hipEventRecord(a, nullptr));
        for (int i = 0; i < iters; ++i) q3_launch_bgemm8(g, nullptr);
        HK(hipEventRecord(b, nullptr)); HK(hipEventSynchronize(b));
        float ms = 0; hipEventElapsedTime(&ms, a, b); *mean_ms = ms / iters;
        hipEventDestroy(a); hipEventDestroy(b);
    }
    return Q3TTS_OK;
}

extern "C" int q3tts_k_bgemm_voc(int32_t device, const uint16_t* xb, int32_t B, int32_t K, const uint16_t* w, int32_t N, int32_t epi, const float* bias,
                                 int32_t bias_n, const float* col_scale, int32_t seg_rows, int32_t gap_rows, float* y, uint16_t* yb, int32_t want_yb) {
    if (!xb || !w || B <= 0 || K % 256 || K < 256 || N % 32 || (epi != Q3_EPI_STORE && epi != Q3_EPI_RESID && epi != Q3_EPI_GELU))
        return q3_set_err(nullptr, Q3TTS_ERR_INVALID, "bgemm_voc hook: K % 256 == 0, N % 32 == 0, epilogue 0 / 1 / 4");
    if ((epi != Q3_EPI_GELU && !y) || ((epi == Q3_EPI_GELU || want_yb) && !yb)) return q3_set_err(nullptr, Q3TTS_ERR_INVALID, "bgemm_voc hook: output missing");
    if (seg_rows < 0 || gap_rows < 0 || (seg_rows > 0 && B % seg_rows) || (bias && bias_n < 1)) return q3_set_err(nullptr, Q3TTS_ERR_INVALID, "bgemm_voc hook: segments / bias");
    HK(hipSetDevice(device));
    const size_t B16 = ((size_t)B + 15) & ~(size_t)15;
    const int T = seg_rows > 0 ? seg_rows : B, S = B / T, P = T + (seg_rows > 0 ? gap_rows : 0);
    DevBuf dx, dw, dwt, db, dc, dy, dyb;
    if (dx.alloc(B16 * K * 2) || dw.alloc((size_t)N * K * 2) || dwt.alloc((size_t)N * K * 2) || db.alloc((size_t)(bias ? bias_n : 1) * 4) || dc.alloc((size_t)N * 4) ||
        dy.alloc((size_t)S * P * N * 4) || dyb.alloc(B16 * N * 2))
        return q3_set_err(nullptr, Q3TTS_ERR_OOM, "hipMalloc");
    { const std::vector<uint16_t> xt = atile_host(xb, B, K); HK(hipMemcpy(dx.p, xt.data(), xt.size() * 2, hipMemcpyHostToDevice)); }
    HK(hipMemcpy(dw.p, w, (size_t)N * K * 2, hipMemcpyHostToDevice));
    if (bias) HK(hipMemcpy(db.p, bias, (size_t)bias_n * 4, hipMemcpyHostToDevice));
    if (col_scale) HK(hipMemcpy(dc.p, col_scale, (size_t)N * 4, hipMemcpyHostToDevice));
    std::vector<float> seg((size_t)S * P * N, -12345.5f);  // sentinel in the gap rows
    if (epi != Q3_EPI_GELU)
        for (int sidx = 0; sidx < S; ++sidx)
            for (int t = 0; t < T; ++t) memcpy(&seg[((size_t)sidx * P + (P - T) + t) * N], y + ((size_t)sidx * T + t) * N, (size_t)N * 4);
    HK(hipMemcpy(dy.p, seg.data(), seg.size() * 4, hipMemcpyHostToDevice));
    Q3Fill f{}; f.dst = (uint4*)dwt.p; f.N = N; f.K = K; f.mode = 0; f.row0 = 0; f.rows = N; f.src_a = (const uint16_t*)dw.p;
    q3_launch_fill_tiled(f, nullptr);
    Q3BGemm g{}; g.a = (const uint16_t*)dx.p; g.B = B; g.w = (const uint4*)dwt.p; g.K = K; g.N = N; g.epi = epi;
    g.y = (float*)dy.p + (size_t)(P - T) * N; g.ldy = N;
    if (seg_rows > 0) { g.seg_rows = T; g.seg_stride = (size_t)P * N; }
    g.bias = bias ? (const float*)db.p : nullptr; g.bias_n = bias_n; g.col_scale = col_scale ? (const float*)dc.p : nullptr;
    if (epi == Q3_EPI_GELU || want_yb) g.yb = (uint16_t*)dyb.p;
    if (q3_launch_bgemm(g, nullptr)) return q3_set_err(nullptr, Q3TTS_ERR_INVALID, "bgemm_voc: shape");
    HK(hipDeviceSynchronize());
    if (epi != Q3_EPI_GELU) {
        HK(hipMemcpy(seg.data(), dy.p, seg.size() * 4, hipMemcpyDeviceToHost));
        for (int sidx = 0; sidx < S; ++sidx) {
            for (int t = 0; t < P - T; ++t)
                for (int c = 0; c < N; ++c)
                    if (seg[((size_t)sidx * P + t) * N + c] != -12345.5f) return q3_set_err(nullptr, Q3TTS_ERR_DEVICE, "bgemm_voc: a gap row was written");
            for (int t = 0; t < T; ++t) memcpy(y + ((size_t)sidx * T + t) * N, &seg[((size_t)sidx * P + (P - T) + t) * N], (size_t)N * 4);
        }
    }
    if (epi == Q3_EPI_GELU || want_yb) { std::vector<uint16_t> t(B16 * N); HK(hipMemcpy(t.data(), dyb.p, t.size() * 2, hipMemcpyDeviceToHost)); untile_host(t, B, N, yb); }
    return Q3TTS_OK;
}

extern "C" int q3tts_k_project(int32_t device, const float* x, int32_t rows, int32_t n_in, const float* w, const float* bias, int32_t n_out, const float* nw,
                               float* y, uint16_t* xb, float* ssp) {
    if (!x || !w || !bias || !y || rows <= 0 || n_in % 64 || n_out % 16 || (nw && (!xb || !ssp))) return q3_set_err(nullptr, Q3TTS_ERR_INVALID, "project hook: n_in % 64 == 0, n_out % 16 == 0");
    HK(hipSetDevice(device));
    DevBuf dx, dw, db, dn, dy, dxb, dss;
    if (dx.alloc((size_t)rows * n_in * 4) || dw.alloc((size_t)n_out * n_in * 4) || db.alloc((size_t)n_out * 4) || dn.alloc((size_t)n_out * 4) ||
        dy.alloc((size_t)rows * n_out * 4) || dxb.alloc((((size_t)rows + 15) & ~(size_t)15) * n_out * 2) || dss.alloc((size_t)rows * (n_out / 16) * 4))
        return q3_set_err(nullptr, Q3TTS_ERR_OOM, "hipMalloc");
    HK(hipMemcpy(dx.p, x, (size_t)rows * n_in * 4, hipMemcpyHostToDevice));
    HK(hipMemcpy(dw.p, w, (size_t)n_out * n_in * 4, hipMemcpyHostToDevice));
    HK(hipMemcpy(db.p, bias, (size_t)n_out * 4, hipMemcpyHostToDevice));
    if (nw) HK(hipMemcpy(dn.p, nw, (size_t)n_out * 4, hipMemcpyHostToDevice));
    Q3Project pj{}; pj.x = (const float*)dx.p; pj.ldx = n_in; pj.rows = rows; pj.w = (const float*)dw.p; pj.bias = (const float*)db.p; pj.n_in = n_in; pj.n_out = n_out;
    pj.y = (float*)dy.p; pj.ldy = n_out; pj.nw = nw ? (const float*)dn.p : nullptr; pj.xb = (uint16_t*)dxb.p; pj.ssp = (float*)dss.p; pj.ld_ssp = n_out / 16;
    if (q3_launch_project(pj, nullptr)) return q3_set_err(nullptr, Q3TTS_ERR_INVALID, "project: shape");
    HK(hipDeviceSynchronize());
    HK(hipMemcpy(y, dy.p, (size_t)rows * n_out * 4, hipMemcpyDeviceToHost));
    if (nw) {
        if (n_out % 32) return q3_set_err(nullptr, Q3TTS_ERR_INVALID, "project hook: norm outputs need n_out % 32 == 0");
        std::vector<uint16_t> t((((size_t)rows + 15) & ~(size_t)15) * n_out);
        HK(hipMemcpy(t.data(), dxb.p, t.size() * 2, hipMemcpyDeviceToHost)); untile_host(t, rows, n_out, xb);
        HK(hipMemcpy(ssp, dss.p, (size_t)rows * (n_out / 16) * 4, hipMemcpyDeviceToHost));
    }
    return Q3TTS_OK;
}

// producer side of the split RMSNorm for plain f32 rows (d % 256 == 0)
extern "C" int q3tts_k_norm_inputs(int32_t device, const float* x, int32_t rows, int32_t d, const float* nw, uint16_t* xb, float* ssp) {
    if (!x || !nw || !xb || !ssp || rows <= 0 || d % 256) return q3_set_err(nullptr, Q3TTS_ERR_INVALID, "norm-inputs hook: d % 256 == 0");
    HK(hipSetDevice(device));
    DevBuf dx, dn, dxb, dss;
    const size_t r16 = ((size_t)rows + 15) & ~(size_t)15;
    if (dx.alloc((size_t)rows * d * 4) || dn.alloc((size_t)d * 4) || dxb.alloc(r16 * d * 2) || dss.alloc((size_t)rows * (d / 16) * 4)) return q3_set_err(nullptr, Q3TTS_ERR_OOM, "hipMalloc");
    HK(hipMemcpy(dx.p, x, (size_t)rows * d * 4, hipMemcpyHostToDevice));
    HK(hipMemcpy(dn.p, nw, (size_t)d * 4, hipMemcpyHostToDevice));
    q3_launch_norm_inputs((const float*)dx.p, d, rows, d, (const float*)dn.p, (uint16_t*)dxb.p, 0, (float*)dss.p, d / 16, nullptr);
    HK(hipDeviceSynchronize());
    { std::vector<uint16_t> t(r16 * d); HK(hipMemcpy(t.data(), dxb.p, t.size() * 2, hipMemcpyDeviceToHost)); untile_host(t, rows, d, xb); }
    HK(hipMemcpy(ssp, dss.p, (size_t)rows * (d / 16) * 4, hipMemcpyDeviceToHost));
    return Q3TTS_OK;
}

// one v_mfma_f32_16x16x32_bf16 chain per case (test hook: pins the instruction's accumulation arithmetic against the
// oracle's integer restatement, oracle/q3_oracle.c q3o_mfma_bf16_dot32)
typedef float q3_f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 q3_bf16x8 __attribute__((ext_vector_type(8)));
__global__ void k_mfma_bf16_cases(const uint16_t* A, const uint16_t* B, const float* C, float* D, int chain) {
    const int l = threadIdx.x, cs = blockIdx.x;
    const uint16_t* a = A + (size_t)cs * chain * 512; const uint16_t* b = B + (size_t)cs * chain * 512;
    q3_f32x4 acc;
    for (int j = 0; j < 4; ++j) acc[j] = C[(size_t)cs * 256 + (4 * (l >> 4) + j) * 16 + (l & 15)];
    for (int st = 0; st < chain; ++st) {
        union { q3_bf16x8 v; uint16_t u[8]; } af, bf;
        for (int j = 0; j < 8; ++j) {  // lane l holds A[row l & 15][k = 8 (l >> 4) + j] and B[k = 8 (l >> 4) + j][col l & 15]
            af.u[j] = a[(size_t)st * 512 + (l & 15) * 32 + 8 * (l >> 4) + j];
            bf.u[j] = b[(size_t)st * 512 + (8 * (l >> 4) + j) * 16 + (l & 15)];
        }
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af.v, bf.v, acc, 0, 0, 0);
    }
    for (int j = 0; j < 4; ++j) D[(size_t)cs * 256 + (4 * (l >> 4) + j) * 16 + (l & 15)] = acc[j];
}
extern "C" int q3tts_k_mfma_bf16(int32_t device, const uint16_t* a, const uint16_t* b, const float* c, float* d, int32_t n_cases, int32_t chain) {
    if (!a || !b || !c || !d || n_cases <= 0 || chain <= 0) return q3_set_err(nullptr, Q3TTS_ERR_INVALID, "mfma hook: bad shape");
    HK(hipSetDevice(device));
    const size_t nab = (size_t)n_cases * chain * 512 * 2, ncd = (size_t)n_cases * 256 * 4;
    DevBuf da, db, dc, dd;
    if (da.alloc(nab) || db.alloc(nab) || dc.alloc(ncd) || dd.alloc(ncd)) return q3_set_err(nullptr, Q3TTS_ERR_OOM, "hipMalloc");
    HK(hipMemcpy(da.p, a, nab, hipMemcpyHostToDevice)); HK(hipMemcpy(db.p, b, nab, hipMemcpyHostToDevice)); HK(hipMemcpy(dc.p, c, ncd, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_mfma_bf16_cases, dim3(n_cases), dim3(64), 0, nullptr, (const uint16_t*)da.p, (const uint16_t*)db.p, (const float*)dc.p, (float*)dd.p, chain);
    HK(hipDeviceSynchronize());
    HK(hipMemcpy(d, dd.p, ncd, hipMemcpyDeviceToHost));
    return Q3TTS_OK;
}

extern "C" int q3tts_k_talker_prefill(q3tts_engine* e, const float* embd, int32_t n_tok, float* hidden_out, float* logits_out) {
    if (!e || !embd || n_tok <= 0) return q3_set_err(e, Q3TTS_ERR_INVALID, "null argument");
    Q3_HIP(e, hipSetDevice(e->cfg.device));
    q3tts_request r{}; r.prompt_embd = embd; r.n_tok = n_tok; r.use_engine_sampler = 0; r.temperature = 0; r.max_steps = 1;
    TRY(plan_rows(e, std::vector<int>{0}));
    TRY(admit(e, 0, &r));
    const q3tts_model_config& m = e->cfg.model;
    hipStream_t s = e->stream;
    if (hidden_out) {
        q3_launch_rmsnorm_rows(e->lanes[0].xT, m.t_d_model, e->T.out_norm, m.rms_eps, m.t_d_model, 1, e->lanes[0].logits_tmp, m.t_d_model, s);
        Q3_HIP(e, hipMemcpyAsync(hidden_out, e->lanes[0].logits_tmp, (size_t)m.t_d_model * 4, hipMemcpyDeviceToHost, s));
    }
    if (logits_out) Q3_HIP(e, hipMemcpyAsync(logits_out, e->lanes[0].logits, (size_t)m.t_vocab * 4, hipMemcpyDeviceToHost, s));
    Q3Slot* stage = e->slots_host + e->B; memset(stage, 0, sizeof(Q3Slot));
    Q3_HIP(e, hipMemcpyAsync(e->slots, stage, sizeof(Q3Slot), hipMemcpyHostToDevice, s));  // retire the slot again
    Q3_HIP(e, hipStreamSynchronize(s));
    return Q3TTS_OK;
}

extern "C" int q3tts_k_probe(q3tts_engine* e, int32_t enable) {
    if (!e) return Q3TTS_ERR_INVALID;
    Q3_HIP(e, hipSetDevice(e->cfg.device));
    if (enable && e->probe_ev.empty()) {
        e->probe_ev.resize(10, nullptr);  // 4 frames x 2 + one empty bracket per chunk (event overhead calibration)
        for (auto& ev : e->probe_ev) Q3_HIP(e, hipEventCreate(&ev));
    }
    const int model = enable & 15, kind = enable >> 4;
    if (model > 2 || kind < 0 || kind > 4) return q3_set_err(e, Q3TTS_ERR_INVALID, "probe: model 0..2, kind 0..4");
    e->probe = model; e->probe_kind = kind;
    return Q3TTS_OK;
}

// The allocator's contract (q3_dev_alloc_zeroed): the zero fill has completed when the pointer is handed out. The hook allocates
// `bytes` through it, uploads a pattern into the first and last 4 KiB on the NULL stream at once (the stream an unsuspecting call
// site would use), and counts the bytes that do not read back as written / as zero. An asynchronous fill on e->stream — the state
// of rounds 1 and 2 — loses this race on large buffers.
extern "C" int q3tts_k_alloc_upload(q3tts_engine* e, int64_t bytes, int64_t* mismatches) {
    if (!e || !mismatches || bytes < 16384) return q3_set_err(e, Q3TTS_ERR_INVALID, "alloc hook: bytes >= 16384");
    Q3_HIP(e, hipSetDevice(e->cfg.device));
    void* q = nullptr;
    TRY(q3_dev_alloc_zeroed(e, &q, (size_t)bytes));
    std::vector<uint8_t> pat(4096), back(12288);
    for (int i = 0; i < 4096; ++i) pat[i] = (uint8_t)(1 + i % 251);
    hipError_t er = hipMemcpyAsync(q, pat.data(), 4096, hipMemcpyHostToDevice, nullptr);
    if (er == hipSuccess) er = hipMemcpyAsync((char*)q + bytes - 4096, pat.data(), 4096, hipMemcpyHostToDevice, nullptr);
    if (er == hipSuccess) er = hipDeviceSynchronize();
    if (er == hipSuccess) er = hipMemcpy(back.data(), q, 8192, hipMemcpyDeviceToHost);
    if (er == hipSuccess) er = hipMemcpy(back.data() + 8192, (char*)q + bytes - 4096, 4096, hipMemcpyDeviceToHost);
    hipFree(q);
    if (er != hipSuccess) return q3_set_err(e, Q3TTS_ERR_DEVICE, hipGetErrorString(er));
    int64_t bad = 0;
    for (int i = 0; i < 4096; ++i) bad += (back[i] != pat[i]) + (back[4096 + i] != 0) + (back[8192 + i] != pat[i]);
    *mismatches = bad;
    return Q3TTS_OK;
}

extern "C" int q3tts_k_bgemm_policy(int32_t big) {
    if (big < -1 || big > 1) return q3_set_err(nullptr, Q3TTS_ERR_INVALID, "bgemm policy: -1, 0 or 1");
    q3_bgemm_big_policy(big);
    return Q3TTS_OK;
}

extern "C" int q3tts_k_attend_policy(int32_t decode, int32_t prefill) {
    if (decode < 0 || decode > 1 || prefill < 0 || prefill > 2) return q3_set_err(nullptr, Q3TTS_ERR_INVALID, "attend policy: decode 0..1, prefill 0..2");
    q3_attend_policy(decode, prefill);
    return Q3TTS_OK;
}

extern "C" int q3tts_k_bgemm_pick(int32_t B, int32_t K, int32_t N, int32_t epilogue, int32_t w_once, int32_t q8, int32_t* out5) {
    if (!out5 || B < 1 || K < 256 || K % 256 || N % 16) return q3_set_err(nullptr, Q3TTS_ERR_INVALID, "bgemm pick: bad shape");
    Q3BGemm g{}; g.B = B; g.K = K; g.N = N; g.epi = epilogue; g.w_once = w_once;
    g.a = (const uint16_t*)16; g.w = (const uint4*)16; g.wscale = q8 ? (const uint16_t*)16 : nullptr;  // (never dereferenced: nothing is launched)
    if (epilogue == Q3_EPI_SWIGLU || epilogue == Q3_EPI_GELU) g.yb = (uint16_t*)16;
    if (epilogue == Q3_EPI_RESID) { g.yb = (uint16_t*)16; g.nw_next = (const float*)16; }
    int rt, nt, d, ntw, big;
    q3_bgemm_pick(g, &rt, &nt, &d, &ntw, &big);
    out5[0] = rt; out5[1] = nt; out5[2] = d; out5[3] = ntw; out5[4] = big;
    return Q3TTS_OK;
}

extern "C" int q3tts_k_rng_f32(uint64_t seed, int32_t n, float* out) {
    if (!out || n < 0) return Q3TTS_ERR_INVALID;
    q3_stdrng_f32(seed, n, out);
    return Q3TTS_OK;
}
