// q3_engine.h — internal engine state of libq3tts (host side, C++). The public surface is include/q3tts.h.
#pragma once
#include <algorithm>
#include <string>
#include <vector>

#include "../../include/q3tts.h"
#include "q3_kernels.h"

struct Q3Voc;  // vocoder (q3_vocoder.hip)
struct Q3Mel;  // log-mel front-end (q3_mel.hip)
struct Q3Clone;  // voice-clone encoders (q3_clone.hip)

struct Q3Tfm {
    int L = 0, d = 0, Hq = 0, Hkv = 0, hd = 0, F = 0, nq = 0, nkv = 0, nqkv = 0, head_n = 0;
    std::vector<float*> attn_norm, ffn_norm, qn, kn;
    std::vector<uint4*> wqkv, wo, wgu, wd;
    float* out_norm = nullptr;
    uint4* head = nullptr;
    // Q8_0 mode (cfg.talker_q8_0, the Talker only): the matrices above hold ggml block quants in the tiled Q8 layout (q3_kernels.h) and
    // these the f16 block scales [N][K/32]; empty / null = bf16 weights
    std::vector<uint16_t*> sqkv, so, sgu, sd; uint16_t* shead = nullptr; bool q8 = false;
    bool a8 = false;  // cfg.talker_q8_0 = 2: the GEMMs' ACTIVATIONS are Q8_0 blocks as well (W8A8, q3_bgemm8.hip): every operand buffer then holds int8 quants + f16 block scales
    uint16_t *kc = nullptr, *vc = nullptr;  // [L][slots][Hkv][n_ctx*hd]
    size_t layer_stride = 0;
    int n_ctx = 0, n_slots = 0;
    float *cs = nullptr, *sn = nullptr;  // RoPE tables [n_ctx][hd/2]
    size_t weight_bytes = 0;              // bf16 matrix bytes of all layers + head
};

struct Q3Scratch {
    float* qkv = nullptr;                   // [rows][nqkv] f32 (the attention kernel's input)
    uint16_t *att = nullptr, *h = nullptr;  // bf16 rows: attention output [rows][nq], SwiGLU output [rows][F] (GEMM operands)
    uint16_t *asc_att = nullptr, *asc_h = nullptr; int rt16 = 0;  // W8A8: the f16 block scales of att / h (q3_q8_scale_idx), row tiles of the buffers
    int rows = 0;
};

// the decode rows: per-row buffers, the row -> slot map and one captured frame-step graph per row-count bucket
struct Q3Lane {
    int nb = 0;                       // row capacity = max_batch
    hipStream_t stream = nullptr;
    float *xT = nullptr, *logits = nullptr, *logits_tmp = nullptr, *fb = nullptr, *px = nullptr;
    // norm inputs of the residual rows: bf16(x * nw) and the per-tile sums of squares, Talker [nb] / Predictor [2 nb]
    uint16_t *xbT = nullptr, *xbP = nullptr;
    float *sspT = nullptr, *sspP = nullptr;
    uint16_t* ascT = nullptr; int rt16T = 0;   // W8A8: block scales of xbT
    unsigned long long* keys = nullptr;
    int *row_pos_t = nullptr, *slot_id = nullptr, *posA = nullptr, *slotA = nullptr, *pos_q = nullptr, *perm = nullptr;
    Q3Scratch sc;
    std::vector<hipGraph_t> graphs;          // per bucket
    std::vector<hipGraphExec_t> execs;
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
};

struct q3tts_engine {
    q3tts_engine_config cfg;
    std::string err;
    hipStream_t stream = nullptr, vstream = nullptr;
    Q3Tfm T, P;
    // assets
    float* text = nullptr;
    std::vector<float*> codec;            // host array of device pointers
    const float** codec_dev = nullptr;    // device array of the same pointers
    std::vector<float*> pproj;            // proj(codec table q) [rows_q][p_d_model] f32: the Predictor's inputs are gathers
    float* proj_w = nullptr;              // f32 [p_d_model][d_embed], as the reference keeps it (src/assets_manager.rs:212-241)
    float* proj_b = nullptr;
    float* tts_pad = nullptr;             // = text[tts_pad_id] (or tts_pad_own: zeros, when the loaded text table is too small)
    float* tts_pad_own = nullptr;
    float* marker_row = nullptr;          // text[tts_pad_id] through the out-of-range rule (the clone prompt's per-frame marker)
    // decode state: B = max_batch slots. A frame step runs on `rows` = the smallest bucket (1, 2, 4, ... B) that holds
    // the live slots: rows [0, n_live) carry the live slots, the rest carry distinct idle slots (row -> slot map on the
    // device), so a draining batch stops paying for rows it no longer has.
    int B = 0;
    std::vector<int> buckets; int cur_bucket = -1;
    std::vector<int> row_of_slot, slot_of_row;
    Q3Slot* slots = nullptr;              // device [B]
    Q3Slot* slots_host = nullptr;         // pinned mirror [B] + staging [B]
    int* codes = nullptr;                 // [B][max_steps_cap][ncb]
    float* rng = nullptr;                 // [B][max_steps_cap]
    std::vector<Q3Lane> lanes;
    Q3Scratch sc_pre;
    // prefill
    float* xp = nullptr;                  // [n_ctx][d]
    uint16_t* xbp = nullptr; float* sspp = nullptr;  // norm inputs of the prefill rows
    uint16_t* ascp = nullptr; int rt16p = 0;         // W8A8: block scales of xbp
    int *pf_pos = nullptr, *pf_slot = nullptr;
    int* pf_seg = nullptr;              // the prefill launch's rows as per-slot runs {first row, n, slot} (device, 3 ints each): admit_group hands it to run_layers
    Q3PromptRow* prow_dev = nullptr; int prow_cap = 0;
    float* spk_dev = nullptr; int* refcodes_dev = nullptr;
    // sampler defaults (SamplerConfig::default: src/tts/engine.rs:25-34)
    float temperature = 0.7f; int top_k = 40; float top_p = 0.9f; int has_seed = 0; uint64_t seed = 0;
    int max_steps = 512;
    // timing
    hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev3 = nullptr;
    std::vector<hipEvent_t> fin_ev;     // per slot: PCM of a finished utterance copied to the host (vocoder stream)
    q3tts_timings tm{};
    Q3Voc* voc = nullptr;
    Q3Mel* mel = nullptr;               // created on first use
    Q3Clone* clone = nullptr;           // q3tts_clone_init
    // q3tts_k_probe: eager frame steps, events around the Predictor gate/up GEMM (pass 1, layer 0) of every frame
    int probe = 0, probe_kind = 0;      // probe: 0 off, 1 Predictor (pass 1, block 0), 2 Talker (block 0); kind: 0 gate/up, 1 QKV, 2 attention, 3 O, 4 down
    std::vector<hipEvent_t> probe_ev;   // 2 per frame of a chunk
    int probe_i = 0;
    double probe_ms = 0, probe_empty_ms = 0; long long probe_cnt = 0, probe_empty_cnt = 0, row_steps = 0;
    // q3tts_set_device_pcm: packed device copy of the last batch's PCM, one row of dev_pcm_stride samples per request
    int dev_pcm_on = 0; float* dev_pcm = nullptr; int dev_pcm_n = 0; size_t dev_pcm_stride = 0;
    double hp_launch = 0, hp_sync = 0;  // Q3TTS_HOST_PROF: host wall of run_chunk's launch part / of its wait (per engine: the node drives several from threads)
    float* first_chunk_host = nullptr;  // pinned landing buffer of the first 4-frame PCM chunk (first-chunk latency)
};

// helpers shared with q3_vocoder.hip
int q3_set_err(q3tts_engine* e, int code, const std::string& msg);
#define Q3_HIP(e, call)                                                                                         \
    do {                                                                                                        \
        hipError_t err__ = (call);                                                                              \
        if (err__ != hipSuccess)                                                                                \
            return q3_set_err((e), Q3TTS_ERR_DEVICE, std::string(#call) + ": " + hipGetErrorString(err__));     \
    } while (0)

// zero-filled device memory whose fill has completed on return (the one allocator of engine, vocoder, mel and clone state)
int q3_dev_alloc_zeroed(q3tts_engine* e, void** p, size_t bytes);

// host ChaCha12 StdRng (q3_rng.cpp)
void q3_stdrng_f32(uint64_t seed, int n, float* out);

void q3_mel_destroy(q3tts_engine* e);
int q3_mel_run(q3tts_engine* e, const float* audio, int64_t n_samples, int32_t* n_frames, float** out_dev);
void q3_clone_destroy(q3tts_engine* e);
// vocoder interface (q3_vocoder.hip)
int q3_voc_create(q3tts_engine* e);
void q3_voc_destroy(q3tts_engine* e);
// reset the streaming state of a slot
int q3_voc_reset(q3tts_engine* e, int slot);
// decode frames [f0, f0+nf) of slot (codes already on device in e->codes) into the slot's PCM buffer on stream
int q3_voc_decode(q3tts_engine* e, int slot, int f0, int nf, int is_last, hipStream_t s);
// batched: nf (<= 4) frames for every listed slot in one set of launches; real[i] <= nf of them are real for slot i (the
// rest are throw-away padding behind a finished utterance's last frame)
// beside_decoder: frame steps will run while this call executes (its long-lived workgroups are then launched one per CU: q3_vocoder.hip, "polite")
int q3_voc_decode_batch(q3tts_engine* e, const int* slots, const int* real, int ns, int nf, hipStream_t s, int beside_decoder);
void q3_voc_mark_last(q3tts_engine* e, int slot);
// PCM buffer of a slot (device) and samples produced so far
float* q3_voc_pcm(q3tts_engine* e, int slot);
int q3_voc_samples(q3tts_engine* e, int slot);
int q3_voc_samples_per_frame(const q3tts_engine* e);
