/*
 * q3tts.h — C ABI of the MI355X-native Qwen3-TTS inference path (libq3tts.so).
 *
 * This is the drop-in boundary for the hot path of cgisky1980/Qwen3-TTS-Rust
 * (TtsEngine::generate_with_voice -> run_inference_stream). It REPLACES the two
 * foreign runtimes the reference crate binds by hand:
 *   - the 29 dlsym'd llama.cpp symbols          (reference: src/models/llama/mod.rs:81-144,240-294)
 *   - the onnxruntime vocoder session           (reference: src/models/onnx.rs:324-459)
 * and keeps the host-visible semantics of
 *   - TtsEngine::{new,set_max_steps,set_sampler_config,generate_with_voice}
 *                                               (reference: src/tts/engine.rs:84,172,177,390)
 *   - SamplerConfig{temperature,top_k,top_p,seed}  (reference: src/tts/engine.rs:14-45)
 *   - PromptBuilder::{build_core,build_clone_prompt} (reference: src/tts/prompt.rs:141,28)
 *
 * Conventions: plain pointers and sizes, no C++ / torch types. Every function
 * returns an int status (0 = ok, <0 = error class) and leaves a message readable
 * through q3tts_last_error(). The library never aborts and never silently
 * truncates (the reference swallows vocoder-thread errors:
 * src/tts/engine.rs:496-502,520). Result buffers are allocated by the callee and
 * released with q3tts_result_free(); inputs are borrowed for the call only.
 * An engine handle is NOT re-entrant (same contract as `&mut self`,
 * src/tts/engine.rs:390); use one engine per GPU / host thread.
 */
#ifndef Q3TTS_H
#define Q3TTS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define Q3TTS_OK 0
#define Q3TTS_ERR_INVALID (-1)     /* bad argument / shape / config */
#define Q3TTS_ERR_DEVICE (-2)      /* HIP runtime error */
#define Q3TTS_ERR_OOM (-3)
#define Q3TTS_ERR_IO (-4)          /* weight / asset file problem */
#define Q3TTS_ERR_STATE (-5)       /* call order / handle state */
#define Q3TTS_ERR_UNSUPPORTED (-6)

#define Q3TTS_MAX_UPSAMPLE 4
#define Q3TTS_MAX_DEC_BLOCKS 8

/* ---- model configuration (every dimension is data, never a literal) -------------------------- */

/* Autoregressive codec-token decoder: Talker + Predictor (reference: llama.cpp GGUF models loaded at
 * src/tts/engine.rs:123-137; dims read at src/models/llama/mod.rs:348-353). */
typedef struct q3tts_model_config {
    /* Talker (Qwen3 decoder: RMSNorm, QK-RMSNorm, M-RoPE, GQA, SwiGLU) */
    int32_t t_n_layer, t_d_model, t_n_head, t_n_kv_head, t_head_dim, t_d_ffn, t_vocab;
    float t_rope_theta;
    int32_t t_mrope_sections[4]; /* rotary pairs per (t,h,w,extra) section; sum == head_dim/2 */
    /* Predictor (same block, plain RoPE, 15 heads of codebook_size logits) */
    int32_t p_n_layer, p_d_model, p_n_head, p_n_kv_head, p_head_dim, p_d_ffn;
    float p_rope_theta;
    int32_t n_codebooks;   /* 16  (reference: src/tts/engine.rs:587) */
    int32_t codebook_size; /* 2048 (reference: src/tts/engine.rs:588-589) */
    float rms_eps;
    /* embedding tables (reference: src/assets_manager.rs:212-249) */
    int32_t d_embed;     /* 2048 */
    int32_t text_vocab;  /* rows of text_embd */
    int32_t codec0_rows; /* rows of codec_embd.0 (holds specials + speaker ids) */
    int32_t codecq_rows; /* rows of codec_embd.1..15 */
    /* protocol (reference: src/tts/prompt.rs:5-16, src/tts/engine.rs:555-558) */
    int32_t sample_limit; /* 2160: Talker samples over logits[0, sample_limit) */
    int32_t eos_code;     /* 2150 */
    int32_t tts_pad_id;   /* 151671: text row used as marker and tts_pad */
} q3tts_model_config;

/* Streaming neural-codec vocoder (reference: ONNX graph behind src/models/onnx.rs:342-459; state shapes
 * src/models/onnx.rs:474-495 pin latent 1024 / pre-conv 512 / 8 layers x 16 heads x 64). */
typedef struct q3tts_vocoder_config {
    int32_t n_codebooks, codebook_size, codebook_dim; /* 16, 2048, 512 */
    int32_t latent_dim;                               /* 1024 */
    int32_t pre_conv_kernel;                          /* 3 */
    int32_t n_layer, n_head, head_dim, d_ffn, sliding_window;
    float rope_theta, rms_eps, layer_scale_init;
    int32_t n_upsample;
    int32_t upsample_ratios[Q3TTS_MAX_UPSAMPLE]; /* ConvTranspose1d(k=r,s=r) + ConvNeXt each */
    int32_t decoder_dim;                         /* 1536 */
    int32_t n_dec_blocks;
    int32_t dec_rates[Q3TTS_MAX_DEC_BLOCKS]; /* 8,5,4,3 -> 1920 samples per frame with 2x2 above */
    int32_t lookahead_frames;                /* frames withheld until more input or is_last (V4) */
    int32_t sample_rate;                     /* 24000 */
} q3tts_vocoder_config;

typedef struct q3tts_engine_config {
    q3tts_model_config model;
    q3tts_vocoder_config vocoder;
    int32_t device;        /* HIP device ordinal */
    int32_t max_batch;     /* concurrent utterance slots, 1..64 */
    int32_t n_ctx;         /* Talker context per slot (reference 4096: src/tts/engine.rs:133) */
    int32_t max_steps_cap; /* upper bound accepted by q3tts_set_max_steps (reference default 512) */
    int32_t with_vocoder;  /* 0: codes only (no vocoder weights allocated) */
    uint64_t synth_seed;   /* seeded synthetic weights when weights_path == NULL */
    const char* weights_path; /* NULL -> synthetic. Else the reference's quant directory (src/tts/engine.rs:91-131):
                               * qwen3_tts_talker.gguf + qwen3_tts_predictor.gguf (llama.cpp qwen3 tensor names; F32, F16,
                               * BF16, Q8_0 or the K-quants Q4_K / Q5_K / Q6_K of the gguf_q5_k_m directory, converted to
                               * bf16 at load) and qwen3_assets.gguf or its NPY fallback
                               * (src/assets_manager.rs:14-26). Shapes must match `model`; the table row counts
                               * (text_vocab, codec0_rows, codecq_rows) are taken from the files. The vocoder stays
                               * synthetic: the reference ships it as ONNX only. */
    int32_t talker_q8_0;      /* 1: the Talker's matrices and lm_head stay ggml Q8_0 blocks ON THE DEVICE (f16 scale + 32 int8 per block,
                               * 1.06 bytes per weight instead of 2) and are multiplied in that form (W8A16: bf16 activations, every block's
                               * MFMA product scaled by f32(d)) — the reference's default quantisation (gguf_q8_0,
                               * src/tts/engine.rs:91-95). Q8_0 tensors of weights_path are kept as stored; other tensor types and the
                               * synthetic weights are quantised with ggml's reference rule. 0 (default): bf16 weights. The Predictor
                               * keeps bf16 weights either way (157 MB, Infinity-Cache resident: its launches are latency-bound). */
    int32_t vocoder_flush_tail; /* Only matters with vocoder.lookahead_frames > 0 (V4). 0 (default): as the reference — its vocoder thread sends
                               * is_last only with a non-empty final buffer (src/tts/engine.rs:510-536), so an utterance of n_frames % 4 == 0
                               * never flushes the withheld look-ahead tail and its audio ends lookahead_frames short (restated by the oracle's
                               * q3o_chunk_plan). 1: always flush at the end of an utterance (every generated frame becomes audio). */
} q3tts_engine_config;

typedef struct q3tts_engine q3tts_engine;
typedef struct q3tts_stream q3tts_stream;

/* Fill cfg with the Qwen3-TTS-12Hz-1.7B shape assumed by SURVEY.md §8 (28x2048 Talker, 5x1024 Predictor). */
void q3tts_default_config(q3tts_engine_config* cfg);

/* TtsEngine::new (reference: src/tts/engine.rs:84-169): allocates weights, KV slabs, vocoder state on the
 * device and builds the replayable frame-step graphs. */
int q3tts_engine_create(const q3tts_engine_config* cfg, q3tts_engine** out);
void q3tts_engine_destroy(q3tts_engine* e);

/* Message for the last failing call on this engine (or on this thread when e == NULL). */
const char* q3tts_last_error(const q3tts_engine* e);

/* set_sampler_config / set_max_steps (reference: src/tts/engine.rs:172-184). has_seed == 0 reproduces
 * `seed: None` (wall-clock nanoseconds, src/tts/engine.rs:473-478). */
int q3tts_set_sampler(q3tts_engine* e, float temperature, int32_t top_k, float top_p, int32_t has_seed, uint64_t seed);
int q3tts_set_max_steps(q3tts_engine* e, int32_t max_steps);

/* ---- prompt (H1: src/tts/prompt.rs:141-277 build_core, :28-118 build_clone_prompt) ------------- */
typedef struct q3tts_prompt_desc {
    const uint32_t* text_ids;     int32_t n_text;     /* tokenizer.encode(text) */
    const uint32_t* instruct_ids; int32_t n_instruct; /* NULL -> instruct == None */
    int32_t lang_id;                                  /* <0 -> None (NOTHINK control block) */
    int32_t spk_id;                                   /* <0 -> None */
    const float* spk_emb;                             /* [d_embed] or NULL */
    const int32_t* ref_codes;     int32_t n_ref_frames; /* clone path: [n_ref_frames*16] or NULL */
    const uint32_t* ref_text_ids; int32_t n_ref_text;
} q3tts_prompt_desc;

/* Builds the prompt embeddings on the device and copies them back: *out_embd = malloc'd [n_tok][d_embed] f32
 * (free with q3tts_free). */
int q3tts_build_prompt(q3tts_engine* e, const q3tts_prompt_desc* p, float** out_embd, int32_t* out_n_tok);
void q3tts_free(void* p);
/* Device-resident results for the multi-GPU gather (the one collective of the path reads device memory). With
 * enable = 1 every q3tts_generate_batch call also keeps request i's PCM in row i of an engine-owned device buffer [n][stride] f32,
 * valid until the next call on this engine; requests with want_pcm = 2 skip the host copy. q3tts_get_device_pcm returns the buffer
 * of the last call (base may be NULL before the first one). The reference has no counterpart: it is one utterance at a time on one
 * device (src/models/llama/mod.rs:413, n_seq_max = 1). */
int q3tts_set_device_pcm(q3tts_engine* e, int32_t enable);
int q3tts_get_device_pcm(q3tts_engine* e, float** base, int64_t* stride_samples, int32_t* n_rows);

/* ---- generation (run_inference_stream: src/tts/engine.rs:445-656) ------------------------------ */
typedef struct q3tts_request {
    const float* prompt_embd; int32_t n_tok; /* [n_tok][d_embed] f32 host rows (PromptData.embd), or NULL ... */
    const q3tts_prompt_desc* prompt;         /* ... to build from ids on the device */
    int32_t use_engine_sampler;              /* 1: ignore the five fields below, use q3tts_set_sampler state */
    float temperature; int32_t top_k; float top_p; int32_t has_seed; uint64_t seed;
    int32_t max_steps;    /* 0 -> engine value */
    int32_t min_frames;   /* bench control: EOS logit masked while n_frames < min_frames (0 = reference) */
    int32_t force_eos_at; /* bench control: EOS forced at this step (<0 = off) */
    int32_t want_pcm;     /* 0: codec ids only; 1: PCM in host memory; 2: PCM kept on the device only (q3tts_set_device_pcm) */
} q3tts_request;

typedef struct q3tts_result {
    int32_t status;
    int32_t n_frames;      /* frames kept (EOS frame excluded, as src/tts/engine.rs:558-562) */
    int32_t hit_eos;
    int32_t* codes;        /* [n_frames][n_codebooks] raw ids (unclamped) */
    float* pcm;            /* [n_samples] mono f32, or NULL; pinned host memory: release with q3tts_result_free only */
    int32_t n_samples;
    int32_t sample_rate;
    float first_chunk_ms;  /* entry -> first 4-frame PCM chunk resident on host (0 if none) */
    float total_ms;
} q3tts_result;

int q3tts_generate(q3tts_engine* e, const q3tts_request* req, q3tts_result* out);
/* Continuous batching over max_batch slots; results are independent of n, of slot assignment and of the
 * number of GPUs the caller shards over (per-utterance RNG stream is a function of req->seed only). */
int q3tts_generate_batch(q3tts_engine* e, const q3tts_request* reqs, int32_t n, q3tts_result* outs);
void q3tts_result_free(q3tts_result* r);

/* Streaming: 4-frame (64-code) chunks as in the reference's vocoder thread (src/tts/engine.rs:507-541). */
int q3tts_stream_begin(q3tts_engine* e, const q3tts_request* req, q3tts_stream** out);
/* Blocks until the next chunk; *chunk is owned by the stream and valid until the next poll/end. */
int q3tts_stream_poll(q3tts_stream* s, const float** chunk, int32_t* n_samples, int32_t* is_final);
int q3tts_stream_end(q3tts_stream* s, q3tts_result* out_codes_optional);

/* ---- one node, several GPUs ----------------------------------------------------------------------
 * The reference is one utterance at a time on one device (n_seq_max = 1, src/models/llama/mod.rs:413; `&mut self`,
 * src/tts/engine.rs:390). Utterances share nothing but read-only weights, so a batch shards over independent units:
 * q3tts_node_create builds one engine (full weight replica) and one host thread per listed device; request i of a
 * q3tts_node_generate_batch call runs on device i mod n_devices (order preserved, results independent of n_devices: the
 * sampler stream is a function of req->seed only). There is no data-path collective. With pcm_i16 != NULL the finished
 * PCM stays on the devices, is converted there to 16-bit samples exactly as the reference saves audio
 * (src/utils/audio.rs:35-37: (x * 32767).clamp(-32768, 32767) as i16) and gathered to the FIRST listed device with RCCL
 * over xGMI — one ncclAllGather of the per-utterance sample counts, one group of ncclSend / ncclRecv — then copied to the
 * host once: pcm_i16[i] = malloc'd [outs[i].n_samples] samples of request i (release with q3tts_free; outs[i].pcm stays
 * NULL). With pcm_i16 == NULL requests behave as in q3tts_generate_batch (want_pcm = 1: f32 PCM in host memory) and RCCL
 * is never loaded. RCCL is resolved at run time (librccl.so.1); a node handle is not re-entrant.
 * Errors: a request that fails by itself (e.g. prompt + max_steps > n_ctx) carries its own outs[i].status and the call still returns
 * Q3TTS_OK with every other result intact. When a device or the gather fails the call returns the error, every outs[i] — including
 * the results of the devices that succeeded — is still valid to pass to q3tts_result_free, and no pcm_i16[i] is left allocated. */
typedef struct q3tts_node q3tts_node;
typedef struct q3tts_node_timings {
    float generate_ms;      /* slowest device: its q3tts_generate_batch wall time */
    float gather_ms;        /* pack + collectives + copy to the host (0 without pcm_i16) */
    float total_ms;
    int64_t gathered_bytes; /* i16 PCM bytes that reached the host through device 0 */
    int32_t n_devices;
} q3tts_node_timings;
int q3tts_node_create(const q3tts_engine_config* cfg /* .device is ignored */, const int32_t* devices, int32_t n_devices, q3tts_node** out);
void q3tts_node_destroy(q3tts_node* n);
int q3tts_node_generate_batch(q3tts_node* n, const q3tts_request* reqs, int32_t n_reqs, q3tts_result* outs, int16_t** pcm_i16 /* [n_reqs] or NULL */);
int q3tts_node_get_timings(const q3tts_node* n, q3tts_node_timings* out);
const char* q3tts_node_last_error(const q3tts_node* n);
int32_t q3tts_node_size(const q3tts_node* n);
q3tts_engine* q3tts_node_engine(q3tts_node* n, int32_t rank);  /* the engine of device `rank` (sampler defaults, timings); owned by the node */
/* Host only: the global request indices device `rank` of `world` owns, in order ({i : i mod world == rank}); returns the count
 * (idx may be NULL to query it), -1 on bad arguments. */
int32_t q3tts_node_shard(int32_t n_total, int32_t world, int32_t rank, int32_t* idx, int32_t cap);

/* Per-stage device timings of the last generate call (hipEvent), ms. */
typedef struct q3tts_timings {
    float prefill_ms, decode_ms, vocoder_ms, total_ms;
    float frame_step_ms;      /* mean device time of one frame-step graph replay */
    float probe_kernel_ms;    /* q3tts_k_probe: mean in-situ device time of the probed kernel (mode 2: the Talker's layer-0 gate/up GEMM; mode 1: the Predictor's), full batch */
    int64_t frame_steps;      /* graph replays timed */
    int64_t algo_bytes_per_step; /* SURVEY.md §8(d) algorithmic bytes of one frame step at the batch run */
    int64_t algo_flops_per_step; /* 2 * (W_T + 15 W_P + 15 h + pj) * mean live utterances per step (decoder GEMMs) */
    float mean_live_slots;       /* utterances generating, averaged over the timed frame steps */
    float mean_rows;             /* decode rows per frame step (row bucket), averaged over the timed frame steps */
    int64_t probe_count;         /* launches behind probe_kernel_ms */
    float probe_empty_ms;        /* mean elapsed time of an EMPTY event bracket on the same stream (event overhead) */
    float mean_ctx_tokens;       /* sum over the live utterances of their Talker context length, averaged over the timed frame steps */
} q3tts_timings;
int q3tts_get_timings(const q3tts_engine* e, q3tts_timings* out);

/* Log-mel front-end of the voice-clone path (replaces SpeakerEncoder::compute_mel, src/models/onnx.rs:166-321): 24 kHz mono
 * f32 in, [n_frames][128] log-mel out (n_fft 1024, hop 256, Slaney mels, the reference's padding rules). */
int32_t q3tts_mel_frames(int64_t n_samples);
int q3tts_mel(q3tts_engine* e, const float* audio, int64_t n_samples, float* out, int32_t cap_frames, int32_t* n_frames);

/* ---- voice-clone encoders (replace AudioEncoder / SpeakerEncoder, src/models/onnx.rs:82-165, and the encoder half of
 * TtsEngine::create_voice_file, src/tts/engine.rs:324-387). The reference runs two ONNX graphs that are not in its
 * repository; the structure here is the model family's (ECAPA-TDNN speaker encoder; SEANet + transformer + split residual
 * VQ codec encoder), every dimension a field, weights seeded-synthetic. ------------------------------- */
typedef struct q3tts_clone_config {
    /* speaker encoder: log-mel [T][mel_dim] -> [se_dim] */
    int32_t mel_dim;                 /* 128 (fixed by q3tts_mel) */
    int32_t se_channels[5];          /* 512,512,512,512,1536: TDNN, 3 SE-Res2Net blocks, aggregation (= 3 x block width) */
    int32_t se_kernels[5];           /* 5,3,3,3,1 */
    int32_t se_dilations[5];         /* 1,2,3,4,1 */
    int32_t se_attn_channels;        /* 128 */
    int32_t se_res2net_scale;        /* 8 */
    int32_t se_se_channels;          /* 128 */
    int32_t se_dim;                  /* 2048 = model.d_embed ("spk_emb" [1,2048], src/models/onnx.rs:149) */
    /* audio encoder: 24 kHz PCM -> [frames][ae_n_codebooks] */
    int32_t ae_filters;              /* 64 */
    int32_t ae_kernel, ae_res_kernel, ae_last_kernel;   /* 7, 3, 3 */
    int32_t ae_n_ratios; int32_t ae_ratios[4];          /* 4: 4,5,6,8 (x960) */
    int32_t ae_hidden;               /* 512 */
    int32_t ae_n_layer, ae_n_head, ae_head_dim, ae_d_ffn, ae_window;  /* 8, 8, 64, 2048, 250 */
    float ae_rope_theta, ae_ln_eps, ae_layer_scale;     /* 10000, 1e-5, 0.01 */
    int32_t ae_down_stride;          /* 2 (x1920 = one 12.5 Hz frame) */
    int32_t ae_vq_dim;               /* 256 */
    int32_t ae_n_codebooks, ae_codebook_size;           /* 16, 2048 ("audio_codes" [1,frames,16], src/models/onnx.rs:107) */
} q3tts_clone_config;
void q3tts_clone_default_config(q3tts_clone_config* cfg);
/* "load" both encoders into the engine (the reference does so when the two ONNX files exist, src/tts/engine.rs:105-119);
 * weights are generated from cfg.synth_seed of the engine. Calling it again replaces them. */
int q3tts_clone_init(q3tts_engine* e, const q3tts_clone_config* cfg);
/* frames the audio encoder produces for n_samples (ceil division through every stride) */
int32_t q3tts_clone_audio_frames(const q3tts_engine* e, int64_t n_samples);
/* AudioEncoder::encode (src/models/onnx.rs:96-121): codes [n_frames][ae_n_codebooks] i64, row-major like "audio_codes".
 * Q3TTS_ERR_STATE with "AudioEncoder not loaded" before q3tts_clone_init (src/tts/engine.rs:330-332). */
int q3tts_clone_audio_encode(q3tts_engine* e, const float* audio, int64_t n_samples, int64_t* codes, int32_t cap_frames,
                             int32_t* n_frames);
/* SpeakerEncoder::encode (src/models/onnx.rs:135-160): log-mel on the device, then the encoder; out [se_dim] */
int q3tts_clone_speaker_encode(q3tts_engine* e, const float* audio, int64_t n_samples, float* spk_emb);
/* test hooks: the speaker encoder on a given log-mel; the audio encoder's pre-quantiser rows [n_frames][ae_hidden] */
int q3tts_k_speaker_from_mel(q3tts_engine* e, const float* mel, int32_t n_frames, float* spk_emb);
int q3tts_k_audio_latent(q3tts_engine* e, const float* audio, int64_t n_samples, float* latent, int32_t cap_frames,
                         int32_t* n_frames);

/* ---- tokenizer (host only; replaces the `tokenizers`-crate wrapper src/utils/tokenizer.rs:1-37 for non-Rust hosts) ----
 * Reads model_dir/tokenizer/tokenizer.json of the Qwen2 family: added tokens, NFC, Split(Qwen2 regex) + ByteLevel, BPE.
 * Any other pipeline is refused at load (Q3TTS_ERR_UNSUPPORTED); input must be valid UTF-8.
 * No engine and no GPU needed; errors go to the caller's buffer. A handle is not thread-safe (it caches words). */
typedef struct q3tts_tokenizer q3tts_tokenizer;
int q3tts_tokenizer_load(const char* tokenizer_json_path, q3tts_tokenizer** out, char* err, int32_t err_cap);
void q3tts_tokenizer_free(q3tts_tokenizer* t);
int32_t q3tts_tokenizer_vocab_size(const q3tts_tokenizer* t);
/* Tokenizer::encode(text) = inner.encode(text, add_special_tokens = false).get_ids() (src/utils/tokenizer.rs:17-25).
 * *n_ids receives the count even when cap is too small (then Q3TTS_ERR_INVALID). */
int q3tts_tokenizer_encode(const q3tts_tokenizer* t, const char* utf8, int64_t n_bytes, uint32_t* ids, int32_t cap,
                           int32_t* n_ids, char* err, int32_t err_cap);
/* Tokenizer::decode(ids) = inner.decode(ids, skip_special_tokens = false) (:27-35), as raw bytes (the crate applies
 * from_utf8_lossy on top) */
int q3tts_tokenizer_decode(const q3tts_tokenizer* t, const uint32_t* ids, int32_t n, char* out, int64_t cap, int64_t* n_bytes,
                           char* err, int32_t err_cap);

/* ---- kernel-level test hooks (host buffers in/out; used only by tests/ and bench.py) ----------- */
/* y[B][N] = exact_gemm(norm?(x)[B][K], W[N][K] bf16 bits) (+bias) — canonical order of DESIGN.md §4.1 */
int q3tts_k_gemm_exact(int32_t device, const float* x, int32_t B, int32_t K, const uint16_t* w_bf16, int32_t N,
                       const float* norm_w /*NULL: no RMSNorm*/, float eps, const float* bias /*NULL*/,
                       int32_t epilogue /*0 store,1 residual(y+=),2 swiglu,3 argmax*/, float* y, uint64_t* argmax_keys,
                       int32_t iters, float* mean_kernel_ms);
/* fused q/k RMSNorm + RoPE + KV append + decode attention for rows of ONE sequence processed in order */
int q3tts_k_attention(int32_t device, const float* qkv /*[n][(Hq+2Hkv)*hd]*/, int32_t n_rows, int32_t pos0,
                      int32_t n_head, int32_t n_kv_head, int32_t head_dim, const float* q_norm_w, const float* k_norm_w,
                      float eps, float rope_theta, const int32_t* mrope_sections, float* out /*[n][Hq*hd]*/);
/* sampler (H4: src/models/llama/mod.rs:666-772) on n rows of logits; r_uniform[n] are the f32 draws */
int q3tts_k_sample(int32_t device, const float* logits, int32_t n, int32_t ld, int32_t limit, float temperature,
                   int32_t top_k, float top_p, const float* r_uniform, int32_t* out_ids);
/* Talker forward over a prompt: hidden[d] (post final norm) and logits[t_vocab] of the LAST row */
int q3tts_k_talker_prefill(q3tts_engine* e, const float* embd, int32_t n_tok, float* hidden_out, float* logits_out);
/* Vocoder: codes [n_frames][n_codebooks] -> pcm; chunk_frames frames per streaming call (0 = one call) */
int q3tts_k_vocoder(q3tts_engine* e, const int32_t* codes, int32_t n_frames, int32_t chunk_frames, float* pcm_out,
                    int32_t* n_samples_out);
/* Measurement (bench.py roofline_vocoder): the batched vocoder alone, n_slots slots x `chunks` 4-frame calls with nothing else on the
 * GPU; *ms_per_chunk = mean duration of one batched call (n_slots x 4 frames of PCM) by HIP events on its stream. */
int q3tts_k_vocoder_bench(q3tts_engine* e, int32_t n_slots, int32_t chunks, float* ms_per_chunk);
/* Host-only: one tensor of a GGUF file (or the array of an .npy file; `tensor` is then ignored) as f32, through the same
 * reader the engine uses for weights_path. out may be NULL to query nelem / dims (ggml order: dims4[0] is the row length)
 * / ggml type (0 F32, 1 F16, 8 Q8_0, 30 BF16). Needs no GPU. */
int q3tts_k_gguf_read(const char* path, const char* tensor, float* out, int64_t cap, int64_t* nelem, int64_t* dims4, int32_t* ggml_type);
/* Measurement mode for bench.py: frame steps are launched eagerly (no graph replay) and ONE launch of every frame is bracketed
 * by HIP events on its own stream. enable = model + 16 * kind; model 2: block 0 of the Talker step, model 1: block 0 of the
 * Predictor's pass 1; kind 0: the gate/up GEMM (so enable = 2 / 1 are the Talker's / Predictor's gate/up as before), 1: QKV GEMM,
 * 2: attention kernel, 3: O projection, 4: down projection. q3tts_timings.probe_kernel_ms / probe_count report it for the frame
 * steps that ran at the full row count. enable = 0 restores graph replay. */
int q3tts_k_probe(q3tts_engine* e, int32_t enable);
/* n_cases independent chains of `chain` v_mfma_f32_16x16x32_bf16 into one accumulator tile: a [n][chain][16][32] bf16 bits,
 * b [n][chain][32][16], c / d [n][16][16] f32. Pins the instruction's accumulation arithmetic. */
int q3tts_k_mfma_bf16(int32_t device, const uint16_t* a, const uint16_t* b, const float* c, float* d, int32_t n_cases, int32_t chain);
/* The decoder's GEMM as the engine launches it (csrc/q3_bgemm.hip): bf16 rows xb [B][K] x W [N][K] (bf16 bit
 * patterns, row-major; tiled on the device) on v_mfma_f32_16x16x32_bf16, K % 256 == 0, N % 16 == 0. ssp [B][ntiles] (or NULL) are
 * the producer's per-tile sums of squares: s_r = 1 / sqrtf(SS(ssp) / d_norm + eps).
 *   epilogue 0: y[B][N] = s_r * RAW            1: y += RAW (y in/out); with nw_next[N] also yb[B][N] = bf16(y * nw_next), ssp_out[B][N/16]
 *            2: yb[B][N/2] = bf16(swiglu(s_r * RAW_gate, s_r * RAW_up)), W = the N/2 gate rows then the N/2 up rows
 *            3: keys[B] = argmax key over s_r * RAW
 * Equals oracle q3o_bgemm bit for bit for every row count (the tile shape the launcher picks never changes a result). */
int q3tts_k_bgemm(int32_t device, const uint16_t* xb, int32_t B, int32_t K, const uint16_t* w_bf16, int32_t N, const float* ssp, int32_t ntiles,
                  int32_t d_norm, float eps, int32_t epilogue, const float* nw_next, float* y, uint16_t* yb, float* ssp_out, uint64_t* keys,
                  int32_t iters, float* mean_kernel_ms);
/* Which kernel serves launches of >= 256 rows: 1 = the many-row kernel (k_bgemm_big) whenever eligible, -1 = never, 0 = when it fills the
 * chip (default; the environment variable Q3TTS_BG_BIG sets the initial value once per process). The results are the same bits. */
int q3tts_k_bgemm_policy(int32_t big);
/* Which kernel variant serves the Talker's decode attention (0 = k_attend_gqa2, default; 1 = k_attend<2, true>) and the prefill of whole
 * prompts (0 = k_attend_prefill when the launch has >= 128 (run, KV head) workgroups, default; 1 = never: k_attend<2, false>; 2 = whenever
 * eligible). Same bits either way (tests compare them in one process); Q3TTS_ATT_OLD / Q3TTS_ATT_PREFILL_OLD set the initial values. */
int q3tts_k_attend_policy(int32_t decode, int32_t prefill);
/* The k_bgemm instance the launcher takes for a shape, without launching: out5 = {row tiles, column tiles, ring depth, non-temporal weight
 * loads, 1 if k_bgemm_big}. bench.py names the kernel symbol of a probed launch from it. */
int q3tts_k_bgemm_pick(int32_t B, int32_t K, int32_t N, int32_t epilogue, int32_t w_once, int32_t q8, int32_t* out5);
/* The same launch with ggml Q8_0 weights kept in block form on the device: q int8 [N][K] row-major (epilogue 2: the
 * N/2 gate rows, then the N/2 up rows), d_f16 the blocks' f16 scales as bit patterns [N][K/32]; K % 512 == 0. Equals oracle q3o_bgemm_q8
 * bit for bit. */
int q3tts_k_bgemm_q8(int32_t device, const uint16_t* xb, int32_t B, int32_t K, const int8_t* q, const uint16_t* d_f16, int32_t N, const float* ssp,
                     int32_t ntiles, int32_t d_norm, float eps, int32_t epilogue, const float* nw_next, float* y, uint16_t* yb, float* ssp_out,
                     uint64_t* keys, int32_t iters, float* mean_kernel_ms);
/* The same launch in ggml's Q8_0 x Q8_0 arithmetic (W8A8: csrc/q3_bgemm8.hip; q3tts_engine_config.talker_q8_0 = 2): the
 * ACTIVATIONS are Q8_0 blocks too — aq int8 [B][K], ad their f16 scales as bit patterns [B][K/32] — a block's product is its exact int32
 * sum times f32(d_w) * f32(d_x). epilogue 0: y = s_r * RAW; 1: y += RAW, then the consumer's operand v = y * nw_next quantised per 32
 * columns by ggml's rule -> yq int8 [B][N], yd f16 [B][N/32], and ssp_out; 2: h = swiglu(s_r * RAW_gate, s_r * RAW_up) quantised ->
 * yq [B][N/2], yd [B][N/64]. K % 512 == 0; N % 32 == 0 (1: % 64, 2: % 128). Equals oracle q3o_bgemm_q8a8 bit for bit. */
int q3tts_k_bgemm_q8a8(int32_t device, const int8_t* aq, const uint16_t* ad, int32_t B, int32_t K, const int8_t* q, const uint16_t* d_f16, int32_t N,
                       const float* ssp, int32_t ntiles, int32_t d_norm, float eps, int32_t epilogue, const float* nw_next, float* y, int8_t* yq,
                       uint16_t* yd, float* ssp_out, int32_t iters, float* mean_kernel_ms);
/* The same GEMM with the epilogue extras only the vocoder uses (nothing in the reference: its vocoder is an ONNX graph, src/models/onnx.rs:342-459):
 * bias[col % bias_n] added to RAW first; epilogue 0: y = RAW + bias; 1: y += col_scale[col] * (RAW + bias), optionally yb = bf16(y);
 * 4: yb = bf16(gelu_erf(RAW + bias)). seg_rows > 0: the f32 rows live in B / seg_rows segments separated by gap_rows rows the kernel
 * must not touch (checked by the hook). y / yb are dense [B][N] on the host side. */
int q3tts_k_bgemm_voc(int32_t device, const uint16_t* xb, int32_t B, int32_t K, const uint16_t* w_bf16, int32_t N, int32_t epilogue, const float* bias,
                      int32_t bias_n, const float* col_scale, int32_t seg_rows, int32_t gap_rows, float* y, uint16_t* yb, int32_t want_yb);
/* H6 — Assets::project (src/assets_manager.rs:383-399) in the reference's own f32 sequence: y[r][o] = bias[o]; y += x[r][i] * w[o][i]
 * for i ascending (w f32 row-major [n_out][n_in]). nw != NULL: also the rows' norm inputs xb = bf16(y * nw), ssp [rows][n_out/16]. */
int q3tts_k_project(int32_t device, const float* x, int32_t rows, int32_t n_in, const float* w, const float* bias, int32_t n_out, const float* nw,
                    float* y, uint16_t* xb, float* ssp);
/* producer side of the split RMSNorm for plain f32 rows: xb = bf16(x * nw), ssp[r][t] = sum of squares of tile t */
int q3tts_k_norm_inputs(int32_t device, const float* x, int32_t rows, int32_t d, const float* nw, uint16_t* xb, float* ssp);
/* Allocator contract: device memory is handed out with its zero fill COMPLETED. Allocates `bytes` through the engine's allocator,
 * uploads a pattern into the first and last 4 KiB on the null stream at once, and reports the bytes that read back wrong (0 expected). */
int q3tts_k_alloc_upload(q3tts_engine* e, int64_t bytes, int64_t* mismatches);
/* rand 0.8 StdRng (ChaCha12) stream: seed_from_u64(seed) then n x gen::<f32>() */
int q3tts_k_rng_f32(uint64_t seed, int32_t n, float* out);

#ifdef __cplusplus
}
#endif
#endif /* Q3TTS_H */
