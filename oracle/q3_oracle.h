/*
 * q3_oracle.h — CPU restatement (plain C) of the Qwen3-TTS inference hot path of
 * cgisky1980/Qwen3-TTS-Rust. TEST INFRASTRUCTURE ONLY: nothing in the product path
 * (qwen3-tts-rust_amd/, include/) may include, link or call this; only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg load libq3oracle.so.
 *
 * PARITY UNPINNED: the reference has no tests, golden vectors or fixtures for this
 * path and its arithmetic lives in llama.cpp b7885 / onnxruntime
 * 1.23.2 binaries that are not in the repository and cannot be fetched. What IS
 * followed line by line is the host logic the crate owns (citations at each
 * function). The transformer math is the standard Qwen3 decoder definition (checked
 * against transformers' Qwen3 in tests/test_decoder_family_cpu.py) in ONE canonical
 * order (bf16 operands on the restated v_mfma_f32_16x16x32_bf16, fixed
 * slice / tile / butterfly orders) that the HIP kernels reproduce bit for bit.
 *
 * The structs below restate include/q3tts.h field for field (same layout) so the
 * tests can fill one ctypes structure for both libraries; the oracle does not
 * include the product header.
 */
#ifndef Q3_ORACLE_H
#define Q3_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct q3o_model_config {
    int32_t t_n_layer, t_d_model, t_n_head, t_n_kv_head, t_head_dim, t_d_ffn, t_vocab;
    float t_rope_theta;
    int32_t t_mrope_sections[4];
    int32_t p_n_layer, p_d_model, p_n_head, p_n_kv_head, p_head_dim, p_d_ffn;
    float p_rope_theta;
    int32_t n_codebooks;
    int32_t codebook_size;
    float rms_eps;
    int32_t d_embed;
    int32_t text_vocab;
    int32_t codec0_rows;
    int32_t codecq_rows;
    int32_t sample_limit;
    int32_t eos_code;
    int32_t tts_pad_id;
} q3o_model_config;

typedef struct q3o_vocoder_config {
    int32_t n_codebooks, codebook_size, codebook_dim;
    int32_t latent_dim;
    int32_t pre_conv_kernel;
    int32_t n_layer, n_head, head_dim, d_ffn, sliding_window;
    float rope_theta, rms_eps, layer_scale_init;
    int32_t n_upsample;
    int32_t upsample_ratios[4];
    int32_t decoder_dim;
    int32_t n_dec_blocks;
    int32_t dec_rates[8];
    int32_t lookahead_frames;
    int32_t sample_rate;
} q3o_vocoder_config;

typedef struct q3o_prompt_desc {
    const uint32_t* text_ids;     int32_t n_text;
    const uint32_t* instruct_ids; int32_t n_instruct;
    int32_t lang_id;
    int32_t spk_id;
    const float* spk_emb;
    const int32_t* ref_codes;     int32_t n_ref_frames;
    const uint32_t* ref_text_ids; int32_t n_ref_text;
} q3o_prompt_desc;

typedef struct q3o_model q3o_model;

/* seeded synthetic weights (identical generator to the device one) */
q3o_model* q3o_create(const q3o_model_config* cfg, uint64_t seed, int32_t n_ctx, int32_t n_threads);
void q3o_destroy(q3o_model* m);

/* primitives ------------------------------------------------------------------------------ */
float q3o_expf(float x);
float q3o_synth(uint64_t seed, uint32_t tensor, uint64_t idx, float scale);
uint16_t q3o_bf16(float x);
void q3o_synth_fill(uint64_t seed, uint32_t tensor, uint64_t n, float base, float std, int32_t round_to_bf16, float* out);
/* y = gemm_exact(norm?(x), W[N][K] bf16) ; epilogue 0 store(+bias) 1 residual 2 swiglu 3 argmax keys */
void q3o_gemm_exact(const float* x, int32_t B, int32_t K, const uint16_t* w_bf16, int32_t N, const float* norm_w,
                    float eps, const float* bias, int32_t epilogue, float* y, uint64_t* argmax_keys);
void q3o_rmsnorm(const float* x, int32_t d, const float* w, float eps, float* y);
/* rows of one sequence processed in order, fresh cache */
void q3o_attention(const float* qkv, int32_t n_rows, int32_t pos0, int32_t n_head, int32_t n_kv_head,
                   int32_t head_dim, const float* q_norm_w, const float* k_norm_w, float eps, float rope_theta,
                   const int32_t* mrope_sections, float* out);
/* one output element of v_mfma_f32_16x16x32_bf16 (gfx950), restated in integers: a[32], b[32] bf16 bits in operand order
 * (position 8g + e = operand e of lane group g), c the accumulator. _ref: the 128-bit form the 64-bit one is tested against. */
float q3o_mfma_bf16_dot32(const uint16_t* a, const uint16_t* b, float c);
float q3o_mfma_bf16_dot32_ref(const uint16_t* a, const uint16_t* b, float c);
/* the decoder's canonical arithmetic (q3_oracle_bf16.c) */
void q3o_permute_rows_bf16(const uint16_t* src, int32_t rows, int32_t K, uint16_t* dst);
void q3o_bgemm_raw_p(const uint16_t* xp, int32_t rows, int32_t K, const uint16_t* wp, int32_t N, float* out, int32_t ldo, int32_t threads);
float q3o_tss16(const float* v16);
void q3o_norm_inputs(const float* x, int32_t d, const float* nw, uint16_t* xb, float* ssp);
float q3o_row_scale(const float* ssp, int32_t ntiles, int32_t d, float eps);
/* kernel-level restatement of q3_bgemm.hip (natural-order operands): epi 0 y = s*raw (s = 1 when ssp == NULL), 1 x += raw
 * (+ norm outputs when nw_next), 2 hb = bf16(swiglu) (w = F gate rows then F up rows), 3 argmax keys */
void q3o_bgemm(const uint16_t* xb, int32_t B, int32_t K, const uint16_t* w, int32_t N, const float* ssp, int32_t ntiles, int32_t d_norm,
               float eps, int32_t epi, const float* nw_next, float* y, uint16_t* yb, float* ssp_out, uint64_t* keys);
/* H6 in the reference's own arithmetic (src/assets_manager.rs:383-399): bias first, sum += h * w in ascending order */
void q3o_project_rows(const float* w, const float* bias, int32_t n_in, int32_t n_out, const float* x, int32_t rows, float* y);
/* 0: canonical bf16-MFMA order (default, what the device computes); 1: plain f32 of the same structure (family pinning) */
void q3o_set_arith(q3o_model* m, int32_t arith);
void q3o_set_talker_q8a8(q3o_model* m);  /* ... and the activations as Q8_0 blocks too: ggml's W8A8 (q3tts_engine_config.talker_q8_0 = 2; q3_oracle_bf16.c) */
void q3o_quantize_rows_q8(const float* v, int32_t rows, int32_t K, int8_t* q, uint16_t* d_f16);
void q3o_bgemm_q8a8(const int8_t* aq, const uint16_t* ad, int32_t B, int32_t K, const int8_t* q, const uint16_t* d_f16, int32_t N, const float* ssp,
                    int32_t ntiles, int32_t d_norm, float eps, int32_t epi, const float* nw_next, float* y, int8_t* yq, uint16_t* yd, float* ssp_out);
void q3o_set_talker_q8(q3o_model* m);  /* the Talker's matrices + lm_head as ggml Q8_0 blocks, canonical Q8 order (q3tts_engine_config.talker_q8_0) */
void q3o_quantize_q8_0(const float* x, int64_t n, int8_t* q, uint16_t* d_f16);  /* ggml's reference quantiser */
float q3o_f16_to_f32(uint16_t h);
void q3o_bgemm_q8(const uint16_t* xb, int32_t B, int32_t K, const int8_t* q, const uint16_t* d_f16, int32_t N, const float* ssp, int32_t ntiles,
                  int32_t d_norm, float eps, int32_t epi, const float* nw_next, float* y, uint16_t* yb, float* ssp_out, uint64_t* keys);
void q3o_set_threads(int32_t n);
const float* q3o_norm_weight(const q3o_model* m, int32_t talker, int32_t layer, int32_t which);
int32_t q3o_matrix(const q3o_model* m, int32_t talker, int32_t layer, int32_t which, float* out);
/* H4 sampler: src/models/llama/mod.rs:666-772 */
int32_t q3o_sample(const float* logits, int32_t limit, float temperature, int32_t top_k, float top_p, float r);
/* rand 0.8 StdRng: seed_from_u64 + gen::<f32>() */
void q3o_rng_f32(uint64_t seed, int32_t n, float* out);
void q3o_chacha_block(const uint32_t in[16], int32_t rounds, uint32_t out[16]);
/* H2 positions: src/tts/engine.rs:306-318 */
void q3o_qwen3_position(int32_t start, int32_t len, int32_t* out4n);

/* H1 prompt: src/tts/prompt.rs:141-277 / :28-118 ; returns n_tok (out may be NULL to size) */
int32_t q3o_build_prompt(const q3o_model* m, const q3o_prompt_desc* p, float* out, int32_t max_tok);
/* table access with the reference's OOB rules (src/assets_manager.rs:419-460) */
void q3o_text_embedding(const q3o_model* m, int64_t id, float* out);
void q3o_codec_embedding(const q3o_model* m, int32_t q, int32_t code, float* out);
/* H6: src/assets_manager.rs:383-399, the reference's own f32 sequence (bias first, sequential sum += h * w) */
void q3o_project(const q3o_model* m, const float* x2048, float* y1024);

/* Talker prefill: hidden (post final norm) and logits of the last prompt row */
void q3o_talker_prefill(q3o_model* m, const float* embd, int32_t n_tok, float* hidden_out, float* logits_out);

/* run_inference_stream (src/tts/engine.rs:445-656) up to the codec ids. Returns n_frames. */
int32_t q3o_generate(q3o_model* m, const float* prompt_embd, int32_t n_tok, float temperature, int32_t top_k,
                     float top_p, uint64_t seed, int32_t max_steps, int32_t min_frames, int32_t force_eos_at,
                     int32_t* codes_out, int32_t* hit_eos);

/* H8 chunker (src/tts/engine.rs:507-541): given n_frames, writes the sequence of vocoder calls as
 * (n_frames_in_call, is_final) pairs; returns the number of calls. */
int32_t q3o_chunk_plan(int32_t n_frames, int32_t* calls_frames, int32_t* calls_final, int32_t max_calls);

/* vocoder (q3_oracle_vocoder.c) ------------------------------------------------------------- */
/* log-mel front-end of the clone path (q3_oracle_mel.c; src/models/onnx.rs:166-321) */
void q3o_mel_tables(float* hann1024, float* cos1024, float* sin1024, float* fb128x513);
int32_t q3o_mel_frames(int64_t n_samples);
int32_t q3o_mel(const float* audio, int64_t n_samples, float* out, float* pre_log);

/* voice-clone encoders (q3_oracle_clone.c; I/O contract src/models/onnx.rs:82-165, caller src/tts/engine.rs:324-387) */
typedef struct q3o_clone_config {
    /* speaker encoder: ECAPA-TDNN family, log-mel [T][mel_dim] -> [se_dim] */
    int32_t mel_dim;
    int32_t se_channels[5], se_kernels[5], se_dilations[5];
    int32_t se_attn_channels, se_res2net_scale, se_se_channels, se_dim;
    /* audio encoder: SEANet conv stack + transformer + stride-2 conv + split residual VQ, 24 kHz PCM -> [frames][ncb] */
    int32_t ae_filters, ae_kernel, ae_res_kernel, ae_last_kernel;
    int32_t ae_n_ratios, ae_ratios[4];
    int32_t ae_hidden, ae_n_layer, ae_n_head, ae_head_dim, ae_d_ffn, ae_window;
    float ae_rope_theta, ae_ln_eps, ae_layer_scale;
    int32_t ae_down_stride, ae_vq_dim, ae_n_codebooks, ae_codebook_size;
} q3o_clone_config;
int32_t q3o_audio_frames(const q3o_clone_config* c, int64_t n_samples);
/* returns 0 / <0; out[se_dim] */
int32_t q3o_speaker_encode(const q3o_clone_config* c, uint64_t seed, const float* mel, int32_t n_frames, float* out);
/* returns n_frames; codes [n_frames][ae_n_codebooks]; latent_out (optional) [n_frames][ae_hidden] = the rows fed to the VQ */
int32_t q3o_audio_encode(const q3o_clone_config* c, uint64_t seed, const float* pcm, int64_t n_samples, int32_t* codes,
                         int32_t cap_frames, float* latent_out);

typedef struct q3o_vocoder q3o_vocoder;
q3o_vocoder* q3o_vocoder_create(const q3o_vocoder_config* cfg, uint64_t seed, int32_t n_threads);
void q3o_vocoder_destroy(q3o_vocoder* v);
void q3o_vocoder_reset(q3o_vocoder* v);
/* 0 (default): GEMM / conv inputs rounded to bf16 (the device's operand precision); 1: plain f32 inputs */
void q3o_vocoder_set_arith(q3o_vocoder* v, int32_t f32_inputs);
void q3o_vocoder_set_arith_mask(q3o_vocoder* v, uint32_t f32_groups_mask);  /* bit g: stage group g keeps f32 GEMM inputs (error budget) */
/* tests: intermediate tensors of one whole decode (stage 1 transformer input, 2 transformer output, 3 up-sampled latent, 4 PCM before
 * the clamp) and the synthetic tensors by id, for loading the same model into the family code */
int32_t q3o_vocoder_stage(q3o_vocoder* v, const int32_t* codes, int32_t n_frames, int32_t stage, float* out);
void q3o_vocoder_mat(const q3o_vocoder* v, int32_t comp, int32_t which, int64_t rows, int64_t cols, int32_t fan_in, float gain, float* out);
void q3o_vocoder_vec(const q3o_vocoder* v, int32_t comp, int32_t which, int64_t n, float base, float std, float* out);
/* streaming call: codes [n_frames][n_codebooks] (clamped by the caller as src/tts/engine.rs:515-519);
 * returns samples written */
int32_t q3o_vocoder_decode(q3o_vocoder* v, const int32_t* codes, int32_t n_frames, int32_t is_last, float* pcm_out,
                           int32_t max_samples);

#ifdef __cplusplus
}
#endif
#endif
