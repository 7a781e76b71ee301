//! `extern "C"` declarations of include/q3tts.h and a safe wrapper that keeps the reference crate's names:
//! `TtsEngine::{new, set_max_steps, set_sampler_config, generate_with_voice}`, `SamplerConfig`, `VoiceFile`,
//! `AudioSample`. UNCOMPILED (no Rust toolchain in the image); field order mirrors the C header one to one.
#![allow(non_camel_case_types)]
use std::ffi::CStr;
use std::os::raw::{c_char, c_float, c_int};

#[repr(C)] #[derive(Clone, Copy)]
pub struct q3tts_model_config {
    pub t_n_layer: i32, pub t_d_model: i32, pub t_n_head: i32, pub t_n_kv_head: i32, pub t_head_dim: i32, pub t_d_ffn: i32, pub t_vocab: i32,
    pub t_rope_theta: c_float, pub t_mrope_sections: [i32; 4],
    pub p_n_layer: i32, pub p_d_model: i32, pub p_n_head: i32, pub p_n_kv_head: i32, pub p_head_dim: i32, pub p_d_ffn: i32,
    pub p_rope_theta: c_float, pub n_codebooks: i32, pub codebook_size: i32, pub rms_eps: c_float,
    pub d_embed: i32, pub text_vocab: i32, pub codec0_rows: i32, pub codecq_rows: i32,
    pub sample_limit: i32, pub eos_code: i32, pub tts_pad_id: i32,
}
#[repr(C)] #[derive(Clone, Copy)]
pub struct q3tts_vocoder_config {
    pub n_codebooks: i32, pub codebook_size: i32, pub codebook_dim: i32, pub latent_dim: i32, pub pre_conv_kernel: i32,
    pub n_layer: i32, pub n_head: i32, pub head_dim: i32, pub d_ffn: i32, pub sliding_window: i32,
    pub rope_theta: c_float, pub rms_eps: c_float, pub layer_scale_init: c_float,
    pub n_upsample: i32, pub upsample_ratios: [i32; 4], pub decoder_dim: i32, pub n_dec_blocks: i32, pub dec_rates: [i32; 8],
    pub lookahead_frames: i32, pub sample_rate: i32,
}
#[repr(C)]
pub struct q3tts_engine_config {
    pub model: q3tts_model_config, pub vocoder: q3tts_vocoder_config,
    pub device: i32, pub max_batch: i32, pub n_ctx: i32, pub max_steps_cap: i32, pub with_vocoder: i32,
    pub synth_seed: u64, pub weights_path: *const c_char,
    pub talker_q8_0: i32,   // 2: the Talker's Q8_0 blocks stay quantised on the device and meet Q8_0 activations (W8A8, the crate's default "q8_0" directory); 1: W8A16; 0: bf16
    pub vocoder_flush_tail: i32,   // 0: the look-ahead tail is flushed as the reference does (only when n_frames % 4 != 0); 1: always
}
#[repr(C)]
pub struct q3tts_prompt_desc {
    pub text_ids: *const u32, pub n_text: i32, pub instruct_ids: *const u32, pub n_instruct: i32,
    pub lang_id: i32, pub spk_id: i32, pub spk_emb: *const c_float,
    pub ref_codes: *const i32, pub n_ref_frames: i32, pub ref_text_ids: *const u32, pub n_ref_text: i32,
}
#[repr(C)]
pub struct q3tts_request {
    pub prompt_embd: *const c_float, pub n_tok: i32, pub prompt: *const q3tts_prompt_desc, pub use_engine_sampler: i32,
    pub temperature: c_float, pub top_k: i32, pub top_p: c_float, pub has_seed: i32, pub seed: u64,
    pub max_steps: i32, pub min_frames: i32, pub force_eos_at: i32, pub want_pcm: i32,
}
#[repr(C)]
pub struct q3tts_result {
    pub status: i32, pub n_frames: i32, pub hit_eos: i32, pub codes: *mut i32, pub pcm: *mut c_float,
    pub n_samples: i32, pub sample_rate: i32, pub first_chunk_ms: c_float, pub total_ms: c_float,
}
pub enum q3tts_engine {}
pub enum q3tts_stream {}
pub enum q3tts_node {}
#[repr(C)] #[derive(Clone, Copy, Default)]
pub struct q3tts_node_timings { pub generate_ms: c_float, pub gather_ms: c_float, pub total_ms: c_float, pub gathered_bytes: i64, pub n_devices: i32 }

#[link(name = "q3tts")]
extern "C" {
    pub fn q3tts_default_config(cfg: *mut q3tts_engine_config);
    pub fn q3tts_engine_create(cfg: *const q3tts_engine_config, out: *mut *mut q3tts_engine) -> c_int;
    pub fn q3tts_engine_destroy(e: *mut q3tts_engine);
    pub fn q3tts_last_error(e: *const q3tts_engine) -> *const c_char;
    pub fn q3tts_set_sampler(e: *mut q3tts_engine, temperature: c_float, top_k: i32, top_p: c_float, has_seed: i32, seed: u64) -> c_int;
    pub fn q3tts_set_max_steps(e: *mut q3tts_engine, max_steps: i32) -> c_int;
    pub fn q3tts_generate(e: *mut q3tts_engine, req: *const q3tts_request, out: *mut q3tts_result) -> c_int;
    pub fn q3tts_generate_batch(e: *mut q3tts_engine, reqs: *const q3tts_request, n: i32, outs: *mut q3tts_result) -> c_int;
    pub fn q3tts_result_free(r: *mut q3tts_result);
    // streaming: 4-frame (64-code) chunks as the reference's vocoder thread produces them (src/tts/engine.rs:507-541)
    pub fn q3tts_stream_begin(e: *mut q3tts_engine, req: *const q3tts_request, out: *mut *mut q3tts_stream) -> c_int;
    pub fn q3tts_stream_poll(s: *mut q3tts_stream, chunk: *mut *const c_float, n_samples: *mut i32, is_final: *mut i32) -> c_int;
    pub fn q3tts_stream_end(s: *mut q3tts_stream, out_codes_optional: *mut q3tts_result) -> c_int;
    pub fn q3tts_free(p: *mut std::ffi::c_void);
    // one node, several GPUs (no counterpart in the reference: n_seq_max = 1, src/models/llama/mod.rs:413): one engine + host thread per device
    // inside the library, request i on device i % n_devices, optional RCCL gather of the i16 PCM to the first device
    pub fn q3tts_node_create(cfg: *const q3tts_engine_config, devices: *const i32, n_devices: i32, out: *mut *mut q3tts_node) -> c_int;
    pub fn q3tts_node_destroy(n: *mut q3tts_node);
    pub fn q3tts_node_generate_batch(n: *mut q3tts_node, reqs: *const q3tts_request, n_reqs: i32, outs: *mut q3tts_result, pcm_i16: *mut *mut i16) -> c_int;
    pub fn q3tts_node_get_timings(n: *const q3tts_node, out: *mut q3tts_node_timings) -> c_int;
    pub fn q3tts_node_last_error(n: *const q3tts_node) -> *const c_char;
    // voice-clone encoders (replace AudioEncoder / SpeakerEncoder, src/models/onnx.rs:82-160)
    pub fn q3tts_clone_default_config(cfg: *mut q3tts_clone_config);
    pub fn q3tts_clone_init(e: *mut q3tts_engine, cfg: *const q3tts_clone_config) -> c_int;
    pub fn q3tts_clone_audio_frames(e: *const q3tts_engine, n_samples: i64) -> i32;
    pub fn q3tts_clone_audio_encode(e: *mut q3tts_engine, audio: *const c_float, n_samples: i64, codes: *mut i64, cap_frames: i32, n_frames: *mut i32) -> c_int;
    pub fn q3tts_clone_speaker_encode(e: *mut q3tts_engine, audio: *const c_float, n_samples: i64, spk_emb: *mut c_float) -> c_int;
}

/// q3tts_clone_config (include/q3tts.h), field for field
#[repr(C)]
#[derive(Clone, Copy)]
pub struct q3tts_clone_config {
    pub mel_dim: i32,
    pub se_channels: [i32; 5], pub se_kernels: [i32; 5], pub se_dilations: [i32; 5],
    pub se_attn_channels: i32, pub se_res2net_scale: i32, pub se_se_channels: i32, pub se_dim: i32,
    pub ae_filters: i32, pub ae_kernel: i32, pub ae_res_kernel: i32, pub ae_last_kernel: i32,
    pub ae_n_ratios: i32, pub ae_ratios: [i32; 4],
    pub ae_hidden: i32, pub ae_n_layer: i32, pub ae_n_head: i32, pub ae_head_dim: i32, pub ae_d_ffn: i32, pub ae_window: i32,
    pub ae_rope_theta: c_float, pub ae_ln_eps: c_float, pub ae_layer_scale: c_float,
    pub ae_down_stride: i32, pub ae_vq_dim: i32, pub ae_n_codebooks: i32, pub ae_codebook_size: i32,
}

// ---- the reference crate's public names on top of the C ABI (src/lib.rs:11-20) ------------------------------------
#[derive(Debug, Clone)]
pub struct SamplerConfig { pub temperature: f32, pub top_k: i32, pub top_p: f32, pub seed: Option<u64> }
impl Default for SamplerConfig { fn default() -> Self { Self { temperature: 0.7, top_k: 40, top_p: 0.9, seed: None } } }

pub struct VoiceFile { pub ref_text: String, pub audio_codes: Vec<i64>, pub speaker_embedding: Vec<f32> }
pub struct AudioSample { pub samples: Vec<f32>, pub sample_rate: u32, pub channels: u16 }

pub struct TtsEngine { raw: *mut q3tts_engine, sampler: SamplerConfig, max_steps: usize }

impl TtsEngine {
    /// TtsEngine::new(model_dir, quant) — src/tts/engine.rs:84-169. `model_dir/<quant dir>` (src/tts/engine.rs:91-95) holds
    /// qwen3_tts_talker.gguf, qwen3_tts_predictor.gguf and qwen3_assets.gguf (or the NPY assets): passed as `weights_path`.
    pub fn new(model_dir: &str, quant: &str) -> Result<Self, String> {
        let quant_dir = match quant { "q5_k_m" => "gguf_q5_k_m", "q8_0" => "gguf_q8_0", _ => "gguf" };
        let dir = std::ffi::CString::new(format!("{}/{}", model_dir, quant_dir)).map_err(|e| e.to_string())?;
        unsafe {
            let mut cfg: q3tts_engine_config = std::mem::zeroed();
            q3tts_default_config(&mut cfg);
            cfg.weights_path = dir.as_ptr();   // borrowed for the call only
            cfg.talker_q8_0 = if quant == "q8_0" { 2 } else { 0 };   // gguf_q8_0: the Talker's blocks stay as stored and are multiplied as llama.cpp does (Q8_0 x Q8_0, W8A8)
            let mut raw = std::ptr::null_mut();
            let rc = q3tts_engine_create(&cfg, &mut raw);
            if rc != 0 { return Err(format!("q3tts_engine_create failed: {}", rc)); }
            Ok(Self { raw, sampler: SamplerConfig::default(), max_steps: 512 })
        }
    }
    pub fn set_max_steps(&mut self, steps: usize) { self.max_steps = steps; }
    pub fn set_sampler_config(&mut self, c: SamplerConfig) { self.sampler = c; }
    pub fn get_sampler_config(&self) -> &SamplerConfig { &self.sampler }

    /// generate_with_voice — src/tts/engine.rs:390. `text_ids` = tokenizer.encode(text) (the tokenizer stays in Rust).
    pub fn generate_with_voice(&mut self, text_ids: &[u32], voice: &VoiceFile, instruct_ids: Option<&[u32]>) -> Result<AudioSample, String> {
        let codes32: Vec<i32> = voice.audio_codes.iter().map(|&c| c as i32).collect();
        let desc = q3tts_prompt_desc {
            text_ids: text_ids.as_ptr(), n_text: text_ids.len() as i32,
            instruct_ids: instruct_ids.map_or(std::ptr::null(), |s| s.as_ptr()), n_instruct: instruct_ids.map_or(0, |s| s.len() as i32),
            lang_id: 2055, spk_id: -1, spk_emb: voice.speaker_embedding.as_ptr(),
            ref_codes: if codes32.is_empty() { std::ptr::null() } else { codes32.as_ptr() }, n_ref_frames: (codes32.len() / 16) as i32,
            ref_text_ids: std::ptr::null(), n_ref_text: 0, // tokenizer.encode(&voice.ref_text) in the real crate
        };
        let req = q3tts_request {
            prompt_embd: std::ptr::null(), n_tok: 0, prompt: &desc, use_engine_sampler: 0,
            temperature: self.sampler.temperature, top_k: self.sampler.top_k, top_p: self.sampler.top_p,
            has_seed: self.sampler.seed.is_some() as i32, seed: self.sampler.seed.unwrap_or(0),
            max_steps: self.max_steps as i32, min_frames: 0, force_eos_at: -1, want_pcm: 1,
        };
        unsafe {
            let mut out: q3tts_result = std::mem::zeroed();
            let rc = q3tts_generate(self.raw, &req, &mut out);
            if rc != 0 { return Err(CStr::from_ptr(q3tts_last_error(self.raw)).to_string_lossy().into_owned()); }
            let samples = std::slice::from_raw_parts(out.pcm, out.n_samples as usize).to_vec();
            q3tts_result_free(&mut out);
            Ok(AudioSample { samples, sample_rate: 24000, channels: 1 })
        }
    }
}
impl TtsEngine {
    /// run_inference_stream with `stream_tx = Some(..)` — src/tts/engine.rs:444-448,520-526: every decoded 4-frame chunk is sent
    /// down the channel as soon as its PCM is on the host (`stx.send(samples.clone())`, send errors ignored as there), and the whole
    /// utterance is returned at the end. The reference threads `stream_tx` through privately (:438-443 pass None); this is the
    /// same hook made callable. Chunks come from q3tts_stream_poll; the engine overlaps the vocoder with the next frames itself.
    pub fn generate_with_voice_stream(&mut self, text_ids: &[u32], voice: &VoiceFile, instruct_ids: Option<&[u32]>,
                                      stream_tx: Option<std::sync::mpsc::Sender<Vec<f32>>>) -> Result<AudioSample, String> {
        let codes32: Vec<i32> = voice.audio_codes.iter().map(|&c| c as i32).collect();
        let desc = q3tts_prompt_desc {
            text_ids: text_ids.as_ptr(), n_text: text_ids.len() as i32,
            instruct_ids: instruct_ids.map_or(std::ptr::null(), |s| s.as_ptr()), n_instruct: instruct_ids.map_or(0, |s| s.len() as i32),
            lang_id: 2055, spk_id: -1, spk_emb: voice.speaker_embedding.as_ptr(),
            ref_codes: if codes32.is_empty() { std::ptr::null() } else { codes32.as_ptr() }, n_ref_frames: (codes32.len() / 16) as i32,
            ref_text_ids: std::ptr::null(), n_ref_text: 0,
        };
        let req = q3tts_request {
            prompt_embd: std::ptr::null(), n_tok: 0, prompt: &desc, use_engine_sampler: 0,
            temperature: self.sampler.temperature, top_k: self.sampler.top_k, top_p: self.sampler.top_p,
            has_seed: self.sampler.seed.is_some() as i32, seed: self.sampler.seed.unwrap_or(0),
            max_steps: self.max_steps as i32, min_frames: 0, force_eos_at: -1, want_pcm: 1,
        };
        unsafe {
            let err = |e: *mut q3tts_engine| CStr::from_ptr(q3tts_last_error(e)).to_string_lossy().into_owned();
            let mut st: *mut q3tts_stream = std::ptr::null_mut();
            if q3tts_stream_begin(self.raw, &req, &mut st) != 0 { return Err(err(self.raw)); }
            let mut full_audio: Vec<f32> = Vec::new();
            loop {
                let (mut chunk, mut n, mut fin) = (std::ptr::null::<c_float>(), 0i32, 0i32);
                if q3tts_stream_poll(st, &mut chunk, &mut n, &mut fin) != 0 { q3tts_stream_end(st, std::ptr::null_mut()); return Err(err(self.raw)); }
                if n > 0 {
                    let samples = std::slice::from_raw_parts(chunk, n as usize).to_vec();  // the chunk is only valid until the next poll
                    if let Some(ref stx) = stream_tx { let _ = stx.send(samples.clone()); }
                    full_audio.extend(samples);
                }
                if fin != 0 { break; }
            }
            if q3tts_stream_end(st, std::ptr::null_mut()) != 0 { return Err(err(self.raw)); }
            Ok(AudioSample { samples: full_audio, sample_rate: 24000, channels: 1 })
        }
    }
}
impl TtsEngine {
    /// The encoder half of create_voice_file (src/tts/engine.rs:375-386): 24 kHz mono samples -> VoiceFile. WAV decoding
    /// (hound, :339-373) stays in the crate.
    pub fn create_voice_from_samples(&mut self, audio: &[f32], ref_text: String) -> Result<VoiceFile, String> {
        unsafe {
            let cap = q3tts_clone_audio_frames(self.raw, audio.len() as i64).max(1);
            let mut codes = vec![0i64; cap as usize * 16];
            let mut nf = 0i32;
            let err = |e: *mut q3tts_engine| CStr::from_ptr(q3tts_last_error(e)).to_string_lossy().into_owned();
            if q3tts_clone_audio_encode(self.raw, audio.as_ptr(), audio.len() as i64, codes.as_mut_ptr(), cap, &mut nf) != 0 { return Err(err(self.raw)); }
            codes.truncate(nf as usize * 16);
            let mut emb = vec![0f32; 2048];
            if q3tts_clone_speaker_encode(self.raw, audio.as_ptr(), audio.len() as i64, emb.as_mut_ptr()) != 0 { return Err(err(self.raw)); }
            Ok(VoiceFile { ref_text, audio_codes: codes, speaker_embedding: emb })
        }
    }
}
impl Drop for TtsEngine { fn drop(&mut self) { unsafe { q3tts_engine_destroy(self.raw) } } }
/// qwen3_tts::cleanup() (src/lib.rs:18-20): nothing global to free.
pub fn cleanup() {}
