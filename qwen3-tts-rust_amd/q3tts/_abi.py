"""ctypes mirror of include/q3tts.h (the C ABI of libq3tts.so).

The structures restate the header field for field; `tests/test_abi_cpu.py` checks that every symbol the
header declares is exported by the built library.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
PKG_ROOT = os.path.dirname(HERE)
REPO_ROOT = os.path.dirname(PKG_ROOT)
LIB_PATH = os.path.join(PKG_ROOT, "csrc", "libq3tts.so")

Q3TTS_MAX_UPSAMPLE = 4
Q3TTS_MAX_DEC_BLOCKS = 8


class ModelConfig(C.Structure):
    _fields_ = [
        ("t_n_layer", C.c_int32), ("t_d_model", C.c_int32), ("t_n_head", C.c_int32), ("t_n_kv_head", C.c_int32),
        ("t_head_dim", C.c_int32), ("t_d_ffn", C.c_int32), ("t_vocab", C.c_int32),
        ("t_rope_theta", C.c_float),
        ("t_mrope_sections", C.c_int32 * 4),
        ("p_n_layer", C.c_int32), ("p_d_model", C.c_int32), ("p_n_head", C.c_int32), ("p_n_kv_head", C.c_int32),
        ("p_head_dim", C.c_int32), ("p_d_ffn", C.c_int32),
        ("p_rope_theta", C.c_float),
        ("n_codebooks", C.c_int32), ("codebook_size", C.c_int32),
        ("rms_eps", C.c_float),
        ("d_embed", C.c_int32), ("text_vocab", C.c_int32), ("codec0_rows", C.c_int32), ("codecq_rows", C.c_int32),
        ("sample_limit", C.c_int32), ("eos_code", C.c_int32), ("tts_pad_id", C.c_int32),
    ]


class VocoderConfig(C.Structure):
    _fields_ = [
        ("n_codebooks", C.c_int32), ("codebook_size", C.c_int32), ("codebook_dim", C.c_int32),
        ("latent_dim", C.c_int32), ("pre_conv_kernel", C.c_int32),
        ("n_layer", C.c_int32), ("n_head", C.c_int32), ("head_dim", C.c_int32), ("d_ffn", C.c_int32),
        ("sliding_window", C.c_int32),
        ("rope_theta", C.c_float), ("rms_eps", C.c_float), ("layer_scale_init", C.c_float),
        ("n_upsample", C.c_int32), ("upsample_ratios", C.c_int32 * Q3TTS_MAX_UPSAMPLE),
        ("decoder_dim", C.c_int32), ("n_dec_blocks", C.c_int32), ("dec_rates", C.c_int32 * Q3TTS_MAX_DEC_BLOCKS),
        ("lookahead_frames", C.c_int32), ("sample_rate", C.c_int32),
    ]


class EngineConfig(C.Structure):
    _fields_ = [
        ("model", ModelConfig), ("vocoder", VocoderConfig),
        ("device", C.c_int32), ("max_batch", C.c_int32), ("n_ctx", C.c_int32), ("max_steps_cap", C.c_int32),
        ("with_vocoder", C.c_int32),
        ("synth_seed", C.c_uint64),
        ("weights_path", C.c_char_p),
        ("talker_q8_0", C.c_int32),
        ("vocoder_flush_tail", C.c_int32),
    ]


class PromptDesc(C.Structure):
    _fields_ = [
        ("text_ids", C.POINTER(C.c_uint32)), ("n_text", C.c_int32),
        ("instruct_ids", C.POINTER(C.c_uint32)), ("n_instruct", C.c_int32),
        ("lang_id", C.c_int32), ("spk_id", C.c_int32),
        ("spk_emb", C.POINTER(C.c_float)),
        ("ref_codes", C.POINTER(C.c_int32)), ("n_ref_frames", C.c_int32),
        ("ref_text_ids", C.POINTER(C.c_uint32)), ("n_ref_text", C.c_int32),
    ]


class Request(C.Structure):
    _fields_ = [
        ("prompt_embd", C.POINTER(C.c_float)), ("n_tok", C.c_int32),
        ("prompt", C.POINTER(PromptDesc)),
        ("use_engine_sampler", C.c_int32),
        ("temperature", C.c_float), ("top_k", C.c_int32), ("top_p", C.c_float), ("has_seed", C.c_int32),
        ("seed", C.c_uint64),
        ("max_steps", C.c_int32), ("min_frames", C.c_int32), ("force_eos_at", C.c_int32), ("want_pcm", C.c_int32),
    ]


class Result(C.Structure):
    _fields_ = [
        ("status", C.c_int32), ("n_frames", C.c_int32), ("hit_eos", C.c_int32),
        ("codes", C.POINTER(C.c_int32)), ("pcm", C.POINTER(C.c_float)),
        ("n_samples", C.c_int32), ("sample_rate", C.c_int32),
        ("first_chunk_ms", C.c_float), ("total_ms", C.c_float),
    ]


class Timings(C.Structure):
    _fields_ = [
        ("prefill_ms", C.c_float), ("decode_ms", C.c_float), ("vocoder_ms", C.c_float), ("total_ms", C.c_float),
        ("frame_step_ms", C.c_float), ("probe_kernel_ms", C.c_float),
        ("frame_steps", C.c_int64), ("algo_bytes_per_step", C.c_int64), ("algo_flops_per_step", C.c_int64),
        ("mean_live_slots", C.c_float), ("mean_rows", C.c_float), ("probe_count", C.c_int64), ("probe_empty_ms", C.c_float),
        ("mean_ctx_tokens", C.c_float),
    ]


class NodeTimings(C.Structure):
    _fields_ = [("generate_ms", C.c_float), ("gather_ms", C.c_float), ("total_ms", C.c_float), ("gathered_bytes", C.c_int64), ("n_devices", C.c_int32)]


class CloneConfig(C.Structure):
    """q3tts_clone_config: the two encoders of the voice-clone front-end (include/q3tts.h)."""
    _fields_ = [
        ("mel_dim", C.c_int32),
        ("se_channels", C.c_int32 * 5), ("se_kernels", C.c_int32 * 5), ("se_dilations", C.c_int32 * 5),
        ("se_attn_channels", C.c_int32), ("se_res2net_scale", C.c_int32), ("se_se_channels", C.c_int32), ("se_dim", C.c_int32),
        ("ae_filters", C.c_int32), ("ae_kernel", C.c_int32), ("ae_res_kernel", C.c_int32), ("ae_last_kernel", C.c_int32),
        ("ae_n_ratios", C.c_int32), ("ae_ratios", C.c_int32 * 4),
        ("ae_hidden", C.c_int32), ("ae_n_layer", C.c_int32), ("ae_n_head", C.c_int32), ("ae_head_dim", C.c_int32),
        ("ae_d_ffn", C.c_int32), ("ae_window", C.c_int32),
        ("ae_rope_theta", C.c_float), ("ae_ln_eps", C.c_float), ("ae_layer_scale", C.c_float),
        ("ae_down_stride", C.c_int32), ("ae_vq_dim", C.c_int32), ("ae_n_codebooks", C.c_int32), ("ae_codebook_size", C.c_int32),
    ]


# every symbol include/q3tts.h declares (tests check the export list against the header text)
SYMBOLS = [
    "q3tts_default_config", "q3tts_engine_create", "q3tts_engine_destroy", "q3tts_last_error", "q3tts_set_sampler",
    "q3tts_set_max_steps", "q3tts_build_prompt", "q3tts_free", "q3tts_generate", "q3tts_generate_batch",
    "q3tts_result_free", "q3tts_stream_begin", "q3tts_stream_poll", "q3tts_stream_end",
    "q3tts_get_timings", "q3tts_k_gemm_exact", "q3tts_k_attention", "q3tts_k_sample", "q3tts_k_talker_prefill",
    "q3tts_k_vocoder", "q3tts_k_vocoder_bench", "q3tts_set_device_pcm", "q3tts_get_device_pcm", "q3tts_k_rng_f32", "q3tts_k_probe", "q3tts_k_gguf_read", "q3tts_mel_frames", "q3tts_mel",
    "q3tts_clone_default_config", "q3tts_clone_init", "q3tts_clone_audio_frames", "q3tts_clone_audio_encode",
    "q3tts_clone_speaker_encode", "q3tts_k_speaker_from_mel", "q3tts_k_audio_latent",
    "q3tts_node_create", "q3tts_node_destroy", "q3tts_node_generate_batch", "q3tts_node_get_timings", "q3tts_node_last_error", "q3tts_node_size", "q3tts_node_engine", "q3tts_node_shard", "q3tts_k_bgemm_q8", "q3tts_k_bgemm_q8a8", "q3tts_k_alloc_upload", "q3tts_k_bgemm_policy", "q3tts_k_attend_policy", "q3tts_k_bgemm_pick", "q3tts_k_mfma_bf16", "q3tts_k_bgemm", "q3tts_k_bgemm_voc", "q3tts_k_project", "q3tts_k_norm_inputs", "q3tts_tokenizer_load", "q3tts_tokenizer_free", "q3tts_tokenizer_vocab_size", "q3tts_tokenizer_encode", "q3tts_tokenizer_decode",
]


class Q3Error(RuntimeError):
    pass


_lib = None


def load_library(path=None):
    """Loads libq3tts.so. Fails loudly when the HIP extension is missing: there is no CPU fallback."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("Q3TTS_LIB") or LIB_PATH  # Q3TTS_LIB: experiment builds of the same HIP library
    if not os.path.exists(p):
        raise Q3Error(
            f"{p} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()'). "
            "There is no CPU fallback for the product path.")
    lib = C.CDLL(p)
    f32p, i32p, u32p = C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_uint32)
    vp = C.c_void_p
    lib.q3tts_default_config.argtypes = [C.POINTER(EngineConfig)]
    lib.q3tts_default_config.restype = None
    lib.q3tts_engine_create.argtypes = [C.POINTER(EngineConfig), C.POINTER(vp)]
    lib.q3tts_engine_destroy.argtypes = [vp]
    lib.q3tts_engine_destroy.restype = None
    lib.q3tts_last_error.argtypes = [vp]
    lib.q3tts_last_error.restype = C.c_char_p
    lib.q3tts_set_sampler.argtypes = [vp, C.c_float, C.c_int32, C.c_float, C.c_int32, C.c_uint64]
    lib.q3tts_set_max_steps.argtypes = [vp, C.c_int32]
    lib.q3tts_build_prompt.argtypes = [vp, C.POINTER(PromptDesc), C.POINTER(f32p), i32p]
    lib.q3tts_free.argtypes = [vp]
    lib.q3tts_free.restype = None
    lib.q3tts_generate.argtypes = [vp, C.POINTER(Request), C.POINTER(Result)]
    lib.q3tts_generate_batch.argtypes = [vp, C.POINTER(Request), C.c_int32, C.POINTER(Result)]
    lib.q3tts_result_free.argtypes = [C.POINTER(Result)]
    lib.q3tts_result_free.restype = None
    lib.q3tts_stream_begin.argtypes = [vp, C.POINTER(Request), C.POINTER(vp)]
    lib.q3tts_stream_poll.argtypes = [vp, C.POINTER(f32p), i32p, i32p]
    lib.q3tts_stream_end.argtypes = [vp, C.POINTER(Result)]
    lib.q3tts_get_timings.argtypes = [vp, C.POINTER(Timings)]
    lib.q3tts_k_gemm_exact.argtypes = [C.c_int32, f32p, C.c_int32, C.c_int32, C.POINTER(C.c_uint16), C.c_int32, f32p,
                                       C.c_float, f32p, C.c_int32, f32p, C.POINTER(C.c_uint64), C.c_int32, f32p]
    lib.q3tts_k_attention.argtypes = [C.c_int32, f32p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, f32p, f32p,
                                      C.c_float, C.c_float, i32p, f32p]
    lib.q3tts_k_sample.argtypes = [C.c_int32, f32p, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_int32, C.c_float,
                                   f32p, i32p]
    lib.q3tts_k_talker_prefill.argtypes = [vp, f32p, C.c_int32, f32p, f32p]
    lib.q3tts_k_vocoder.argtypes = [vp, i32p, C.c_int32, C.c_int32, f32p, i32p]
    lib.q3tts_k_vocoder_bench.argtypes = [vp, C.c_int32, C.c_int32, f32p]
    lib.q3tts_set_device_pcm.argtypes = [vp, C.c_int32]
    lib.q3tts_get_device_pcm.argtypes = [vp, C.POINTER(f32p), C.POINTER(C.c_int64), i32p]
    lib.q3tts_k_rng_f32.argtypes = [C.c_uint64, C.c_int32, f32p]
    lib.q3tts_k_probe.argtypes = [C.c_void_p, C.c_int32]
    lib.q3tts_mel_frames.argtypes = [C.c_int64]
    lib.q3tts_mel_frames.restype = C.c_int32
    lib.q3tts_mel.argtypes = [C.c_void_p, f32p, C.c_int64, f32p, C.c_int32, C.POINTER(C.c_int32)]
    lib.q3tts_k_gguf_read.argtypes = [C.c_char_p, C.c_char_p, f32p, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int32)]
    lib.q3tts_clone_default_config.argtypes = [C.POINTER(CloneConfig)]
    lib.q3tts_clone_default_config.restype = None
    lib.q3tts_clone_init.argtypes = [vp, C.POINTER(CloneConfig)]
    lib.q3tts_clone_audio_frames.argtypes = [vp, C.c_int64]
    lib.q3tts_clone_audio_frames.restype = C.c_int32
    lib.q3tts_clone_audio_encode.argtypes = [vp, f32p, C.c_int64, C.POINTER(C.c_int64), C.c_int32, i32p]
    lib.q3tts_clone_speaker_encode.argtypes = [vp, f32p, C.c_int64, f32p]
    lib.q3tts_k_speaker_from_mel.argtypes = [vp, f32p, C.c_int32, f32p]
    lib.q3tts_k_audio_latent.argtypes = [vp, f32p, C.c_int64, f32p, C.c_int32, i32p]
    lib.q3tts_k_mfma_bf16.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]
    lib.q3tts_k_bgemm.argtypes = [C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_float,
                                  C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, f32p]
    lib.q3tts_k_project.argtypes = [C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.q3tts_k_norm_inputs.argtypes = [C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.q3tts_k_bgemm_voc.argtypes = [C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p,
                                      C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32]
    lib.q3tts_node_create.argtypes = [C.POINTER(EngineConfig), i32p, C.c_int32, C.POINTER(vp)]
    lib.q3tts_node_destroy.argtypes = [vp]
    lib.q3tts_node_destroy.restype = None
    lib.q3tts_node_generate_batch.argtypes = [vp, C.POINTER(Request), C.c_int32, C.POINTER(Result), C.POINTER(C.POINTER(C.c_int16))]
    lib.q3tts_node_get_timings.argtypes = [vp, C.POINTER(NodeTimings)]
    lib.q3tts_node_last_error.argtypes = [vp]
    lib.q3tts_node_last_error.restype = C.c_char_p
    lib.q3tts_node_size.argtypes = [vp]
    lib.q3tts_node_size.restype = C.c_int32
    lib.q3tts_node_engine.argtypes = [vp, C.c_int32]
    lib.q3tts_node_engine.restype = vp
    lib.q3tts_node_shard.argtypes = [C.c_int32, C.c_int32, C.c_int32, i32p, C.c_int32]
    lib.q3tts_node_shard.restype = C.c_int32
    lib.q3tts_k_bgemm_q8.argtypes = [C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_float,
                                     C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, f32p]
    lib.q3tts_k_bgemm_q8a8.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_float,
                                       C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, f32p]
    lib.q3tts_k_bgemm_policy.argtypes = [C.c_int32]
    lib.q3tts_k_attend_policy.argtypes = [C.c_int32, C.c_int32]
    lib.q3tts_k_bgemm_pick.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, i32p]
    lib.q3tts_k_alloc_upload.argtypes = [vp, C.c_int64, C.POINTER(C.c_int64)]
    lib.q3tts_tokenizer_load.argtypes = [C.c_char_p, C.POINTER(vp), C.c_char_p, C.c_int32]
    lib.q3tts_tokenizer_free.argtypes = [vp]
    lib.q3tts_tokenizer_free.restype = None
    lib.q3tts_tokenizer_vocab_size.argtypes = [vp]
    lib.q3tts_tokenizer_vocab_size.restype = C.c_int32
    lib.q3tts_tokenizer_encode.argtypes = [vp, C.c_char_p, C.c_int64, u32p, C.c_int32, i32p, C.c_char_p, C.c_int32]
    lib.q3tts_tokenizer_decode.argtypes = [vp, u32p, C.c_int32, C.c_char_p, C.c_int64, C.POINTER(C.c_int64), C.c_char_p, C.c_int32]
    if path is None:
        _lib = lib
    return lib


def default_config():
    cfg = EngineConfig()
    load_library().q3tts_default_config(C.byref(cfg))
    return cfg


def tiny_config(max_batch=4, n_ctx=256, with_vocoder=1):
    """A small shape the CPU oracle finishes in well under a second (used by parity tests)."""
    cfg = EngineConfig()
    m = cfg.model
    m.t_n_layer, m.t_d_model, m.t_n_head, m.t_n_kv_head, m.t_head_dim, m.t_d_ffn, m.t_vocab = 2, 512, 4, 2, 128, 1024, 3072
    m.t_rope_theta = 1000000.0
    m.t_mrope_sections[:] = [24, 20, 20, 0]
    m.p_n_layer, m.p_d_model, m.p_n_head, m.p_n_kv_head, m.p_head_dim, m.p_d_ffn = 2, 512, 4, 2, 128, 512
    m.p_rope_theta = 1000000.0
    m.n_codebooks, m.codebook_size = 16, 64
    m.rms_eps = 1e-6
    m.d_embed, m.text_vocab, m.codec0_rows, m.codecq_rows = 512, 151936, 3072, 64
    m.sample_limit, m.eos_code, m.tts_pad_id = 2160, 2150, 151671
    v = cfg.vocoder
    v.n_codebooks, v.codebook_size, v.codebook_dim = 16, 64, 32
    v.latent_dim, v.pre_conv_kernel = 64, 3
    v.n_layer, v.n_head, v.head_dim, v.d_ffn, v.sliding_window = 2, 2, 32, 128, 8
    v.rope_theta, v.rms_eps, v.layer_scale_init = 10000.0, 1e-5, 0.01
    v.n_upsample = 2
    v.upsample_ratios[:] = [2, 2, 0, 0]
    v.decoder_dim, v.n_dec_blocks = 512, 4
    v.dec_rates[:] = [8, 5, 4, 3, 0, 0, 0, 0]
    v.lookahead_frames, v.sample_rate = 0, 24000
    cfg.device, cfg.max_batch, cfg.n_ctx, cfg.max_steps_cap, cfg.with_vocoder = 0, max_batch, n_ctx, 64, with_vocoder
    cfg.synth_seed = 0
    cfg.weights_path = None
    cfg.talker_q8_0 = 0
    cfg.vocoder_flush_tail = 0
    return cfg


def full_config_py():
    """The Qwen3-TTS-12Hz-1.7B shape of SURVEY.md §8 without touching the library (CPU-only tests)."""
    cfg = EngineConfig()
    m = cfg.model
    m.t_n_layer, m.t_d_model, m.t_n_head, m.t_n_kv_head, m.t_head_dim, m.t_d_ffn, m.t_vocab = 28, 2048, 16, 8, 128, 6144, 3072
    m.t_rope_theta = 1000000.0
    m.t_mrope_sections[:] = [24, 20, 20, 0]
    m.p_n_layer, m.p_d_model, m.p_n_head, m.p_n_kv_head, m.p_head_dim, m.p_d_ffn = 5, 1024, 16, 8, 128, 3072
    m.p_rope_theta = 1000000.0
    m.n_codebooks, m.codebook_size = 16, 2048
    m.rms_eps = 1e-6
    m.d_embed, m.text_vocab, m.codec0_rows, m.codecq_rows = 2048, 151936, 3072, 2048
    m.sample_limit, m.eos_code, m.tts_pad_id = 2160, 2150, 151671
    v = cfg.vocoder
    v.n_codebooks, v.codebook_size, v.codebook_dim = 16, 2048, 512
    v.latent_dim, v.pre_conv_kernel = 1024, 3
    v.n_layer, v.n_head, v.head_dim, v.d_ffn, v.sliding_window = 8, 16, 64, 3072, 72
    v.rope_theta, v.rms_eps, v.layer_scale_init = 10000.0, 1e-5, 0.01
    v.n_upsample = 2
    v.upsample_ratios[:] = [2, 2, 0, 0]
    v.decoder_dim, v.n_dec_blocks = 1536, 4
    v.dec_rates[:] = [8, 5, 4, 3, 0, 0, 0, 0]
    v.lookahead_frames, v.sample_rate = 0, 24000
    cfg.device, cfg.max_batch, cfg.n_ctx, cfg.max_steps_cap, cfg.with_vocoder = 0, 64, 4096, 512, 1
    cfg.synth_seed = 0
    cfg.weights_path = None
    cfg.talker_q8_0 = 0
    cfg.vocoder_flush_tail = 0
    return cfg


def tiny_clone_config(d_embed=512):
    """Small encoder shapes for the parity tests (every structural feature of the full shape, narrower)."""
    c = CloneConfig()
    c.mel_dim = 128
    c.se_channels[:] = [64, 64, 64, 64, 192]
    c.se_kernels[:] = [5, 3, 3, 3, 1]
    c.se_dilations[:] = [1, 2, 3, 4, 1]
    c.se_attn_channels, c.se_res2net_scale, c.se_se_channels, c.se_dim = 32, 4, 32, d_embed
    c.ae_filters, c.ae_kernel, c.ae_res_kernel, c.ae_last_kernel = 32, 7, 3, 3
    c.ae_n_ratios = 4
    c.ae_ratios[:] = [4, 5, 6, 8]
    c.ae_hidden, c.ae_n_layer, c.ae_n_head, c.ae_head_dim, c.ae_d_ffn, c.ae_window = 128, 2, 4, 32, 256, 6
    c.ae_rope_theta, c.ae_ln_eps, c.ae_layer_scale = 10000.0, 1e-5, 0.5
    c.ae_down_stride, c.ae_vq_dim, c.ae_n_codebooks, c.ae_codebook_size = 2, 32, 16, 64
    return c


def full_clone_config_py():
    """The family's full encoder shapes (what q3tts_clone_default_config fills), without touching the library."""
    c = CloneConfig()
    c.mel_dim = 128
    c.se_channels[:] = [512, 512, 512, 512, 1536]
    c.se_kernels[:] = [5, 3, 3, 3, 1]
    c.se_dilations[:] = [1, 2, 3, 4, 1]
    c.se_attn_channels, c.se_res2net_scale, c.se_se_channels, c.se_dim = 128, 8, 128, 2048
    c.ae_filters, c.ae_kernel, c.ae_res_kernel, c.ae_last_kernel = 64, 7, 3, 3
    c.ae_n_ratios = 4
    c.ae_ratios[:] = [4, 5, 6, 8]
    c.ae_hidden, c.ae_n_layer, c.ae_n_head, c.ae_head_dim, c.ae_d_ffn, c.ae_window = 512, 8, 8, 64, 2048, 250
    c.ae_rope_theta, c.ae_ln_eps, c.ae_layer_scale = 10000.0, 1e-5, 0.01
    c.ae_down_stride, c.ae_vq_dim, c.ae_n_codebooks, c.ae_codebook_size = 2, 256, 16, 2048
    return c
