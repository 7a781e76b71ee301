"""GPU parity tests of the voice-clone encoders: HIP path through the C ABI vs the CPU oracle.

The audio encoder's output is integer (codec ids), so the whole front-end follows canonical summation orders
 and every comparison is `==` on raw bits — floats included. PARITY UNPINNED against the reference: its two
encoder graphs exist only as ONNX files outside the repository; what is pinned is the I/O contract
(src/models/onnx.rs:96-160) and the caller (src/tts/engine.rs:275-302, 324-387).
"""
import struct
import wave

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _clip(n, seed=0):
    """speech-shaped synthetic clip: a few drifting harmonics plus noise, in [-1, 1]"""
    rng = np.random.default_rng(seed)
    t = np.arange(n, dtype=np.float64) / 24000.0
    f0 = 120.0 + 30.0 * np.sin(2 * np.pi * 1.3 * t)
    ph = 2 * np.pi * np.cumsum(f0) / 24000.0
    x = sum(np.sin(k * ph) / k for k in range(1, 6)) * 0.2 + rng.standard_normal(n) * 0.02
    return np.clip(x, -1, 1).astype(np.float32)


@pytest.fixture(scope="module")
def tiny_clone(oracle):
    from q3tts import _abi, native
    cfg = _abi.tiny_config(max_batch=2, n_ctx=256, with_vocoder=1)
    eng = native.NativeEngine(cfg)
    ccfg = _abi.tiny_clone_config(cfg.model.d_embed)
    eng.clone_init(ccfg)
    yield cfg, ccfg, eng
    eng.close()


@pytest.mark.parametrize("T", [1, 2, 5, 46, 282])
def test_speaker_encoder_bits(oracle, tiny_clone, T):
    """TDNN (reflect padding narrower than the dilated kernel at T = 1, 2, 5), Res2Net chains, squeeze-excitation, attentive
    statistics pooling, final 1x1: bit-exact on a log-mel-like input."""
    cfg, ccfg, eng = tiny_clone
    rng = np.random.default_rng(T)
    mel = (rng.standard_normal((T, 128)) * 2.0 - 4.0).astype(np.float32)
    ref = oracle.speaker_encode(ccfg, 0, mel)
    out = eng.speaker_from_mel(mel)
    assert np.isfinite(out).all() and np.abs(out).max() > 1e-3
    assert np.array_equal(_bits(out), _bits(ref))


def test_speaker_encode_from_audio_uses_the_device_mel(oracle, tiny_clone):
    """SpeakerEncoder::encode takes audio (src/models/onnx.rs:135-160): the device log-mel feeds the encoder directly."""
    cfg, ccfg, eng = tiny_clone
    a = _clip(20000, 3)
    mel = eng.mel(a)
    assert mel.shape == (78, 128)
    out = eng.speaker_encode(a)
    assert np.array_equal(_bits(out), _bits(oracle.speaker_encode(ccfg, 0, mel)))
    # and the oracle's own mel (identical up to the last logf, test_mel_matches_oracle) gives the same embedding to ~1e-4
    ref = oracle.speaker_encode(ccfg, 0, oracle.mel(a))
    assert np.abs(out - ref).max() <= 1e-3 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("n", [1, 959, 960, 1921, 7000, 30000])
def test_audio_encoder_codes_and_latent_bits(oracle, tiny_clone, n):
    """causal SEANet stack (strides 4,5,6,8), transformer with a sliding window shorter than the clip, replicate-padded
    stride-2 conv, split residual VQ: ids equal and pre-quantiser rows bit-exact, ragged lengths included."""
    cfg, ccfg, eng = tiny_clone
    a = _clip(n, n)
    codes_ref, lat_ref = oracle.audio_encode(ccfg, 0, a)
    assert codes_ref.shape[0] == eng.clone_audio_frames(n) == -(-(-(-n // 960)) // 2)
    lat = eng.audio_latent(a)
    assert np.array_equal(_bits(lat), _bits(lat_ref))
    codes = eng.audio_encode(a)
    assert codes.dtype == np.int64 and codes.shape == codes_ref.shape
    assert np.array_equal(codes, codes_ref.astype(np.int64))
    assert codes.min() >= 0 and codes.max() < ccfg.ae_codebook_size
    if n >= 7000:
        assert len(np.unique(codes)) > 8  # the quantiser is not stuck on one codeword


def test_audio_encoder_is_causal(tiny_clone):
    """Every convolution is causal and the attention is causal, so the codes of a prefix are a prefix of the codes
    (up to the frames that still see the right-hand zero padding)."""
    cfg, ccfg, eng = tiny_clone
    a = _clip(24000, 9)
    full = eng.audio_encode(a)
    part = eng.audio_encode(a[:1920 * 6])
    assert np.array_equal(part[:5], full[:5])
    assert eng.audio_encode(np.zeros(0, dtype=np.float32)).shape == (0, ccfg.ae_n_codebooks)


def test_clone_errors_are_loud(oracle):
    from q3tts import _abi, native
    cfg = _abi.tiny_config(max_batch=1, n_ctx=128, with_vocoder=0)
    eng = native.NativeEngine(cfg)
    try:
        with pytest.raises(_abi.Q3Error, match="AudioEncoder not loaded"):
            eng.audio_encode(np.zeros(4000, dtype=np.float32))
        with pytest.raises(_abi.Q3Error, match="SpeakerEncoder not loaded"):
            eng.speaker_encode(np.zeros(4000, dtype=np.float32))
        bad = _abi.tiny_clone_config(cfg.model.d_embed)
        bad.se_channels[4] = 128
        with pytest.raises(_abi.Q3Error, match="3 x the block width"):
            eng.clone_init(bad)
        bad = _abi.tiny_clone_config(cfg.model.d_embed)
        bad.ae_ratios[3] = 200  # 2 x 200 x 256 channels: deeper than the exact GEMM takes
        with pytest.raises(_abi.Q3Error, match="exceeds 8192"):
            eng.clone_init(bad)
        eng.clone_init(_abi.tiny_clone_config(cfg.model.d_embed))
        with pytest.raises(_abi.Q3Error, match="shorter than one mel frame"):
            eng.speaker_encode(np.zeros(100, dtype=np.float32))
        import ctypes as C
        a = np.zeros(5000, dtype=np.float32)
        out = np.zeros((1, 16), dtype=np.int64)
        n = C.c_int32(0)
        rc = eng.lib.q3tts_clone_audio_encode(eng.h, a.ctypes.data_as(C.POINTER(C.c_float)), a.size, out.ctypes.data_as(C.POINTER(C.c_int64)), 1, C.byref(n))
        assert rc != 0 and n.value == 3 and b"too small" in eng.lib.q3tts_last_error(eng.h)
    finally:
        eng.close()


def _write_wav(path, samples, kind):
    n = len(samples)
    if kind == "i16":
        with wave.open(str(path), "wb") as w:
            w.setnchannels(1); w.setsampwidth(2); w.setframerate(24000)
            w.writeframes(np.trunc(np.clip(samples * 32767.0, -32768, 32767)).astype("<i2").tobytes())
        return
    tag, bits, body = (3, 32, samples.astype("<f4").tobytes())
    if kind == "f32_stereo":
        body = np.stack([samples, -samples], axis=1).astype("<f4").tobytes()
    ch = 2 if kind == "f32_stereo" else 1
    hdr = b"RIFF" + struct.pack("<I", 36 + len(body)) + b"WAVE" + b"fmt " + struct.pack("<IHHIIHH", 16, tag, ch, 24000, 24000 * ch * 4, ch * 4, bits)
    with open(path, "wb") as f:
        f.write(hdr + b"data" + struct.pack("<I", len(body)) + body)
    assert n


def test_create_voice_file_then_generate_matches_oracle(oracle, tiny_clone, tmp_path):
    """TtsEngine::create_voice_file (src/tts/engine.rs:324-387) and generate(ref_audio) (:243-302) end to end: WAV ->
    codes + speaker embedding -> ICL clone prompt (src/tts/prompt.rs:28-118) -> codec ids, equal to the oracle's."""
    from q3tts import api
    cfg, ccfg, eng = tiny_clone
    te = api.TtsEngine.__new__(api.TtsEngine)
    te._native, te.cfg, te.tokenizer, te.speakers, te.max_steps, te.sampler_config = eng, cfg, None, {}, 6, api.SamplerConfig(0.0, 40, 0.9, 7)
    a = _clip(9000, 4)
    _write_wav(tmp_path / "ref_f32.wav", a, "f32")
    _write_wav(tmp_path / "ref_st.wav", a, "f32_stereo")
    _write_wav(tmp_path / "ref_i16.wav", a, "i16")
    v = te.create_voice_file(tmp_path / "ref_f32.wav", [9, 8, 7])
    codes_ref, _ = oracle.audio_encode(ccfg, 0, a)
    emb_ref = oracle.speaker_encode(ccfg, 0, eng.mel(a))
    assert v.audio_codes == codes_ref.reshape(-1).tolist() and len(v.audio_codes) == 5 * 16
    assert np.array_equal(_bits(np.asarray(v.speaker_embedding, dtype=np.float32)), _bits(emb_ref))
    v2 = te.create_voice_file(tmp_path / "ref_st.wav", [9, 8, 7])  # stereo: channel 1 only (:369-373)
    assert v2.audio_codes == v.audio_codes and v2.speaker_embedding == v.speaker_embedding
    # the clone prompt built from that voice, generated greedily, equals the oracle run on the oracle's encoders' outputs
    om = oracle.OracleModel(cfg.model, seed=0, n_ctx=256, n_threads=4)
    try:
        text = list(range(100, 108))
        desc, keep = oracle.make_prompt_desc(text, spk_emb=emb_ref, ref_codes=np.clip(codes_ref.reshape(-1), 0, None).astype(np.int32),
                                             ref_text_ids=[9, 8, 7], lang_id=2055)
        ref, _ = om.generate(om.build_prompt(desc), temperature=0.0, max_steps=6)
    finally:
        om.close()
    d2, keep2 = te._desc(text, v, None)
    res = eng.generate(desc=d2, temperature=0.0, max_steps=6)
    assert np.array_equal(res.codes, ref)
    # generate(text, ref_audio, ref_text): i16 WAV through load_wav, result cached beside the file as TTSC (:275-302)
    out = te.generate(text, tmp_path / "ref_i16.wav", [9, 8, 7])
    assert len(out.samples) == 6 * 1920
    c2, e2 = api.load_cache(tmp_path / "ref_i16.cache")
    a16 = api.AudioSample.load_wav(tmp_path / "ref_i16.wav").samples
    assert c2 == oracle.audio_encode(ccfg, 0, a16)[0].reshape(-1).tolist() and len(e2) == cfg.model.d_embed
    out2 = te.generate(text, tmp_path / "ref_i16.wav", [9, 8, 7])  # second call is served from the cache
    assert np.array_equal(np.asarray(out2.samples), np.asarray(out.samples))
    with pytest.raises(Exception, match="24000Hz"):
        with wave.open(str(tmp_path / "bad.wav"), "wb") as w:
            w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000); w.writeframes(b"\0\0" * 400)
        te.create_voice_file(tmp_path / "bad.wav", [1])


def test_full_shape_three_second_clip(oracle):
    """The family's full encoder shapes on a 3 s reference clip (BASELINE config 5's clone variant): 72 000 samples ->
    38 code frames x 16, 282 mel frames -> 2048-d embedding; equal to the oracle, and timed."""
    import time
    from q3tts import _abi, native
    cfg = _abi.tiny_config(max_batch=1, n_ctx=128, with_vocoder=0)
    eng = native.NativeEngine(cfg)
    try:
        ccfg = _abi.CloneConfig()
        eng.lib.q3tts_clone_default_config(ccfg)
        ref_cfg = _abi.full_clone_config_py()
        assert bytes(ccfg) == bytes(ref_cfg)
        eng.clone_init(ccfg)
        a = _clip(72000, 11)
        codes = eng.audio_encode(a)
        emb = eng.speaker_encode(a)
        assert codes.shape == (38, 16) and emb.shape == (2048,)
        t0 = time.perf_counter()
        for _ in range(3):
            eng.audio_encode(a); eng.speaker_encode(a)
        ms = (time.perf_counter() - t0) / 3 * 1e3
        print(f"clone front-end, 3 s clip, full shape: {ms:.1f} ms (audio encoder + mel + speaker encoder, host to host)")
        codes_ref, _ = oracle.audio_encode(ccfg, 0, a)
        assert np.array_equal(codes, codes_ref.astype(np.int64))
        assert np.array_equal(_bits(emb), _bits(oracle.speaker_encode(ccfg, 0, eng.mel(a))))
    finally:
        eng.close()


def test_text_in_wav_out_from_a_model_directory(oracle, tmp_path):
    """TtsEngine::new(model_dir, quant) + generate_with_voice(text, speaker) as a reference user writes it
    (README.md:117-169): model_dir/gguf_q8_0/{talker,predictor,assets}.gguf, model_dir/tokenizer/tokenizer.json,
    model_dir/preset_speakers/*.json -> text in, WAV out. Ids equal the oracle's on the crate's token ids."""
    from test_tokenizer_cpu import _train
    from q3tts import _abi, api
    cfg = _abi.tiny_config(max_batch=2, n_ctx=128, with_vocoder=1)
    root = tmp_path / "models"
    (root / "gguf_q8_0").mkdir(parents=True)
    (root / "tokenizer").mkdir()
    (root / "preset_speakers").mkdir()
    oracle.write_model_dir(str(root / "gguf_q8_0"), cfg.model, 0, matrix_type=30, assets="gguf", with_text=True)
    hf, _ = _train(420, False, root / "tokenizer")
    spk = ((np.arange(cfg.model.d_embed) % 11 - 5) * 0.0625).astype(np.float32)
    api.VoiceFile.new("", [], spk.tolist()).with_metadata(name="Vivian").save(root / "preset_speakers" / "vivian.json")
    te = api.TtsEngine.new(str(root), "q8_0", config=cfg)
    try:
        assert te.cfg.weights_path == str(root / "gguf_q8_0").encode() and te.tokenizer is not None
        te.set_sampler_config(api.SamplerConfig(0.0, 40, 0.9, 1))
        te.set_max_steps(5)
        text = "你好，world! it's 2024."
        audio = te.generate_with_voice(text, te.get_speaker("vivian"))
        assert len(audio.samples) == 5 * 1920
        audio.save_wav(tmp_path / "out.wav")
        ids = hf.encode(text, add_special_tokens=False).ids
        om = oracle.OracleModel(cfg.model, seed=0, n_ctx=128, n_threads=4)
        try:
            desc, keep = oracle.make_prompt_desc(ids, spk_emb=spk)
            ref, _ = om.generate(om.build_prompt(desc), temperature=0.0, max_steps=5)
        finally:
            om.close()
        d2, keep2 = te._desc(text, te.get_speaker("vivian"), None)
        assert np.array_equal(te._native.generate(desc=d2, temperature=0.0, max_steps=5).codes, ref)
    finally:
        te.close()
