"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle, bit-exact.

Every comparison here is `==` on raw bits: the device kernels and the oracle follow the same canonical
summation orders, so there is no tolerance to state.
"""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def _rand(rng, shape, scale=1.0):
    return (rng.standard_normal(shape) * scale).astype(np.float32)


def _bf16_bits(a):
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = u + 0x7FFF + ((u >> 16) & 1)
    return (u >> 16).astype(np.uint16)


@pytest.fixture(scope="module")
def native():
    from q3tts import native
    return native


def _oracle_gemm(O, x, w, norm_w, eps, bias, epi, y_in=None):
    L = O.lib()
    B, K = x.shape
    N = w.shape[0]
    ny = N // 2 if epi == 2 else N
    y = np.zeros((B, ny if epi == 2 else N), dtype=np.float32) if y_in is None else y_in.copy()
    if epi == 3:
        y = np.zeros((B, N), dtype=np.float32)
    keys = np.zeros(B, dtype=np.uint64)
    L.q3o_gemm_exact(O.ptr(x, O.f32p), B, K, w.ctypes.data_as(C.POINTER(C.c_uint16)), N,
                     None if norm_w is None else O.ptr(norm_w, O.f32p), eps, None if bias is None else O.ptr(bias, O.f32p), epi,
                     O.ptr(y, O.f32p), keys.ctypes.data_as(C.POINTER(C.c_uint64)))
    return y, keys


@pytest.mark.parametrize("B,K,N", [(1, 512, 16), (1, 2048, 256), (3, 512, 48), (6, 2048, 64), (7, 2048, 64), (16, 1024, 64), (17, 2048, 32), (33, 1536, 64),
                                   (64, 2048, 128), (70, 512, 32), (2, 6144, 64), (3, 6144, 32), (40, 1024, 12288), (64, 3072, 8192), (31, 2048, 96),
                                   (64, 6144, 2048), (64, 2048, 2048), (50, 1024, 2048), (200, 2048, 2048), (300, 2048, 4096), (257, 1024, 6144)])
def test_gemm_exact_store_bias(oracle, native, B, K, N):
    rng = np.random.default_rng(B * 1000 + K + N)
    x = _rand(rng, (B, K))
    w = _bf16_bits(_rand(rng, (N, K), 0.02))
    bias = _rand(rng, (N,), 0.1)
    y_ref, _ = _oracle_gemm(oracle, x, w, None, 0.0, bias, 0)
    y, _, _ = native.k_gemm_exact(x, w, bias=bias, epilogue=0)
    assert np.array_equal(_bits(y), _bits(y_ref))


# fused RMSNorm through every launcher branch: small path, ring RT = 1/2/4 x NT = 1/2/3, unrolled and
# runtime-K instances (K = 512/1024/2048 vs 1536/3072), ragged last row tile, SwiGLU / argmax epilogues below
@pytest.mark.parametrize("B,K,N", [(1, 2048, 64), (5, 1024, 32), (12, 1024, 32), (13, 1024, 32), (40, 512, 64), (64, 2048, 12288),
                                   (64, 1024, 6144), (64, 1024, 4096), (64, 2048, 4096), (64, 2048, 3072), (32, 1024, 6144), (17, 1024, 96),
                                   (128, 1024, 6144), (100, 2048, 4096), (37, 1536, 4096), (64, 3072, 6144), (200, 512, 48), (64, 1024, 2048), (320, 2048, 4096), (1000, 1024, 4096)])
def test_gemm_exact_norm_prologue(oracle, native, B, K, N):
    rng = np.random.default_rng(7 + B)
    x = _rand(rng, (B, K), 3.0)
    w = _bf16_bits(_rand(rng, (N, K), 0.02))
    nw = (1.0 + _rand(rng, (K,), 0.05)).astype(np.float32)
    y_ref, _ = _oracle_gemm(oracle, x, w, nw, 1e-6, None, 0)
    y, _, _ = native.k_gemm_exact(x, w, norm_w=nw, eps=1e-6, epilogue=0)
    assert np.array_equal(_bits(y), _bits(y_ref))


def test_gemm_exact_residual_swiglu_argmax(oracle, native):
    rng = np.random.default_rng(11)
    B, K, N = 9, 1024, 96  # small-batch kernel
    x = _rand(rng, (B, K))
    w = _bf16_bits(_rand(rng, (N, K), 0.05))
    y0 = _rand(rng, (B, N))
    y_ref, _ = _oracle_gemm(oracle, x, w, None, 0.0, None, 1, y_in=y0)
    y, _, _ = native.k_gemm_exact(x, w, epilogue=1, y_in=y0)
    assert np.array_equal(_bits(y), _bits(y_ref))
    y_ref, _ = _oracle_gemm(oracle, x, w, None, 0.0, None, 2)
    y, _, _ = native.k_gemm_exact(x, w, epilogue=2)
    assert np.array_equal(_bits(y), _bits(y_ref))
    _, k_ref = _oracle_gemm(oracle, x, w, None, 0.0, None, 3)
    _, k, _ = native.k_gemm_exact(x, w, epilogue=3)
    assert np.array_equal(k, k_ref)
    # ties resolve to the smallest index (first max, strict >): duplicate weight rows
    w2 = w.copy(); w2[40] = w2[7]; w2[90] = w2[7]
    _, k_ref = _oracle_gemm(oracle, x, w2, None, 0.0, None, 3)
    _, k, _ = native.k_gemm_exact(x, w2, epilogue=3)
    assert np.array_equal(k, k_ref)


@pytest.mark.parametrize("n_rows,pos0,Hq,Hkv", [(1, 0, 2, 1), (5, 0, 4, 2), (3, 70, 4, 2), (2, 300, 16, 8), (1, 1000, 4, 4)])
def test_attention_exact(oracle, native, n_rows, pos0, Hq, Hkv):
    # the hook starts from an empty cache, so rows before pos0 are zero keys in both implementations
    rng = np.random.default_rng(100 + pos0 + Hq)
    hd = 128
    total = pos0 + n_rows
    qkv = _rand(rng, (total, (Hq + 2 * Hkv) * hd))
    qn = (1.0 + _rand(rng, (hd,), 0.05)).astype(np.float32)
    kn = (1.0 + _rand(rng, (hd,), 0.05)).astype(np.float32)
    sec = np.array([24, 20, 20, 0], dtype=np.int32)
    L = oracle.lib()
    ref = np.zeros((total, Hq * hd), dtype=np.float32)
    L.q3o_attention(oracle.ptr(qkv, oracle.f32p), total, 0, Hq, Hkv, hd, oracle.ptr(qn, oracle.f32p), oracle.ptr(kn, oracle.f32p), 1e-6,
                    1e6, oracle.ptr(sec, oracle.i32p), oracle.ptr(ref, oracle.f32p))
    out = native.k_attention(qkv, 0, Hq, Hkv, hd, qn, kn, 1e-6, 1e6, sec)
    assert np.array_equal(_bits(out), _bits(ref))


def test_sampler_matches_oracle(oracle, native):
    rng = np.random.default_rng(5)
    n, ld, limit = 64, 3072, 2160
    lg = _rand(rng, (n, ld), 2.0)
    lg[3, 100] = lg[3, 50] = lg[3].max() + 1.0  # exact tie at the top: stable order keeps index 50 first
    r = rng.random(n).astype(np.float32)
    L = oracle.lib()
    for (T, k, p) in [(0.0, 40, 0.9), (0.7, 40, 0.9), (1.0, 0, 1.0), (0.5, 5, 0.5), (0.7, -1, 0.95), (1.3, 3000, 0.3)]:
        ref = np.array([L.q3o_sample(oracle.ptr(lg[i], oracle.f32p), limit, T, k, p, float(r[i])) for i in range(n)], dtype=np.int32)
        out = native.k_sample(lg, limit, T, k, p, r)
        assert np.array_equal(out, ref), (T, k, p)


def test_rng_stream(oracle, native):
    L = oracle.lib()
    for seed in (0, 42, 2**63 + 12345):
        ref = np.zeros(200, dtype=np.float32)
        L.q3o_rng_f32(seed, 200, oracle.ptr(ref, oracle.f32p))
        assert np.array_equal(native.k_rng_f32(seed, 200), ref)


@pytest.fixture(scope="module")
def tiny(oracle):
    from q3tts import _abi, native
    cfg = _abi.tiny_config(max_batch=4, n_ctx=256, with_vocoder=0)
    eng = native.NativeEngine(cfg)
    om = oracle.OracleModel(cfg.model, seed=0, n_ctx=256, n_threads=4)
    yield cfg, eng, om
    eng.close()
    om.close()


def _spk(d):
    return ((np.arange(d) % 13 - 6) * 0.03125).astype(np.float32)


def test_prompt_builder_paths(oracle, tiny):
    cfg, eng, om = tiny
    d = cfg.model.d_embed
    cases = [
        dict(text_ids=np.arange(1000, 1020), spk_emb=_spk(d)),
        dict(text_ids=[151936 + 5, 7], spk_emb=_spk(d), lang_id=-1),             # OOB text id -> fallback pattern; NOTHINK block
        dict(text_ids=[1, 2, 3], spk_id=3065, instruct_ids=[10, 11, 12]),          # preset id row (OOB for codec0 here -> zeros)
        dict(text_ids=np.arange(40), spk_emb=_spk(d), ref_codes=(np.arange(5 * 16) * 37) % 64, ref_text_ids=[9, 8, 7]),
        dict(text_ids=[], spk_emb=_spk(d)),
    ]
    for kw in cases:
        desc, keep = oracle.make_prompt_desc(**kw)
        ref = om.build_prompt(desc)
        out = eng.build_prompt(desc)
        assert out.shape == ref.shape
        assert np.array_equal(_bits(out), _bits(ref)), kw.keys()


def test_talker_prefill_bits(oracle, tiny):
    cfg, eng, om = tiny
    desc, keep = oracle.make_prompt_desc(np.arange(500, 520), spk_emb=_spk(cfg.model.d_embed))
    pe = om.build_prompt(desc)
    h_ref, l_ref = om.talker_prefill(pe)
    h, l = eng.talker_prefill(pe)
    assert np.array_equal(_bits(h), _bits(h_ref))
    assert np.array_equal(_bits(l), _bits(l_ref))


def test_generate_greedy_ids_bit_exact(oracle, tiny):
    cfg, eng, om = tiny
    desc, keep = oracle.make_prompt_desc(np.arange(100, 120), spk_emb=_spk(cfg.model.d_embed))
    pe = om.build_prompt(desc)
    ref, eos_ref = om.generate(pe, temperature=0.0, max_steps=12)
    res = eng.generate(embd=pe, temperature=0.0, max_steps=12)
    assert res.codes.shape == ref.shape and res.hit_eos == eos_ref
    assert np.array_equal(res.codes, ref)
    # same through the device prompt builder
    res2 = eng.generate(desc=desc, temperature=0.0, max_steps=12)
    assert np.array_equal(res2.codes, ref)


@pytest.mark.parametrize("n_text", [60, 250, 600])
def test_generate_long_contexts_through_the_decode_attention(oracle, n_text):
    """The Talker's decode attention (k_attend_gqa2: four waves per (slot, KV head), wave sw owns key blocks sw, sw + 4, ...) across its
    block structure: 71 prompt rows (two blocks on two waves), 261 (every wave one block, the first a second one), 611 (a third trip of
    the value pass, blocks 8 and 9 on waves 0 and 1) — two utterances of different length side by side, greedy ids equal the oracle's."""
    from q3tts import _abi, native
    cfg = _abi.tiny_config(max_batch=2, n_ctx=1024, with_vocoder=0)
    eng = native.NativeEngine(cfg)
    om = oracle.OracleModel(cfg.model, seed=0, n_ctx=1024, n_threads=4)
    try:
        rng = np.random.default_rng(n_text)
        reqs, refs = [], []
        for nt in (n_text, n_text // 2 + 3):
            desc, keep = oracle.make_prompt_desc(rng.integers(0, 151643, size=nt), spk_emb=_spk(cfg.model.d_embed))
            pe = om.build_prompt(desc)
            assert pe.shape[0] == nt + 11
            refs.append(om.generate(pe, temperature=0.0, max_steps=9, min_frames=9)[0])
            reqs.append(dict(embd=pe, temperature=0.0, max_steps=9, min_frames=9))
        for o, r in zip(eng.generate_batch(reqs), refs):
            assert o.status == 0 and r.shape == (9, 16) and np.array_equal(o.codes, r)
    finally:
        eng.close()
        om.close()


def test_generate_sampled_ids_and_eos_controls(oracle, tiny):
    cfg, eng, om = tiny
    desc, keep = oracle.make_prompt_desc(np.arange(300, 310), spk_emb=_spk(cfg.model.d_embed))
    pe = om.build_prompt(desc)
    for seed in (1, 1234):
        ref, eos_ref = om.generate(pe, temperature=0.7, top_k=40, top_p=0.9, seed=seed, max_steps=10)
        res = eng.generate(embd=pe, temperature=0.7, top_k=40, top_p=0.9, seed=seed, max_steps=10)
        assert np.array_equal(res.codes, ref) and res.hit_eos == eos_ref
    ref, eos_ref = om.generate(pe, temperature=0.9, top_k=0, top_p=1.0, seed=9, max_steps=9, min_frames=9, force_eos_at=6)
    res = eng.generate(embd=pe, temperature=0.9, top_k=0, top_p=1.0, seed=9, max_steps=9, min_frames=9, force_eos_at=6)
    assert eos_ref and res.hit_eos and ref.shape[0] == 6
    assert np.array_equal(res.codes, ref)


def test_batch_is_invariant_and_matches_oracle(oracle, tiny):
    """Continuous batching over 4 slots with 7 mixed-length requests == one-at-a-time == oracle."""
    cfg, eng, om = tiny
    reqs, refs = [], []
    for i in range(7):
        desc, keep = oracle.make_prompt_desc(np.arange(50 * i, 50 * i + 5 + 3 * i), spk_emb=_spk(cfg.model.d_embed))
        pe = om.build_prompt(desc)
        kw = dict(temperature=0.7, top_k=40, top_p=0.9, seed=1000 + i, max_steps=16, min_frames=3 + i, force_eos_at=3 + i)
        refs.append(om.generate(pe, **kw)[0])
        reqs.append(dict(embd=pe, **kw))
    outs = eng.generate_batch(reqs)
    for i, (o, r) in enumerate(zip(outs, refs)):
        assert o.status == 0 and np.array_equal(o.codes, r), i
    single = eng.generate(**reqs[5])
    assert np.array_equal(single.codes, refs[5])


def test_errors_are_loud(tiny):
    from q3tts import _abi
    cfg, eng, om = tiny
    with pytest.raises(_abi.Q3Error):
        eng.generate(embd=np.zeros((cfg.n_ctx - 2, cfg.model.d_embed), dtype=np.float32), max_steps=16)  # prompt + steps > n_ctx
    with pytest.raises(_abi.Q3Error):
        eng.generate(embd=np.zeros((4, cfg.model.d_embed), dtype=np.float32), max_steps=4, want_pcm=1)  # no vocoder in this engine (with_vocoder=0)
    bad = _abi.tiny_config()
    bad.model.t_head_dim = 64
    from q3tts import native
    with pytest.raises(_abi.Q3Error):
        native.NativeEngine(bad)


# ---- vocoder (V1-V6): PCM within an RMS tolerance of the CPU restatement ------------------------------------------
PCM_RMS_TOL = 2e-3  # of full scale (+-1.0), small vocoder shapes; bf16 MFMA vs scalar f32 accumulation order + libm vs device sin/erf/exp
# The full shape is deeper and wider (8 transformer layers, 4 decoder blocks from 1536 channels): each bf16-operand GEMM adds its
# 2^-9 relative rounding, so the device (bf16 operands, MFMA accumulation) sits further from BOTH oracles — the one that rounds GEMM
# inputs to bf16 like the device, and the plain-f32 one (the reference's ORT CPU arithmetic up to summation order). Measured on
# MI355X, 8 frames of the synthetic model: 2.4-2.6e-3 / see the printed values; signal RMS is ~0.23, i.e. ~39 dB below the signal.
# The error budget (tests/test_vocoder_family_cpu.py::test_full_shape_pcm_error_budget_of_the_bf16_operand_rounding): every stage group
# contributes 0.4-1.2e-3 and they add in quadrature to 2.4e-3; no single stage — in particular not the HBM-bound last blocks, where wider
# operands would be free — can be fixed to reach 2e-3, so the full-shape bound is 3.5e-3 (was 5e-3 in round 2).
PCM_RMS_TOL_FULL = 3.5e-3


@pytest.fixture(scope="module")
def tiny_voc(oracle):
    from q3tts import _abi, native
    cfg = _abi.tiny_config(max_batch=4, n_ctx=256, with_vocoder=1)
    eng = native.NativeEngine(cfg)
    L = oracle.lib()
    v = L.q3o_vocoder_create(C.byref(cfg.vocoder), 0, 4)
    yield cfg, eng, v
    eng.close()
    L.q3o_vocoder_destroy(v)


def _oracle_pcm(oracle, v, codes, spf=1920):
    L = oracle.lib()
    L.q3o_vocoder_reset(v)
    pcm = np.zeros(codes.shape[0] * spf + 64, dtype=np.float32)
    n = L.q3o_vocoder_decode(v, oracle.ptr(codes, oracle.i32p), codes.shape[0], 1, oracle.ptr(pcm, oracle.f32p), pcm.size)
    return pcm[:n].copy()


@pytest.mark.parametrize("n_frames", [1, 4, 7, 13])
def test_vocoder_pcm_vs_oracle(oracle, tiny_voc, n_frames):
    cfg, eng, v = tiny_voc
    rng = np.random.default_rng(n_frames)
    codes = rng.integers(0, cfg.vocoder.codebook_size, size=(n_frames, 16)).astype(np.int32)
    ref = _oracle_pcm(oracle, v, codes)
    out = eng.vocoder(codes)
    assert out.shape == ref.shape == (n_frames * 1920,)
    rms = float(np.sqrt(np.mean((out - ref) ** 2)))
    assert rms <= PCM_RMS_TOL, rms
    assert np.abs(out).max() <= 1.0


def test_vocoder_streaming_equals_one_shot(tiny_voc):
    """Causal convs + resident state: any chunking gives the same PCM as one call (bit for bit on the device)."""
    cfg, eng, v = tiny_voc
    codes = np.random.default_rng(3).integers(0, cfg.vocoder.codebook_size, size=(11, 16)).astype(np.int32)
    one = eng.vocoder(codes, chunk_frames=0)
    for ch in (1, 3, 4):
        assert np.array_equal(eng.vocoder(codes, chunk_frames=ch), one), ch


def test_vocoder_attention_kernels_agree(tiny_voc):
    """The sliding-window attention with the (slot, head) window staged in LDS (k_voc_attn_lds) against the row-walking kernel it replaces
    (Q3TTS_VOC_ATTN_OLD=1): same chains in the same order, so the PCM is the same bits — one-shot and chunked (1, 3 and 4 tokens per call,
    call positions before and after the sliding window has filled)."""
    import os
    cfg, eng, v = tiny_voc
    codes = np.random.default_rng(12).integers(0, cfg.vocoder.codebook_size, size=(min(80, cfg.max_steps_cap), 16)).astype(np.int32)
    new = [eng.vocoder(codes, chunk_frames=ch) for ch in (0, 1, 3, 4)]
    os.environ["Q3TTS_VOC_ATTN_OLD"] = "1"
    try:
        old = [eng.vocoder(codes, chunk_frames=ch) for ch in (0, 1, 3, 4)]
    finally:
        del os.environ["Q3TTS_VOC_ATTN_OLD"]
    for a, b in zip(new, old):
        assert np.array_equal(a, b) and np.array_equal(a, new[0])


def test_vocoder_clamps_out_of_range_codes(oracle, tiny_voc):
    """Codes outside [0, codebook_size) are clamped like the reference's vocoder thread (src/tts/engine.rs:515-519)."""
    cfg, eng, v = tiny_voc
    codes = np.random.default_rng(4).integers(0, cfg.vocoder.codebook_size, size=(4, 16)).astype(np.int32)
    wild = codes.copy(); wild[0, 0] = 2150; wild[1, 3] = -7
    clamped = np.clip(wild, 0, cfg.vocoder.codebook_size - 1)
    assert np.array_equal(eng.vocoder(wild), eng.vocoder(clamped))


def test_vocoder_narrow_block_kernels(oracle):
    """A vocoder shaped so the narrow-block kernels run (fused residual units at C = 192 / 96, the 96-wide LDS tile,
    the small-M ring): PCM vs the oracle, chunked == one-shot, and the un-fused path gives the same bits."""
    import os
    from q3tts import _abi, native
    cfg = _abi.tiny_config(max_batch=2, n_ctx=128, with_vocoder=1)
    vc = cfg.vocoder
    vc.decoder_dim, vc.n_dec_blocks = 768, 3
    for i, r in enumerate((8, 5, 3)):
        vc.dec_rates[i] = r
    spf = 2 * 2 * 8 * 5 * 3
    L = oracle.lib()
    v = L.q3o_vocoder_create(C.byref(vc), 0, 4)
    eng = native.NativeEngine(cfg)
    try:
        codes = np.random.default_rng(21).integers(0, vc.codebook_size, size=(9, 16)).astype(np.int32)
        ref = _oracle_pcm(oracle, v, codes, spf=spf)
        one = eng.vocoder(codes)
        assert one.shape == ref.shape == (9 * spf,)
        assert float(np.sqrt(np.mean((one - ref) ** 2))) <= PCM_RMS_TOL
        for ch in (1, 4):
            assert np.array_equal(eng.vocoder(codes, chunk_frames=ch), one), ch
        os.environ["Q3TTS_VOC_NOFUSE"] = "1"   # conv k7 -> snake -> conv k1 as separate GEMM launches
        try:
            assert np.array_equal(eng.vocoder(codes), one)
        finally:
            del os.environ["Q3TTS_VOC_NOFUSE"]
        os.environ["Q3TTS_VOC_NORING"] = "1"   # the register-staged GEMM instead of the LDS-DMA ring: same K-step order, same bits
        try:
            assert np.array_equal(eng.vocoder(codes), one)
        finally:
            del os.environ["Q3TTS_VOC_NORING"]
        os.environ["Q3TTS_VOC_TAP_MIN"] = "1"   # the 384-channel 7-tap convolutions on k_vconv_tap (a batch of 64 takes it by itself): same bits
        try:
            for ch in (0, 3):
                assert np.array_equal(eng.vocoder(codes, chunk_frames=ch), one), ch
        finally:
            del os.environ["Q3TTS_VOC_TAP_MIN"]
        os.environ["Q3TTS_VOC_POLITE"] = "1"   # one workgroup per CU (81 KiB of LDS declared), 4-wave units: what runs beside the decoder; same bits
        try:
            assert np.array_equal(eng.vocoder(codes), one)
            os.environ["Q3TTS_VOC_TAP_MIN"] = "1"   # ... with the 4-wave form of k_vconv_tap
            assert np.array_equal(eng.vocoder(codes), one)
            assert np.array_equal(eng.vocoder(codes, chunk_frames=3), one)
        finally:
            del os.environ["Q3TTS_VOC_POLITE"]
            os.environ.pop("Q3TTS_VOC_TAP_MIN", None)
    finally:
        eng.close()
        L.q3o_vocoder_destroy(v)


def test_slot_reuse_starts_from_a_clean_vocoder_state(oracle, tiny_voc):
    """The same request on the same slot, before and after a different utterance used that slot: identical PCM. (The vocoder's conv
    histories are zeroed by one kernel at admission; a reset that does nothing shows up here as a different first chunk.)"""
    cfg, eng, v = tiny_voc
    om = oracle.OracleModel(cfg.model, seed=0, n_ctx=256, n_threads=4)

    def req(seed, n_ids, frames):
        desc, keep = oracle.make_prompt_desc(np.arange(3, 3 + n_ids), spk_emb=_spk(cfg.model.d_embed))
        return dict(embd=om.build_prompt(desc), want_pcm=1, temperature=0.7, top_k=40, top_p=0.9, seed=seed, max_steps=frames, min_frames=frames, force_eos_at=frames)

    a, b = req(11, 6, 9), req(12, 9, 14)
    first = eng.generate_batch([a])[0]
    other = eng.generate_batch([b])[0]
    again = eng.generate_batch([a])[0]
    assert first.status == other.status == again.status == 0
    assert np.array_equal(first.codes, again.codes) and np.array_equal(first.pcm, again.pcm)
    ref = _oracle_pcm(oracle, v, np.clip(first.codes, 0, cfg.vocoder.codebook_size - 1).astype(np.int32))
    assert float(np.sqrt(np.mean((again.pcm - ref) ** 2))) <= PCM_RMS_TOL
    om.close()


def test_batch_with_pcm_crosses_row_buckets(oracle, tiny_voc):
    """Mixed lengths on 4 slots with the vocoder on: rows are re-packed 4 -> 2 -> 1 mid-utterance, slots are re-used."""
    cfg, eng, v = tiny_voc
    om = oracle.OracleModel(cfg.model, seed=0, n_ctx=256, n_threads=4)
    reqs, refs = [], []
    for i in range(7):
        desc, keep = oracle.make_prompt_desc(np.arange(7 * i, 7 * i + 5 + 2 * i), spk_emb=_spk(cfg.model.d_embed))
        pe = om.build_prompt(desc)
        kw = dict(temperature=0.7, top_k=40, top_p=0.9, seed=500 + i, max_steps=24, min_frames=3 + 3 * i, force_eos_at=3 + 3 * i)
        refs.append(om.generate(pe, **kw)[0])
        reqs.append(dict(embd=pe, want_pcm=1, **kw))
    outs = eng.generate_batch(reqs)
    for i, (o, r) in enumerate(zip(outs, refs)):
        assert o.status == 0 and np.array_equal(o.codes, r), i
        ref_pcm = _oracle_pcm(oracle, v, np.clip(r, 0, cfg.vocoder.codebook_size - 1).astype(np.int32))
        assert o.pcm.shape == ref_pcm.shape and float(np.sqrt(np.mean((o.pcm - ref_pcm) ** 2))) <= PCM_RMS_TOL, i
    om.close()


def test_end_to_end_pcm_and_batch(oracle, tiny_voc):
    """generate with want_pcm: ids equal the oracle, PCM within tolerance, batched == single."""
    cfg, eng, v = tiny_voc
    om = oracle.OracleModel(cfg.model, seed=0, n_ctx=256, n_threads=4)
    reqs, refs = [], []
    for i in range(6):
        desc, keep = oracle.make_prompt_desc(np.arange(10 * i, 10 * i + 6 + i), spk_emb=_spk(cfg.model.d_embed))
        pe = om.build_prompt(desc)
        kw = dict(temperature=0.7, seed=77 + i, max_steps=16, min_frames=2 + 2 * i, force_eos_at=2 + 2 * i)
        refs.append(om.generate(pe, **kw)[0])
        reqs.append(dict(embd=pe, want_pcm=1, **kw))
    outs = eng.generate_batch(reqs)
    for i, (o, r) in enumerate(zip(outs, refs)):
        assert np.array_equal(o.codes, r), i
        ref_pcm = _oracle_pcm(oracle, v, np.clip(r, 0, cfg.vocoder.codebook_size - 1).astype(np.int32))
        assert o.pcm.shape == ref_pcm.shape
        assert float(np.sqrt(np.mean((o.pcm - ref_pcm) ** 2))) <= PCM_RMS_TOL, i
    single = eng.generate(**reqs[4])
    assert np.array_equal(single.pcm, outs[4].pcm)
    assert single.first_chunk_ms > 0
    om.close()


def test_streaming_api_chunks(oracle, tiny_voc):
    """q3tts_stream_*: 4-frame chunks (src/tts/engine.rs:507-541) concatenate to exactly the non-streaming PCM."""
    from q3tts import native
    cfg, eng, v = tiny_voc
    desc, keep = oracle.make_prompt_desc(np.arange(40, 52), spk_emb=_spk(cfg.model.d_embed))
    kw = dict(desc=desc, temperature=0.7, seed=5, max_steps=16, min_frames=10, force_eos_at=10)
    whole = eng.generate(want_pcm=1, **kw)
    chunks = list(native.stream_chunks(eng, want_pcm=1, **kw))
    assert [c.size for c, _ in chunks] == [4 * 1920, 4 * 1920, 2 * 1920] and [f for _, f in chunks] == [False, False, True]
    assert np.array_equal(np.concatenate([c for c, _ in chunks]), whole.pcm)
    assert np.array_equal(eng.last_stream_result.codes, whole.codes)


def _oracle_pcm_by_chunk_plan(oracle, v, codes, lookahead, spf=1920):
    """What the reference's vocoder thread delivers for an utterance (src/tts/engine.rs:507-541): the calls of q3o_chunk_plan — 4-frame
    chunks, a final call with is_last only when frames are left over — fed to q3o_vocoder_decode one by one; returns the per-call PCM."""
    L = oracle.lib()
    n = codes.shape[0]
    cf, cl = np.zeros(n + 2, dtype=np.int32), np.zeros(n + 2, dtype=np.int32)
    k = L.q3o_chunk_plan(n, oracle.ptr(cf, oracle.i32p), oracle.ptr(cl, oracle.i32p), cf.size)
    L.q3o_vocoder_reset(v)
    parts, f = [], 0
    for i in range(k):
        buf = np.zeros((int(cf[i]) + lookahead) * spf + 64, dtype=np.float32)
        m = L.q3o_vocoder_decode(v, oracle.ptr(codes[f:f + cf[i]].copy(), oracle.i32p), int(cf[i]), int(cl[i]), oracle.ptr(buf, oracle.f32p), buf.size)
        parts.append(buf[:m].copy())
        f += int(cf[i])
    assert f == n
    return parts


def test_vocoder_lookahead_on_the_device(oracle):
    """V4 with lookahead_frames = 2 (src/models/onnx.rs:364-366: the vocoder withholds a tail until is_last; src/tts/engine.rs:510-536:
    is_last is only ever sent with a non-empty final buffer, so an utterance of n_frames % 4 == 0 keeps its tail for good). One-shot,
    4-frame streaming and whole utterances through generate_batch against q3o_vocoder_decode driven by q3o_chunk_plan: the same number
    of samples per call as the reference's thread would see, PCM within tolerance; vocoder_flush_tail = 1 delivers every frame."""
    from q3tts import _abi, native
    LA = 2
    cfg = _abi.tiny_config(max_batch=2, n_ctx=256, with_vocoder=1)
    cfg.vocoder.lookahead_frames = LA
    L = oracle.lib()
    v = L.q3o_vocoder_create(C.byref(cfg.vocoder), 0, 4)
    eng = native.NativeEngine(cfg)
    cfg_f = _abi.tiny_config(max_batch=2, n_ctx=256, with_vocoder=1)
    cfg_f.vocoder.lookahead_frames = LA
    cfg_f.vocoder_flush_tail = 1
    eng_f = native.NativeEngine(cfg_f)
    om = oracle.OracleModel(cfg.model, seed=0, n_ctx=256, n_threads=4)
    try:
        # the hook with an explicit is_last at the end: every frame, in any chunking (the withheld tail is only a delay)
        codes = np.random.default_rng(8).integers(0, cfg.vocoder.codebook_size, size=(9, 16)).astype(np.int32)
        ref_all = _oracle_pcm(oracle, v, codes)
        one = eng.vocoder(codes)
        assert one.shape == ref_all.shape == (9 * 1920,) and float(np.sqrt(np.mean((one - ref_all) ** 2))) <= PCM_RMS_TOL
        assert np.array_equal(eng.vocoder(codes, chunk_frames=4), one)
        desc, keep = oracle.make_prompt_desc(np.arange(60, 70), spk_emb=_spk(cfg.model.d_embed))
        pe = om.build_prompt(desc)
        reqs, want = [], []
        for n in (8, 6, 3, 4, 12, 13):   # n % 4 == 0: the reference never flushes -> n - 2 frames of audio
            kw = dict(temperature=0.7, top_k=40, top_p=0.9, seed=40 + n, max_steps=16, min_frames=n, force_eos_at=n)
            ref_codes = om.generate(pe, **kw)[0]
            assert ref_codes.shape[0] == n
            parts = _oracle_pcm_by_chunk_plan(oracle, v, np.clip(ref_codes, 0, cfg.vocoder.codebook_size - 1).astype(np.int32), LA)
            want.append((n, ref_codes, parts))
            reqs.append(dict(embd=pe, want_pcm=1, **kw))
        outs = eng.generate_batch(reqs)      # 6 utterances over 2 slots: batched 4-frame vocoder calls, slot reuse
        outs_f = eng_f.generate_batch(reqs)
        for o, of, (n, ref_codes, parts) in zip(outs, outs_f, want):
            ref = np.concatenate(parts)
            kept = n - LA if n % 4 == 0 else n
            assert o.status == 0 and np.array_equal(o.codes, ref_codes), n
            assert ref.size == kept * 1920 and o.pcm.size == ref.size, (n, o.pcm.size, ref.size)
            assert float(np.sqrt(np.mean((o.pcm - ref) ** 2))) <= PCM_RMS_TOL, n
            assert of.status == 0 and np.array_equal(of.codes, ref_codes) and of.pcm.size == n * 1920, n
            assert np.array_equal(of.pcm[:o.pcm.size], o.pcm), n   # the flush only adds the tail
            full = _oracle_pcm(oracle, v, np.clip(ref_codes, 0, cfg.vocoder.codebook_size - 1).astype(np.int32))
            assert float(np.sqrt(np.mean((of.pcm - full) ** 2))) <= PCM_RMS_TOL, n
        # streaming: per-call sample counts equal the reference thread's
        for (n, ref_codes, parts), r in zip(want, reqs):
            chunks = list(native.stream_chunks(eng, **r))
            sizes = [p.size for p in parts if p.size]
            assert [c.size for c, _ in chunks] == sizes, (n, [c.size for c, _ in chunks], sizes)
            got = np.concatenate([c for c, _ in chunks])
            assert float(np.sqrt(np.mean((got - np.concatenate(parts)) ** 2))) <= PCM_RMS_TOL, n
            assert np.array_equal(eng.last_stream_result.codes, ref_codes)
            chunks_f = list(native.stream_chunks(eng_f, **r))
            assert sum(c.size for c, _ in chunks_f) == n * 1920 and chunks_f[-1][1], n
    finally:
        eng.close()
        eng_f.close()
        om.close()
        L.q3o_vocoder_destroy(v)


def test_api_mirror_generate_with_voice(tiny_voc, tmp_path):
    """TtsEngine / VoiceFile / SamplerConfig / AudioSample mirror over the same engine handle (token ids in, WAV out)."""
    from q3tts import api
    cfg, eng, v = tiny_voc
    te = api.TtsEngine.__new__(api.TtsEngine)
    te._native, te.cfg, te.tokenizer, te.speakers, te.max_steps, te.sampler_config = eng, cfg, None, {}, 6, api.SamplerConfig(0.0, 40, 0.9, 7)
    voice = api.VoiceFile.new("", [], _spk(cfg.model.d_embed).tolist())
    a = te.generate_with_voice(list(range(100, 110)), voice)
    assert a.sample_rate == 24000 and a.channels == 1 and len(a.samples) == 6 * 1920
    te.set_language(None)  # no-language control block (NOTHINK variant): a different prompt, hence different audio
    b = te.generate_with_voice(list(range(100, 110)), voice)
    assert len(b.samples) == 6 * 1920 and not np.array_equal(np.asarray(a.samples), np.asarray(b.samples))
    te.set_language(2055)
    assert np.array_equal(np.asarray(te.generate_with_voice(list(range(100, 110)), voice).samples), np.asarray(a.samples))
    a.save_wav(tmp_path / "o.wav")
    assert api.AudioSample.load_wav(tmp_path / "o.wav").samples.size == 6 * 1920
    te.speakers = {"vivian": voice}
    assert te.get_speaker("nobody") is voice  # fallback chain: id -> name -> vivian (src/tts/engine.rs:211-231)
    with pytest.raises(Exception):
        te.generate_with_voice("plain text needs a tokenizer.json", voice)


# ---------------------------------------------------------------------------------------------------------------
# loader row: an engine built from model FILES equals the engine built from the generator
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("matrix_type,assets,with_text", [(30, "gguf", True), (0, "npy", True), (1, "gguf", False)])
def test_engine_from_model_files(oracle, tmp_path, matrix_type, assets, with_text):
    """weights_path = the reference's quant directory (llama.cpp-style talker / predictor GGUFs + qwen3_assets.gguf or
    NPY). The files hold the synthetic model (bf16-exact matrices), so BF16, F32 and F16 containers must all reproduce
    the oracle's ids bit for bit; without a text table every text id follows the out-of-range formula and tts_pad is 0."""
    from q3tts import _abi, native
    cfg = _abi.tiny_config(max_batch=2, n_ctx=128, with_vocoder=0)
    oracle.write_model_dir(str(tmp_path), cfg.model, 0, matrix_type=matrix_type, assets=assets, with_text=with_text)
    cfg.weights_path = str(tmp_path).encode()
    eng = native.NativeEngine(cfg)
    ocfg = _abi.tiny_config(max_batch=2, n_ctx=128, with_vocoder=0)
    if not with_text:
        ocfg.model.text_vocab = 0
    om = oracle.OracleModel(ocfg.model, seed=0, n_ctx=128, n_threads=4)
    try:
        desc, keep = oracle.make_prompt_desc(np.arange(900, 912), spk_emb=_spk(cfg.model.d_embed))
        pe = om.build_prompt(desc)
        assert np.array_equal(_bits(eng.build_prompt(desc)), _bits(pe))
        for kw in (dict(temperature=0.0, max_steps=6), dict(temperature=0.7, top_k=40, top_p=0.9, seed=11, max_steps=6)):
            ref, _ = om.generate(pe, **kw)
            assert np.array_equal(eng.generate(desc=desc, **kw).codes, ref)
    finally:
        eng.close()
        om.close()


@pytest.mark.parametrize("matrix_type", [13, 14])
def test_engine_from_k_quant_files(oracle, tmp_path, matrix_type):
    """The reference's gguf_q5_k_m directory holds Q5_K / Q6_K matrices (src/tts/engine.rs:91-95). An engine created from a K-quant
    container must behave exactly like one created from an F32 container that holds the de-quantised values (tests/_gguf.py, the
    numpy restatement of the block layouts): same prompt rows, same greedy and sampled ids. (The K-quant layouts themselves are
    PARITY UNPINNED: llama.cpp is not in the reference.)"""
    import _gguf as G
    from q3tts import _abi, native
    cfg = _abi.tiny_config(max_batch=2, n_ctx=128, with_vocoder=0)
    dq, df = tmp_path / "kq", tmp_path / "f32"
    oracle.write_model_dir(str(dq), cfg.model, 0, matrix_type=matrix_type)
    oracle.write_model_dir(str(df), cfg.model, 0, matrix_type=G.F32)
    for fname in ("qwen3_tts_talker.gguf", "qwen3_tts_predictor.gguf"):   # the F32 twin holds what the K-quant file decodes to
        tens = G.read(str(dq / fname))
        assert any(ty == matrix_type for _, ty in tens.values())
        G.write(str(df / fname), [(k, v, G.F32) for k, (v, ty) in tens.items()], meta={"general.architecture": "qwen3", "general.alignment": 32})
    engs = []
    try:
        for d in (dq, df):
            c = _abi.tiny_config(max_batch=2, n_ctx=128, with_vocoder=0)
            c.weights_path = str(d).encode()
            engs.append(native.NativeEngine(c))
        desc, keep = oracle.make_prompt_desc(np.arange(900, 912), spk_emb=_spk(cfg.model.d_embed))
        assert np.array_equal(_bits(engs[0].build_prompt(desc)), _bits(engs[1].build_prompt(desc)))
        for kw in (dict(temperature=0.0, max_steps=6), dict(temperature=0.7, top_k=40, top_p=0.9, seed=11, max_steps=6)):
            a, b = engs[0].generate(desc=desc, **kw), engs[1].generate(desc=desc, **kw)
            assert a.codes.shape[0] > 0 and np.array_equal(a.codes, b.codes)
    finally:
        for e in engs:
            e.close()


def test_model_files_errors_are_loud(oracle, tmp_path):
    import _gguf as G
    from q3tts import _abi, native
    cfg = _abi.tiny_config(max_batch=1, n_ctx=128, with_vocoder=0)
    cfg.weights_path = str(tmp_path / "nowhere").encode()
    with pytest.raises(_abi.Q3Error, match="cannot open"):
        native.NativeEngine(cfg)
    oracle.write_model_dir(str(tmp_path), cfg.model, 0, with_text=False)
    t = oracle.synth_transformer_tensors(cfg.model, 0, True)
    bad = dict(t); bad.pop("blk.1.ffn_up.weight")
    G.write(str(tmp_path / "qwen3_tts_talker.gguf"), [(k, v, G.BF16 if v.ndim == 2 else G.F32) for k, v in bad.items()])
    cfg.weights_path = str(tmp_path).encode()
    with pytest.raises(_abi.Q3Error, match="blk.1.ffn_up.weight.*missing"):
        native.NativeEngine(cfg)
    bad = dict(t); bad["blk.0.attn_q.weight"] = bad["blk.0.attn_q.weight"][:, :-32]
    G.write(str(tmp_path / "qwen3_tts_talker.gguf"), [(k, v, G.BF16 if v.ndim == 2 else G.F32) for k, v in bad.items()])
    with pytest.raises(_abi.Q3Error, match="attn_q.weight.*shape"):
        native.NativeEngine(cfg)


# ---------------------------------------------------------------------------------------------------------------
# clone-path front-end (the part pinned in-repo): log-mel on the device vs the oracle
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [0, 255, 256, 5000, 72000])
def test_mel_matches_oracle(oracle, tiny, n):
    cfg, eng, om = tiny
    rng = np.random.default_rng(100 + n)
    t = np.arange(n) / 24000.0
    audio = (0.25 * np.sin(2 * np.pi * 180.0 * t) * (1 + 0.5 * np.sin(2 * np.pi * 3.0 * t)) + 0.02 * rng.standard_normal(n)).astype(np.float32)
    ref = oracle.mel(audio)
    got = eng.mel(audio)
    assert got.shape == ref.shape
    if n >= 256:
        # same DFT / filterbank order on both sides: only the final logf (device vs libm) may differ in the last place
        assert np.abs(got - ref).max() <= 4e-6, float(np.abs(got - ref).max())


def test_probe_mode_is_transparent(oracle, tiny):
    """q3tts_k_probe (bench.py's in-situ kernel timer): eager frame steps with event brackets give the same ids as graph
    replay and report a positive mean duration for the probed launches."""
    cfg, eng, om = tiny
    desc, keep = oracle.make_prompt_desc(np.arange(70, 80), spk_emb=_spk(cfg.model.d_embed))
    reqs = [dict(desc=desc, temperature=0.7, seed=40 + i, max_steps=12, min_frames=8, force_eos_at=8) for i in range(4)]
    ref = [o.codes for o in eng.generate_batch(reqs)]
    eng.probe(True)
    try:
        outs = eng.generate_batch(reqs)
        tm = eng.timings()
    finally:
        eng.probe(False)
    assert all(np.array_equal(o.codes, r) for o, r in zip(outs, ref))
    assert tm.probe_count > 0 and tm.probe_kernel_ms > 0 and tm.probe_empty_ms > 0
    assert all(np.array_equal(o.codes, r) for o, r in zip(eng.generate_batch(reqs), ref))


# ---------------------------------------------------------------------------------------------------------------
# the bf16 MFMA's accumulation arithmetic: hardware vs the oracle's integer restatement
# ---------------------------------------------------------------------------------------------------------------
def test_bf16_mfma_arithmetic_model(oracle):
    """v_mfma_f32_16x16x32_bf16 is not an fmaf chain: per lane group it adds eight truncated products and the truncated
    accumulator in a fixed-point window and rounds once. oracle/q3_oracle_bf16.c::q3o_mfma_bf16_dot32 restates that in integers;
    here fresh seeded cases — narrow and wide exponent spreads, cancelling +-2^E pairs that expose the window, accumulators far
    above and far below the products, chains of two and four instructions — must agree bit for bit on every output."""
    from q3tts import _abi
    lib = _abi.load_library()
    L = oracle.lib()
    L.q3o_mfma_bf16_dot32.argtypes = [C.POINTER(C.c_uint16), C.POINTER(C.c_uint16), C.c_float]
    L.q3o_mfma_bf16_dot32.restype = C.c_float
    rng = np.random.default_rng(20261004)

    def val(shape, s):
        return (np.ldexp(1.0 + rng.integers(0, 128, shape) / 128.0, rng.integers(-s, s + 1, shape)) * rng.choice([-1.0, 1.0], shape)).astype(np.float32)
    n = 24
    total = 0
    for s, cs, chain, cancel in [(2, 2, 1, 0), (8, 8, 1, 0), (20, 30, 1, 0), (6, 40, 1, 0), (30, 4, 1, 0), (4, 4, 1, 14), (4, 4, 1, 25), (4, 6, 1, 33), (8, 8, 2, 0), (5, 5, 4, 21)]:
        A = val((n, chain, 16, 32), s); B = val((n, chain, 32, 16), s)
        Cm = (val((n, 16, 16), cs) * (rng.random((n, 16, 16)) < 0.8)).astype(np.float32)
        if cancel:
            for c in range(n):
                st = int(rng.integers(chain)); k1, k2 = rng.choice(32, 2, replace=False)
                A[c, st, :, k1] = np.ldexp(1.0, cancel // 2); A[c, st, :, k2] = -np.ldexp(1.0, cancel // 2)
                B[c, st, k1, :] = np.ldexp(1.0, cancel - cancel // 2); B[c, st, k2, :] = np.ldexp(1.0, cancel - cancel // 2)
        Ab, Bb = _bf16_bits(A), _bf16_bits(B)
        D = np.zeros_like(Cm)
        assert lib.q3tts_k_mfma_bf16(0, Ab.ctypes.data, Bb.ctypes.data, Cm.ctypes.data, D.ctypes.data, n, chain) == 0
        for c in range(n):
            for i in range(0, 16, 3):
                for j in range(0, 16, 5):
                    acc = float(Cm[c, i, j])
                    for st in range(chain):
                        a = np.ascontiguousarray(Ab[c, st, i, :]); b = np.ascontiguousarray(Bb[c, st, :, j])
                        acc = L.q3o_mfma_bf16_dot32(a.ctypes.data_as(C.POINTER(C.c_uint16)), b.ctypes.data_as(C.POINTER(C.c_uint16)), acc)
                    assert _bits(np.float32(acc)) == _bits(D[c, i, j]), (s, cs, chain, cancel, c, i, j, acc, float(D[c, i, j]))
                    total += 1
    assert total == 10 * n * 6 * 4


# ---------------------------------------------------------------------------------------------------------------
# the decoder's GEMM (csrc/q3_bgemm.hip) against the oracle's restatement (oracle/q3_oracle_bf16.c), every epilogue, every
# tile instance the launcher can pick (rows 1..64 -> RT 1..4, N -> NT 1..3, many rows -> 64-row chunks), K/256 = 1..24 steps
# ---------------------------------------------------------------------------------------------------------------
def _bgemm_case(oracle, native, B, K, N, epi, scaled, seed):
    rng = np.random.default_rng(seed)
    x = _rand(rng, (B, K), 1.5); x[:, :7] *= 300.0; x[:, 100:140] *= 1e-3   # outlier and tiny channels, as activations have
    xb, wb = _bf16_bits(x), _bf16_bits(_rand(rng, (N, K), 0.02))
    d_norm = K
    ssp = (np.abs(_rand(rng, (B, d_norm // 16), 4.0)) + 0.5).astype(np.float32) if scaled else None
    nw_next = (1.0 + _rand(rng, (N,), 0.05)).astype(np.float32) if epi == 1 else None
    y0 = _rand(rng, (B, N), 2.0) if epi == 1 else None
    ref = oracle.bgemm(xb, wb, ssp, d_norm, 1e-6, epi, nw_next, y0)
    got = native.k_bgemm(xb, wb, ssp, d_norm, 1e-6, epi, nw_next, y0)
    return ref, got, xb, wb


@pytest.mark.parametrize("B,K,N", [(64, 2048, 12288), (64, 2048, 4096), (48, 2048, 4096), (33, 1024, 6144), (17, 2048, 2048), (16, 6144, 2048), (1, 2048, 12288),
                                   (1, 1024, 4096), (2, 512, 1536), (5, 256, 16), (64, 3072, 1024), (50, 1024, 96), (300, 2048, 4096), (100, 1024, 6144), (70, 512, 32),
                                   (64, 1024, 3072), (31, 2048, 3072), (40, 512, 1024)])
def test_bgemm_scaled_store_matches_oracle(oracle, native, B, K, N):
    ref, got, xb, wb = _bgemm_case(oracle, native, B, K, N, 0, True, B + K + N)
    assert np.array_equal(_bits(got["y"]), _bits(ref["y"]))


def test_bgemm_is_a_gemm(oracle, native):
    ref, got, xb, wb = _bgemm_case(oracle, native, 33, 2048, 1008, 0, False, 5)
    assert np.array_equal(_bits(got["y"]), _bits(ref["y"]))
    xf = (xb.astype(np.uint32) << 16).view(np.float32).astype(np.float64); wf = (wb.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
    assert np.abs(got["y"] - xf @ wf.T).max() <= 2e-5 * np.abs(xf @ wf.T).max()


@pytest.mark.parametrize("B,K,N", [(64, 2048, 2048), (64, 6144, 2048), (64, 2048, 1024), (64, 3072, 1024), (1, 6144, 2048), (23, 512, 512), (7, 1024, 96), (130, 2048, 2048), (36, 1024, 512)])
def test_bgemm_residual_and_norm_outputs_match_oracle(oracle, native, B, K, N):
    """O / down projections: x += RAW, and the consumer's norm inputs (bf16(x * nw_next), per-tile sums of squares) out of the same epilogue."""
    ref, got, _, _ = _bgemm_case(oracle, native, B, K, N, 1, False, 300 + B + K)
    assert np.array_equal(_bits(got["y"]), _bits(ref["y"]))
    assert np.array_equal(got["yb"], ref["yb"])
    assert np.array_equal(_bits(got["ssp_out"]), _bits(ref["ssp_out"]))


@pytest.mark.parametrize("B,K,N", [(64, 2048, 12288), (64, 1024, 6144), (37, 1024, 6144), (1, 2048, 12288), (128, 1024, 6144), (9, 512, 1024), (64, 512, 1024), (20, 512, 192)])
def test_bgemm_swiglu_matches_oracle(oracle, native, B, K, N):
    ref, got, _, _ = _bgemm_case(oracle, native, B, K, N, 2, True, 100 + B + N)
    assert np.array_equal(got["yb"], ref["yb"])
    assert np.count_nonzero(got["yb"] & 0x7fff) > got["yb"].size // 2


@pytest.mark.parametrize("B,K,N,epi", [(256, 2048, 4096, 0), (300, 2048, 4096, 0), (1000, 1024, 256, 0), (513, 256, 128, 0), (257, 2048, 2048, 1), (640, 6144, 2048, 1),
                                       (300, 512, 128, 1), (256, 2048, 12288, 2), (391, 1024, 6144, 2), (1984, 2048, 4096, 0)])
def test_bgemm_many_rows_kernel_matches_oracle(oracle, native, B, K, N, epi):
    """Prefill-sized launches (>= 256 rows, N % 128 == 0) run k_bgemm_big: 128 x 128 tiles, the whole K in one wave, the 8 K-slices summed
    in the canonical order inside the wave. Same bits as the oracle (and therefore as k_bgemm), ragged last row tiles included."""
    lib = native._abi.load_library()
    assert lib.q3tts_k_bgemm_policy(1) == 0   # (the launcher's own rule wants >= 256 tiles: the oracle would take minutes at such sizes)
    try:
        ref, got, _, _ = _bgemm_case(oracle, native, B, K, N, epi, epi != 1, 900 + B + N + epi)
    finally:
        assert lib.q3tts_k_bgemm_policy(0) == 0
    if epi == 0:
        assert np.array_equal(_bits(got["y"]), _bits(ref["y"]))
    elif epi == 1:
        assert np.array_equal(_bits(got["y"]), _bits(ref["y"]))
        assert np.array_equal(got["yb"], ref["yb"])
        assert np.array_equal(_bits(got["ssp_out"]), _bits(ref["ssp_out"]))
    else:
        assert np.array_equal(got["yb"], ref["yb"])


def _bf16_round_bits(a):
    """RNE to bf16 of finite f32 values, as bits."""
    u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32).astype(np.uint64)
    return ((u + 0x7fff + ((u >> 16) & 1)) >> 16).astype(np.uint16)


@pytest.mark.parametrize("B,K,N,seg,gap", [(64, 1024, 2048, 0, 0), (48, 1024, 1024, 8, 6), (512, 1024, 4096, 16, 6), (40, 4096, 1024, 8, 3), (7, 256, 96, 0, 0)])
def test_bgemm_vocoder_epilogue_extras(oracle, native, B, K, N, seg, gap):
    """bias, LayerScale column scale, GELU, per-slot row segments and the bf16 copy of the residual result (what the vocoder's transformer and
    up-sampling stages ask of k_bgemm): RAW comes from the oracle-checked store epilogue, the rest is element-wise f32 arithmetic restated
    here in numpy (no fused multiply-add on either side: the library is built with -ffp-contract=off)."""
    from scipy.special import erf
    rng = np.random.default_rng(B + K + N)
    xb, wb = _bf16_bits(_rand(rng, (B, K), 1.0)), _bf16_bits(_rand(rng, (N, K), 0.03))
    raw = native.k_bgemm(xb, wb, None, K, 1e-6, 0)["y"]
    ref = oracle.bgemm(xb, wb, None, K, 1e-6, 0, None, None)["y"]
    assert np.array_equal(_bits(raw), _bits(ref))
    bias_n = N // 2 if N % 2 == 0 and N >= 64 else N            # (the ConvTranspose bias repeats every d of its r * d outputs)
    bias = _rand(rng, (bias_n,), 0.05)
    cs = (0.01 + np.abs(_rand(rng, (N,), 0.01))).astype(np.float32)
    y0 = _rand(rng, (B, N), 1.0)
    bcol = bias[np.arange(N) % bias_n][None, :]
    v = (raw + bcol).astype(np.float32)
    # store + bias, into row segments with untouched gap rows in between
    got = native.k_bgemm_voc(xb, wb, 0, bias=bias, seg_rows=seg, gap_rows=gap)
    assert np.array_equal(_bits(got["y"]), _bits(v))
    # residual with column scale + bias, and the bf16 copy the next stage's GEMM reads
    got = native.k_bgemm_voc(xb, wb, 1, bias=bias, col_scale=cs, seg_rows=seg, gap_rows=gap, y0=y0, want_yb=True)
    exp = (y0 + (cs[None, :] * v).astype(np.float32)).astype(np.float32)
    assert np.array_equal(_bits(got["y"]), _bits(exp))
    assert np.array_equal(got["yb"], _bf16_round_bits(exp))
    # GELU -> bf16, the device's expression in f32: (0.5 x) * (1 + erff(x / sqrt 2)). erff and scipy's f32 erf may differ in the last
    # place, which 1 + erf amplifies for negative x: absolute slack 2^-20 |x| on top of half a bf16 step
    got = native.k_bgemm_voc(xb, wb, 4, bias=bias)
    e32 = erf((v * np.float32(0.70710678118654752)).astype(np.float32)).astype(np.float32)
    g32 = ((np.float32(0.5) * v).astype(np.float32) * (np.float32(1.0) + e32).astype(np.float32)).astype(np.float32)
    dev = (got["yb"].astype(np.uint32) << 16).view(np.float32)
    assert np.all(np.abs(dev.astype(np.float64) - g32.astype(np.float64)) <= np.abs(g32) * 2.0 ** -8 + np.abs(v) * 2.0 ** -20)
    assert np.count_nonzero(got["yb"] != _bf16_round_bits(g32)) <= max(4, got["yb"].size // 200)


@pytest.mark.parametrize("B,K,N", [(64, 1024, 2048), (2, 1024, 2048), (33, 512, 64), (64, 2048, 3072)])
def test_bgemm_argmax_matches_oracle(oracle, native, B, K, N):
    ref, got, _, _ = _bgemm_case(oracle, native, B, K, N, 3, True, 7 + B + N)
    assert np.array_equal(got["keys"], ref["keys"])


def test_bgemm_argmax_ties_resolve_to_the_first_index(oracle, native):
    rng = np.random.default_rng(3)
    xb, wb = _bf16_bits(_rand(rng, (9, 512))), _bf16_bits(_rand(rng, (96, 512), 0.05))
    wb[40] = wb[7]; wb[90] = wb[7]
    assert np.array_equal(native.k_bgemm(xb, wb, None, 512, 1e-6, 3)["keys"], oracle.bgemm(xb, wb, None, 512, 1e-6, 3)["keys"])


# ---- Q8_0 weights kept in block form on the device (q3tts_engine_config.talker_q8_0) -----------------------------------
@pytest.mark.parametrize("B,K,N,epi", [(64, 2048, 12288, 2), (64, 2048, 4096, 0), (64, 6144, 2048, 1), (64, 2048, 2048, 1), (1, 2048, 3072, 0), (1, 6144, 2048, 1),
                                       (37, 512, 1024, 2), (48, 2048, 4096, 0), (9, 1024, 96, 3), (130, 512, 512, 1), (300, 2048, 4096, 0), (5, 1536, 32, 0)])
def test_bgemm_q8_matches_oracle(oracle, native, B, K, N, epi):
    """The decoder's GEMM on ggml Q8_0 blocks as stored (f16 d, 32 x int8): per block P = the bf16 MFMA of (x, q) from a zero accumulator,
    t = fmaf(f32(d), P, t) over the blocks of a K slice, the 8 slices added in order — bit for bit the oracle's q3o_bgemm_q8, every epilogue,
    the Talker's shapes at 64 rows, ragged row tiles, many rows (k_bgemm_big does not take Q8: 64-row chunks), blocks of zeros, q = -128."""
    rng = np.random.default_rng(B + K + N + epi)
    x = _rand(rng, (B, K), 1.5); x[:, :7] *= 300.0; x[:, 100:140] *= 1e-3
    w = _rand(rng, (N, K), 0.02); w[3, 64:96] = 0.0; w[5 % N, :32] *= 40.0
    q, d16 = oracle.quantize_q8_0(w)
    q[1, 0] = -128   # valid in a file, never produced by the quantiser
    xb = _bf16_bits(x)
    scaled = epi in (0, 2, 3)
    ssp = (np.abs(_rand(rng, (B, K // 16), 4.0)) + 0.5).astype(np.float32) if scaled else None
    nw_next = (1.0 + _rand(rng, (N,), 0.05)).astype(np.float32) if epi == 1 and N % 32 == 0 else None
    y0 = _rand(rng, (B, N), 2.0) if epi == 1 else None
    ref = oracle.bgemm_q8(xb, q, d16, ssp, K, 1e-6, epi, nw_next, y0)
    got = native.k_bgemm_q8(xb, q, d16, ssp, K, 1e-6, epi, nw_next, y0)
    if epi in (0, 1):
        assert np.array_equal(_bits(got["y"]), _bits(ref["y"]))
    if epi == 1 and nw_next is not None:
        assert np.array_equal(got["yb"], ref["yb"]) and np.array_equal(_bits(got["ssp_out"]), _bits(ref["ssp_out"]))
    if epi == 2:
        assert np.array_equal(got["yb"], ref["yb"]) and np.count_nonzero(got["yb"] & 0x7fff) > got["yb"].size // 2
    if epi == 3:
        assert np.array_equal(got["keys"], ref["keys"])
    if epi == 0 and B == 64:   # and it IS the product with the de-quantised weights, to f32 rounding
        xf = (xb.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
        wd = (q.astype(np.float64).reshape(N, K // 32, 32) * d16.view(np.float16).astype(np.float64)[:, :, None]).reshape(N, K)
        s = np.array([oracle.row_scale(ssp[r], K, 1e-6) for r in range(B)], dtype=np.float64)
        want = (xf @ wd.T) * s[:, None]
        assert np.abs(got["y"] - want).max() <= 3e-5 * np.abs(want).max()


@pytest.mark.parametrize("B,K,N,epi", [(64, 2048, 12288, 2), (64, 2048, 4096, 0), (64, 6144, 2048, 1), (64, 2048, 2048, 1), (1, 2048, 3072, 0), (1, 6144, 2048, 1),
                                       (37, 512, 1024, 2), (48, 2048, 4096, 0), (17, 1024, 256, 2), (130, 512, 512, 1), (300, 2048, 4096, 0), (5, 1536, 32, 0), (200, 1024, 512, 2)])
def test_bgemm_q8a8_matches_oracle(oracle, native, B, K, N, epi):
    """W8A8 on v_mfma_i32_16x16x32_i8 (csrc/q3_bgemm8.hip): activations AND weights as ggml Q8_0 blocks — what llama.cpp multiplies for the
    reference's gguf_q8_0 directory (src/tts/engine.rs:91-95). A block's product = its exact int32 sum x (f32(d_w) * f32(d_x)), blocks added in
    order inside a K slice, the 8 slices in order: bit for bit q3o_bgemm_q8a8 — RAW with the row scale, the residual epilogue whose NEXT operand is
    quantised in the kernel (int8 quants + f16 scales + tile sums of squares), SwiGLU quantised in the kernel; the Talker's shapes at 64 rows,
    ragged row tiles, many rows, blocks of zeros, q = -128 on both sides."""
    rng = np.random.default_rng(7 * B + K + N + epi)
    a = _rand(rng, (B, K), 1.5); a[:, :7] *= 300.0; a[:, 100:140] *= 1e-3; a[B // 2, 64:96] = 0.0
    w = _rand(rng, (N, K), 0.02); w[3, 64:96] = 0.0; w[5 % N, :32] *= 40.0
    aq, ad = oracle.quantize_q8_0(a)
    q, d16 = oracle.quantize_q8_0(w)
    q[1, 0] = -128; aq[0, 1] = -128
    scaled = epi in (0, 2)
    ssp = (np.abs(_rand(rng, (B, K // 16), 4.0)) + 0.5).astype(np.float32) if scaled else None
    nw_next = (1.0 + _rand(rng, (N,), 0.05)).astype(np.float32) if epi == 1 else None
    y0 = _rand(rng, (B, N), 2.0) if epi == 1 else None
    ref = oracle.bgemm_q8a8(aq, ad, q, d16, ssp, K, 1e-6, epi, nw_next, y0)
    got = native.k_bgemm_q8a8(aq, ad, q, d16, ssp, K, 1e-6, epi, nw_next, y0)
    if epi in (0, 1):
        assert np.array_equal(_bits(got["y"]), _bits(ref["y"]))
    if epi == 1:
        assert np.array_equal(_bits(got["ssp_out"]), _bits(ref["ssp_out"]))
    if epi in (1, 2):
        assert np.array_equal(got["yd"], ref["yd"]) and np.array_equal(got["yq"], ref["yq"])
        assert np.count_nonzero(got["yq"]) > got["yq"].size // 2
    if epi == 0 and B == 64:   # and it IS the product of the de-quantised operands, to f32 rounding
        ad_f = ad.view(np.float16).astype(np.float64); wd_f = d16.view(np.float16).astype(np.float64)
        af = (aq.astype(np.float64).reshape(B, K // 32, 32) * ad_f[:, :, None]).reshape(B, K)
        wf = (q.astype(np.float64).reshape(N, K // 32, 32) * wd_f[:, :, None]).reshape(N, K)
        s = np.array([oracle.row_scale(ssp[r], K, 1e-6) for r in range(B)], dtype=np.float64)
        want = (af @ wf.T) * s[:, None]
        assert np.abs(got["y"] - want).max() <= 3e-5 * np.abs(want).max()


def _q8_engine_and_oracle(oracle, cfg, n_ctx, mode=1):
    """mode 1: W8A16 (Q8_0 weights, bf16 activations); mode 2: W8A8 — the activations are Q8_0 blocks too (ggml's vec_dot_q8_0_q8_0)."""
    from q3tts import native
    cfg.talker_q8_0 = mode
    eng = native.NativeEngine(cfg)
    om = oracle.OracleModel(cfg.model, seed=0, n_ctx=n_ctx, n_threads=min(16, os.cpu_count() or 4))
    if mode == 2:
        om.set_talker_q8a8()
    else:
        om.set_talker_q8()
    return eng, om


@pytest.mark.parametrize("mode", [1, 2])
def test_talker_q8_0_on_the_device_ids_match_the_oracle(oracle, mode):
    """q3tts_engine_config.talker_q8_0 = 1 / 2 with the synthetic model: device and oracle quantise the same bf16 weights with ggml's reference
    rule and multiply the blocks in the canonical Q8 order (mode 2: against activations quantised where they are produced — the GEMM epilogues,
    the attention kernels, the feedback row, the prompt rows — in ggml's Q8_0 x Q8_0 arithmetic) — prompt rows, prefill logits / hidden, greedy
    and sampled ids are equal, on one slot and on several of different length, with every attention kernel variant; and the ids DIFFER from
    the bf16 engine's and from the other mode's (the quantisation is really in the path)."""
    from q3tts import _abi, native
    cfg = _abi.tiny_config(max_batch=4, n_ctx=256, with_vocoder=0)
    eng, om = _q8_engine_and_oracle(oracle, cfg, 256, mode)
    try:
        desc, keep = oracle.make_prompt_desc(np.arange(100, 120), spk_emb=_spk(cfg.model.d_embed))
        pe = om.build_prompt(desc)
        h_ref, l_ref = om.talker_prefill(pe)
        h, l = eng.talker_prefill(pe)
        assert np.array_equal(_bits(l), _bits(l_ref)) and np.array_equal(_bits(h), _bits(h_ref))
        ref, _ = om.generate(pe, temperature=0.0, max_steps=12, min_frames=12)
        res = eng.generate(embd=pe, temperature=0.0, max_steps=12, min_frames=12)
        assert np.array_equal(res.codes, ref)
        rng = np.random.default_rng(8)
        reqs, refs = [], []
        for i in range(6):
            d2, k2 = oracle.make_prompt_desc(rng.integers(0, 151643, size=int(rng.integers(3, 40))), spk_emb=_spk(cfg.model.d_embed))
            p2 = om.build_prompt(d2)
            t = int(rng.integers(3, 15))
            kw = dict(temperature=0.7, top_k=40, top_p=0.9, seed=70 + i, max_steps=20, min_frames=t, force_eos_at=t)
            refs.append(om.generate(p2, **kw)[0]); reqs.append(dict(embd=p2, **kw))
        for o, r in zip(eng.generate_batch(reqs), refs):
            assert o.status == 0 and np.array_equal(o.codes, r)
        try:   # the other attention kernels (k_attend<2, true> for decode, k_attend<2, false> / k_attend_prefill for whole prompts) write the same operand
            for pol in ((1, 1), (0, 2)):
                _attend_policy(*pol)
                for o, r in zip(eng.generate_batch(reqs), refs):
                    assert o.status == 0 and np.array_equal(o.codes, r), pol
        finally:
            _attend_policy(0, 0)
        cfg16 = _abi.tiny_config(max_batch=4, n_ctx=256, with_vocoder=0)
        cfg16.talker_q8_0 = 0 if mode == 1 else 1
        e16 = native.NativeEngine(cfg16)
        try:
            assert not np.array_equal(e16.generate(embd=pe, temperature=0.0, max_steps=12, min_frames=12).codes, ref)
        finally:
            e16.close()
    finally:
        eng.close()
        om.close()


def test_talker_q8a8_64_slots_row_buckets(oracle):
    """W8A8 with max_batch = 64 on the small shape: 80 sampled requests of mixed prompt and target length — the decode rows run every row-tile
    instance of k_bgemm8 (64 -> 48 -> 32 -> 16 -> 8 ... 1 rows as the batch drains, (2,2) / (4,2) / (2,4) / (1,x) tiles, prefill at > 64 rows in
    64-row chunks), slots are re-used — every request's ids equal the oracle's W8A8 replay."""
    from q3tts import _abi
    cfg = _abi.tiny_config(max_batch=64, n_ctx=256, with_vocoder=0)
    eng, om = _q8_engine_and_oracle(oracle, cfg, 256, 2)
    try:
        rng = np.random.default_rng(88)
        reqs, refs = [], []
        for i in range(80):
            n_text = int(rng.integers(3, 30)); target = int(rng.integers(2, 33))
            desc, keep = oracle.make_prompt_desc(rng.integers(0, 151643, size=n_text), spk_emb=_spk(cfg.model.d_embed))
            pe = om.build_prompt(desc)
            kw = dict(temperature=0.7, top_k=40, top_p=0.9, seed=3000 + i, max_steps=40, min_frames=target, force_eos_at=target)
            refs.append(om.generate(pe, **kw)[0])
            reqs.append(dict(embd=pe, **kw))
        outs = eng.generate_batch(reqs)
        tm = eng.timings()
        for i, (o, r) in enumerate(zip(outs, refs)):
            assert o.status == 0 and o.codes.shape == r.shape and np.array_equal(o.codes, r), i
        assert tm.mean_rows < 60.0   # the batch did drain through smaller row buckets
    finally:
        eng.close()
        om.close()


@pytest.mark.parametrize("talker_type,mode", [(8, 1), (30, 1), (8, 2)])
def test_talker_q8_0_from_model_files(oracle, tmp_path, talker_type, mode):
    """weights_path + talker_q8_0 = 1. A Q8_0 Talker container (type 8: the reference's gguf_q8_0 directory, src/tts/engine.rs:91-95) goes
    to the device AS STORED — the file's own f16 scales and int8 quants, never widened to bf16 — and a BF16 container (type 30) is quantised
    on the device with ggml's rule; either way the ids equal the oracle's Q8 mode (tests/_gguf.py's writer and the oracle's quantiser are
    the same rule: checked in the CPU suite). The Predictor file is BF16 in both cases (it keeps bf16 weights)."""
    import _gguf as G
    from q3tts import _abi, native
    cfg = _abi.tiny_config(max_batch=2, n_ctx=128, with_vocoder=0)
    oracle.write_model_dir(str(tmp_path), cfg.model, 0, matrix_type=talker_type, predictor_type=G.BF16)
    cfg.weights_path = str(tmp_path).encode()
    cfg.talker_q8_0 = mode   # (2: W8A8 — the file's blocks against activations quantised on the device, as llama.cpp multiplies them)
    eng = native.NativeEngine(cfg)
    om = oracle.OracleModel(_abi.tiny_config(max_batch=2, n_ctx=128, with_vocoder=0).model, seed=0, n_ctx=128, n_threads=4)
    if mode == 2:
        om.set_talker_q8a8()
    else:
        om.set_talker_q8()
    try:
        desc, keep = oracle.make_prompt_desc(np.arange(900, 912), spk_emb=_spk(cfg.model.d_embed))
        pe = om.build_prompt(desc)
        assert np.array_equal(_bits(eng.build_prompt(desc)), _bits(pe))
        for kw in (dict(temperature=0.0, max_steps=6, min_frames=6), dict(temperature=0.7, top_k=40, top_p=0.9, seed=11, max_steps=6, min_frames=6)):
            ref, _ = om.generate(pe, **kw)
            assert ref.shape[0] == 6 and np.array_equal(eng.generate(desc=desc, **kw).codes, ref)
    finally:
        eng.close()
        om.close()


@pytest.mark.parametrize("mode", [1, 2])
def test_talker_q8_0_full_shape_prefill_and_frames(oracle, mode):
    """The same at the benchmarked shape (28 x 2048 Talker in Q8_0 blocks, 1.5 GB instead of 2.8 GB of weights): prefill logits / hidden
    and 4 greedy frames equal the oracle's (mode 2: W8A8 on the int8 MFMA)."""
    import time
    from q3tts import _abi
    cfg = _abi.full_config_py()
    cfg.max_batch, cfg.n_ctx, cfg.max_steps_cap, cfg.with_vocoder = 2, 128, 16, 0
    eng, om = _q8_engine_and_oracle(oracle, cfg, 128, mode)
    try:
        t0 = time.time()
        desc, keep = oracle.make_prompt_desc(np.random.default_rng(1234).integers(0, 151643, size=12), spk_emb=_spk(cfg.model.d_embed))
        pe = om.build_prompt(desc)
        h_ref, l_ref = om.talker_prefill(pe)
        h, l = eng.talker_prefill(pe)
        assert np.array_equal(_bits(l), _bits(l_ref)) and np.array_equal(_bits(h), _bits(h_ref))
        ref, _ = om.generate(pe, temperature=0.0, max_steps=4, min_frames=4)
        res = eng.generate(desc=desc, temperature=0.0, max_steps=4, min_frames=4)
        assert ref.shape == (4, 16) and np.array_equal(res.codes, ref)
        print(f"full shape, Talker in Q8_0 blocks (mode {mode}): prefill + 4 frames equal; oracle {time.time() - t0:.0f} s")
    finally:
        eng.close()
        om.close()


@pytest.mark.parametrize("rows,n_in,n_out", [(1, 2048, 1024), (64, 2048, 1024), (37, 512, 512), (130, 256, 96)])
def test_projection_follows_the_reference_sequence(oracle, native, rows, n_in, n_out):
    """H6 (src/assets_manager.rs:383-399): `sum = bias; sum += h * w` in ascending input order, f32 weights. The device kernel
    and the oracle both follow that sequence, and so does a literal numpy float32 loop on a few outputs."""
    rng = np.random.default_rng(rows + n_out)
    x = _rand(rng, (rows, n_in), 2.0); w = _rand(rng, (n_out, n_in), 0.02); b = _rand(rng, (n_out,), 0.02)
    nw = (1.0 + _rand(rng, (n_out,), 0.05)).astype(np.float32)
    y, xb, ssp = native.k_project(x, w, b, nw)
    y_ref = oracle.project_rows(w, b, x)
    assert np.array_equal(_bits(y), _bits(y_ref))
    for (r, o) in [(0, 0), (rows - 1, n_out - 1), (rows // 2, 17 % n_out)]:
        acc = np.float32(b[o])
        for i in range(n_in):
            acc = np.float32(acc + np.float32(x[r, i] * w[o, i]))
        assert _bits(acc) == _bits(y[r, o])
    xb_ref, ssp_ref = oracle.norm_inputs(y_ref, nw)
    assert np.array_equal(xb, xb_ref) and np.array_equal(_bits(ssp), _bits(ssp_ref))


@pytest.mark.parametrize("rows,d", [(1, 2048), (31, 1024), (5, 512)])
def test_norm_inputs_match_oracle(oracle, native, rows, d):
    rng = np.random.default_rng(rows + d)
    x = _rand(rng, (rows, d), 3.0); nw = (1.0 + _rand(rng, (d,), 0.05)).astype(np.float32)
    xb, ssp = native.k_norm_inputs(x, nw)
    xb_ref, ssp_ref = oracle.norm_inputs(x, nw)
    assert np.array_equal(xb, xb_ref) and np.array_equal(_bits(ssp), _bits(ssp_ref))
    # and together with the consumer's reduction it is an RMSNorm
    s = np.array([oracle.row_scale(ssp_ref[r], d, 1e-6) for r in range(rows)])
    assert np.allclose(s, 1.0 / np.sqrt((x.astype(np.float64) ** 2).mean(axis=1) + 1e-6), rtol=1e-6)


# ---------------------------------------------------------------------------------------------------------------
# parity at the benchmarked shape and batch (BASELINE.json configs[1] / configs[2]): the full 1.7B shape, and 64 slots
# ---------------------------------------------------------------------------------------------------------------
def test_full_shape_single_utterance_ids_and_pcm(oracle):
    """configs[1] at the shape bench.py runs (28 x 2048 Talker, 5 x 1024 Predictor, 15 heads x 2048, full vocoder): one utterance,
    n_text = 20 (31 prompt rows), greedy, 8 frames — prompt rows, prefill logits and all 8 x 16 codec ids equal the oracle's bit for
    bit, PCM within the RMS tolerance (the achieved value is printed). The oracle needs ~1 s per frame on 16 threads."""
    import os
    import time
    from q3tts import _abi, native
    cfg = _abi.full_config_py()
    cfg.max_batch, cfg.n_ctx, cfg.max_steps_cap = 2, 256, 64
    threads = min(16, os.cpu_count() or 4)
    om = oracle.OracleModel(cfg.model, seed=0, n_ctx=256, n_threads=threads)
    eng = native.NativeEngine(cfg)
    try:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "speakers", "vivian.json")) as f:
            import json
            spk = np.asarray(json.load(f)["spk_emb"], dtype=np.float32)
        ids = np.random.default_rng(1234).integers(0, 151643, size=20)
        desc, keep = oracle.make_prompt_desc(ids, spk_emb=spk)
        pe = om.build_prompt(desc)
        assert pe.shape == (31, 2048) and np.array_equal(_bits(eng.build_prompt(desc)), _bits(pe))
        t0 = time.time()
        h_ref, l_ref = om.talker_prefill(pe)
        h, l = eng.talker_prefill(pe)
        assert np.array_equal(_bits(l), _bits(l_ref)) and np.array_equal(_bits(h), _bits(h_ref))
        ref, _ = om.generate(pe, temperature=0.0, max_steps=8, min_frames=8)
        res = eng.generate(desc=desc, temperature=0.0, max_steps=8, min_frames=8, want_pcm=1)
        assert ref.shape == (8, 16) and np.array_equal(res.codes, ref)
        L = oracle.lib()
        v = L.q3o_vocoder_create(C.byref(cfg.vocoder), 0, threads)
        try:
            ref_pcm = _oracle_pcm(oracle, v, np.clip(ref, 0, cfg.vocoder.codebook_size - 1).astype(np.int32))
        finally:
            L.q3o_vocoder_destroy(v)
        rms = float(np.sqrt(np.mean((res.pcm - ref_pcm) ** 2)))
        print(f"full shape, 8 greedy frames: ids equal; PCM RMS error {rms:.2e} (tolerance {PCM_RMS_TOL_FULL:.0e}, signal RMS {float(np.sqrt(np.mean(ref_pcm ** 2))):.2f}); "
              f"oracle {time.time() - t0:.1f} s on {threads} threads")
        assert res.pcm.shape == ref_pcm.shape == (8 * 1920,) and rms <= PCM_RMS_TOL_FULL
        # a sampled pair on two slots at once (row bucket 2), same shape
        reqs, refs = [], []
        for i in range(2):
            kw = dict(temperature=0.7, top_k=40, top_p=0.9, seed=1000 + i, max_steps=5, min_frames=3 + i, force_eos_at=3 + i)
            refs.append(om.generate(pe, **kw)[0])
            reqs.append(dict(desc=desc, **kw))
        for o, r in zip(eng.generate_batch(reqs), refs):
            assert o.status == 0 and np.array_equal(o.codes, r)
    finally:
        eng.close()
        om.close()


def test_64_slots_mixed_lengths_sampled_ids_and_pcm(oracle):
    """configs[2] on the small shape the oracle finishes quickly: max_batch = 64, 100 mixed-length sampled requests — the row buckets
    cross 64 -> 48 -> 32 -> 16 -> 8 -> ... -> 1, slots are re-used by the 36 requests that wait, results are handed over deferred. Every
    request's ids equal the oracle's; PCM of a sample of them within tolerance."""
    from q3tts import _abi, native
    cfg = _abi.tiny_config(max_batch=64, n_ctx=256, with_vocoder=1)
    eng = native.NativeEngine(cfg)
    om = oracle.OracleModel(cfg.model, seed=0, n_ctx=256, n_threads=4)
    L = oracle.lib()
    v = L.q3o_vocoder_create(C.byref(cfg.vocoder), 0, 4)
    try:
        rng = np.random.default_rng(64)
        reqs, refs = [], []
        for i in range(100):
            n_text = int(rng.integers(3, 30)); target = int(rng.integers(2, 41))
            desc, keep = oracle.make_prompt_desc(rng.integers(0, 151643, size=n_text), spk_emb=_spk(cfg.model.d_embed))
            pe = om.build_prompt(desc)
            kw = dict(temperature=0.7, top_k=40, top_p=0.9, seed=1000 + i, max_steps=48, min_frames=target, force_eos_at=target)
            refs.append(om.generate(pe, **kw)[0])
            reqs.append(dict(embd=pe, want_pcm=1, **kw))
        outs = eng.generate_batch(reqs)
        tm = eng.timings()
        worst = 0.0
        for i, (o, r) in enumerate(zip(outs, refs)):
            assert o.status == 0 and o.codes.shape == r.shape and np.array_equal(o.codes, r), i
            if i % 9 == 0:
                ref_pcm = _oracle_pcm(oracle, v, np.clip(r, 0, cfg.vocoder.codebook_size - 1).astype(np.int32))
                assert o.pcm.shape == ref_pcm.shape
                worst = max(worst, float(np.sqrt(np.mean((o.pcm - ref_pcm) ** 2))))
        print(f"64 slots, 100 requests: ids equal; worst PCM RMS {worst:.2e}; mean live {tm.mean_live_slots:.1f}, mean rows {tm.mean_rows:.1f}")
        assert worst <= PCM_RMS_TOL
        assert tm.mean_rows - tm.mean_live_slots <= 8.0  # row buckets in multiples of 16: at most 15 padding rows, fewer on average
    finally:
        eng.close()
        om.close()
        L.q3o_vocoder_destroy(v)


def test_full_shape_64_slots_sampled_mixed_lengths_ids_and_pcm(oracle):
    """configs[2] at the BENCHMARKED shape and batch (VERDICT r02, next #1a): the full 1.7B shape, max_batch = 64, 72 sampled requests of
    mixed prompt and target lengths with the vocoder on. The first 64 start together on 64 rows; targets of 4 / 8 / 12 frames drain the
    batch through the row buckets 64 -> 48 -> 32 -> 16 (k_gather_rows moves the surviving rows' logits / hidden state); requests 64..71
    wait and take re-used slots (stale KV, vocoder state reset); every result is handed over deferred. The oracle (one utterance at a
    time, ~1 s per frame) replays five of them: the first, a long one that outlives every bucket change, one from the middle, and two
    on re-used slots including the last — ids equal, PCM within the full-shape tolerance. /root/reference/src/tts/engine.rs:545-642."""
    import json
    import time
    from q3tts import _abi, native
    cfg = _abi.full_config_py()
    cfg.max_batch, cfg.n_ctx, cfg.max_steps_cap = 64, 128, 32
    threads = min(16, os.cpu_count() or 4)
    eng = native.NativeEngine(cfg)
    om = oracle.OracleModel(cfg.model, seed=0, n_ctx=128, n_threads=threads)
    L = oracle.lib()
    v = L.q3o_vocoder_create(C.byref(cfg.vocoder), 0, threads)
    try:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "speakers", "vivian.json")) as f:
            spk = np.asarray(json.load(f)["spk_emb"], dtype=np.float32)
        rng = np.random.default_rng(6464)
        reqs, metas, keep_all = [], [], []
        for i in range(72):
            n_text = int(rng.integers(2, 9))
            target = (4, 8, 12)[i % 3] if i < 64 else 4 + (i % 2)
            ids = rng.integers(0, 151643, size=n_text)
            desc, keep = oracle.make_prompt_desc(ids, spk_emb=spk)
            keep_all.append((desc, keep))
            kw = dict(temperature=0.7, top_k=40, top_p=0.9, seed=1000 + i, max_steps=16, min_frames=target, force_eos_at=target)
            reqs.append(dict(desc=desc, want_pcm=1, **kw)); metas.append((desc, kw, target))
        outs = eng.generate_batch(reqs)
        tm = eng.timings()
        assert all(o.status == 0 for o in outs) and [o.n_frames for o in outs] == [m[2] for m in metas]
        assert tm.mean_rows < 60.0 and tm.mean_live_slots > 16.0   # the batch did drain through smaller row buckets
        t0 = time.time()
        worst = 0.0
        for i in (0, 2, 37, 64, 71):
            desc, kw, target = metas[i]
            pe = om.build_prompt(desc)
            ref, _ = om.generate(pe, **kw)
            assert ref.shape == (target, 16) and np.array_equal(outs[i].codes, ref), i
            ref_pcm = _oracle_pcm(oracle, v, np.clip(ref, 0, cfg.vocoder.codebook_size - 1).astype(np.int32))
            assert outs[i].pcm.shape == ref_pcm.shape == (target * 1920,), i
            worst = max(worst, float(np.sqrt(np.mean((outs[i].pcm - ref_pcm) ** 2))))
        print(f"full shape, 64 slots, 72 sampled requests: ids of 5 replayed utterances equal; worst PCM RMS {worst:.2e} (tolerance {PCM_RMS_TOL_FULL:.0e}); "
              f"mean live {tm.mean_live_slots:.1f}, mean rows {tm.mean_rows:.1f}, frame step {tm.frame_step_ms:.2f} ms; oracle {time.time() - t0:.0f} s on {threads} threads")
        assert worst <= PCM_RMS_TOL_FULL
    finally:
        eng.close()
        om.close()
        L.q3o_vocoder_destroy(v)


def _attend_policy(decode, prefill):
    from q3tts import _abi
    rc = _abi.load_library().q3tts_k_attend_policy(decode, prefill)
    assert rc == 0


@pytest.mark.parametrize("n_rows", [1, 2, 63, 64, 65, 100, 127, 128])
def test_attention_kernel_variants_agree_on_prompt_runs(oracle, tiny, n_rows):
    """k_attend_prefill (a whole prompt run served from LDS: one key block for runs of <= 64 rows, two for 65..128, the value loop in
    trips of 64) against k_attend<2, false> (a workgroup per row and KV head) IN ONE PROCESS through q3tts_k_attend_policy, and both
    against the oracle: hidden row and logits of the last prompt row, bit for bit (the comparison the kernels' comments cite)."""
    cfg, eng, om = tiny
    desc, keep = oracle.make_prompt_desc(np.random.default_rng(n_rows).integers(0, 151643, size=130), spk_emb=_spk(cfg.model.d_embed))
    pe = np.ascontiguousarray(om.build_prompt(desc)[:n_rows])
    h_ref, l_ref = om.talker_prefill(pe)
    try:
        for prefill in (2, 1):   # 2: k_attend_prefill even for one run; 1: never
            _attend_policy(0, prefill)
            h, l = eng.talker_prefill(pe)
            assert np.array_equal(_bits(h), _bits(h_ref)) and np.array_equal(_bits(l), _bits(l_ref)), prefill
    finally:
        _attend_policy(0, 0)


def test_decode_attention_kernel_variants_agree(oracle, tiny):
    """k_attend_gqa2 (four waves per (slot, KV head)) against k_attend<2, true> in one process: greedy ids of two utterances with
    contexts that cross a key block (60 + frames, 130 + frames) equal each other and the oracle's."""
    cfg, eng, om = tiny
    reqs, refs = [], []
    for nt in (52, 122):
        desc, keep = oracle.make_prompt_desc(np.random.default_rng(nt).integers(0, 151643, size=nt), spk_emb=_spk(cfg.model.d_embed))
        pe = om.build_prompt(desc)
        refs.append(om.generate(pe, temperature=0.0, max_steps=7, min_frames=7)[0])
        reqs.append(dict(embd=pe, temperature=0.0, max_steps=7, min_frames=7))
    try:
        for decode in (1, 0):
            _attend_policy(decode, 0)
            for o, r in zip(eng.generate_batch(reqs), refs):
                assert o.status == 0 and np.array_equal(o.codes, r), decode
    finally:
        _attend_policy(0, 0)


def test_full_shape_long_prompts_batched_prefill_and_long_decode_context(oracle):
    """The full 1.7B shape (16 query / 8 KV heads) where round 3's value checks stopped at 70-row contexts:
    seventeen prompts admitted by one q3tts_generate_batch call — sixteen runs of 1, 63, 64, 65, 127, 128 and ten times 128 rows
    (1 728 rows: 16 x 8 = 128 (run, KV head) workgroups, so the launcher takes k_attend_prefill by itself, at its longest LDS-resident runs
    and both key-block counts; the GEMMs run k_bgemm_big at 1 728 rows), then — the group is full — a run of 330 rows alone (the
    k_attend<2, false> fallback, k_bgemm's 64-row chunks at 330 rows), two greedy frames each: decode attention (k_attend_gqa2) over
    contexts of 2 .. 332 keys. The oracle replays the 330-row, one 128-row, the 65-row and the 1-row utterance: ids equal. The same batch
    under the old kernels (k_attend<2, false> everywhere, k_attend<2, true> for decode) gives the same ids for all seventeen; the last
    prompt row's hidden state and logits of the 65- and 128-row prompts equal the oracle's bit for bit under both prefill kernels."""
    import time
    from q3tts import _abi, native
    cfg = _abi.full_config_py()
    cfg.max_batch, cfg.n_ctx, cfg.max_steps_cap, cfg.with_vocoder = 18, 2048, 8, 0
    threads = min(64, os.cpu_count() or 4)
    eng = native.NativeEngine(cfg)
    om = oracle.OracleModel(cfg.model, seed=0, n_ctx=512, n_threads=threads)
    try:
        lens = [1, 63, 64, 65, 127, 128] + [128] * 10 + [330]
        assert sum(lens[:16]) == 1728 and sum(lens) > cfg.n_ctx   # the 330-row run does not fit behind the sixteen: a group of its own
        reqs, pes = [], []
        for i, n in enumerate(lens):
            desc, keep = oracle.make_prompt_desc(np.random.default_rng(4000 + i).integers(0, 151643, size=max(n, 11) - 11 + 8), spk_emb=_spk(cfg.model.d_embed))
            pe = np.ascontiguousarray(om.build_prompt(desc)[-n:] if n >= 11 else om.build_prompt(desc)[:n])
            assert pe.shape == (n, cfg.model.d_embed)
            pes.append(pe)
            reqs.append(dict(embd=pe, temperature=0.0, max_steps=2, min_frames=2))
        outs = eng.generate_batch(reqs)
        assert all(o.status == 0 and o.codes.shape == (2, 16) for o in outs)
        t0 = time.time()
        for i in (16, 15, 3, 0):
            ref, _ = om.generate(pes[i], temperature=0.0, max_steps=2, min_frames=2)
            assert np.array_equal(outs[i].codes, ref), (i, lens[i])
        t_or = time.time() - t0
        try:
            _attend_policy(1, 1)
            old = eng.generate_batch(reqs)
            for i, (a, b) in enumerate(zip(outs, old)):
                assert b.status == 0 and np.array_equal(a.codes, b.codes), (i, lens[i])
            for i in (3, 15):
                h_ref, l_ref = om.talker_prefill(pes[i])
                for prefill in (2, 1):
                    _attend_policy(0, prefill)
                    h, l = eng.talker_prefill(pes[i])
                    assert np.array_equal(_bits(h), _bits(h_ref)) and np.array_equal(_bits(l), _bits(l_ref)), (i, prefill)
        finally:
            _attend_policy(0, 0)
        print(f"full shape, 17 prompts of {lens[0]}..{lens[-1]} rows: ids of 4 replayed utterances equal, old and new attention kernels agree on all 17; oracle {t_or:.0f} s on {threads} threads")
    finally:
        eng.close()
        om.close()


def test_allocator_hands_out_memory_with_the_zero_fill_completed():
    """Rounds 1 and 2 zero-filled asynchronously on a non-blocking stream and fixed three call sites whose uploads were overwritten by
    the fill. The allocator now returns after the fill: a null-stream upload right behind a LARGE allocation (the fill of 1 GiB takes
    ~0.2 ms, the upload is issued microseconds after the allocation returns) must read back intact, and the rest must read zero."""
    from q3tts import _abi, native
    cfg = _abi.tiny_config(max_batch=1, n_ctx=128, with_vocoder=1)
    eng = native.NativeEngine(cfg)
    try:
        for nbytes in (1 << 30, 16384, 256 << 20, 1 << 30):
            assert eng.alloc_upload_mismatches(nbytes) == 0, nbytes
    finally:
        eng.close()


@pytest.mark.parametrize("n_frames", [4, 7])
def test_full_shape_vocoder_pcm_vs_oracle(oracle, n_frames):
    """V1-V6 at the full shape (d 1024, 8 layers x 16 heads x 64, window 72, 1536 -> 768 -> 384 -> 192 -> 96): PCM vs the oracle,
    streaming in 4-frame chunks == one call."""
    import os
    from q3tts import _abi, native
    cfg = _abi.full_config_py()
    cfg.max_batch, cfg.n_ctx, cfg.max_steps_cap = 1, 128, 32
    # the decoder is not under test here: shrink it so the engine comes up quickly
    m = cfg.model
    m.t_n_layer, m.p_n_layer = 1, 1
    eng = native.NativeEngine(cfg)
    L = oracle.lib()
    v = L.q3o_vocoder_create(C.byref(cfg.vocoder), 0, min(16, os.cpu_count() or 4))
    try:
        codes = np.random.default_rng(n_frames).integers(0, cfg.vocoder.codebook_size, size=(n_frames, 16)).astype(np.int32)
        ref = _oracle_pcm(oracle, v, codes)
        L.q3o_vocoder_set_arith(v, 1)
        try:
            ref32 = _oracle_pcm(oracle, v, codes)   # plain f32 GEMM inputs: the reference's ORT CPU arithmetic up to summation order
        finally:
            L.q3o_vocoder_set_arith(v, 0)
        one = eng.vocoder(codes)
        e = lambda a, b: float(np.sqrt(np.mean((a - b) ** 2)))
        print(f"full-shape vocoder, {n_frames} frames, PCM RMS: device vs bf16-input oracle {e(one, ref):.2e}, device vs f32 oracle {e(one, ref32):.2e}, "
              f"bf16-input oracle vs f32 oracle {e(ref, ref32):.2e}; signal RMS {float(np.sqrt(np.mean(ref32 ** 2))):.2f}")
        assert one.shape == ref.shape == (n_frames * 1920,) and e(one, ref) <= PCM_RMS_TOL_FULL and e(one, ref32) <= PCM_RMS_TOL_FULL
        assert np.array_equal(eng.vocoder(codes, chunk_frames=4), one)
        os.environ["Q3TTS_VOC_NORING"] = "1"   # big convolutions on the register-staged GEMM: same bits as the LDS-DMA ring
        try:
            assert np.array_equal(eng.vocoder(codes), one)
        finally:
            del os.environ["Q3TTS_VOC_NORING"]
        os.environ["Q3TTS_VOC_TAP_MIN"] = "1"   # the 768- and 384-channel 7-tap convolutions on k_vconv_tap (what a batch of 64 runs): same bits
        try:
            assert np.array_equal(eng.vocoder(codes), one)
            assert np.array_equal(eng.vocoder(codes, chunk_frames=3), one)   # partial tiles: 96 / 480 rows per call
        finally:
            del os.environ["Q3TTS_VOC_TAP_MIN"]
        os.environ["Q3TTS_VOC_POLITE"] = "1"   # the launches a wide call gets beside the decoder (one workgroup per CU, 4-wave units): same bits
        try:
            assert np.array_equal(eng.vocoder(codes), one)
            os.environ["Q3TTS_VOC_TAP_MIN"] = "1"   # ... with the 4-wave form of k_vconv_tap (768 and 384 channels)
            assert np.array_equal(eng.vocoder(codes), one)
            assert np.array_equal(eng.vocoder(codes, chunk_frames=3), one)
        finally:
            del os.environ["Q3TTS_VOC_POLITE"]
            os.environ.pop("Q3TTS_VOC_TAP_MIN", None)
    finally:
        eng.close()
        L.q3o_vocoder_destroy(v)
