"""CPU tests of the oracle (the checker itself): published known-answer vectors, hand-derived expectations from the
cited reference lines, and committed golden self-vectors (tests/golden/oracle_tiny.json, made by make_golden.py).

PARITY UNPINNED: the reference ships no tests / fixtures for this path; the only external vectors are
the ChaCha block-function KATs below, everything else pins the restatement against the reference's source text.
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _u32(a):
    return np.asarray(a, dtype=np.uint32)


def test_chacha20_rfc7539_block(oracle):
    """RFC 7539 §2.3.2 block-function test vector (key 00..1f, counter 1, nonce 00:00:00:09:00:00:00:4a:00:00:00:00)."""
    L = oracle.lib()
    st = _u32([0x61707865, 0x3320646e, 0x79622d32, 0x6b206574] + [int.from_bytes(bytes(range(4 * i, 4 * i + 4)), "little") for i in range(8)] +
              [1, 0x09000000, 0x4a000000, 0])
    out = np.zeros(16, dtype=np.uint32)
    L.q3o_chacha_block(oracle.ptr(st, oracle.u32p), 20, oracle.ptr(out, oracle.u32p))
    assert out.tobytes().hex() == ("10f1e7e4d13b5915500fdd1fa32071c4c7d1f4c733c068030422aa9ac3d46c4e"
                                   "d2826446079faa0914c2d705d98b02a2b5129cd1de164eb9cbd083e8a2503c4e")


def test_chacha_zero_key_keystreams(oracle):
    """Zero key / zero nonce keystream heads: ChaCha20 (the vector rand_chacha's own test suite uses) and ChaCha12."""
    L = oracle.lib()
    st = _u32([0x61707865, 0x3320646e, 0x79622d32, 0x6b206574] + [0] * 12)
    out = np.zeros(16, dtype=np.uint32)
    L.q3o_chacha_block(oracle.ptr(st, oracle.u32p), 20, oracle.ptr(out, oracle.u32p))
    assert [int(x) for x in out[:8]] == [0xade0b876, 0x903df1a0, 0xe56a5d40, 0x28bd8653, 0xb819d2bd, 0x1aed8da0, 0xccef36a8, 0xc70d778b]
    L.q3o_chacha_block(oracle.ptr(st, oracle.u32p), 12, oracle.ptr(out, oracle.u32p))
    assert out.tobytes()[:32].hex() == "9bf49a6a0755f953811fce125f2683d50429c3bb49e074147e0089a52eae155f"


def test_stdrng_stream_shape(oracle):
    L = oracle.lib()
    a = np.zeros(200, dtype=np.float32)
    L.q3o_rng_f32(42, 200, oracle.ptr(a, oracle.f32p))
    assert (a >= 0).all() and (a < 1).all()
    assert np.all(a * 16777216.0 == np.floor(a * 16777216.0))  # 24-bit fractions: (u32 >> 8) * 2^-24
    b = np.zeros(70, dtype=np.float32)
    L.q3o_rng_f32(42, 70, oracle.ptr(b, oracle.f32p))
    assert np.array_equal(a[:70], b)  # 64-word buffer refill boundary is seamless
    c = np.zeros(8, dtype=np.float32)
    L.q3o_rng_f32(43, 8, oracle.ptr(c, oracle.f32p))
    assert not np.array_equal(a[:8], c)


def test_positions_h2(oracle):
    """qwen3_position: src/tts/engine.rs:306-314 — [t.. | h.. | w.. | 0..]."""
    L = oracle.lib()
    out = np.zeros(12, dtype=np.int32)
    L.q3o_qwen3_position(5, 3, oracle.ptr(out, oracle.i32p))
    assert out.tolist() == [5, 6, 7, 5, 6, 7, 5, 6, 7, 0, 0, 0]


@pytest.mark.parametrize("n,expect", [(0, []), (1, [(1, 1)]), (3, [(3, 1)]), (4, [(4, 0)]), (5, [(4, 0), (1, 1)]), (8, [(4, 0), (4, 0)]),
                                      (9, [(4, 0), (4, 0), (1, 1)])])
def test_chunker_h8(oracle, n, expect):
    """Vocoder-thread chunking: src/tts/engine.rs:507-541. Note n % 4 == 0: no call ever carries is_last (reference quirk)."""
    L = oracle.lib()
    cf, cl = np.zeros(16, dtype=np.int32), np.zeros(16, dtype=np.int32)
    k = L.q3o_chunk_plan(n, oracle.ptr(cf, oracle.i32p), oracle.ptr(cl, oracle.i32p), 16)
    assert [(int(cf[i]), int(cl[i])) for i in range(k)] == expect


def _sample(oracle, logits, T, k, p, r):
    lg = np.asarray(logits, dtype=np.float32)
    return oracle.lib().q3o_sample(oracle.ptr(lg, oracle.f32p), lg.size, T, k, p, r)


def test_sampler_h4_hand_cases(oracle):
    """src/models/llama/mod.rs:666-772."""
    # greedy: first max, strict '>' (:690-701)
    assert _sample(oracle, [1.0, 5.0, 5.0, 2.0], 0.0, 40, 0.9, 0.0) == 1
    assert _sample(oracle, [float("nan"), 1.0, 0.5], 0.0, 0, 1.0, 0.0) == 1  # NaN never wins
    # probabilities 8/15, 4/15, 2/15, 1/15 for logits ln(1,2,4,8) at temperature 1
    lg = np.log(np.array([1.0, 2.0, 4.0, 8.0]))
    for r, want in [(0.0, 3), (0.5, 3), (0.6, 2), (0.79, 2), (0.85, 1), (0.95, 0), (0.9999, 0)]:
        assert _sample(oracle, lg, 1.0, 0, 1.0, r) == want, r
    assert _sample(oracle, lg, 1.0, 1, 1.0, 0.99) == 3            # top_k = 1 (:711-713)
    assert _sample(oracle, lg, 1.0, 2, 1.0, 0.99) == 2            # renormalised over the top 2
    assert _sample(oracle, lg, 1.0, 0, 0.7, 0.7) == 2             # top-p: cut after cumsum >= 0.7 (inclusive), renormalise
    assert _sample(oracle, lg, 1.0, 0, 0.5, 0.99) == 3            # first candidate alone already reaches p
    assert _sample(oracle, lg, 1.0, -1, 1.0, 0.95) == 0           # negative top_k: `as usize` -> disabled (:646)
    assert _sample(oracle, lg, 0.01, 0, 1.0, 0.9999) == 3         # low temperature ~ greedy
    # stable sort: equal logits keep index order (:708)
    assert _sample(oracle, [2.0, 7.0, 7.0, 1.0], 1.0, 1, 1.0, 0.5) == 1
    # r beyond the (rounded) cumulative sum falls back to the first candidate (:767-770)
    assert _sample(oracle, lg, 1.0, 0, 1.0, 2.0) == 3


def test_bf16_rounding_and_expf(oracle):
    L = oracle.lib()
    assert L.q3o_bf16(1.0) == 0x3F80
    assert L.q3o_bf16(1.0 + 2.0 ** -8) == 0x3F80       # tie -> even
    assert L.q3o_bf16(1.0 + 3 * 2.0 ** -8) == 0x3F82   # tie -> even (up)
    assert L.q3o_bf16(-2.5) == 0xC020
    xs = np.linspace(-80, 20, 4001, dtype=np.float32)
    ys = np.array([L.q3o_expf(float(x)) for x in xs], dtype=np.float64)
    rel = np.abs(ys - np.exp(xs.astype(np.float64))) / np.exp(xs.astype(np.float64))
    assert rel.max() < 4e-7
    assert L.q3o_expf(-100.0) == 0.0 and L.q3o_expf(0.0) == 1.0


@pytest.fixture(scope="module")
def tiny_model(oracle):
    from q3tts import _abi
    cfg = _abi.tiny_config()
    m = oracle.OracleModel(cfg.model, seed=0, n_ctx=256, n_threads=4)
    yield cfg, m
    m.close()


def _text(oracle, m, i, d):
    out = np.zeros(d, dtype=np.float32)
    oracle.lib().q3o_text_embedding(m.h, i, oracle.ptr(out, oracle.f32p))
    return out


def _codec(oracle, m, q, c, d):
    out = np.zeros(d, dtype=np.float32)
    oracle.lib().q3o_codec_embedding(m.h, q, c, oracle.ptr(out, oracle.f32p))
    return out


def test_prompt_layout_h1(oracle, tiny_model):
    """src/tts/prompt.rs:141-277: row count and row contents of the x-vector prompt."""
    cfg, m = tiny_model
    d = cfg.model.d_embed
    spk = ((np.arange(d) % 13 - 6) * 0.03125).astype(np.float32)
    ids = [11, 22, 33, 44, 55]
    desc, keep = oracle.make_prompt_desc(ids, spk_emb=spk)
    pe = m.build_prompt(desc)
    assert pe.shape == (len(ids) + 11, d)
    marker, pad0 = _text(oracle, m, 151671, d), _codec(oracle, m, 0, 2148, d)
    for r, tid in enumerate([151644, 77091, 198]):                       # role block :171-175
        assert np.array_equal(pe[r], _text(oracle, m, tid, d))
    for r, cid in zip(range(3, 7), [2154, 2156, 2055, 2157]):            # THINK, THINK_BOS, lang, THINK_EOS :180-191
        assert np.array_equal(pe[r], marker + _codec(oracle, m, 0, cid, d))
    assert np.array_equal(pe[7], marker + spk)                           # speaker embedding :215-222
    assert np.array_equal(pe[8], _text(oracle, m, 151672, d) + pad0)     # BOS_TOKEN + PAD :229-239
    for j, t in enumerate(ids):
        assert np.array_equal(pe[9 + j], _text(oracle, m, t, d) + pad0)
    assert np.array_equal(pe[9 + len(ids)], _text(oracle, m, 151673, d) + pad0)
    assert np.array_equal(pe[10 + len(ids)], marker + _codec(oracle, m, 0, 2149, d))  # activation BOS :256-264
    # variants: instruct block (+5+k rows), NOTHINK (3 control rows), preset speaker id, clone prompt
    desc, keep = oracle.make_prompt_desc(ids, spk_emb=spk, instruct_ids=[7, 8])
    assert m.build_prompt(desc).shape[0] == len(ids) + 11 + 5 + 2
    desc, keep = oracle.make_prompt_desc(ids, spk_emb=spk, lang_id=-1)
    assert m.build_prompt(desc).shape[0] == len(ids) + 10
    desc, keep = oracle.make_prompt_desc(ids, spk_id=3065)
    pe2 = m.build_prompt(desc)
    assert np.array_equal(pe2[7], marker + _codec(oracle, m, 0, 3065, d))
    codes = (np.arange(3 * 16) * 5) % 64
    desc, keep = oracle.make_prompt_desc(ids, spk_emb=spk, ref_codes=codes, ref_text_ids=[1, 2])
    pe3 = m.build_prompt(desc)
    assert pe3.shape[0] == len(ids) + 11 + (2 + 2) + 1 + 3 + 1
    fr = np.zeros(d, dtype=np.float32)
    for q in range(16):
        fr = fr + _codec(oracle, m, q, int(codes[16 + q]), d)
    assert np.array_equal(pe3[8 + 4 + 1 + 1], marker + fr)               # second reference frame :79-96
    # table edge rules: src/assets_manager.rs:419-460
    assert np.all(_codec(oracle, m, 3, 999999, d) == 0) and np.array_equal(_codec(oracle, m, 0, -5, d), _codec(oracle, m, 0, 0, d))
    fb = _text(oracle, m, 200000, d)
    assert np.array_equal(fb, (np.mod((200000 * 17 + np.arange(d)).astype(np.float32), 2.0) - 1.0).astype(np.float32))


def test_project_h6(oracle, tiny_model):
    """Assets::project (src/assets_manager.rs:383-399) is one of the few floating-point sequences the crate spells out itself:
    `let mut sum = bias[o]; for i: sum += h[i] * w[o * n_in + i]` in f32. The oracle follows it literally — checked here against
    a numpy float32 loop with one rounding per multiply and per add — so for this row the restatement IS pinned."""
    cfg, m = tiny_model
    L = oracle.lib()
    d, dp = cfg.model.d_embed, cfg.model.p_d_model
    x = np.random.default_rng(0).standard_normal(d).astype(np.float32)
    y = np.zeros(dp, dtype=np.float32)
    L.q3o_project(m.h, oracle.ptr(x, oracle.f32p), oracle.ptr(y, oracle.f32p))
    W = oracle.synth_tensor(0, (3 << 16) | 1, (dp, d), 0.0, 0.02, True)
    b = oracle.synth_tensor(0, (3 << 16) | 2, (dp,), 0.0, 0.02, False)
    for o in (0, 1, 7, dp - 1):
        acc = np.float32(b[o])
        for i in range(d):
            acc = np.float32(acc + np.float32(x[i] * W[o, i]))
        assert acc.view(np.uint32) == y[o].view(np.uint32), o
    assert np.array_equal(oracle.project_rows(W, b, x[None, :])[0].view(np.uint32), y.view(np.uint32))


def test_mfma_restatement_64bit_form_equals_128bit_form(oracle):
    """q3o_mfma_bf16_dot32 (64-bit integers, what the GEMMs run) against q3o_mfma_bf16_dot32_ref (the 128-bit form measured against
    the hardware): random exponent spreads, zeros, accumulators from far below to far above the products (the fall-back branch)."""
    import ctypes as C
    L = oracle.lib()
    u16p = C.POINTER(C.c_uint16)
    rng = np.random.default_rng(1)
    n = 40000

    def rb(count, spread):
        e = rng.integers(127 - spread, 127 + spread, size=count).astype(np.uint16)
        v = (rng.integers(0, 2, size=count).astype(np.uint16) << 15) | (e << 7) | rng.integers(0, 128, size=count).astype(np.uint16)
        v[rng.random(count) < 0.05] = 0
        return v
    for spread, cexp in ((3, 4), (20, 30), (40, 60)):
        A = rb(n * 32, spread).reshape(n, 32); B = rb(n * 32, spread).reshape(n, 32)
        Cc = (rng.standard_normal(n) * np.exp2(rng.integers(-cexp, cexp, size=n))).astype(np.float32)
        Cc[rng.random(n) < 0.1] = 0
        for i in range(n):
            r = np.float32(L.q3o_mfma_bf16_dot32(A[i].ctypes.data_as(u16p), B[i].ctypes.data_as(u16p), float(Cc[i])))
            r2 = np.float32(L.q3o_mfma_bf16_dot32_ref(A[i].ctypes.data_as(u16p), B[i].ctypes.data_as(u16p), float(Cc[i])))
            assert r.view(np.uint32) == r2.view(np.uint32), (spread, i)


def test_split_rmsnorm_is_an_rmsnorm(oracle):
    """producer (bf16(x * nw), per-tile sums of squares) + consumer (lane-strided sums, butterfly, 1/sqrt) against
    float64, and the tile / lane orders against literal numpy float32 restatements."""
    rng = np.random.default_rng(5)
    for d in (512, 1024, 2048):
        x = (rng.standard_normal((3, d)) * 3).astype(np.float32); nw = (1 + 0.05 * rng.standard_normal(d)).astype(np.float32)
        xb, ssp = oracle.norm_inputs(x, nw)
        for r in range(3):
            sq = (x[r] * x[r]).astype(np.float32).reshape(d // 16, 16)
            for m in (1, 2, 4, 8):
                sq = (sq + sq[:, np.arange(16) ^ m]).astype(np.float32)
            assert np.array_equal(sq[:, 0].view(np.uint32), ssp[r].view(np.uint32))
            v = np.zeros(64, dtype=np.float32)
            for t in range(d // 16):
                v[t % 64] = ssp[r, t] if t < 64 else np.float32(v[t % 64] + ssp[r, t])
            for m in (32, 16, 8, 4, 2, 1):
                v = (v + v[np.arange(64) ^ m]).astype(np.float32)
            s_ref = np.float32(1.0) / np.sqrt(np.float32(v[0] / np.float32(d) + np.float32(1e-6)), dtype=np.float32)
            s = np.float32(oracle.row_scale(ssp[r], d, 1e-6))
            assert s.view(np.uint32) == s_ref.view(np.uint32)
            assert abs(float(s) - 1.0 / np.sqrt((x[r].astype(np.float64) ** 2).mean() + 1e-6)) <= 1e-6 * float(s)
        ref = (x * nw).astype(np.float32)
        back = (xb.astype(np.uint32) << 16).view(np.float32)
        assert np.abs(back - ref).max() <= 2.0 ** -8 * np.abs(ref).max()


def test_golden_self_vectors(oracle, tiny_model):
    """Regression pins produced by tests/golden/make_golden.py from this same restatement (self-vectors, labelled so)."""
    cfg, m = tiny_model
    with open(os.path.join(HERE, "golden", "oracle_tiny.json")) as f:
        g = json.load(f)
    L = oracle.lib()
    r = np.zeros(len(g["rng_seed42"]), dtype=np.float32)
    L.q3o_rng_f32(42, r.size, oracle.ptr(r, oracle.f32p))
    assert r.view(np.uint32).tolist() == g["rng_seed42"]
    spk = ((np.arange(cfg.model.d_embed) % 13 - 6) * 0.03125).astype(np.float32)
    desc, keep = oracle.make_prompt_desc(g["text_ids"], spk_emb=spk)
    pe = m.build_prompt(desc)
    assert int(pe.view(np.uint32).astype(np.uint64).sum()) == g["prompt_bits_sum"]
    hid, lg = m.talker_prefill(pe)
    assert lg.view(np.uint32)[:8].tolist() == g["prefill_logits_bits_head"]
    codes, eos = m.generate(pe, temperature=0.0, max_steps=g["max_steps"])
    assert codes.tolist() == g["greedy_codes"]
    codes, eos = m.generate(pe, temperature=0.7, top_k=40, top_p=0.9, seed=1234, max_steps=g["max_steps"])
    assert codes.tolist() == g["sampled_codes_seed1234"]


def test_vocoder_oracle_properties(oracle):
    """Causality: chunked streaming == one call; look-ahead withholds frames until flushed (V4)."""
    from q3tts import _abi
    cfg = _abi.tiny_config()
    L = oracle.lib()
    codes = np.random.default_rng(0).integers(0, 64, size=(6, 16)).astype(np.int32)
    v = L.q3o_vocoder_create(C.byref(cfg.vocoder), 0, 4)
    pcm = np.zeros(6 * 1920 + 8, dtype=np.float32)
    n = L.q3o_vocoder_decode(v, oracle.ptr(codes, oracle.i32p), 6, 1, oracle.ptr(pcm, oracle.f32p), pcm.size)
    assert n == 6 * 1920 and np.abs(pcm).max() <= 1.0
    L.q3o_vocoder_reset(v)
    parts = []
    for a, b, last in ((0, 4, 0), (4, 6, 1)):
        buf = np.zeros(4 * 1920, dtype=np.float32)
        k = L.q3o_vocoder_decode(v, oracle.ptr(codes[a:b].copy(), oracle.i32p), b - a, last, oracle.ptr(buf, oracle.f32p), buf.size)
        parts.append(buf[:k].copy())
    assert np.array_equal(np.concatenate(parts), pcm[:n])
    L.q3o_vocoder_destroy(v)
    cfg.vocoder.lookahead_frames = 2
    v = L.q3o_vocoder_create(C.byref(cfg.vocoder), 0, 4)
    buf = np.zeros(6 * 1920, dtype=np.float32)
    assert L.q3o_vocoder_decode(v, oracle.ptr(codes[:4].copy(), oracle.i32p), 4, 0, oracle.ptr(buf, oracle.f32p), buf.size) == 2 * 1920
    assert L.q3o_vocoder_decode(v, oracle.ptr(codes[4:].copy(), oracle.i32p), 2, 1, oracle.ptr(buf, oracle.f32p), buf.size) == 4 * 1920
    L.q3o_vocoder_destroy(v)


def _mel_numpy_f64(audio):
    """Independent restatement of the SPEC (src/models/onnx.rs:166-321) in float64 with numpy's rFFT: pins the padding rules,
    window, Slaney filterbank and log floor of the oracle (the oracle's own DFT order is pinned by the GPU parity test)."""
    x = np.asarray(audio, dtype=np.float64)
    n, pad = x.size, 384
    head = [x[i] if i < n else 0.0 for i in range(pad, 0, -1)]
    tail = [x[max(n - 1 - i, 0)] if n > 0 else 0.0 for i in range(1, pad + 1)]
    p = np.concatenate([head, x, tail])
    if p.size < 1024:
        return np.zeros((0, 128))
    nf = (p.size - 1024) // 256 + 1
    win = 0.5 * (1.0 - np.cos(2.0 * np.pi * np.arange(1024) / 1024.0))
    fr = np.stack([p[f * 256:f * 256 + 1024] * win for f in range(nf)])
    mag = np.sqrt(np.abs(np.fft.rfft(fr, axis=1)) ** 2 + 1e-9)

    def hz_to_mel(f):
        return np.where(f >= 1000.0, 15.0 + np.log(np.maximum(f, 1e-9) / 1000.0) / (np.log(6.4) / 27.0), f / (200.0 / 3.0))

    def mel_to_hz(m):
        return np.where(m >= 15.0, 1000.0 * np.exp((np.log(6.4) / 27.0) * (m - 15.0)), (200.0 / 3.0) * m)

    edges = mel_to_hz(np.linspace(hz_to_mel(np.float64(0.0)), hz_to_mel(np.float64(12000.0)), 130))
    freqs = np.arange(513) * 24000.0 / 1024.0
    fb = np.zeros((128, 513))
    for m in range(128):
        fl, fc, frr = edges[m], edges[m + 1], edges[m + 2]
        up = (freqs - fl) / (fc - fl)
        dn = (frr - freqs) / (frr - fc)
        w = np.where((freqs >= fl) & (freqs <= fc), up, np.where((freqs > fc) & (freqs <= frr), dn, 0.0))
        fb[m] = w * (2.0 / (frr - fl))
    return np.log(np.maximum(mag @ fb.T, 1e-5))


@pytest.mark.parametrize("n", [0, 100, 255, 256, 1000, 24000])
def test_mel_oracle_follows_the_spec(oracle, n):
    rng = np.random.default_rng(n)
    t = np.arange(n) / 24000.0
    audio = (0.3 * np.sin(2 * np.pi * 220.0 * t) + 0.1 * np.sin(2 * np.pi * 3100.0 * t) + 0.01 * rng.standard_normal(n)).astype(np.float32)
    got = oracle.mel(audio)
    ref = _mel_numpy_f64(audio)
    assert got.shape == ref.shape == ((0, 128) if n < 256 else ((n + 768 - 1024) // 256 + 1, 128))
    if n >= 256:
        assert np.abs(got - ref).max() <= 2e-3   # f32 DFT + f32 filterbank vs float64; log floor region included


def test_bf16_mfma_restatement_against_hardware_vectors(oracle):
    """tests/golden/bf16_mfma_mi355x.npz holds inputs and results of v_mfma_f32_16x16x32_bf16 measured on an MI355X
    (tests/golden/make_bf16_mfma_golden.py): ~1 000 vectors from random exponent spreads, cancelling +-2^E pairs, one and
    two active lane groups. The integer restatement in the oracle has to reproduce every one of them bit for bit."""
    import ctypes as C
    import os
    Z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bf16_mfma_mi355x.npz"))
    L = oracle.lib()
    a, b, c, d = Z["a"], Z["b"], Z["c"], Z["d"]
    assert a.shape[0] >= 900 and len(set(Z["set"].tolist())) >= 20
    for i in range(a.shape[0]):
        ai, bi = np.ascontiguousarray(a[i]), np.ascontiguousarray(b[i])
        for f in (L.q3o_mfma_bf16_dot32, L.q3o_mfma_bf16_dot32_ref):
            r = np.float32(f(ai.ctypes.data_as(C.POINTER(C.c_uint16)), bi.ctypes.data_as(C.POINTER(C.c_uint16)), float(c[i])))
            assert r.view(np.uint32) == d[i].view(np.uint32), (str(Z["set"][i]), i, float(r), float(d[i]))


def test_sampler_expf_vs_libm_boundary_flips(oracle):
    """the sampler evaluates exp with the spec'd q3_expf (max 2.2e-7 relative) where the reference calls libm's
    f32 exp (src/models/llama/mod.rs:720). A literal Python restatement of :666-772 with float32 arithmetic is run twice — once with
    the oracle's q3o_expf (and must then reproduce q3o_sample on every draw: that validates the restatement), once with libm's exp — over
    200 logit rows x 500 draws = 1e5 draws at temperature 0.7 / top-k 40 / top-p 0.9. The flips are counted and bounded."""
    L = oracle.lib()
    f32 = np.float32
    rng = np.random.default_rng(2026)

    def cdf(logits, T, k, p, expf):
        order = np.argsort(-logits, kind="stable")[:k]                      # :705-713 (stable descending, top-k)
        v = logits[order]
        e = np.array([expf(f32((x - v[0]) / f32(T))) for x in v], dtype=np.float32)   # :716-723
        s = f32(0)
        for x in e:
            s = f32(s + x)
        pr = np.array([f32(x / s) for x in e], dtype=np.float32)             # :726-731
        cum, cut = f32(0), len(pr)
        for i, x in enumerate(pr):                                           # :734-753
            cum = f32(cum + x)
            if cum >= f32(p):
                cut = i + 1
                break
        pr = pr[:cut]
        ns = f32(0)
        for x in pr:
            ns = f32(ns + x)
        pr = np.array([f32(x / ns) for x in pr], dtype=np.float32)
        c, out = f32(0), []
        for x in pr:
            c = f32(c + x)
            out.append(c)
        return order[:cut], np.array(out, dtype=np.float32)

    def pick(order, cum, r):                                                 # :756-770: first i with r < cum[i], else the first candidate
        i = int(np.searchsorted(cum, r, side="right"))
        return int(order[i]) if i < len(order) else int(order[0])
    spec = lambda x: f32(L.q3o_expf(float(x)))
    libm = lambda x: f32(np.exp(f32(x), dtype=np.float32))
    flips = total = 0
    for row in range(200):
        lg = (rng.standard_normal(2160) * 2.5).astype(np.float32)
        o1, c1 = cdf(lg, 0.7, 40, 0.9, spec)
        o2, c2 = cdf(lg, 0.7, 40, 0.9, libm)
        draws = rng.random(500).astype(np.float32)
        for j, r in enumerate(draws):
            a = pick(o1, c1, r)
            if j < 25:
                assert a == L.q3o_sample(oracle.ptr(lg, oracle.f32p), 2160, 0.7, 40, 0.9, float(r)), (row, j)
            flips += a != pick(o2, c2, r)
            total += 1
    print(f"sampler exp: q3_expf vs libm over {total} draws: {flips} different ids")
    assert total == 100000 and flips <= 20   # expected ~ 2 * 40 * 2.2e-7 * 1e5 < 2


def test_q8_0_quantiser_and_canonical_gemm(oracle):
    """Q8_0 row of the oracle: its C quantiser is ggml's reference rule (d = amax / 127 as f16, q = roundf(x / d), ties away
    from zero) and agrees with the numpy writer of tests/_gguf.py block for block; the canonical Q8 GEMM (per block: bf16 MFMA from zero,
    then fmaf with f32(d)) is the product with the de-quantised weights to f32 rounding; the Talker switched to Q8 still generates."""
    import _gguf as G
    rng = np.random.default_rng(0)
    w = (rng.standard_normal((48, 1024)) * 0.02).astype(np.float32)
    w[3, :32] = 0.0
    w[5, 32:64] = np.float32(0.5) * np.arange(32, dtype=np.float32)   # x / d lands on .5 ties
    q, d = oracle.quantize_q8_0(w)
    raw = G.quantize_q8_0(w).reshape(-1, 34)
    assert np.array_equal(d, raw[:, :2].copy().view(np.uint16).reshape(d.shape)) and np.array_equal(q, raw[:, 2:].copy().view(np.int8).reshape(q.shape))
    assert np.abs(q).max() == 127 and not q[3, :32].any() and d[3, 0] == 0
    deq = (q.astype(np.float64).reshape(48, 32, 32) * d.view(np.float16).astype(np.float64)[:, :, None]).reshape(48, 1024)
    assert np.abs(deq - w).max() <= np.abs(w).max() / 127.0 * 0.51 + 1e-4
    x = rng.standard_normal((7, 1024)).astype(np.float32)
    u = x.view(np.uint32).astype(np.uint64)
    xb = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)
    xf = (xb.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
    y = oracle.bgemm_q8(xb, q, d, None, 1024, 1e-6, 0)["y"]
    want = xf @ deq.T
    assert np.abs(y - want).max() <= 3e-5 * np.abs(want).max()
    from q3tts import _abi
    cfg = _abi.tiny_config(max_batch=1, n_ctx=64, with_vocoder=0)
    om = oracle.OracleModel(cfg.model, seed=0, n_ctx=64, n_threads=4)
    try:
        desc, keep = oracle.make_prompt_desc(np.arange(100, 108), spk_emb=((np.arange(cfg.model.d_embed) % 13 - 6) * 0.03125).astype(np.float32))
        pe = om.build_prompt(desc)
        bf, _ = om.generate(pe, temperature=0.0, max_steps=4, min_frames=4)
        om.set_talker_q8()
        q8, _ = om.generate(pe, temperature=0.0, max_steps=4, min_frames=4)
        assert q8.shape == bf.shape == (4, 16) and not np.array_equal(q8, bf)
    finally:
        om.close()
