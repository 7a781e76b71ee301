"""CPU tests of the host logic: C-ABI export list, API mirror types (VoiceFile / AudioSample / TTSC cache / SamplerConfig),
speaker presets, utterance sharding and the PCM gather over gloo (world_size 2)."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)


def test_abi_exports_every_declared_symbol():
    from q3tts import _abi
    hdr = open(os.path.join(REPO, "include", "q3tts.h")).read()
    declared = sorted(set(re.findall(r"\b(q3tts_[a-z0-9_]+)\s*\(", hdr)))
    assert declared == sorted(_abi.SYMBOLS)
    lib = _abi.load_library()  # loads without a GPU; no compute call is made here
    for s in declared:
        assert hasattr(lib, s), s
    cfg = _abi.default_config()
    assert (cfg.model.t_n_layer, cfg.model.t_d_model, cfg.model.p_n_layer, cfg.model.n_codebooks, cfg.model.sample_limit) == (28, 2048, 5, 16, 2160)
    py = _abi.full_config_py()
    assert bytes(py.model) == bytes(cfg.model) and bytes(py.vocoder) == bytes(cfg.vocoder)


def test_product_path_fails_loudly_without_gpu_or_library(tmp_path):
    from q3tts import _abi, native
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_abi.Q3Error, match="no HIP device|hipGetDeviceCount|no CPU fallback"):
        native.NativeEngine(_abi.tiny_config())
    with pytest.raises(_abi.Q3Error, match="not found"):
        _abi.load_library(str(tmp_path / "missing.so"))


def test_voice_file_and_presets(tmp_path):
    from q3tts.api import VoiceFile
    v = VoiceFile.load(os.path.join(HERE, "golden", "speakers", "vivian.json"))  # preset: `spk_emb` alias, extra `spk_id` ignored
    assert v.name == "vivian" and len(v.speaker_embedding) == 2048 and v.audio_codes == [] and v.ref_text == ""
    emb = np.asarray(v.speaker_embedding, dtype=np.float32)
    assert np.array_equal((emb.view(np.uint32) & 0xFFFF), np.zeros(2048, dtype=np.uint32))  # bf16-representable values
    p = tmp_path / "v.json"
    VoiceFile.new("hi", [1, 2, 3], [0.5, -1.0]).with_metadata(name="x").save(p)
    w = VoiceFile.load(p)
    assert (w.ref_text, w.audio_codes, w.speaker_embedding, w.name) == ("hi", [1, 2, 3], [0.5, -1.0], "x")
    (tmp_path / "bad.json").write_text('{"name": "n"}')
    with pytest.raises(ValueError):
        VoiceFile.load(tmp_path / "bad.json")


def test_audio_sample_wav_roundtrip(tmp_path):
    from q3tts.api import AudioSample
    s = np.array([0.0, 0.5, -0.5, 1.0, -1.0, 2.0, -2.0, 1e-5], dtype=np.float32)
    a = AudioSample(s, 24000, 1)
    a.save_wav(tmp_path / "a.wav")
    b = AudioSample.load_wav(tmp_path / "a.wav")
    want = np.trunc(np.clip(s * np.float32(32767.0), -32768, 32767)).astype(np.int16)  # src/utils/audio.rs:35-37
    assert np.array_equal((b.samples * 32768.0).astype(np.int16), want)
    assert b.sample_rate == 24000 and b.channels == 1 and abs(a.duration() - 8 / 24000) < 1e-9


def test_ttsc_cache_roundtrip(tmp_path):
    from q3tts.api import load_cache, save_cache
    p = tmp_path / "ref.cache"
    save_cache(p, [1, -2, 2047, 2 ** 40], [0.25, -3.5])
    raw = p.read_bytes()
    assert raw[:4] == b"TTSC" and raw[4:8] == (1).to_bytes(4, "little") and len(raw) == 8 + 8 + 4 * 8 + 8 + 2 * 4
    assert load_cache(p) == ([1, -2, 2047, 2 ** 40], [0.25, -3.5])
    p.write_bytes(b"XXXX" + raw[4:])
    with pytest.raises(ValueError):
        load_cache(p)


def test_sampler_config_defaults():
    from q3tts.api import SamplerConfig
    c = SamplerConfig()
    assert (c.temperature, c.top_k, c.top_p, c.seed) == (0.7, 40, 0.9, None)  # src/tts/engine.rs:25-34


def test_sharding_is_a_partition_and_seeds_ignore_world_size():
    from q3tts import dist
    for n in (0, 1, 7, 64, 512):
        for world in (1, 2, 4, 8):
            parts = [dist.shard_indices(n, r, world) for r in range(world)]
            assert sorted(sum(parts, [])) == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1
    assert [dist.global_seed(1000, i) for i in dist.shard_indices(8, 1, 2)] == [1001, 1003, 1005, 1007]
    assert dist.global_seed(1000, 5) == 1005


def test_weak_scaling_workload_is_the_same_on_every_rank():
    """bench.py's workload: an utterance's prompt and forced length come from its length class
    c(i) = (i + i // 64) mod 64, its sampler seed from its global index. For G in {1, 2, 4, 8} and 64 utterances per rank every rank's
    indices {r + G j} hit every class exactly once — each rank (and N = 1, whose classes are its indices) runs the same 64 lengths, so
    value(G) / (G value(1)) measures the hardware and not the draw; seeds stay distinct across the whole job."""
    import bench
    from q3tts import dist as qd
    spk = np.zeros(2048, dtype=np.float32)
    base = None
    for G in (1, 2, 4, 8):
        seeds = []
        for r in range(G):
            idx = qd.shard_indices(64 * G, r, G)
            assert sorted(qd.workload_class(i, 64) for i in idx) == list(range(64)), (G, r)
            keep = []
            reqs, frames = bench.make_workload(64, r, G, spk, keep, 0)
            by_class = {qd.workload_class(gi, 64): (len(ids), int(t), tuple(int(x) for x in ids[:4])) for (gi, ids, t, sd) in bench.make_workload.meta}
            per_rank = [by_class[c] for c in range(64)]
            if base is None:
                base = per_rank
                assert sum(frames) == 9865 and [m[0] for m in bench.make_workload.meta] == list(range(64))   # N = 1: the round-3 workload, unchanged
            assert per_rank == base, (G, r)
            assert sorted(frames) == sorted(t for _, t, _ in base) and max(frames) == max(t for _, t, _ in base)
            seeds += [sd for (_, _, _, sd) in bench.make_workload.meta]
        assert sorted(seeds) == [1000 + i for i in range(64 * G)]
    # the node leg (one process, all utterances): same classes through n_classes
    keep = []
    reqs, frames = bench.make_workload(64 * 2, 0, 1, spk, keep, 0)
    assert sum(frames) == 2 * 9865
    # an utterance is a function of its global index alone (never of the rank count or of --batch): 2 ranks x 3 == 1 rank x 6
    keep = []
    bench.make_workload(6, 0, 1, spk, keep, 0)
    one = {m[0]: (tuple(m[1]), m[2], m[3]) for m in bench.make_workload.meta}
    for r in range(2):
        bench.make_workload(3, r, 2, spk, keep, 0)
        for m in bench.make_workload.meta:
            assert one[m[0]] == (tuple(m[1]), m[2], m[3])


def test_pcm_gather_over_gloo_world2(tmp_path):
    """The N > 1 path on CPU: 2 ranks, gloo, variable-length PCM gathered to rank 0 and re-assembled in global order."""
    out = tmp_path / "ok"
    env = dict(os.environ, Q3_GLOO_OUT=str(out), MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29617", os.path.join(HERE, "_gloo_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert out.read_text() == "ok 7"


# ---------------------------------------------------------------------------------------------------------------
# loader row: the engine's GGUF / NPY reader against the numpy restatement (tests/_gguf.py).
# Host-only entry point of the C ABI: needs no GPU.
# ---------------------------------------------------------------------------------------------------------------
def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def test_gguf_reader_matches_numpy_restatement(tmp_path):
    import _gguf as G
    from q3tts import native
    rng = np.random.default_rng(5)
    a = (rng.standard_normal((48, 96)) * 0.05).astype(np.float32)
    a[0, :4] = [0.0, -0.0, 6.1e-5, -65504.0]  # zero / f16 subnormal boundary / f16 max
    v = rng.standard_normal(96).astype(np.float32)
    path = str(tmp_path / "t.gguf")
    G.write(path, [("blk.0.attn_q.weight", a, G.F32), ("m.f16", a, G.F16), ("m.bf16", a, G.BF16), ("m.q8", a, G.Q8_0), ("v", v, G.F32)],
            meta={"general.architecture": "qwen3", "general.alignment": 32, "qwen3.block_count": 2, "tokenizer.ggml.tokens": ["a", "bc"],
                  "tokenizer.ggml.token_type": [1, 2, 3], "some.float": 1.5, "some.flag": True, "big": 2 ** 40})
    ref = G.read(path)
    for name, (want, ty) in ref.items():
        got, gty = native.k_gguf_read(path, name)
        assert gty == ty and got.shape == want.shape and np.array_equal(_bits(got), _bits(want)), name
    assert np.array_equal(_bits(ref["blk.0.attn_q.weight"][0]), _bits(a))                       # F32 is a byte copy
    assert np.array_equal(G.f32_to_bf16_bits(ref["m.bf16"][0]), G.f32_to_bf16_bits(a))            # BF16 = RNE of the f32 value
    assert np.abs(ref["m.q8"][0] - a).max() <= np.abs(a).max() / 127.0                          # Q8_0 block error bound
    G.write(str(tmp_path / "v2.gguf"), [("v", v, G.F32)], version=2)                                # v2 has the same layout
    assert np.array_equal(native.k_gguf_read(str(tmp_path / "v2.gguf"), "v")[0], v)
    np.save(tmp_path / "x.npy", a)                                                                # NPY fallback reader
    got, _ = native.k_gguf_read(str(tmp_path / "x.npy"))
    assert got.shape == a.shape and np.array_equal(_bits(got), _bits(a))


def test_gguf_k_quants_match_numpy_restatement(tmp_path):
    """Q4_K / Q5_K / Q6_K (the tensor types of the reference's gguf_q5_k_m directory, src/tts/engine.rs:91-95): the C++ reader against
    the independent numpy restatement of the published super-block layouts (PARITY UNPINNED: llama.cpp is not in the reference),
    bit for bit, plus the block error bounds and hand-built blocks that exercise every bit field."""
    import _gguf as G
    from q3tts import native
    rng = np.random.default_rng(9)
    a = (rng.standard_normal((12, 512)) * 0.05).astype(np.float32)
    a[3] = 0.0; a[4, :256] = np.abs(a[4, :256]); a[5, 256:] = -np.abs(a[5, 256:])      # all-zero block, all-positive / all-negative blocks
    path = str(tmp_path / "k.gguf")
    G.write(path, [("q4", a, G.Q4_K), ("q5", a, G.Q5_K), ("q6", a, G.Q6_K)])
    ref = G.read(path)
    for name, bound in (("q4", 1.0 / 15), ("q5", 1.0 / 31), ("q6", 1.0 / 31)):
        got, gty = native.k_gguf_read(path, name)
        want, ty = ref[name]
        assert gty == ty and np.array_equal(_bits(got), _bits(want)), name
        span = (a.reshape(-1, 32).max(axis=1) - np.minimum(a.reshape(-1, 32).min(axis=1), 0)).max()
        assert np.abs(got - a).max() <= 1.2 * bound * max(span, np.abs(a).max()), name
    # every bit field: random raw super-blocks (finite f16 scales) decoded by both readers
    for ty, bs in ((G.Q4_K, 144), (G.Q5_K, 176), (G.Q6_K, 210)):
        raw = rng.integers(0, 256, size=(6, bs), dtype=np.uint8)
        for b in raw:   # finite f16 scale fields, everything else random
            if ty == G.Q6_K:
                b[208:210] = np.array([np.float16(0.37)]).view(np.uint8)
            else:
                b[0:4] = np.array([np.float16(0.013), np.float16(0.21)]).view(np.uint8)
        want = G.decode(raw.tobytes(), ty, 6 * 256)
        p2 = str(tmp_path / ("raw%d.gguf" % ty))
        G.write(p2, [("t", np.zeros((6, 256), np.float32), ty)])
        blob = bytearray(open(p2, "rb").read())
        data_off = len(blob) - (raw.size + ((-raw.size) % 32))
        blob[data_off:data_off + raw.size] = raw.tobytes()
        open(p2, "wb").write(bytes(blob))
        got, _ = native.k_gguf_read(p2, "t")
        assert np.array_equal(_bits(got.reshape(-1)), _bits(want)), ty


def test_gguf_reader_errors_are_loud(tmp_path):
    import struct
    import _gguf as G
    from q3tts import _abi, native
    v = np.arange(64, dtype=np.float32)
    good = str(tmp_path / "g.gguf")
    G.write(good, [("v", v, G.F32)])
    with pytest.raises(_abi.Q3Error, match="missing"):
        native.k_gguf_read(good, "nope")
    with pytest.raises(_abi.Q3Error, match="cannot open"):
        native.k_gguf_read(str(tmp_path / "absent.gguf"), "v")
    raw = open(good, "rb").read()
    (tmp_path / "magic.gguf").write_bytes(b"GGML" + raw[4:])
    with pytest.raises(_abi.Q3Error, match="not a GGUF"):
        native.k_gguf_read(str(tmp_path / "magic.gguf"), "v")
    (tmp_path / "v1.gguf").write_bytes(raw[:4] + struct.pack("<I", 1) + raw[8:])
    with pytest.raises(_abi.Q3Error, match="version"):
        native.k_gguf_read(str(tmp_path / "v1.gguf"), "v")
    (tmp_path / "cut.gguf").write_bytes(raw[:-100])
    with pytest.raises(_abi.Q3Error, match="outside the file"):
        native.k_gguf_read(str(tmp_path / "cut.gguf"), "v")
    # a tensor type the reader does not take (10 = Q2_K) is listed but refused when asked for; a K-quant row that is not a
    # multiple of the 256-element super-block likewise
    G.write(str(tmp_path / "kq.gguf"), [("v", v, G.F32)], raw_types={"v": 10})
    with pytest.raises(_abi.Q3Error, match="unsupported ggml type 10"):
        native.k_gguf_read(str(tmp_path / "kq.gguf"), "v")
    G.write(str(tmp_path / "kq2.gguf"), [("v", v, G.F32)], raw_types={"v": 13})
    with pytest.raises(_abi.Q3Error, match="unsupported ggml type 13"):
        native.k_gguf_read(str(tmp_path / "kq2.gguf"), "v")
    np.save(tmp_path / "f64.npy", v.astype(np.float64))
    with pytest.raises(_abi.Q3Error, match="f32"):
        native.k_gguf_read(str(tmp_path / "f64.npy"))


def test_bench_spawn_ranks_fails_fast_when_a_rank_dies():
    """bench.py --gpus 2 without a launcher: rank 1 exits at start-up; rank 0 would wait in the rendezvous until its timeout. The
    parent polls every child, terminates the survivor and returns non-zero within seconds (no GPU is touched before the rendezvous)."""
    import subprocess
    import sys
    import time
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, Q3TTS_DIST_BACKEND="gloo", Q3TTS_BENCH_FAIL_RANK="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--tiny", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=240)
    assert p.returncode != 0 and "rank 1 exited with 3" in p.stderr, (p.returncode, p.stderr[-500:])
    assert time.time() - t0 < 120


def test_node_shard_order_matches_the_python_sharding():
    """q3tts_node_shard (host only, no GPU): device r of G owns {i : i mod G == r} in order — the partition q3tts/dist.py and bench.py
    use across processes; sharding followed by reassembly is the identity for every (n, G), empty shards included."""
    from q3tts import dist as qd
    from q3tts import native
    for n in (0, 1, 7, 64, 65, 512):
        for G in (1, 2, 3, 8):
            parts = [native.node_shard(n, G, r) for r in range(G)]
            assert parts == [qd.shard_indices(n, r, G) for r in range(G)]
            assert sorted(i for p in parts for i in p) == list(range(n))
            assert qd.reassemble([[f"u{i}" for i in p] for p in parts], n, G) == [f"u{i}" for i in range(n)]
    with pytest.raises(ValueError):
        native.node_shard(4, 2, 2)
