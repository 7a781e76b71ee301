"""The N > 1 code of bench.py executed on the 1-GPU box (VERDICT r02, next #1c/d).

* backend nccl (RCCL) at WORLD_SIZE = 1: process group on the device, zero-copy view of the engine's packed device PCM, device-side i16
  conversion, uint8 all_gather / gather — every line of `gather_pcm_device` runs, and what it delivers must equal the reference's i16
  conversion (src/utils/audio.rs:35-37) of the host PCM of the same run.
* two ranks over gloo, both engines on GPU 0: per-utterance codec ids equal the one-rank run's for the same GLOBAL indices (sharding and
  global-index seeds: results do not depend on the number of ranks). This run also goes through `spawn_ranks`.
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(REPO, "bench.py")
COMMON = ["--tiny", "--steps", "1", "--warmup", "0", "--no-single", "--no-probe", "--no-cpu-baseline"]


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _run(args, env_extra, timeout=600):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), **env_extra)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        if k not in env_extra:
            env.pop(k, None)
    p = subprocess.run([sys.executable, BENCH] + args + COMMON, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_dist_branch_with_rccl_at_world_size_1():
    line = _run(["--gpus", "1", "--batch", "6"],
                {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port()),
                 "Q3TTS_FORCE_DIST": "1", "Q3TTS_BENCH_CHECK_GATHER": "1"})
    gc = line["gather_check"]
    print("RCCL world-1 gather:", gc, "gather ms/step", line["gather_ms_per_step"])
    assert gc["backend"] == "nccl" and gc["utterances"] == 6 and gc["samples"] > 6 * 25 * 1920 - 1 and gc["mismatching_samples"] == 0
    assert line["value_without_gather"] >= line["value"] > 0


def test_two_ranks_give_the_ids_of_one_rank_for_the_same_global_indices(tmp_path):
    a, b = str(tmp_path / "two"), str(tmp_path / "one")
    _run(["--gpus", "2", "--batch", "3"], {"Q3TTS_DIST_BACKEND": "gloo", "Q3TTS_ONE_GPU": "1", "Q3TTS_BENCH_DUMP": a})   # via spawn_ranks
    _run(["--gpus", "1", "--batch", "6"], {"Q3TTS_BENCH_DUMP": b})
    one = np.load(b + ".rank0.npz")
    assert sorted(one.files) == [f"g{i}" for i in range(6)]
    seen = set()
    for r in range(2):
        two = np.load(f"{a}.rank{r}.npz")
        assert sorted(two.files) == [f"g{i}" for i in range(r, 6, 2)]   # rank r owns {i : i mod 2 == r}
        for k in two.files:
            assert two[k].shape == one[k].shape and two[k].shape[0] >= 25 and np.array_equal(two[k], one[k]), k
            seen.add(k)
    assert len(seen) == 6


def test_node_api_one_device_gathers_the_reference_i16_pcm():
    """Multi-GPU behind the C ABI (q3tts_node_*) with the one device of this box: the same code path an 8-GPU node runs — engine
    thread, device-side i16 packing, ncclCommInitAll, the all-gather of the sample counts, the (here empty) send / receive group, one
    copy to the host. ids equal a plain engine's for the same requests; the gathered i16 equals the reference's conversion
    (src/utils/audio.rs:35-37) of that engine's f32 PCM; without the gather the node hands back the f32 PCM itself."""
    sys.path.insert(0, os.path.join(REPO, "qwen3-tts-rust_amd")); sys.path.insert(0, os.path.join(REPO, "tests"))
    import _oracle as O
    from q3tts import _abi, native
    cfg = _abi.tiny_config(max_batch=4, n_ctx=256, with_vocoder=1)
    rng = np.random.default_rng(77)
    reqs, keep = [], []
    for i in range(7):   # more requests than slots: slot re-use inside the node's engine
        spk = ((np.arange(cfg.model.d_embed) % 13 - 6) * 0.03125).astype(np.float32)
        desc, k = O.make_prompt_desc(rng.integers(0, 151643, size=int(rng.integers(3, 12))), spk_emb=spk)
        keep.append((desc, k))
        t = int(rng.integers(3, 14))
        reqs.append(dict(desc=desc, temperature=0.7, top_k=40, top_p=0.9, seed=500 + i, max_steps=20, min_frames=t, force_eos_at=t, want_pcm=1))
    eng = native.NativeEngine(cfg)
    try:
        plain = eng.generate_batch(reqs)
    finally:
        eng.close()
    node = native.NativeNode(cfg, [0])
    try:
        got = node.generate_batch(reqs, gather_i16=True)
        tm = node.timings()
        again = node.generate_batch(reqs, gather_i16=False)
    finally:
        node.close()
    assert tm.n_devices == 1 and tm.gathered_bytes == 2 * sum(o.n_samples for o in plain) and tm.gather_ms > 0
    for a, b, c in zip(plain, got, again):
        assert b.status == 0 and np.array_equal(a.codes, b.codes) and np.array_equal(a.codes, c.codes)
        want = np.trunc(np.clip(a.pcm.astype(np.float32) * np.float32(32767.0), -32768.0, 32767.0)).astype(np.int16)
        assert b.pcm is None and b.pcm_i16 is not None and np.array_equal(b.pcm_i16, want)
        assert c.pcm is not None and np.array_equal(c.pcm, a.pcm)
    print(f"node API, 1 device: 7 utterances, {tm.gathered_bytes} bytes gathered in {tm.gather_ms:.2f} ms (generate {tm.generate_ms:.1f} ms)")


def test_node_error_paths_leave_every_result_intact_and_freeable():
    """q3tts_node_generate_batch's error contract (include/q3tts.h). A request that fails by itself — prompt + max_steps beyond n_ctx —
    carries its own status while every other utterance of the batch comes back complete (ids equal a plain engine's, gathered i16 PCM
    present) and the failed one gathers nothing; a device-level failure (want_pcm on a node without a vocoder) returns the error with
    every result still valid to free and no i16 buffer allocated (NativeNode frees them all before raising)."""
    sys.path.insert(0, os.path.join(REPO, "qwen3-tts-rust_amd")); sys.path.insert(0, os.path.join(REPO, "tests"))
    import _oracle as O
    from q3tts import _abi, native
    cfg = _abi.tiny_config(max_batch=2, n_ctx=128, with_vocoder=1)
    spk = ((np.arange(cfg.model.d_embed) % 13 - 6) * 0.03125).astype(np.float32)
    rng = np.random.default_rng(5)
    keep, reqs = [], []
    for i in range(5):
        n_text = 110 if i == 2 else int(rng.integers(3, 9))    # request 2: 121 prompt rows + 12 steps > n_ctx = 128
        desc, k = O.make_prompt_desc(rng.integers(0, 151643, size=n_text), spk_emb=spk)
        keep.append((desc, k))
        reqs.append(dict(desc=desc, temperature=0.7, top_k=40, top_p=0.9, seed=900 + i, max_steps=12, min_frames=5 + i, force_eos_at=5 + i, want_pcm=1))
    eng = native.NativeEngine(cfg)
    try:
        plain = eng.generate_batch([r for i, r in enumerate(reqs) if i != 2])
    finally:
        eng.close()
    node = native.NativeNode(cfg, [0])
    try:
        for gather in (True, False):
            got = node.generate_batch(reqs, gather_i16=gather)
            assert got[2].status != 0 and got[2].codes.shape[0] == 0 and got[2].pcm is None and got[2].pcm_i16 is None
            for a, b in zip(plain, [g for i, g in enumerate(got) if i != 2]):
                assert b.status == 0 and np.array_equal(a.codes, b.codes)
                if gather:
                    want = np.trunc(np.clip(a.pcm.astype(np.float32) * np.float32(32767.0), -32768.0, 32767.0)).astype(np.int16)
                    assert np.array_equal(b.pcm_i16, want)
                else:
                    assert np.array_equal(b.pcm, a.pcm)
        again = node.generate_batch([r for i, r in enumerate(reqs) if i != 2], gather_i16=True)   # the node is usable after a failed request
        assert all(o.status == 0 for o in again) and all(np.array_equal(a.codes, b.codes) for a, b in zip(plain, again))
    finally:
        node.close()
    cfg0 = _abi.tiny_config(max_batch=2, n_ctx=128, with_vocoder=0)
    node0 = native.NativeNode(cfg0, [0])
    try:
        with pytest.raises(_abi.Q3Error):
            node0.generate_batch(reqs[:2], gather_i16=False)    # want_pcm without a vocoder: the device's whole call fails
        assert node0.last_failed_statuses == [-5, -5]   # Q3TTS_ERR_STATE: untouched results, valid to free
        ok = node0.generate_batch([dict(r, want_pcm=0) for r in reqs[:2]], gather_i16=False)    # ... and the node goes on working
        assert all(o.status == 0 and o.codes.shape[0] > 0 for o in ok)
    finally:
        node0.close()


def test_bench_node_mode_one_gpu():
    """bench.py --node (the q3tts_node_* leg) on the one GPU of this box."""
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--node", "--tiny", "--batch", "5", "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    line = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["node"]["gathered_bytes_per_step"] > 5 * 25 * 1920 * 2 - 1
