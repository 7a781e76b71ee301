#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X-native Qwen3-TTS hot path.

  python bench.py --gpus N --steps K --warmup W
  N > 1: one rank per GPU. Under torch.distributed.run the ranks come from the environment; a plain `python bench.py --gpus N`
  starts the N rank processes itself (children, before this process touches a GPU) and relays rank 0's JSON line.

A "step" is one pass of the hot path over one batch of synthetic utterances: BASELINE.json configs[2] per GPU (batch = 64
mixed-length prompts, temperature 0.7 / top-k 40 / top-p 0.9, prompt ids -> codec ids -> 24 kHz PCM); with N GPUs that is
configs[3] (N x 64 utterances sharded by global index, RCCL only gathers the PCM). Weights are seeded synthetic bf16 tensors of the
Qwen3-TTS-12Hz-1.7B shape (no checkpoints offline). Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REPO, "qwen3-tts-rust_amd"))
sys.path.insert(0, os.path.join(REPO, "tests"))

FRAME_SEC = 0.08  # 1 frame = 16 codes = 1920 samples @ 24 kHz (reference: src/tts/engine.rs:509-512,653)
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
BF16_MFMA_PEAK_TF = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak (the 5 PF headline includes 2:1 sparsity)
ROUND = "r04"


def vivian():
    with open(os.path.join(REPO, "tests", "golden", "speakers", "vivian.json")) as f:
        return np.asarray(json.load(f)["spk_emb"], dtype=np.float32)


def make_workload(n_per_gpu, rank, world, spk, keepalive, want_pcm, n_classes=64):
    """the GLOBAL list of n_per_gpu*world utterances is a function of the global index only; rank r
    owns {i : i mod world == r}, so per-utterance results do not depend on the number of GPUs. An utterance's prompt
    (n_text ~ U{8..64}) and forced length (n_frames ~ U{25..250}, EOS forced at the target) are drawn from its length CLASS
    c(i) = (i + i // 64) mod 64 (q3tts.dist.workload_class; 64 = the benchmark's utterances per GPU, a constant so that an utterance does
    not depend on the number of ranks or on --batch), its sampler seed is 1000 + i: with 64 utterances per GPU every rank's indices hit
    every class exactly once, so every rank — and N = 1, whose classes are its indices: the round-3 workload unchanged — runs the same
    64 lengths with different sampler streams (weak scaling measures the hardware, not the draw)."""
    from q3tts import dist as qd
    from q3tts.native import make_prompt_desc
    reqs, frames = [], []
    meta = []  # (global index, prompt ids, target frames, sampler seed): what the oracle needs to replay an utterance
    for gi in qd.shard_indices(n_per_gpu * world, rank, world):
        r = np.random.default_rng(977 * qd.workload_class(gi, n_classes) + 1)
        n_text = int(r.integers(8, 65))
        ids = r.integers(0, 151643, size=n_text)
        target = int(r.integers(25, 251))
        desc, keep = make_prompt_desc(ids, spk_emb=spk)
        keepalive.append((desc, keep))
        reqs.append(dict(desc=desc, temperature=0.7, top_k=40, top_p=0.9, seed=qd.global_seed(1000, gi), max_steps=256,
                         min_frames=target, force_eos_at=target, want_pcm=want_pcm))
        frames.append(target)
        meta.append((gi, ids, target, qd.global_seed(1000, gi)))
    make_workload.meta = meta
    return reqs, frames


def vocoder_macs_per_frame(v):
    """Multiply-accumulates of the vocoder per 12.5 Hz frame, from the loaded configuration (~2.56 G at the default shape)."""
    d, H, F = v.latent_dim, v.n_head * v.head_dim, v.d_ffn
    macs = v.pre_conv_kernel * v.codebook_dim * d
    macs += v.n_layer * (4 * d * H + 3 * d * F + 2 * min(v.sliding_window, 72) * H)
    pos = 1
    for u in range(v.n_upsample):
        r = v.upsample_ratios[u]
        pos *= r
        macs += pos * (d * d + 7 * d + 8 * d * d)          # ConvTranspose (per output position) + depthwise k7 + two pointwise convs
    macs += pos * 7 * d * v.decoder_dim
    ch = v.decoder_dim
    for b in range(v.n_dec_blocks):
        r = v.dec_rates[b]
        pos *= r
        co = ch // 2
        macs += pos * (2 * ch * co + 3 * (7 * co * co + co * co))
        ch = co
    macs += pos * 7 * ch
    return float(macs)


def cpu_baseline(cfg, spk, with_voc, gpu_codes, gpu_pcm):
    """The oracle (CPU restatement, kind "port") on bounded samples of config 1 (one utterance, n_text = 20, greedy): at 4 threads
    (what the reference configures: src/models/llama/mod.rs:420-428, src/tts/engine.rs:136) and at all host cores. The all-cores leg
    doubles as the parity check of the run: its codes must equal the GPU's single-utterance codes, and the PCM RMS is reported."""
    import ctypes as C
    import _oracle as O
    L = O.lib()
    ncores = os.cpu_count() or 1
    all_thr = min(ncores, 64)
    t0 = time.time()
    om = O.OracleModel(cfg.model, seed=cfg.synth_seed, n_ctx=256, n_threads=all_thr)
    t_load = time.time() - t0
    ids = np.random.default_rng(1234).integers(0, 151643, size=20)
    desc, keep = O.make_prompt_desc(ids, spk_emb=spk)
    pe = om.build_prompt(desc)
    legs = []
    codes_all = None
    for thr, nfr in ((all_thr, 16), (4, 6)):
        L.q3o_set_threads(thr)
        t0 = time.time()
        codes, _ = om.generate(pe, temperature=0.0, max_steps=nfr, min_frames=nfr)
        t_ar = time.time() - t0
        t_voc, pcm = 0.0, None
        if with_voc:
            v = L.q3o_vocoder_create(C.byref(cfg.vocoder), cfg.synth_seed, thr)
            cc = np.clip(codes, 0, cfg.vocoder.codebook_size - 1).astype(np.int32)
            pcm = np.zeros(cc.shape[0] * 1920 + 64, dtype=np.float32)
            t0 = time.time()
            n = L.q3o_vocoder_decode(v, O.ptr(cc, O.i32p), cc.shape[0], 1, O.ptr(pcm, O.f32p), pcm.size)
            t_voc = time.time() - t0
            L.q3o_vocoder_destroy(v)
            pcm = pcm[:n]
        wall = t_ar + t_voc
        legs.append({"value": round(codes.shape[0] * FRAME_SEC / wall, 4), "unit": "audio-sec/s", "cores": thr, "rtf": round(wall / (codes.shape[0] * FRAME_SEC), 3),
                     "sample": f"config 1: 1 utterance, n_text=20 (31 prompt rows), greedy, {codes.shape[0]} frames: decoder {t_ar:.1f}s (prefill included) + vocoder {t_voc:.1f}s"})
        if thr == all_thr:
            codes_all, pcm_all = codes, pcm
    om.close()
    parity = None
    if gpu_codes is not None:
        n = min(gpu_codes.shape[0], codes_all.shape[0])
        ids_equal = bool(np.array_equal(gpu_codes[:n], codes_all[:n]))
        parity = {"what": f"single-utterance leg vs the oracle, first {n} greedy frames of config 1 at the full shape", "ids_equal": ids_equal}
        if gpu_pcm is not None and pcm_all is not None:
            m = min(gpu_pcm.size, pcm_all.size, n * 1920)
            parity["pcm_rms_error"] = float(np.sqrt(np.mean((gpu_pcm[:m] - pcm_all[:m]) ** 2)))
            parity["pcm_signal_rms"] = float(np.sqrt(np.mean(pcm_all[:m] ** 2)))
        assert ids_equal, "codec ids of the GPU path differ from the CPU oracle at the benchmarked shape"
        # the tolerance tests/test_parity_gpu.py states for the full-shape vocoder (PCM_RMS_TOL_FULL): bf16 GEMM inputs against the oracle
        assert parity.get("pcm_rms_error", 0.0) <= 3.5e-3, f"PCM of the GPU path is off the CPU oracle: RMS error {parity.get('pcm_rms_error')}"
    main_leg = dict(legs[0])
    main_leg.update({"kind": "port", "host_cores": ncores,
                     "note": f"CPU restatement (oracle/), not llama.cpp/ORT: the reference's CPU path cannot run here; synthetic weight generation {t_load:.1f}s excluded",
                     "four_threads": legs[1]})
    return main_leg, parity


def batch_parity(cfg, spk, meta, outs, picks, n_frames=8):
    """Value check of the TIMED batch leg (BASELINE configs[2], sampled, 64 slots, row buckets, batched vocoder): the first `n_frames`
    frames of a few of its utterances against the oracle replaying the same prompt ids, sampler seed and controls. ids must be equal;
    the PCM of those frames (the vocoder is causal) within the full-shape tolerance. /root/reference/src/tts/engine.rs:545-642."""
    import ctypes as C
    import _oracle as O
    L = O.lib()
    thr = min(os.cpu_count() or 1, 64)
    om = O.OracleModel(cfg.model, seed=cfg.synth_seed, n_ctx=512, n_threads=thr)
    v = L.q3o_vocoder_create(C.byref(cfg.vocoder), cfg.synth_seed, thr) if cfg.with_vocoder else None
    rep = {"what": f"timed batch leg vs the oracle: first {n_frames} sampled frames of utterances {picks} (prompt ids, seed 1000 + index, temperature 0.7 / top-k 40 / top-p 0.9)",
           "ids_equal": True, "utterances": []}
    t0 = time.time()
    try:
        for i in picks:
            gi, ids, target, seed = meta[i]
            nf = min(n_frames, target)
            desc, keep = O.make_prompt_desc(ids, spk_emb=spk)
            pe = om.build_prompt(desc)
            ref, _ = om.generate(pe, temperature=0.7, top_k=40, top_p=0.9, seed=seed, max_steps=nf, min_frames=target, force_eos_at=target)
            got = outs[i].codes[:nf]
            eq = bool(ref.shape == got.shape and np.array_equal(ref, got))
            ent = {"index": int(gi), "prompt_rows": int(pe.shape[0]), "frames": int(nf), "ids_equal": eq}
            if v is not None and outs[i].pcm is not None and eq:
                cc = np.clip(ref, 0, cfg.vocoder.codebook_size - 1).astype(np.int32)
                pcm = np.zeros(cc.shape[0] * 1920 + 64, dtype=np.float32)
                L.q3o_vocoder_reset(v)
                n = L.q3o_vocoder_decode(v, O.ptr(cc, O.i32p), cc.shape[0], 1, O.ptr(pcm, O.f32p), pcm.size)
                ent["pcm_rms_error"] = float(np.sqrt(np.mean((outs[i].pcm[:n] - pcm[:n]) ** 2)))
                ent["pcm_signal_rms"] = float(np.sqrt(np.mean(pcm[:n] ** 2)))
            rep["utterances"].append(ent)
            rep["ids_equal"] = rep["ids_equal"] and eq
    finally:
        if v is not None:
            L.q3o_vocoder_destroy(v)
        om.close()
    rep["oracle_seconds"] = round(time.time() - t0, 1)
    assert rep["ids_equal"], f"codec ids of the timed batch leg differ from the CPU oracle: {rep['utterances']}"
    worst = max([u.get("pcm_rms_error", 0.0) for u in rep["utterances"]] or [0.0])
    assert worst <= 3.5e-3, f"PCM of the timed batch leg is off the CPU oracle: RMS error {worst}"
    return rep


def node_main(args):
    """`python bench.py --gpus N --node`: the same workload (N x batch utterances, the global list of make_workload) through the
    library's own multi-GPU entry points (q3tts_node_*, include/q3tts.h): ONE process, one engine + one host thread per GPU, requests
    sharded by index (i mod N), the i16 PCM gathered to GPU 0 with RCCL over xGMI. The driver's N > 1 runs use one process per GPU
    (torch.distributed); this leg measures the path a Rust host would call."""
    from q3tts import _abi, native
    n = args.gpus
    cfg = _abi.tiny_config(max_batch=min(64, args.batch), n_ctx=1024) if args.tiny else _abi.full_config_py()
    cfg.max_batch, cfg.n_ctx, cfg.max_steps_cap = min(64, args.batch), (1024 if args.tiny else args.n_ctx), 512
    cfg.with_vocoder = 1
    spk = vivian()[:cfg.model.d_embed]
    keepalive = []
    reqs, frames = make_workload(args.batch * n, 0, 1, spk, keepalive, 1)
    node = native.NativeNode(cfg, list(range(n)))
    for _ in range(args.warmup):
        node.generate_batch(reqs, gather_i16=True)
    t0 = time.perf_counter()
    gather_ms = gen_ms = 0.0
    for _ in range(args.steps):
        outs = node.generate_batch(reqs, gather_i16=True)
        tm = node.timings()
        gather_ms += tm.gather_ms; gen_ms += tm.generate_ms
    elapsed = time.perf_counter() - t0
    assert all(o.status == 0 for o in outs) and [o.n_frames for o in outs] == frames
    assert all(o.pcm_i16 is not None and o.pcm_i16.size == o.n_frames * 1920 for o in outs)
    audio = sum(frames) * FRAME_SEC * args.steps
    line = {"metric": "audio_sec_per_s", "value": round(audio / elapsed, 2), "unit": "audio-sec/s", "n_gpus": n, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[%d] through q3tts_node_* (one process, one engine + host thread per GPU): %d x %d mixed-length prompts sharded by index, "
                                   "temperature=0.7 top-k=40 top-p=0.9, prompt ids -> codec ids -> 24 kHz PCM -> i16 gathered to GPU 0 (RCCL) -> host" % (3 if n > 1 else 2, n, args.batch),
                       "utterances_per_gpu": args.batch, "n_ctx": cfg.n_ctx, "with_vocoder": True},
            "node": {"generate_ms_per_step": round(gen_ms / args.steps, 2), "gather_ms_per_step": round(gather_ms / args.steps, 3), "gathered_bytes_per_step": int(tm.gathered_bytes)},
            "value_without_gather": round(audio / max(elapsed - gather_ms * 1e-3, 1e-9), 2)}
    if args.tiny:
        line["invalid"] = "test run on the tiny shape"
    node.close()
    print(json.dumps(line), flush=True)


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N ranks as child processes (this parent never initialises a GPU) and
    relay rank 0's JSON line. Every child is polled: if one exits non-zero (bad device, out of memory) the others — which would sit in
    the RCCL rendezvous until its timeout — are terminated and the bench fails at once. Children are always fresh processes."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    out_path = os.path.join(REPO, "gpurun_out", f"bench_rank0_{os.getpid()}.out") if os.path.isdir(os.path.join(REPO, "gpurun_out")) else f"/tmp/bench_rank0_{os.getpid()}.out"
    out_f = open(out_path, "w+")
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=out_f if r == 0 else subprocess.DEVNULL, text=True))
    rc = 0
    live = list(range(n))
    while live:
        time.sleep(0.2)
        for r in list(live):
            code = procs[r].poll()
            if code is None:
                continue
            live.remove(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                sys.stderr.write(f"bench.py: rank {r} exited with {code}; stopping the other ranks\n")
                for o in live:
                    procs[o].terminate()
    if rc != 0:
        for p in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()
    out_f.seek(0)
    sys.stdout.write(out_f.read())
    sys.stdout.flush()
    out_f.close()
    try:
        os.remove(out_path)
    except OSError:
        pass
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=64, help="utterances per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-vocoder", action="store_true", help="codes only (diagnostic; the JSON line is then marked invalid)")
    ap.add_argument("--no-single", action="store_true", help="skip the batch=1 RTF / first-chunk leg")
    ap.add_argument("--n-ctx", type=int, default=4096)
    ap.add_argument("--no-q8", action="store_true", help="skip the Talker-in-Q8_0 leg (a second engine after the probe legs)")
    ap.add_argument("--no-probe", action="store_true", help="skip the in-situ dominant-kernel measurement (roofline.achieved falls back to the whole frame step)")
    ap.add_argument("--probe-only", nargs="?", const="talker", default=None, choices=["talker", "predictor", "vocoder"],
                    help=f"run only one probe leg (the commands profiled for profiles/{ROUND}/*): the Talker's gate/up (default), the Predictor's, or the vocoder alone")
    ap.add_argument("--probe-kind", type=int, default=0, help="with --probe-only talker / predictor: 0 gate/up GEMM (default), 1 QKV GEMM, 2 attention, 3 O projection, 4 down projection")
    ap.add_argument("--tiny", action="store_true", help="tests only: the small shape of the parity tests instead of the 1.7B shape (the JSON line is marked invalid)")
    ap.add_argument("--node", action="store_true", help="one process: the library's q3tts_node_* entry points drive all --gpus devices (instead of one rank process per GPU)")
    args = ap.parse_args()

    if args.node:
        return node_main(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    if os.environ.get("Q3TTS_BENCH_FAIL_RANK", "") == str(rank):  # tests/test_host_cpu.py: a rank that dies at start-up
        sys.exit(3)
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    # rehearsal switches for a 1-GPU box: Q3TTS_DIST_BACKEND=gloo keeps the collectives on the host, Q3TTS_ONE_GPU=1 puts
    # every rank's engine on GPU 0 (the driver's multi-GPU runs use neither: RCCL, one GPU per rank)
    backend = os.environ.get("Q3TTS_DIST_BACKEND", "nccl")
    if os.environ.get("Q3TTS_ONE_GPU", "0") not in ("", "0"):
        local_rank = 0
    # tests/test_dist_gpu.py: Q3TTS_FORCE_DIST=1 runs the N > 1 code (process group, device-side i16 gather) at WORLD_SIZE = 1 — RCCL
    # accepts a single rank, so a 1-GPU box executes every line of the collective path; Q3TTS_BENCH_CHECK_GATHER=1 keeps the host PCM
    # as well and compares what the gather delivered with it; Q3TTS_BENCH_DUMP=<prefix> writes every rank's codec ids by global index
    use_dist = world > 1 or os.environ.get("Q3TTS_FORCE_DIST", "0") not in ("", "0")
    check_gather = os.environ.get("Q3TTS_BENCH_CHECK_GATHER", "0") not in ("", "0")
    dump_prefix = os.environ.get("Q3TTS_BENCH_DUMP", "")
    if use_dist:
        import torch
        import torch.distributed as dist
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    from q3tts import _abi, native
    from q3tts import dist as qd
    cfg = _abi.tiny_config(max_batch=min(64, args.batch), n_ctx=1024) if args.tiny else _abi.full_config_py()
    cfg.device, cfg.max_batch, cfg.n_ctx, cfg.max_steps_cap = local_rank, min(64, args.batch), (1024 if args.tiny else args.n_ctx), 512
    cfg.with_vocoder = 0 if args.no_vocoder else 1
    eng = native.NativeEngine(cfg)
    spk = vivian()[:cfg.model.d_embed]
    keepalive = []
    dev_gather = bool(use_dist and cfg.with_vocoder)
    if dev_gather:
        eng.set_device_pcm(True)   # the gather reads the PCM where the vocoder wrote it; no per-rank host copy
    reqs, frames = make_workload(args.batch, rank, world, spk, keepalive, 0 if args.no_vocoder else (2 if dev_gather and not check_gather else 1))

    def sync_all():
        if dist is not None:
            import torch
            if backend == "nccl":
                torch.cuda.synchronize()
            dist.barrier()

    def gather_pcm(outs):
        """RCCL over xGMI: one all_gather of the lengths, one padded gather of the i16 PCM from device memory to rank 0, which then
        holds every utterance on its host (the only collective on the path). Returns the seconds spent (0 when N = 1)."""
        if dist is None or not dev_gather:
            return 0.0
        import torch
        t0 = time.perf_counter()
        rows = qd.device_pcm_tensor(eng, torch.device("cuda", local_rank))
        if backend != "nccl":
            rows = rows.cpu()
        g = qd.gather_pcm_device(dist, rows, [o.n_samples for o in outs], rank, world, as_i16=True)
        if rank == 0:
            host = [t.cpu() for t in g[0]]  # (rank 0 ends with every utterance's PCM in host memory)
            gather_pcm.last = (host, g[1].cpu())
        elif backend == "nccl":
            torch.cuda.synchronize()
        return time.perf_counter() - t0

    KINDS = {0: "gate/up GEMM", 1: "QKV GEMM", 2: "attention", 3: "O projection", 4: "down projection"}

    def probe_leg(mode=2, kind=0, engine=None, q8=False):
        """One launch, in situ: the same batch of 64 utterances for 24 forced frames (codes only, so nothing else shares the GPU), frame
        steps launched eagerly with HIP events on the decode stream around ONE launch per frame — block 0 of the Talker step (mode 2)
        or of the Predictor's pass 1 (mode 1); kind 0 gate/up GEMM, 1 QKV GEMM, 2 attention, 3 O projection, 4 down projection.
        Algorithmic bytes per launch follow SURVEY.md §8(d)'s accounting (weights streamed once per launch + the operand / result rows
        + for attention the K / V bytes of the live context), stated per kind below."""
        pe = engine or eng
        pe.probe(mode + 16 * kind)
        preqs = [dict(r, min_frames=24, force_eos_at=24, max_steps=32, want_pcm=0) for r in reqs]
        for _ in range(2):
            pouts = pe.generate_batch(preqs)
        ptm = pe.timings()
        pe.probe(0)
        wb = 1.0625 if q8 else 2.0   # bytes per weight: ggml Q8_0 = 34 bytes per 32 weights
        ab = 1.0625 if q8 else 2.0   # bytes per activation element of a GEMM operand (W8A8: the rows are Q8_0 blocks as well)
        assert all(o.status == 0 and o.n_frames == 24 for o in pouts)
        m = cfg.model
        M = len(preqs)
        if mode == 2:
            d, F, nq, nkv, hd, L, T_ctx = m.t_d_model, m.t_d_ffn, m.t_n_head * m.t_head_dim, m.t_n_kv_head * m.t_head_dim, m.t_head_dim, m.t_n_layer, ptm.mean_ctx_tokens
        else:
            d, F, nq, nkv, hd, L = m.p_d_model, m.p_d_ffn, m.p_n_head * m.p_head_dim, m.p_n_kv_head * m.p_head_dim, m.p_head_dim, m.p_n_layer
            T_ctx = 3.0 * M  # pass 1: 3 keys per utterance in the per-frame cache
        nqkv = nq + 2 * nkv
        if kind == 0:
            K, N = d, 2 * F; nbytes = wb * N * K + ab * M * K + 4.0 * M * (K // 16) + ab * M * (N // 2)   # weights + bf16 rows + tile partials + bf16 SwiGLU rows
        elif kind == 1:
            K, N = d, nqkv; nbytes = wb * N * K + ab * M * K + 4.0 * M * (K // 16) + 4.0 * M * N            # ... + f32 q/k/v rows
        elif kind == 3:
            K, N = nq, d; nbytes = wb * N * K + ab * M * K + 8.0 * M * N + ab * M * N + 4.0 * M * (N // 16)  # weights + bf16 rows + residual read/write + next norm inputs
        elif kind == 4:
            K, N = F, d; nbytes = wb * N * K + ab * M * K + 8.0 * M * N + ab * M * N + 4.0 * M * (N // 16)
        else:
            K, N = hd, nq; nbytes = 2.0 * 2.0 * nkv * T_ctx + 4.0 * M * nqkv + 2.0 * 2.0 * M * nkv + 2.0 * M * nq   # K + V of the context, f32 q/k/v rows, bf16 K/V append, bf16 out rows
        flops = 2.0 * M * K * N if kind != 2 else 2.0 * 2.0 * nq * T_ctx
        per_step = L * (1 if mode == 2 else (m.n_codebooks - 1))   # launches of this kind per frame step (pass A of the Predictor runs 2 M rows; its attention is k_attend_pair)
        return {"model": "Talker" if mode == 2 else "Predictor", "kind": KINDS[kind], "kernel_ms": ptm.probe_kernel_ms, "empty_ms": ptm.probe_empty_ms, "launches": int(ptm.probe_count),
                "rows": M, "K": K, "N": N, "flops": flops, "bytes": nbytes, "frame_step_ms": ptm.frame_step_ms, "per_step": per_step, "mean_ctx_tokens": ptm.mean_ctx_tokens}

    def vocoder_leg():
        ms = eng.vocoder_bench(min(64, cfg.max_batch), 8)
        fl = 2.0 * vocoder_macs_per_frame(cfg.vocoder)
        nfr = min(64, cfg.max_batch) * 4
        tf = fl * nfr / (ms * 1e-3) / 1e12
        return {"bound": "mfma", "achieved": round(tf, 1), "peak": BF16_MFMA_PEAK_TF, "unit": "TFLOP/s", "frac": round(tf / BF16_MFMA_PEAK_TF, 4), "traffic": None,
                "kernel": "the vocoder's kernels together (k_vgemm_lds / k_vgemm_small / k_voc_resunit / k_voc_*): one batched 4-frame call for 64 slots, "
                          "nothing else on the GPU", "ms_per_call": round(ms, 3), "frames_per_call": nfr, "algorithmic_flops_per_frame": int(fl),
                "how": "HIP events on the stream around 8 batched calls (q3tts_k_vocoder_bench); FLOPs from the loaded vocoder configuration"}

    if args.probe_only:
        pr = vocoder_leg() if args.probe_only == "vocoder" else probe_leg(2 if args.probe_only == "talker" else 1, args.probe_kind)
        if rank == 0:
            print(json.dumps({"probe": pr}), flush=True)
        eng.close()
        return

    for _ in range(args.warmup):
        gather_pcm(eng.generate_batch(reqs))
    sync_all()
    t0 = time.perf_counter()
    step_frames = 0
    dec_ms = steps_dev = bytes_step = flops_step = live = 0
    t_gather = 0.0
    utt_rtf = []
    for _ in range(args.steps):
        outs = eng.generate_batch(reqs)
        t_gather += gather_pcm(outs)
        step_frames += sum(o.n_frames for o in outs)
        utt_rtf += [o.total_ms / (o.n_frames * 80.0) for o in outs if o.n_frames > 0]
        tm = eng.timings()
        dec_ms += tm.decode_ms
        steps_dev += tm.frame_steps
        bytes_step, flops_step, live = tm.algo_bytes_per_step, tm.algo_flops_per_step, tm.mean_live_slots
        tm_last = tm   # (the informational legs below overwrite the engine's timings)
    sync_all()
    elapsed = time.perf_counter() - t0
    assert all(o.status == 0 for o in outs)
    assert [o.n_frames for o in outs] == frames, "forced lengths not honoured"
    timed_outs = outs  # the last timed step's results: value-checked against the oracle below (batch_parity)
    # Outside the timed region, for information: the same utterances fed CONTINUOUSLY — 3 x the step's requests in one q3tts_generate_batch
    # call, so freed slots are refilled at once and the batch drains only at the very end (a step of `value` above admits its 64 utterances
    # together and runs its mixed lengths down to one live row: mean live utterances ~40 of 64). Not the contract's metric: steps overlap.
    cont = None
    if rank == 0 and not args.no_probe and not args.tiny:
        t1 = time.perf_counter()
        couts = eng.generate_batch(reqs * 3)
        cdt = time.perf_counter() - t1
        ctm = eng.timings()
        assert all(o.status == 0 for o in couts)
        cont = {"what": "3 x the step's utterances in ONE generate_batch call over the same 64 slots (continuous batching: freed slots refilled at once); outside the timed region, "
                        "not the contract's metric", "utterances": len(couts), "audio_sec_per_s": round(sum(o.n_frames for o in couts) * FRAME_SEC / cdt, 2),
                "mean_live_utterances": round(ctm.mean_live_slots, 2), "frame_step_ms": round(ctm.frame_step_ms, 4)}
    gather_check = None
    if check_gather and dev_gather and rank == 0:
        # what the collective delivered for rank 0's own utterances == the reference's i16 conversion (src/utils/audio.rs:35-37) of the
        # host PCM of the same run
        host, lens = gather_pcm.last
        assert int(lens[0, 0]) == len(outs)
        bad = 0
        for j, o in enumerate(outs):
            want = np.trunc(np.clip(o.pcm.astype(np.float32) * np.float32(32767.0), -32768.0, 32767.0)).astype(np.int16)
            got = host[0][j, :int(lens[0, 1 + j])].numpy()
            assert int(lens[0, 1 + j]) == want.size == o.n_samples, (j, int(lens[0, 1 + j]), want.size)
            bad += int(np.count_nonzero(got != want))
        gather_check = {"backend": backend, "world": world, "utterances": len(outs), "samples": int(sum(o.n_samples for o in outs)), "mismatching_samples": bad}
        assert bad == 0, gather_check
    if dump_prefix:
        np.savez(f"{dump_prefix}.rank{rank}.npz", **{f"g{make_workload.meta[j][0]}": o.codes for j, o in enumerate(outs)})
    if dist is not None:
        import torch
        tdev = "cuda" if backend == "nccl" else "cpu"
        t = torch.tensor([elapsed, float(step_frames), elapsed - t_gather], dtype=torch.float64, device=tdev)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        elapsed, total_frames, elapsed_nog = float(tmax[0].item()), float(t[1].item()), float(tmax[2].item())
        # per-rank view of the weak-scaling workload: every rank runs the same 64 lengths (make_workload), so these rows must agree
        mine = torch.tensor([float(sum(frames)), float(steps_dev) / max(1, args.steps), float(max(frames)), dec_ms / max(1, steps_dev)], dtype=torch.float64, device=tdev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [{"rank": r, "frames": int(x[0].item()), "frame_steps": int(round(x[1].item())), "longest_utterance_frames": int(x[2].item()),
                     "frame_step_ms": round(float(x[3].item()), 4)} for r, x in enumerate(allr)]
    else:
        total_frames, elapsed_nog = float(step_frames), elapsed
        per_rank = None

    line = None
    if rank == 0:
        audio_sec = total_frames * FRAME_SEC
        value = audio_sec / elapsed
        frame_step_ms = dec_ms / max(1, steps_dev)
        hbm_gbs = bytes_step / (frame_step_ms * 1e-3) / 1e9 if frame_step_ms > 0 else 0.0
        mfma_tf = flops_step / (frame_step_ms * 1e-3) / 1e12 if frame_step_ms > 0 else 0.0
        tm = tm_last
        rt = np.asarray(utt_rtf) if utt_rtf else np.zeros(1)
        line = {
            "metric": "audio_sec_per_s", "value": round(value, 2), "unit": "audio-sec/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[%d]: batch=%d mixed-length prompts per GPU (n_text~U{8..64}, n_frames~U{25..250} "
                                   "EOS-forced), temperature=0.7 top-k=40 top-p=0.9, prompt ids -> codec ids -> 24 kHz PCM%s" %
                                   (3 if world > 1 else 2, args.batch, " on %d GPUs, utterances sharded by global index, one %s gather of the i16 PCM to rank 0" %
                                   (world, "RCCL" if backend == "nccl" else backend + " (host rehearsal, NOT RCCL)") if world > 1 else ""),
                       "shape": "Qwen3-TTS-12Hz-1.7B (28x2048 Talker, 5x1024 Predictor, 8-layer codec vocoder), seeded synthetic bf16 weights",
                       "utterances_per_gpu": args.batch, "n_ctx": args.n_ctx, "with_vocoder": not args.no_vocoder},
            "rtf_per_utterance": round(frame_step_ms / 80.0, 5),
            "rtf_per_utterance_what": "frame_step_ms / 80 ms: the real-time factor every live utterance of the batch advances at",
            "utterance_latency_rtf": {"what": "(batch start -> this utterance's PCM complete) / its audio seconds, over the utterances of the timed steps on rank 0",
                                      "mean": round(float(rt.mean()), 4), "p50": round(float(np.median(rt)), 4), "p95": round(float(np.percentile(rt, 95)), 4)},
            "frame_step_ms": round(frame_step_ms, 4),
            "stage_ms_last_step": {"prefill": round(tm.prefill_ms, 2), "decode": round(tm.decode_ms, 2), "vocoder_host_wait": round(tm.vocoder_ms, 2)},
            "frame_step": {
                "what": "one frame step = sample + 15 Predictor passes + Talker step over the live row bucket (graph replay; the vocoder shares the GPU on its own stream)",
                "ms": round(frame_step_ms, 4), "mean_live_utterances": round(live, 2), "mean_rows": round(tm.mean_rows, 2),
                "algorithmic_flops": int(flops_step), "algorithmic_bytes": int(bytes_step),
                "tflops": round(mfma_tf, 2), "hbm_GBs": round(hbm_gbs, 1), "frac_of_8TBs": round(hbm_gbs / HBM_PEAK_GBS, 4)},
        }
        if use_dist:
            line["per_rank"] = per_rank
            line["value_without_gather"] = round(audio_sec / elapsed_nog, 2)
            line["gather_ms_per_step"] = round(t_gather / args.steps * 1e3, 3)
        if cont:
            line["continuous_batching"] = cont
        if gather_check:
            line["gather_check"] = gather_check
        if args.tiny:
            line["invalid"] = "test run on the tiny shape"
        if not args.no_probe:
            # Every kernel kind of a decoder block, measured in situ (q3tts_k_probe): the bracket also times the closing event packet; an
            # EMPTY bracket on the same stream (empty_ms) bounds that overhead from above, so a launch's own period lies in
            # [kernel_ms - empty_ms, kernel_ms]; rocprofv3 and the in-kernel timestamps of tools/chain_stamps.hip put it at the lower end
            # (profiles/README.md). `achieved` uses bracket - empty bracket, both measured in this run.
            legs = [probe_leg(mode, kind) for mode in (2, 1) for kind in (0, 1, 2, 3, 4)]
            step_us = float(np.mean([p["frame_step_ms"] for p in legs])) * 1e3   # the probe legs' own (eager, codes-only, 64 rows) frame step
            by_kernel = []
            for p in legs:
                us = max(p["kernel_ms"] - p["empty_ms"], 1e-6) * 1e3
                gbs = p["bytes"] / (us * 1e-6) / 1e9
                by_kernel.append({"kernel": "%s %s" % (p["model"], p["kind"]), "launches_per_frame_step": p["per_step"], "us_per_launch": round(us, 2),
                                  "us_per_launch_whole_bracket": round(p["kernel_ms"] * 1e3, 2), "launches_timed": p["launches"],
                                  "algorithmic_bytes_per_launch": int(p["bytes"]), "achieved_GBs": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4),
                                  "frac_lower_bound": round(p["bytes"] / max(p["kernel_ms"] * 1e-3, 1e-12) / 1e9 / HBM_PEAK_GBS, 4),
                                  "share_of_step": round(p["per_step"] * us / step_us, 4), "M": p["rows"], "K": p["K"], "N": p["N"]})
            by_kernel.sort(key=lambda k: -k["share_of_step"])
            # The headline is quoted per kernel SYMBOL, the unit rocprofv3's kernel stats (profiles/r03/*_kernel_stats.csv) are in: kinds that run the
            # same k_bgemm instance at 64 rows (the launcher's cost model, q3_bgemm.hip) are one symbol. Launch-weighted means over its kinds.
            # The symbol of a GEMM kind is what the launcher's cost model picks for its shape (q3tts_k_bgemm_pick: asked, not assumed — a change of
            # the model, of the batch size or of Q3TTS_BG_LDS_CAP moves the names with it); kinds that share an instance are one symbol.
            import ctypes as _C
            EPI = {"gate/up GEMM": 2, "QKV GEMM": 0, "O projection": 1, "down projection": 1}
            sym_kinds = {}
            for p in legs:
                name = "%s %s" % (p["model"], p["kind"])
                if p["kind"] == "attention":
                    sym = "k_attend_gqa2" if p["model"] == "Talker" else "k_attend_small<2>"
                else:
                    o5 = (_C.c_int32 * 5)()
                    rc = eng.lib.q3tts_k_bgemm_pick(p["rows"], p["K"], p["N"], EPI[p["kind"]], 1 if p["model"] == "Talker" else 0, 0, o5)
                    assert rc == 0
                    sym = "k_bgemm_big" if o5[4] else "k_bgemm<%d, %d, %d, %s, false>" % (o5[0], o5[1], o5[2], "true" if o5[3] else "false")
                sym_kinds.setdefault(sym, []).append(name)
            symbols = [(sym, tuple(kinds)) for sym, kinds in sym_kinds.items()]
            traffic_src = step_traffic = None
            tj = {}
            tround = ROUND if os.path.exists(os.path.join(REPO, "profiles", ROUND, "pmc_traffic.json")) else "r03"   # (until this round's PMC passes are collected)
            tpath = os.path.join(REPO, "profiles", tround, "pmc_traffic.json")
            if os.path.exists(tpath):
                with open(tpath) as f:
                    tj = json.load(f)
                step_traffic = tj.get("frame_step")
                traffic_src = f"profiled offline (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over `bench.py --probe-only`, profiles/{tround}/pmc_traffic.json), not measured in this run"
            by_symbol = []
            for sym, kinds in symbols:
                ks = [k for k in by_kernel if k["kernel"] in kinds]
                if not ks:
                    continue
                n = float(sum(k["launches_per_frame_step"] for k in ks))
                wmean = lambda f: sum(k["launches_per_frame_step"] * f(k) for k in ks) / n
                ls = [[p for p in legs if "%s %s" % (p["model"], p["kind"]) == k["kernel"]][0] for k in ks]
                ents = [tj.get("by_kernel", {}).get(k["kernel"]) for k in ks]
                by_symbol.append({
                    "symbol": sym, "kinds": [k["kernel"] for k in ks], "launches_per_frame_step": int(n), "us": wmean(lambda k: k["us_per_launch"]),
                    "us_bracket": wmean(lambda k: k["us_per_launch_whole_bracket"]), "bytes": wmean(lambda k: k["algorithmic_bytes_per_launch"]),
                    "share": sum(k["share_of_step"] for k in ks), "launches_timed": sum(k["launches_timed"] for k in ks),
                    "empty_us": sum(k["launches_per_frame_step"] * l["empty_ms"] * 1e3 for k, l in zip(ks, ls)) / n,
                    "intensity": sum(l["flops"] for l in ls) / sum(l["bytes"] for l in ls),
                    "traffic": int(sum(k["launches_per_frame_step"] * e["hbm_bytes_per_launch"] for k, e in zip(ks, ents)) / n) if all(ents) else None,
                    "shapes": ["M=%d K=%d N=%d" % (k["M"], k["K"], k["N"]) for k in ks]})
            by_symbol.sort(key=lambda k: -k["share"])
            top = by_symbol[0]
            gbs, gbs_lb = top["bytes"] / (top["us"] * 1e-6) / 1e9, top["bytes"] / (top["us_bracket"] * 1e-6) / 1e9   # (bracket - empty: upper end; whole bracket: lower end)
            line["roofline_dominant_symbol"] = {
                "bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                "achieved_lower_bound": round(gbs_lb, 1), "frac_lower_bound": round(gbs_lb / HBM_PEAK_GBS, 4),
                "traffic": top["traffic"], "traffic_source": traffic_src, "share_of_step": round(top["share"], 4),
                "kernel": "%s = %s (%s), %d launches per frame step: the kernel SYMBOL with the largest share of the frame step's time (launches x period, summed over the "
                          "kinds that run this instance; launch-weighted means); bf16 ridge 2500 TF / 8 TB/s = 312 flop/B > %.0f flop/B => HBM-bound by SURVEY.md §8(d)'s streaming "
                          "accounting (the Predictor's 157 MB of weights are re-read 15 x per frame and stay Infinity-Cache resident, so its launches are latency-, not bandwidth-limited)" %
                          (top["symbol"], " + ".join(top["kinds"]), "; ".join(top["shapes"]), top["launches_per_frame_step"], top["intensity"]),
                "launch_us": round(top["us_bracket"], 2), "empty_bracket_us": round(top["empty_us"], 2), "launch_us_minus_empty_bracket": round(top["us"], 2),
                "launches_timed": top["launches_timed"], "algorithmic_bytes_per_launch": int(top["bytes"]),
                "how": "achieved = algorithmic bytes / (launch_us - empty_bracket_us); HIP events on the decode stream around this launch in block 0 of every frame step, eager frame steps, "
                       f"64 live utterances, codes only (q3tts_k_probe); rocprofv3 of the same leg: profiles/{ROUND}/probe_kernel_stats.csv"}
            # rocprofv3's own average duration per symbol over the same probe leg (profiles/<round>/probe_kernel_stats.csv, collected offline with
            # `rocprofv3 --kernel-trace --stats -- python3 bench.py --probe-only`): the figure quoted as us_per_launch when the file is there; the
            # live event brackets bound it from both sides (us_lower_bound = bracket - empty bracket, us_upper_bound = the whole bracket)
            prof_us = {}
            spath = os.path.join(REPO, "profiles", tround, "probe_kernel_stats.csv")
            if os.path.exists(spath):
                import csv
                with open(spath) as f:
                    for row in csv.DictReader(f):
                        nm = row["Name"].replace("void ", "").split("(")[0].strip()
                        prof_us[nm] = float(row["AverageNs"]) * 1e-3
            def sym_us(g):
                return prof_us.get(g["symbol"], g["us_bracket"])
            line["roofline_by_symbol"] = [{"symbol": g["symbol"], "kinds": g["kinds"], "launches_per_frame_step": g["launches_per_frame_step"],
                                           "us_per_launch": round(sym_us(g), 2), "us_per_launch_source": (f"rocprofv3 average, profiles/{tround}/probe_kernel_stats.csv (offline)" if g["symbol"] in prof_us else "whole event bracket of this run (upper bound)"),
                                           "us_lower_bound": round(g["us"], 2), "us_upper_bound": round(g["us_bracket"], 2),
                                           "algorithmic_bytes_per_launch": int(g["bytes"]), "frac": round(g["bytes"] / (sym_us(g) * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                                           "frac_upper_bound": round(g["bytes"] / (g["us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                                           "share_of_step": round(g["launches_per_frame_step"] * sym_us(g) / step_us, 4), "traffic": g["traffic"]} for g in by_symbol]
            line["roofline_by_kernel"] = by_kernel
            line["roofline_by_kernel_what"] = ("every kernel kind of a decoder block, sorted by share of the frame step (launches_per_frame_step x us_per_launch / the probe legs' frame step of "
                                               "%.0f us); us_per_launch = event bracket - empty bracket (the lower end of a launch's period: the in-kernel timestamps of "
                                               "tools/chain_stamps.hip, profiles/r03, put the periods 1-2 us higher), frac_lower_bound uses the whole bracket; the shares leave %.0f %% "
                                               "for that difference and for the kernels not listed (heads, sampler, projection, k_pred_next, pass-A attention: ~7 %%)" %
                                               (step_us, 100.0 * (1.0 - sum(k["share_of_step"] for k in by_kernel))))
            # Headline: the WHOLE frame step — the unit the hot path repeats (one hipGraph replay of 547 dependent launches). Per kernel symbol the
            # step's time is spread thin (the largest two symbols hold ~15 % each and swap places from box to box), so a single kernel's figure
            # says little about the step; the symbols are all listed in roofline_by_symbol / roofline_by_kernel, the largest one again in
            # roofline_dominant_symbol.
            fs_gbs = bytes_step / (step_us * 1e-6) / 1e9
            line["roofline"] = {"bound": "hbm", "achieved": round(fs_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(fs_gbs / HBM_PEAK_GBS, 4),
                                "traffic": step_traffic, "traffic_source": traffic_src, "share_of_step": 1.0,
                                "kernel": "one frame step = 547 dependent launches (sampler + projection, 15 Predictor passes, Talker step) at 64 rows, codes only, eager launches with nothing "
                                          "else on the GPU: SURVEY.md §8(d) algorithmic bytes of one step / its duration. By symbol: roofline_by_symbol (the largest: roofline_dominant_symbol)",
                                "launch_us": round(step_us, 1), "launches_timed": int(sum(p["launches"] for p in legs)), "algorithmic_bytes_per_launch": int(bytes_step),
                                "how": "HIP events on the decode stream around every frame step of the probe legs (q3tts_get_timings: frame_step_ms), mean over the legs; "
                                       f"rocprofv3 of one such step: profiles/{ROUND}/frame_step_timeline.txt; traffic: FETCH_SIZE / WRITE_SIZE of all its kernels (profiles/{ROUND}/pmc_traffic.json)"}
            line["roofline_frame_step"] = dict(line["roofline"], what="same object as `roofline` (kept under its round-2 name)", us=round(step_us, 1), algorithmic_bytes=int(bytes_step))
            line["frame_step_64_rows_codes_only_ms"] = round(step_us * 1e-3, 4)
            if cfg.with_vocoder:
                line["roofline_vocoder"] = vocoder_leg()
            if not args.no_q8:
                # The Talker in ggml Q8_0 x Q8_0 arithmetic on the device (q3tts_engine_config.talker_q8_0 = 2: W8A8; the reference's default
                # quantisation, src/tts/engine.rs:91-95): NOT the headline configuration (BASELINE configs name bf16) — a second engine, same
                # probe legs, so the halved weight stream is measured next to the bf16 one. ids bit-exact vs the oracle: tests/test_parity_gpu.py.
                eng.close()
                cfg8 = _abi.full_config_py()
                cfg8.device, cfg8.max_batch, cfg8.n_ctx, cfg8.max_steps_cap, cfg8.with_vocoder, cfg8.talker_q8_0 = local_rank, cfg.max_batch, args.n_ctx, 512, 0, 2
                e8 = native.NativeEngine(cfg8)
                l8 = [probe_leg(2, kind, engine=e8, q8=True) for kind in (0, 1, 3, 4)]
                preqs = [dict(r, min_frames=24, force_eos_at=24, max_steps=32, want_pcm=0) for r in reqs]
                for _ in range(2):
                    e8.generate_batch(preqs)
                g8 = e8.timings().frame_step_ms
                e8.close()
                ent = []
                for p in l8:
                    us = max(p["kernel_ms"] - p["empty_ms"], 1e-6) * 1e3
                    bf = [k for k in by_kernel if k["kernel"] == "%s %s" % (p["model"], p["kind"])][0]
                    ent.append({"kernel": "Talker %s" % p["kind"], "us_per_launch": round(us, 2), "bf16_us_per_launch": bf["us_per_launch"], "algorithmic_bytes_per_launch": int(p["bytes"]),
                                "bytes_per_us": round(p["bytes"] / us, 0), "achieved_GBs": round(p["bytes"] / (us * 1e-6) / 1e9, 1), "frac": round(p["bytes"] / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)})
                line["talker_q8_0"] = {"what": "the same 64-row frame step with the Talker in ggml's Q8_0 x Q8_0 arithmetic (talker_q8_0 = 2, W8A8: weights AND activations as Q8_0 blocks, "
                                               "1.0625 bytes per element, a block's product = its exact int32 sum on v_mfma_i32_16x16x32_i8 x f32(d_w) * f32(d_x): what llama.cpp computes for the "
                                               "reference's gguf_q8_0 directory); second engine, codes only; not the headline configuration",
                                       "frame_step_64_rows_codes_only_ms": round(g8, 4), "bf16_frame_step_64_rows_codes_only_ms": round(step_us * 1e-3, 4), "by_kernel": ent}
        else:
            line["roofline"] = {"bound": "hbm", "achieved": round(hbm_gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(hbm_gbs / HBM_PEAK_GBS, 4), "traffic": None,
                                "kernel": "whole frame step (no per-kernel probe in this run)"}
        if args.no_vocoder:
            line["invalid"] = "diagnostic run without the vocoder"

    # batch = 1 leg (BASELINE configs[1] / configs[4]): RTF and p50 first-chunk latency, outside the timed region
    gpu_codes = gpu_pcm = None
    if rank == 0 and not args.no_single:
        eng.close()
        cfg1 = _abi.full_config_py()
        cfg1.device, cfg1.max_batch, cfg1.n_ctx, cfg1.max_steps_cap = local_rank, 1, args.n_ctx, 512
        cfg1.with_vocoder = cfg.with_vocoder
        e1 = native.NativeEngine(cfg1)
        ids = np.random.default_rng(1234).integers(0, 151643, size=20)
        desc, keep = native.make_prompt_desc(ids, spk_emb=spk)
        firsts, rtfs = [], []
        for it in range(12):
            t1 = time.perf_counter()
            o = e1.generate(desc=desc, temperature=0.0, max_steps=64, min_frames=64, want_pcm=cfg.with_vocoder)
            dt = time.perf_counter() - t1
            if it >= 2:
                rtfs.append(dt / (o.n_frames * FRAME_SEC))
        gpu_codes, gpu_pcm = o.codes, o.pcm
        t1m = e1.timings()
        for it in range(54):   # first-chunk latency: median over >= 50 runs; 8 frames each
            o = e1.generate(desc=desc, temperature=0.0, max_steps=8, min_frames=8, want_pcm=cfg.with_vocoder)
            if it >= 2:
                firsts.append(o.first_chunk_ms)
        line["single_utterance"] = {"workload": "BASELINE.json configs[1]: 1 utterance, n_text=20, greedy, 64 frames (RTF over 10 runs); first chunk over %d runs" % len(firsts),
                                    "rtf_p50": round(float(np.median(rtfs)), 5), "first_chunk_ms_p50": round(float(np.median(firsts)), 2),
                                    "first_chunk_ms_p95": round(float(np.percentile(firsts, 95)), 2),
                                    "frame_step_ms": round(t1m.frame_step_ms, 4),
                                    "hbm_GBs": round(t1m.algo_bytes_per_step / (t1m.frame_step_ms * 1e-3) / 1e9, 1) if t1m.frame_step_ms else 0}
        # BASELINE configs[4], clone variant: a 3 s (72 000-sample) synthetic reference clip goes through the device
        # front-end (log-mel + speaker encoder + audio encoder, q3_clone.hip), then the ICL clone prompt is generated
        ccfg = _abi.CloneConfig()
        e1.lib.q3tts_clone_default_config(ccfg)
        e1.clone_init(ccfg)
        tt = np.arange(72000) / 24000.0
        clip = (0.2 * np.sin(2 * np.pi * 140.0 * tt) + 0.02 * np.random.default_rng(5).standard_normal(72000)).astype(np.float32)
        fe, cf = [], []
        for it in range(12):
            t1 = time.perf_counter()
            ref_codes = e1.audio_encode(clip)
            ref_emb = e1.speaker_encode(clip)
            fe_ms = (time.perf_counter() - t1) * 1e3
            dsc, keep2 = native.make_prompt_desc(ids, spk_emb=ref_emb, ref_codes=ref_codes.reshape(-1).astype(np.int32),
                                                 ref_text_ids=np.arange(1000, 1012))
            o = e1.generate(desc=dsc, temperature=0.0, max_steps=16, min_frames=16, want_pcm=cfg.with_vocoder)
            if it >= 2:
                fe.append(fe_ms)
                cf.append(o.first_chunk_ms)
        line["single_utterance"]["clone_variant"] = {
            "workload": "BASELINE.json configs[4] clone variant: 3 s reference clip -> mel + speaker encoder + audio encoder "
                        "(family-structure encoders, synthetic weights) -> ICL clone prompt (38 ref frames, 12 ref-text ids) -> first 4-frame chunk",
            "front_end_ms_p50": round(float(np.median(fe)), 2), "first_chunk_ms_p50": round(float(np.median(cf)), 2),
            "first_chunk_incl_front_end_ms_p50": round(float(np.median(np.asarray(fe) + np.asarray(cf))), 2)}
        e1.close()
    if rank == 0 and not args.no_cpu_baseline:  # (rank 0 only, after the timed region and its barrier; the other ranks wait at the last barrier)
        line["cpu_baseline"], parity = cpu_baseline(cfg, spk, cfg.with_vocoder, gpu_codes, gpu_pcm)
        if parity:
            line["parity_in_this_run"] = parity
        # (N > 1: the PCM stays on the device for the gather, so the batch leg is checked on ids only there)
        picks = sorted({0, len(timed_outs) // 2, len(timed_outs) - 1})
        line["parity_batch_leg"] = batch_parity(cfg, spk, make_workload.meta, timed_outs, picks)
    if rank == 0:
        # the scalars a reader needs, LAST in the object (a log tail keeps them): repeated from the entries above
        su = line.get("single_utterance", {})
        line["headline"] = {
            "frame_step_64_rows_codes_only_ms": line.get("frame_step_64_rows_codes_only_ms"),
            "roofline_frac": line.get("roofline", {}).get("frac"),
            "roofline_vocoder_frac": line.get("roofline_vocoder", {}).get("frac"), "vocoder_ms_per_call": line.get("roofline_vocoder", {}).get("ms_per_call"),
            "single_rtf_p50": su.get("rtf_p50"), "first_chunk_ms_p50": su.get("first_chunk_ms_p50"),
            "clone_first_chunk_ms_p50": su.get("clone_variant", {}).get("first_chunk_ms_p50"),
            "utterance_latency_rtf_mean": line["utterance_latency_rtf"]["mean"],
            "ms_per_step": line["ms_per_step"], "frame_step_ms": line["frame_step_ms"], "rtf_per_utterance": line["rtf_per_utterance"],
            "audio_sec_per_s": line["value"]}
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
