#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X-native Qwen3-TTS hot path.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU; RCCL only gathers the output PCM)

A "step" is one pass of the hot path over one batch of synthetic utterances: BASELINE.json configs[2]
(batch = 64 mixed-length prompts per GPU, temperature 0.7 / top-k 40 / top-p 0.9, prompt -> codec ids -> 24 kHz PCM).
Weights are seeded synthetic bf16 tensors of the Qwen3-TTS-12Hz-1.7B shape (no checkpoints offline).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REPO, "qwen3-tts-rust_amd"))
sys.path.insert(0, os.path.join(REPO, "tests"))

FRAME_SEC = 0.08  # 1 frame = 16 codes = 1920 samples @ 24 kHz (reference: src/tts/engine.rs:509-512,653)
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
BF16_MFMA_PEAK_TF = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA peak (the 5 PF headline includes 2:1 sparsity)


def vivian():
    with open(os.path.join(REPO, "tests", "golden", "speakers", "vivian.json")) as f:
        return np.asarray(json.load(f)["spk_emb"], dtype=np.float32)


def make_workload(n_per_gpu, rank, world, spk, keepalive):
    """the GLOBAL list of n_per_gpu*world utterances is a function of the global index
    only (n_text ~ U{8..64}, n_frames ~ U{25..250} with EOS forced at the target, sampler seed 1000 + index);
    rank r owns {i : i mod world == r}, so per-utterance results do not depend on the number of GPUs."""
    from q3tts import dist as qd
    from q3tts.native import make_prompt_desc
    reqs, frames = [], []
    for gi in qd.shard_indices(n_per_gpu * world, rank, world):
        r = np.random.default_rng(977 * gi + 1)
        n_text = int(r.integers(8, 65))
        ids = r.integers(0, 151643, size=n_text)
        target = int(r.integers(25, 251))
        desc, keep = make_prompt_desc(ids, spk_emb=spk)
        keepalive.append((desc, keep))
        reqs.append(dict(desc=desc, temperature=0.7, top_k=40, top_p=0.9, seed=qd.global_seed(1000, gi), max_steps=256,
                         min_frames=target, force_eos_at=target, want_pcm=1))
        frames.append(target)
    return reqs, frames


def cpu_baseline(cfg, spk, with_voc, threads):
    """The oracle (CPU restatement, kind "port") on a bounded sample of config 1: one utterance, n_text = 20, greedy."""
    import ctypes as C
    import _oracle as O
    n_frames = 16  # ~10 s of CPU work on the box's host cores (bounded sample)
    t0 = time.time()
    om = O.OracleModel(cfg.model, seed=cfg.synth_seed, n_ctx=256, n_threads=threads)
    t_load = time.time() - t0
    ids = np.random.default_rng(1234).integers(0, 151643, size=20)
    desc, keep = O.make_prompt_desc(ids, spk_emb=spk)
    t0 = time.time()
    pe = om.build_prompt(desc)
    codes, _ = om.generate(pe, temperature=0.0, max_steps=n_frames, min_frames=n_frames)
    t_ar = time.time() - t0
    t_voc = 0.0
    L = O.lib()
    if with_voc and hasattr(L, "q3o_vocoder_create"):
        v = L.q3o_vocoder_create(C.byref(cfg.vocoder), cfg.synth_seed, threads)
        cc = np.clip(codes, 0, cfg.vocoder.codebook_size - 1).astype(np.int32)
        pcm = np.zeros(cc.shape[0] * 1920 + 64, dtype=np.float32)
        t0 = time.time()
        L.q3o_vocoder_decode(v, O.ptr(cc, O.i32p), cc.shape[0], 1, O.ptr(pcm, O.f32p), pcm.size)
        t_voc = time.time() - t0
        L.q3o_vocoder_destroy(v)
    om.close()
    wall = t_ar + t_voc
    return {"value": round(codes.shape[0] * FRAME_SEC / wall, 4), "unit": "audio-sec/s", "cores": threads, "kind": "port",
            "sample": f"config 1: 1 utterance, n_text=20 (31 prompt rows), greedy, {codes.shape[0]} frames, AR {t_ar:.1f}s + vocoder "
                      f"{t_voc:.1f}s (synthetic weight generation {t_load:.1f}s excluded); CPU restatement, not llama.cpp/ORT",
            "rtf": round(wall / (codes.shape[0] * FRAME_SEC), 3)}


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
    ap.add_argument("--no-probe", action="store_true", help="skip the in-situ dominant-kernel measurement (roofline.achieved falls back to the whole frame step)")
    ap.add_argument("--probe-only", nargs="?", const="talker", default=None, choices=["talker", "predictor"],
                    help="run only the probe leg (the command profiled for profiles/*/probe_kernel_stats.csv): the Talker's gate/up (default) or the Predictor's")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    # rehearsal switches for a 1-GPU box: Q3TTS_DIST_BACKEND=gloo keeps the collectives on the host, Q3TTS_ONE_GPU=1 puts
    # every rank's engine on GPU 0 (the driver's multi-GPU runs use neither: RCCL, one GPU per rank)
    backend = os.environ.get("Q3TTS_DIST_BACKEND", "nccl")
    if os.environ.get("Q3TTS_ONE_GPU", "0") not in ("", "0"):
        local_rank = 0
    tdev = "cuda" if backend == "nccl" else "cpu"
    if world > 1:
        import torch
        import torch.distributed as dist
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    from q3tts import _abi, native
    cfg = _abi.full_config_py()
    cfg.device, cfg.max_batch, cfg.n_ctx, cfg.max_steps_cap = local_rank, min(64, args.batch), args.n_ctx, 512
    cfg.with_vocoder = 0 if args.no_vocoder else 1
    eng = native.NativeEngine(cfg)
    spk = vivian()
    keepalive = []
    reqs, frames = make_workload(args.batch, rank, world, spk, keepalive)
    if args.no_vocoder:
        for r in reqs:
            r["want_pcm"] = 0

    def sync_all():
        if dist is not None:
            import torch
            if tdev == "cuda":
                torch.cuda.synchronize()
            dist.barrier()

    def gather_pcm(outs):
        """RCCL over xGMI: lengths all-gather + padded gather of the PCM to rank 0 (the only collective on the path)."""
        if dist is None:
            return
        import torch
        from q3tts import dist as qd
        qd.gather_pcm(dist, [o.pcm if o.pcm is not None else np.zeros(0, dtype=np.float32) for o in outs], rank, world,
                      device=tdev, dtype=torch.float16, to_numpy=False)

    def probe_leg(mode=2):
        """One GEMM, in situ: the same batch of 64 utterances for 24 forced frames (codes only, so nothing else shares the GPU),
        frame steps launched eagerly with HIP events on the decode stream around it. mode 2: the Talker's layer-0 gate/up GEMM
        (k_bgemm, M = 64, K = 2048, N = 12288 — the largest GEMM of the frame step); mode 1: the Predictor's pass-1 / layer-0 gate/up
        (k_bgemm, M = 64, K = 1024, N = 6144)."""
        eng.probe(mode)
        preqs = [dict(r, min_frames=24, force_eos_at=24, max_steps=32, want_pcm=0) for r in reqs]
        for _ in range(2):
            pouts = eng.generate_batch(preqs)
        ptm = eng.timings()
        eng.probe(0)
        assert all(o.status == 0 and o.n_frames == 24 for o in pouts)
        m = cfg.model
        rows = len(preqs)
        K, N = (m.t_d_model, 2 * m.t_d_ffn) if mode == 2 else (m.p_d_model, 2 * m.p_d_ffn)
        flops = 2.0 * rows * K * N
        nbytes = 2.0 * N * K + 2.0 * rows * K + 4.0 * rows * (K // 16) + 2.0 * rows * (N // 2)  # weights + bf16 rows + tile partials + bf16 SwiGLU rows
        return {"kernel_ms": ptm.probe_kernel_ms, "empty_ms": ptm.probe_empty_ms, "launches": int(ptm.probe_count), "rows": rows, "K": K, "N": N, "flops": flops, "bytes": nbytes}

    if args.probe_only:
        pr = probe_leg(2 if args.probe_only == "talker" else 1)
        if rank == 0:
            print(json.dumps({"probe": pr}), flush=True)
        eng.close()
        return

    for _ in range(args.warmup):
        outs = eng.generate_batch(reqs)
        gather_pcm(outs)
    sync_all()
    t0 = time.perf_counter()
    step_frames = 0
    dec_ms = steps_dev = bytes_step = flops_step = live = 0
    for _ in range(args.steps):
        outs = eng.generate_batch(reqs)
        gather_pcm(outs)
        step_frames += sum(o.n_frames for o in outs)
        tm = eng.timings()
        dec_ms += tm.decode_ms
        steps_dev += tm.frame_steps
        bytes_step, flops_step, live = tm.algo_bytes_per_step, tm.algo_flops_per_step, tm.mean_live_slots
    sync_all()
    elapsed = time.perf_counter() - t0
    assert all(o.status == 0 for o in outs)
    assert [o.n_frames for o in outs] == frames, "forced lengths not honoured"
    if dist is not None:
        import torch
        t = torch.tensor([elapsed, float(step_frames)], dtype=torch.float64, device=tdev)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        elapsed, total_frames = float(tmax[0].item()), float(t[1].item())
    else:
        total_frames = float(step_frames)

    line = None
    if rank == 0:
        audio_sec = total_frames * FRAME_SEC
        value = audio_sec / elapsed
        frame_step_ms = dec_ms / max(1, steps_dev)
        hbm_gbs = bytes_step / (frame_step_ms * 1e-3) / 1e9 if frame_step_ms > 0 else 0.0
        mfma_tf = flops_step / (frame_step_ms * 1e-3) / 1e12 if frame_step_ms > 0 else 0.0
        # arithmetic intensity = flops/bytes against the bf16 ridge 2500 TF / 8 TB/s = 312 flop/B: the decoder is HBM-bound at every batch in scope
        mfma_bound = bytes_step > 0 and flops_step / bytes_step > BF16_MFMA_PEAK_TF * 1e3 / HBM_PEAK_GBS
        tm = eng.timings()
        line = {
            "metric": "audio_sec_per_s", "value": round(value, 2), "unit": "audio-sec/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[2]: batch=%d mixed-length prompts per GPU (n_text~U{8..64}, n_frames~U{25..250} "
                                   "EOS-forced), temperature=0.7 top-k=40 top-p=0.9, prompt ids -> codec ids -> 24 kHz PCM%s" %
                                   (args.batch, " + RCCL PCM gather" if world > 1 else ""),
                       "shape": "Qwen3-TTS-12Hz-1.7B (28x2048 Talker, 5x1024 Predictor, 8-layer codec vocoder), seeded synthetic bf16 weights",
                       "utterances_per_gpu": args.batch, "n_ctx": args.n_ctx, "with_vocoder": not args.no_vocoder},
            "rtf_per_utterance": round(frame_step_ms / 80.0, 5),
            "frame_step_ms": round(frame_step_ms, 4),
            "stage_ms_last_step": {"prefill": round(tm.prefill_ms, 2), "decode": round(tm.decode_ms, 2), "vocoder_host_wait": round(tm.vocoder_ms, 2)},
            "frame_step": {
                "what": "one frame step = sample + 15 Predictor passes + Talker step over the live row bucket (graph replay)",
                "ms": round(frame_step_ms, 4), "mean_live_utterances": round(live, 2), "mean_rows": round(tm.mean_rows, 2),
                "algorithmic_flops": int(flops_step), "algorithmic_bytes": int(bytes_step),
                "tflops": round(mfma_tf, 2), "frac_of_bf16_mfma_peak": round(mfma_tf / BF16_MFMA_PEAK_TF, 4),
                "hbm_GBs": round(hbm_gbs, 1), "frac_of_8TBs": round(hbm_gbs / HBM_PEAK_GBS, 4), "mfma_bound": bool(mfma_bound)},
        }
        if not args.no_probe:
            pr = probe_leg(2)
            # The bracket also times the closing event packet. An EMPTY bracket on the same stream (empty_ms) bounds that
            # overhead from above, so the kernel's own duration lies in [kernel_ms - empty_ms, kernel_ms]; rocprofv3 puts it
            # in between (profiles/README.md). `achieved` uses the whole bracket: a lower bound on the kernel's rate.
            k_ms = pr["kernel_ms"]
            k_gbs = pr["bytes"] / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
            traffic, traffic_src = None, None
            tpath = os.path.join(REPO, "profiles", "r02", "pmc_traffic.json")
            if os.path.exists(tpath):
                with open(tpath) as f:
                    traffic = json.load(f).get("hbm_bytes_per_launch")
                traffic_src = "profiled offline (two rocprofv3 --pmc passes over `bench.py --probe-only`, profiles/r02/pmc_traffic.json), not measured in this run"
            line["roofline"] = {
                "bound": "hbm", "achieved": round(k_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(k_gbs / HBM_PEAK_GBS, 4),
                "traffic": traffic, "traffic_source": traffic_src,
                "kernel": "k_bgemm: Talker gate/up GEMM (row scale of the split RMSNorm + SwiGLU epilogue) on v_mfma_f32_16x16x32_bf16, M=%d K=%d N=%d, "
                          "28 launches per frame step, the largest GEMM of the step; bf16 ridge 2500 TF / 8 TB/s = 312 flop/B > %.0f flop/B => HBM-bound" %
                          (pr["rows"], pr["K"], pr["N"], pr["flops"] / pr["bytes"]),
                "launch_us": round(k_ms * 1e3, 2), "empty_bracket_us": round(pr["empty_ms"] * 1e3, 2),
                "launch_us_minus_empty_bracket": round((pr["kernel_ms"] - pr["empty_ms"]) * 1e3, 2), "launches_timed": pr["launches"],
                "algorithmic_flops_per_launch": int(pr["flops"]), "algorithmic_bytes_per_launch": int(pr["bytes"]),
                "tflops": round(pr["flops"] / (k_ms * 1e-3) / 1e12, 2) if k_ms > 0 else 0.0,
                "how": "HIP events on the decode stream around every launch of this kernel in layer 0 of the Talker step, eager frame steps, "
                       "64 live utterances, codes only (q3tts_k_probe mode 2); rocprofv3 of the same leg: profiles/r02/probe_kernel_stats.csv"}
            pb = probe_leg(1)
            b_ms = pb["kernel_ms"]
            b_gbs = pb["bytes"] / (b_ms * 1e-3) / 1e9 if b_ms > 0 else 0.0
            line["roofline_predictor_kernel"] = {
                "bound": "hbm", "achieved": round(b_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(b_gbs / HBM_PEAK_GBS, 4), "traffic": None,
                "kernel": "k_bgemm: Predictor gate/up GEMM, M=%d K=%d N=%d (75 launches per frame step; weights re-read 15x per frame, Infinity-Cache resident)" % (pb["rows"], pb["K"], pb["N"]),
                "launch_us": round(b_ms * 1e3, 2), "empty_bracket_us": round(pb["empty_ms"] * 1e3, 2),
                "launch_us_minus_empty_bracket": round((pb["kernel_ms"] - pb["empty_ms"]) * 1e3, 2), "launches_timed": pb["launches"],
                "algorithmic_flops_per_launch": int(pb["flops"]), "algorithmic_bytes_per_launch": int(pb["bytes"]),
                "tflops": round(pb["flops"] / (b_ms * 1e-3) / 1e12, 2) if b_ms > 0 else 0.0}
        else:
            line["roofline"] = {"bound": "mfma" if mfma_bound else "hbm", "achieved": round(mfma_tf if mfma_bound else hbm_gbs, 2),
                                "peak": BF16_MFMA_PEAK_TF if mfma_bound else HBM_PEAK_GBS, "unit": "TFLOP/s" if mfma_bound else "GB/s",
                                "frac": round(mfma_tf / BF16_MFMA_PEAK_TF if mfma_bound else hbm_gbs / HBM_PEAK_GBS, 4), "traffic": None,
                                "kernel": "whole frame step (no per-kernel probe in this run)"}
        if args.no_vocoder:
            line["invalid"] = "diagnostic run without the vocoder"

    # batch = 1 leg (BASELINE configs[1] / configs[4]): RTF and p50 first-chunk latency, outside the timed region
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
                firsts.append(o.first_chunk_ms)
                rtfs.append(dt / (o.n_frames * FRAME_SEC))
        t1m = e1.timings()
        line["single_utterance"] = {"workload": "BASELINE.json configs[1]: 1 utterance, n_text=20, greedy, 64 frames",
                                    "rtf_p50": round(float(np.median(rtfs)), 5), "first_chunk_ms_p50": round(float(np.median(firsts)), 2),
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
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        threads = min(16, os.cpu_count() or 1)
        line["cpu_baseline"] = cpu_baseline(cfg, spk, cfg.with_vocoder, threads)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
