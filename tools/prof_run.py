"""Small driver for rocprofv3: one engine, a few utterances. Usage: python tools/prof_run.py BATCH FRAMES [with_voc]"""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "qwen3-tts-rust_amd"))
from q3tts import _abi, native  # noqa: E402

batch, frames = int(sys.argv[1]), int(sys.argv[2])
voc = int(sys.argv[3]) if len(sys.argv) > 3 else 0
cfg = _abi.full_config_py()
cfg.max_batch, cfg.n_ctx, cfg.with_vocoder = batch, 1024, voc
eng = native.NativeEngine(cfg)
spk = ((np.arange(2048) % 13 - 6) * 0.03125).astype(np.float32)
keep = []
reqs = []
for i in range(batch):
    ids = np.random.default_rng(i).integers(0, 151643, size=20)
    d, k = native.make_prompt_desc(ids, spk_emb=spk)
    keep.append((d, k))
    reqs.append(dict(desc=d, temperature=0.7, seed=i, max_steps=frames, min_frames=frames, want_pcm=voc))
for it in range(2):
    outs = eng.generate_batch(reqs)
    t = eng.timings()
    print(f"iter {it}: frame_step_ms={t.frame_step_ms:.4f} prefill_ms={t.prefill_ms:.2f} steps={t.frame_steps} "
          f"GB/s={t.algo_bytes_per_step / (t.frame_step_ms * 1e-3) / 1e9:.1f}", flush=True)
eng.close()
