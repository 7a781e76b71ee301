"""Clone front-end (log-mel + speaker encoder + audio encoder, q3_clone.hip) at the family's full shape on a 3 s clip:
host-to-host time per call and, under rocprofv3 --kernel-trace --stats, the per-kernel summary (profiles/r01/clone_*).

  python3 tools/clone_bench.py [iters]
"""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "qwen3-tts-rust_amd"))
from q3tts import _abi, native  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cfg = _abi.tiny_config(max_batch=1, n_ctx=128, with_vocoder=0)
eng = native.NativeEngine(cfg)
ccfg = _abi.CloneConfig()
eng.lib.q3tts_clone_default_config(ccfg)
eng.clone_init(ccfg)
t = np.arange(72000) / 24000.0
clip = (0.2 * np.sin(2 * np.pi * 140.0 * t) + 0.02 * np.random.default_rng(5).standard_normal(72000)).astype(np.float32)
for _ in range(3):
    eng.audio_encode(clip); eng.speaker_encode(clip)
ta, ts = [], []
for _ in range(iters):
    t0 = time.perf_counter(); eng.audio_encode(clip); t1 = time.perf_counter(); eng.speaker_encode(clip); t2 = time.perf_counter()
    ta.append((t1 - t0) * 1e3); ts.append((t2 - t1) * 1e3)
print(f"clone front-end, 3 s clip, full shape, {iters} calls: audio encoder p50 {np.median(ta):.2f} ms, mel + speaker encoder p50 {np.median(ts):.2f} ms")
eng.close()
