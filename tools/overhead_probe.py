import os, sys, time, json
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "qwen3-tts-rust_amd"); sys.path.insert(0, "tests")
sys.argv = ["bench.py"]
import bench as B
from q3tts import _abi, native
cfg = _abi.full_config_py(); cfg.device, cfg.max_batch, cfg.n_ctx, cfg.max_steps_cap, cfg.with_vocoder = 0, 64, 4096, 512, 1
eng = native.NativeEngine(cfg)
keep = []; reqs, frames = B.make_workload(64, 0, 1, B.vivian(), keep)
for it in range(3):
    t0 = time.perf_counter(); outs = eng.generate_batch(reqs); t1 = time.perf_counter()
    tm = eng.timings()
    print(f"python wall {1e3*(t1-t0):.1f} ms | C total {tm.total_ms:.1f} | prefill {tm.prefill_ms:.1f} decode {tm.decode_ms:.1f} voc_wait {tm.vocoder_ms:.1f}", flush=True)
eng.close()
