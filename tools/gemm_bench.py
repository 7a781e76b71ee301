"""Micro-benchmark of the exact GEMM through the C-ABI test hook (back-to-back launches, weights may be MALL-resident)."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "qwen3-tts-rust_amd"))
from q3tts import native  # noqa: E402

shapes = [(1, 2048, 4096, 1), (1, 2048, 2048, 0), (1, 2048, 12288, 1), (1, 6144, 2048, 0), (1, 1024, 4096, 1), (1, 2048, 1024, 0),
          (1, 3072, 1024, 0), (16, 2048, 4096, 1), (64, 2048, 4096, 1), (64, 2048, 12288, 1), (64, 6144, 2048, 0), (64, 1024, 4096, 1),
          (64, 2048, 2048, 0), (31, 2048, 4096, 1)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
rng = np.random.default_rng(0)
for (B, K, N, norm) in shapes:
    x = rng.standard_normal((B, K)).astype(np.float32)
    w = (rng.integers(0, 65536, size=(N, K)) & 0xBFFF).astype(np.uint16)  # finite bf16 bit patterns
    nw = np.ones(K, dtype=np.float32) if norm else None
    _, _, ms = native.k_gemm_exact(x, w, norm_w=nw, epilogue=0, iters=300)
    us = ms * 1e3
    print(f"B={B:3d} K={K:5d} N={N:6d} norm={norm}: {us:8.2f} us  {N * K * 2 / ms / 1e6:8.1f} GB/s  {2.0 * B * N * K / ms / 1e9:8.2f} TFLOP/s "
          f"(padded MFMA rate {2.0 * max(16, (B + 15) // 16 * 16) * N * K / ms / 1e9:7.2f})", flush=True)
