"""Builds tests/golden/bf16_mfma_mi355x.npz from the probe outputs measured on an MI355X (gpurun_out/probe_bf16/*.npz,
produced there by tools/probe_bf16_mfma.py and tools/probe_bf16_steps.py). Unlike oracle_tiny.json these are HARDWARE
vectors: inputs (bf16 bit patterns of one A row and one B column, the f32 accumulator) and the device's result for
v_mfma_f32_16x16x32_bf16, sampled from every case set (random spreads, cancelling pairs, one / two active lane groups).
The CPU suite checks oracle/q3_oracle.c::q3o_mfma_bf16_dot32 against them without a GPU.
  python tests/golden/make_bf16_mfma_golden.py"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(os.path.dirname(os.path.dirname(HERE)), "gpurun_out", "probe_bf16")
rng = np.random.default_rng(0)
a, b, c, d, tag = [], [], [], [], []
for f in ("cases.npz", "steps.npz"):
    Z = np.load(os.path.join(SRC, f))
    for nm in sorted({k[:-2] for k in Z.files}):
        A, B, Cm, D = Z[nm + "_A"], Z[nm + "_B"], Z[nm + "_C"], Z[nm + "_D"]
        if A.ndim != 3:
            continue
        for _ in range(48):
            cs, i, j = int(rng.integers(A.shape[0])), int(rng.integers(16)), int(rng.integers(16))
            a.append(A[cs, i, :].copy()); b.append(B[cs, :, j].copy()); c.append(Cm[cs, i, j]); d.append(D[cs, i, j]); tag.append(nm)
np.savez_compressed(os.path.join(HERE, "bf16_mfma_mi355x.npz"), a=np.array(a, np.uint16), b=np.array(b, np.uint16), c=np.array(c, np.float32),
                    d=np.array(d, np.float32), set=np.array(tag))
print(len(a), "vectors")
