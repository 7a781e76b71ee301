"""Single-step / two-step case sets for the bf16-MFMA accumulation probe: only lane group 0 (or 0 and 1) carries products.
Stores gpurun_out/probe_bf16/steps.npz.   python3 tools/probe_bf16_steps.py (from tools/)"""
import os

import numpy as np

from probe_bf16_mfma_lib import R, bf16_bits, run

rng = np.random.default_rng(7)
out = {}


def val(shape, s):
    return (np.ldexp(1.0 + rng.integers(0, 128, shape) / 128.0, rng.integers(-s, s + 1, shape)) * rng.choice([-1.0, 1.0], shape)).astype(np.float32)


for name, groups, nprod, s, cmode in [("s1_8_r6", 1, 8, 6, "rand"), ("s1_8_r6z", 1, 8, 6, "zero"), ("s1_8_r14", 1, 8, 14, "rand"), ("s1_2_r10", 1, 2, 10, "rand"),
                                      ("s1_1_r12", 1, 1, 12, "rand"), ("s2_8_r6", 2, 8, 6, "rand"), ("s1_8_r3", 1, 8, 3, "rand"), ("s1_3_r8z", 1, 3, 8, "zero")]:
    n = 32
    A = np.zeros((n, 16, 32), np.float32); B = np.zeros((n, 32, 16), np.float32)
    for g in range(groups):
        A[:, :, 8 * g:8 * g + nprod] = val((n, 16, nprod), s); B[:, 8 * g:8 * g + nprod, :] = val((n, nprod, 16), s)
    Cm = (val((n, 16, 16), 2 * s) * (1.0 + rng.random((n, 16, 16)).astype(np.float32))).astype(np.float32) if cmode == "rand" else np.zeros((n, 16, 16), np.float32)
    Ab, Bb = bf16_bits(A), bf16_bits(B)
    out[name + "_A"], out[name + "_B"], out[name + "_C"], out[name + "_D"] = Ab, Bb, Cm, run(Ab, Bb, Cm)
d = os.path.join(R, "gpurun_out", "probe_bf16")
os.makedirs(d, exist_ok=True)
np.savez_compressed(os.path.join(d, "steps.npz"), **out)
print("saved", len(out) // 4, "sets")
