"""Case generator for the bf16-MFMA accumulation probe; see probe_bf16_mfma_lib.py / probe_bf16_analyze.py."""
import os

import numpy as np

from probe_bf16_mfma_lib import R, bf16_bits, run

rng = np.random.default_rng(0)
out = {}
# random sets: magnitudes 2^U(-s, s) with 8-bit mantissas, random signs; C random f32 of comparable size or zero
for name, s, cmode in [("r2", 2, "rand"), ("r8", 8, "rand"), ("r20", 20, "rand"), ("r40", 40, "zero"), ("r12z", 12, "zero"), ("r30c", 30, "rand")]:
    n = 64
    def val(shape):
        return (np.ldexp(1.0 + rng.integers(0, 128, shape) / 128.0, rng.integers(-s, s + 1, shape)) * rng.choice([-1.0, 1.0], shape)).astype(np.float32)
    A = bf16_bits(val((n, 16, 32))); B = bf16_bits(val((n, 32, 16)))
    Cm = val((n, 16, 16)) * rng.standard_normal((n, 16, 16)).astype(np.float32) if cmode == "rand" else np.zeros((n, 16, 16), np.float32)
    out[name + "_A"], out[name + "_B"], out[name + "_C"], out[name + "_D"] = A, B, Cm.astype(np.float32), run(A, B, Cm.astype(np.float32))
# cancellation sets: k and k' carry +x and -x (same magnitude 2^E above the rest), the rest small: exposes window width and grouping
for name, E in [("c10", 10), ("c20", 20), ("c24", 24), ("c26", 26), ("c30", 30), ("c40", 40), ("c60", 60)]:
    n = 64
    a = np.ldexp(1.0 + rng.integers(0, 128, (n, 16, 32)) / 128.0, rng.integers(-2, 3, (n, 16, 32))).astype(np.float32) * rng.choice([-1.0, 1.0], (n, 16, 32))
    b = np.ldexp(1.0 + rng.integers(0, 128, (n, 32, 16)) / 128.0, rng.integers(-2, 3, (n, 32, 16))).astype(np.float32) * rng.choice([-1.0, 1.0], (n, 32, 16))
    for c in range(n):  # per case: positions (k1, k2) get a = +-2^(E/2 ...) for every row, b = 1 for every col -> products +-2^E
        k1, k2 = rng.choice(32, 2, replace=False)
        e1 = E // 2; e2 = E - e1
        a[c, :, k1] = np.ldexp(1.0, e1); a[c, :, k2] = -np.ldexp(1.0, e1)
        b[c, k1, :] = np.ldexp(1.0, e2); b[c, k2, :] = np.ldexp(1.0, e2)
    A = bf16_bits(a); B = bf16_bits(b)
    Cm = (rng.standard_normal((n, 16, 16)) * (rng.random((n, 16, 16)) < 0.5)).astype(np.float32)
    out[name + "_A"], out[name + "_B"], out[name + "_C"], out[name + "_D"] = A, B, Cm, run(A, B, Cm)
# two MFMAs chained into the same accumulator (K = 64)
n = 64
val2 = lambda shape: (np.ldexp(1.0 + rng.integers(0, 128, shape) / 128.0, rng.integers(-8, 9, shape)) * rng.choice([-1.0, 1.0], shape)).astype(np.float32)
A = bf16_bits(val2((n, 2, 16, 32))); B = bf16_bits(val2((n, 2, 32, 16))); Cm = val2((n, 16, 16))
out["ch2_A"], out["ch2_B"], out["ch2_C"], out["ch2_D"] = A, B, Cm, run(A, B, Cm, chain=2)
d = os.path.join(R, "gpurun_out", "probe_bf16")
os.makedirs(d, exist_ok=True)
np.savez_compressed(os.path.join(d, "cases.npz"), **out)
print("saved", len(out) // 4, "sets")
