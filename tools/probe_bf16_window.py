"""How far below a large accumulator does a product still count?  acc = 2^G (ulp 2^(G-23)), p1 = 2^(G-24) (exactly half an
ulp: a tie that RNE resolves downwards), p2 = 2^(G-24-d): the result moves up to acc + ulp iff p2 is still in the sum.
Also the mirrored question for products far below the largest product of their lane group (no accumulator).
  python3 probe_bf16_window.py (from tools/)"""
import numpy as np

from probe_bf16_mfma_lib import bf16_bits, run

G = 24
ds = list(range(1, 65))
n = (len(ds) + 15) // 16
for label, same_group in (("p1 and p2 in lane group 0", True), ("p1 in lane group 0, p2 in lane group 1", False)):
    A = np.zeros((n, 16, 32), np.float32); B = np.zeros((n, 32, 16), np.float32); Cm = np.full((n, 16, 16), np.ldexp(1.0, G), np.float32)
    for idx, d in enumerate(ds):
        c, i = idx // 16, idx % 16
        A[c, i, 0] = np.ldexp(1.0, G - 24); B[c, 0, :] = 1.0
        k2 = 1 if same_group else 8
        e = G - 24 - d
        A[c, i, k2] = np.ldexp(1.0, e // 2); B[c, k2, :] = np.ldexp(1.0, e - e // 2)
    D = run(bf16_bits(A), bf16_bits(B), Cm)
    kept = [d for idx, d in enumerate(ds) if D[idx // 16, idx % 16, 0] > np.ldexp(1.0, G)]
    print(label, ": p2 = 2^(G-24-d) still moves the result for d in", kept)
# products only: big = 2^0 + tie construction at its 24-bit ulp: p0 = 1, p1 = 2^-24 (half ulp of 1), p2 = 2^(-24-d)
for label, k1, k2 in (("all in lane group 0", 1, 2), ("tie term in group 0, tiny term in group 1", 1, 8), ("tie term in group 1, tiny term in group 0", 8, 2)):
    A = np.zeros((n, 16, 32), np.float32); B = np.zeros((n, 32, 16), np.float32); Cm = np.zeros((n, 16, 16), np.float32)
    for idx, d in enumerate(ds):
        c, i = idx // 16, idx % 16
        A[c, i, 0] = 1.0; B[c, 0, :] = 1.0
        A[c, i, k1] = np.ldexp(1.0, -12); B[c, k1, :] = np.ldexp(1.0, -12)
        e = -24 - d
        A[c, i, k2] = np.ldexp(1.0, e // 2); B[c, k2, :] = np.ldexp(1.0, e - e // 2)
    D = run(bf16_bits(A), bf16_bits(B), Cm)
    kept = [d for idx, d in enumerate(ds) if D[idx // 16, idx % 16, 0] > 1.0]
    print("products only,", label, ": 2^(-24-d) still moves 1 + 2^-24 up for d in", kept)
