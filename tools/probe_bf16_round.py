"""Final rounding of one accumulate step: acc = 2^24 (ulp 2) plus one or two products in lane group 0; prints D - 2^24.
  python3 probe_bf16_round.py (from tools/)"""
import numpy as np

from probe_bf16_mfma_lib import bf16_bits, run

cases = [("p=0.5", [(0.5, 1.0)]), ("p=1", [(1.0, 1.0)]), ("p=1.25", [(1.25, 1.0)]), ("p=1.5", [(1.5, 1.0)]), ("p=1.75", [(1.75, 1.0)]), ("p=2", [(2.0, 1.0)]),
         ("p=2.5", [(2.5, 1.0)]), ("p=3", [(3.0, 1.0)]), ("p=3.5", [(3.5, 1.0)]), ("p=1+0.5", [(1.0, 1.0), (0.5, 1.0)]), ("p=1+2^-8", [(1.0, 1.0), (2.0 ** -8, 1.0)]),
         ("p=1+2^-20", [(1.0, 1.0), (2.0 ** -10, 2.0 ** -10)]), ("p=3+2^-8", [(3.0, 1.0), (2.0 ** -8, 1.0)]), ("p=-1", [(-1.0, 1.0)]), ("p=-1-0.5", [(-1.0, 1.0), (-0.5, 1.0)]),
         ("p=-0.5", [(-0.5, 1.0)]), ("p=-0.25", [(-0.25, 1.0)]), ("p=1.0078125^2", [(1.0078125, 1.0078125)]), ("p=0.9921875*1.0078125", [(0.9921875, 1.0078125)])]
for accv, name in ((2.0 ** 24, "acc=2^24"), (-(2.0 ** 24), "acc=-2^24"), (2.0 ** 24 + 2, "acc=2^24+2")):
    n = (len(cases) + 15) // 16
    A = np.zeros((n, 16, 32), np.float32); B = np.zeros((n, 32, 16), np.float32); Cm = np.full((n, 16, 16), accv, np.float32)
    for idx, (_, prods) in enumerate(cases):
        c, i = idx // 16, idx % 16
        for k, (a, b) in enumerate(prods):
            A[c, i, k] = a; B[c, k, :] = b
    D = run(bf16_bits(A), bf16_bits(B), Cm)
    print(name, " ".join("%s:%+g" % (cases[idx][0], float(D[idx // 16, idx % 16, 0]) - accv) for idx in range(len(cases))))
