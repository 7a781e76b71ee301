"""Offline analysis of gpurun_out/probe_bf16/cases.npz: which exact-arithmetic model reproduces v_mfma_f32_16x16x32_bf16?
  python3 tools/probe_bf16_analyze.py [set ...]
"""
import os
import struct
import sys

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
Z = np.load(os.path.join(R, "gpurun_out", "probe_bf16", "cases.npz"))
S = 400  # fixed-point scale: value = N / 2^S


def f32_to_fix(bits):
    """float32 bit pattern -> exact integer N with value = N / 2^S"""
    s = -1 if bits >> 31 else 1
    e = (bits >> 23) & 0xFF; m = bits & 0x7FFFFF
    if e == 0:
        return s * m << (S - 149)
    return s * ((m | 0x800000) << (S + e - 150))


def fix_to_f32(N, mode="rne"):
    """exact N / 2^S -> float32 bits (normal range assumed; RNE / truncation toward zero)"""
    if N == 0:
        return 0
    s = 0x80000000 if N < 0 else 0
    N = abs(N)
    msb = N.bit_length() - 1  # value in [2^(msb-S), 2^(msb-S+1))
    sh = msb - 23
    if sh > 0:
        q, r = N >> sh, N & ((1 << sh) - 1)
        half = 1 << (sh - 1)
        if mode == "rne" and (r > half or (r == half and (q & 1))):
            q += 1
        if q >> 24:
            q >>= 1; msb += 1
    else:
        q = N << -sh
    e = msb - S + 127
    if e <= 0 or e >= 255:
        return None
    return s | (e << 23) | (q & 0x7FFFFF)


def bf(bits16):
    return f32_to_fix(int(bits16) << 16)


def models(a, b, c):
    """a, b: 32 exact ints (scale S each) -> products at scale 2S; returns dict name -> f32 bits"""
    p = [(x * y) >> S for x, y in zip(a, b)]  # exact: the low S bits are zero (bf16 x bf16 has 16 significant bits)
    out = {}
    out["exact_rne"] = fix_to_f32(c + sum(p))
    out["exact_trunc"] = fix_to_f32(c + sum(p), "trunc")
    acc = c
    for k in range(32):
        acc = f32_to_fix(fix_to_f32(acc + p[k]) or 0)
    out["seq_fma"] = fix_to_f32(acc)
    for g in (2, 4, 8, 16):
        acc = c
        for k0 in range(0, 32, g):
            r = fix_to_f32(acc + sum(p[k0:k0 + g]))
            acc = f32_to_fix(r or 0)
        out["blk%d_rne" % g] = fix_to_f32(acc)
        acc = c
        for k0 in range(0, 32, g):
            r = fix_to_f32(acc + sum(p[k0:k0 + g]), "trunc")
            acc = f32_to_fix(r or 0)
        out["blk%d_trunc" % g] = fix_to_f32(acc, "trunc")
    # products summed first (no c), rounded, then added to c
    t = fix_to_f32(sum(p)); out["sum_then_c"] = fix_to_f32(c + f32_to_fix(t or 0))
    # interleaved grouping: lanes groups hold k = 8g + j; hardware might pair j across groups: blocks {j, 8+j, 16+j, 24+j}
    acc = c
    for j in range(8):
        acc = f32_to_fix(fix_to_f32(acc + p[j] + p[8 + j] + p[16 + j] + p[24 + j]) or 0)
    out["strided4_rne"] = fix_to_f32(acc)
    return out


names = sys.argv[1:] or sorted({k[:-2] for k in Z.files})
for nm in names:
    A, B, Cm, D = Z[nm + "_A"], Z[nm + "_B"], Z[nm + "_C"], Z[nm + "_D"]
    if A.ndim == 4:
        continue
    n = A.shape[0]
    hits, tot = {}, 0
    rng = np.random.default_rng(1)
    for c in range(min(n, 24)):
        for _ in range(24):
            i, j = int(rng.integers(16)), int(rng.integers(16))
            a = [bf(A[c, i, k]) for k in range(32)]; b = [bf(B[c, k, j]) for k in range(32)]
            cc = f32_to_fix(int(Cm[c, i, j].view(np.uint32)))
            d = int(D[c, i, j].view(np.uint32))
            ms = models(a, b, cc)
            tot += 1
            for k, v in ms.items():
                hits[k] = hits.get(k, 0) + (v == d)
    print(nm, tot, " ".join("%s=%.3f" % (k, v / tot) for k, v in sorted(hits.items(), key=lambda kv: -kv[1])[:8]))
