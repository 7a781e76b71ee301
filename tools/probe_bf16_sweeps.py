"""Targeted sweeps on v_mfma_f32_16x16x32_bf16 (see tools/probe_bf16_mfma.py): granularity at which a small product / the
accumulator survives next to a cancelling pair +-2^E, per lane group. Prints tables.   python3 tools/probe_bf16_sweeps.py
"""
import numpy as np
from probe_bf16_mfma_lib import run, bf16_bits  # noqa


def lowbit(x):
    if x == 0:
        return None
    u = np.float32(x).view(np.uint32)
    e = int((u >> 23) & 0xFF) - 127; m = int(u & 0x7FFFFF) | 0x800000
    return e - 23 + ((m & -m).bit_length() - 1)


def sweep(E, grp_big, grp_small, use_c, mant=1.0 + 127 / 128.0):
    """cancelling pair in lane group grp_big (k = 8*grp_big + 0, 1); small product x = mant^2 * 2^t at k = 8*grp_small + 2
    (or x as the accumulator when use_c); returns [(t, exact x, device result)]"""
    ts = list(range(E - 40, E + 1))
    n = (len(ts) + 255) // 256
    A = np.zeros((n, 16, 32), np.float32); B = np.zeros((n, 32, 16), np.float32); Cm = np.zeros((n, 16, 16), np.float32)
    e1 = E // 2; e2 = E - e1
    k0 = 8 * grp_big
    A[:, :, k0] = np.ldexp(1.0, e1); A[:, :, k0 + 1] = -np.ldexp(1.0, e1)
    B[:, k0, :] = np.ldexp(1.0, e2); B[:, k0 + 1, :] = np.ldexp(1.0, e2)
    exact = {}
    for idx, t in enumerate(ts):
        c, i, j = idx // 256, (idx // 16) % 16, idx % 16
        if use_c:
            x = np.float32(np.ldexp(1.0 + (2 ** 23 - 1) / 2 ** 23, t))
            Cm[c, i, j] = x
        else:
            pass
        exact[idx] = (c, i, j)
    if not use_c:  # x_(i,j) = (mant * 2^ti) * (mant * 2^tj): choose per (i, j) exponents so that ti + tj = t
        ks = 8 * grp_small + 2
        # rows carry mant * 2^(t_row), cols carry mant * 2^(t_col); t = t_row + t_col; t_row = -20 + 2*i ... simple: put everything in A per row, B col = mant
        # each (c, i, j) needs its own t: use j to select among 16 k positions? keep simple: one t per row i (16 per case), all cols equal
        ts = list(range(E - 40, E + 1))
        n = (len(ts) + 15) // 16
        A = np.zeros((n, 16, 32), np.float32); B = np.zeros((n, 32, 16), np.float32); Cm = np.zeros((n, 16, 16), np.float32)
        A[:, :, k0] = np.ldexp(1.0, e1); A[:, :, k0 + 1] = -np.ldexp(1.0, e1)
        B[:, k0, :] = np.ldexp(1.0, e2); B[:, k0 + 1, :] = np.ldexp(1.0, e2)
        exact = {}
        for idx, t in enumerate(ts):
            c, i = idx // 16, idx % 16
            A[c, i, ks] = np.ldexp(mant, t); B[c, ks, :] = mant
            exact[idx] = (c, i, 0)
    D = run(bf16_bits(A), bf16_bits(B), Cm)
    out = []
    for idx, t in enumerate(ts):
        c, i, j = exact[idx]
        x = float(Cm[c, i, j]) if use_c else float(np.float32(bf16(A[c, i, 8 * grp_small + 2])) * np.float32(bf16(B[c, 8 * grp_small + 2, 0])))
        out.append((t, x, float(D[c, i, j])))
    return out


def bf16(x):
    return np.array([bf16_bits(np.array([x], np.float32))[0]], np.uint16).astype(np.uint32).__lshift__(16).view(np.float32)[0]


if __name__ == "__main__":
    for E in (24, 30):
        for (gb, gs, uc, label) in [(0, 0, False, "product, same group as the pair"), (1, 0, False, "product in group 0, pair in group 1"),
                                    (0, 1, False, "product in group 1, pair in group 0"), (3, 3, False, "product and pair in group 3"),
                                    (0, 0, True, "accumulator, pair in group 0"), (3, 0, True, "accumulator, pair in group 3")]:
            rows = sweep(E, gb, gs, uc)
            print("== E=%d  %s" % (E, label))
            for t, x, d in rows:
                if x != 0:
                    print("  t=%4d  x=%.9g  dev=%.9g  dev/x=%.7f  lowbit(dev)=%s  lowbit(x)=%s" % (t, x, d, d / x, lowbit(d), lowbit(x)))
