"""GGUF v3 writer + reader in numpy — test infrastructure (the CPU restatement of the file format for the loader row,
SURVEY.md §8f rank 2). Written from the public GGUF layout the reference's own reader follows
(/root/reference/src/assets_manager.rs:28-265): magic, version, counts, metadata KV, tensor infos, 32-byte aligned data.
Independent of csrc/q3_gguf.cpp: the tests write files with this module and read them back through the C ABI, and vice versa.
"""
import struct

import numpy as np

F32, F16, Q8_0, Q4_K, Q5_K, Q6_K, BF16 = 0, 1, 8, 12, 13, 14, 30
ALIGN = 32


def f32_to_bf16_bits(x):
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)


def bf16_bits_to_f32(b):
    return (np.asarray(b, dtype=np.uint16).astype(np.uint32) << 16).view(np.float32)


def quantize_q8_0(x):
    """ggml's reference quantizer: per block of 32, d = max|x| / 127 (stored as f16), q = round(x / d)."""
    x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1, 32)
    amax = np.abs(x).max(axis=1)
    d = (amax / 127.0).astype(np.float32)
    inv = np.where(d > 0, 1.0 / np.where(d > 0, d, 1.0), 0.0).astype(np.float32)
    v = x * inv[:, None]
    q = (np.sign(v) * np.floor(np.abs(v) + np.float32(0.5))).astype(np.int8)   # C roundf (half away from zero), as ggml's quantize_row_q8_0_ref
    out = np.zeros((x.shape[0], 34), dtype=np.uint8)
    out[:, :2] = d.astype(np.float16).view(np.uint8).reshape(-1, 2)
    out[:, 2:] = q.view(np.uint8)
    return out.reshape(-1)


def dequantize_q8_0(raw, n):
    blk = np.frombuffer(raw, dtype=np.uint8, count=n // 32 * 34).reshape(-1, 34)
    d = blk[:, :2].copy().view(np.float16).astype(np.float32).reshape(-1)
    q = blk[:, 2:].copy().view(np.int8).astype(np.float32)
    return (q * d[:, None]).reshape(-1)


# ---- ggml K-quants (super-blocks of 256): an independent restatement of the published block layouts (llama.cpp ggml-quants.c:
# block_q4_K / block_q5_K / block_q6_K and their dequantize_row_* loops). llama.cpp is not in /root/reference: PARITY UNPINNED.
# The quantisers below are simple min/max ones (any valid block decodes deterministically; ggml's search for the best scales is
# not restated, it does not change what a reader must do).
def _f16(x):
    return np.asarray(x, dtype=np.float32).astype(np.float16)


def _pack_scales_k4(sc, mn):
    """8 six-bit scales + 8 six-bit mins -> 12 bytes (inverse of get_scale_min_k4)."""
    q = np.zeros(12, dtype=np.uint8)
    for j in range(4):
        q[j] = (sc[j] & 63) | ((sc[j + 4] >> 4) << 6)
        q[j + 4] = (mn[j] & 63) | ((mn[j + 4] >> 4) << 6)
        q[j + 8] = (sc[j + 4] & 0xF) | ((mn[j + 4] & 0xF) << 4)
    return q


def _unpack_scales_k4(q):
    sc, mn = np.zeros(8, dtype=np.int32), np.zeros(8, dtype=np.int32)
    for j in range(8):
        if j < 4:
            sc[j], mn[j] = q[j] & 63, q[j + 4] & 63
        else:
            sc[j] = (q[j + 4] & 0xF) | ((q[j - 4] >> 6) << 4)
            mn[j] = (q[j + 4] >> 4) | ((q[j] >> 6) << 4)
    return sc, mn


def _quantize_k45(x, bits):
    """Q4_K (bits = 4) / Q5_K (bits = 5): value = d * sc_j * q - dmin * m_j over 8 sub-blocks of 32."""
    qmax = (1 << bits) - 1
    x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1, 8, 32)
    out = []
    for blk in x:
        lo = np.minimum(blk.min(axis=1), 0.0)
        scale = (blk.max(axis=1) - lo) / qmax
        d = np.float32(_f16(scale.max() / 63.0)) if scale.max() > 0 else np.float32(0)
        dmin = np.float32(_f16((-lo).max() / 63.0)) if (-lo).max() > 0 else np.float32(0)
        sc = np.clip(np.rint(scale / d) if d > 0 else np.zeros(8), 0, 63).astype(np.int32)
        mn = np.clip(np.rint(-lo / dmin) if dmin > 0 else np.zeros(8), 0, 63).astype(np.int32)
        den = (d * sc.astype(np.float32))[:, None]
        q = np.clip(np.rint((blk + (dmin * mn.astype(np.float32))[:, None]) / np.where(den > 0, den, 1.0)), 0, qmax).astype(np.uint8)
        qs = np.zeros(128, dtype=np.uint8); qh = np.zeros(32, dtype=np.uint8)
        for c in range(4):
            a, b = q[2 * c], q[2 * c + 1]
            qs[32 * c:32 * c + 32] = (a & 0xF) | ((b & 0xF) << 4)
            if bits == 5:
                qh |= ((a >> 4) & 1) << (2 * c)
                qh |= ((b >> 4) & 1) << (2 * c + 1)
        head = np.concatenate([_f16([d]).view(np.uint8), _f16([dmin]).view(np.uint8), _pack_scales_k4(sc, mn)])
        out.append(np.concatenate([head, qh, qs]) if bits == 5 else np.concatenate([head, qs]))
    return np.concatenate(out).astype(np.uint8)


def _dequantize_k45(raw, n, bits):
    bs = 176 if bits == 5 else 144
    blk = np.frombuffer(raw, dtype=np.uint8, count=n // 256 * bs).reshape(-1, bs)
    y = np.zeros((blk.shape[0], 256), dtype=np.float32)
    for i, b in enumerate(blk):
        d = np.float32(b[0:2].copy().view(np.float16)[0]); dmin = np.float32(b[2:4].copy().view(np.float16)[0])
        sc, mn = _unpack_scales_k4(b[4:16])
        qh = b[16:48] if bits == 5 else None
        qs = b[48:176] if bits == 5 else b[16:144]
        for c in range(4):
            lo = (qs[32 * c:32 * c + 32] & 0xF).astype(np.int32); hi = (qs[32 * c:32 * c + 32] >> 4).astype(np.int32)
            if bits == 5:
                lo = lo + np.where(qh & (1 << (2 * c)), 16, 0); hi = hi + np.where(qh & (2 << (2 * c)), 16, 0)
            d1 = np.float32(d * np.float32(sc[2 * c])); m1 = np.float32(dmin * np.float32(mn[2 * c]))
            d2 = np.float32(d * np.float32(sc[2 * c + 1])); m2 = np.float32(dmin * np.float32(mn[2 * c + 1]))
            y[i, 64 * c:64 * c + 32] = (d1 * lo.astype(np.float32)).astype(np.float32) - m1
            y[i, 64 * c + 32:64 * c + 64] = (d2 * hi.astype(np.float32)).astype(np.float32) - m2
    return y.reshape(-1)


def _q6_index(e):
    """element e of a Q6_K super-block -> (ql index, high nibble?, qh index, qh shift, scale index)"""
    h, r = divmod(e, 128)
    grp, l = divmod(r, 32)
    return 64 * h + l + (32 if grp in (1, 3) else 0), grp >= 2, 32 * h + l, 2 * grp, 8 * h + l // 16 + 2 * grp


_Q6 = np.array([_q6_index(e) for e in range(256)], dtype=np.int64)  # columns: ql index, high nibble, qh index, qh shift, scale index


def quantize_q6_k(x):
    x = np.ascontiguousarray(x, dtype=np.float32).reshape(-1, 256)
    nb = x.shape[0]
    qi, hi, hq, sh, si = _Q6.T
    sub = np.zeros((nb, 16), dtype=np.float32)
    np.maximum.at(sub, (np.arange(nb)[:, None], si[None, :]), np.abs(x))
    sub /= 32.0
    d = np.where(sub.max(axis=1) > 0, (sub.max(axis=1) / 127.0).astype(np.float16).astype(np.float32), 0).astype(np.float32)
    sc = np.clip(np.rint(np.divide(sub, d[:, None], out=np.zeros_like(sub), where=d[:, None] > 0)), -128, 127).astype(np.int8)
    den = d[:, None] * sc.astype(np.float32)[:, si]
    q = (np.clip(np.rint(np.divide(x, den, out=np.zeros_like(x), where=den != 0)), -32, 31) + 32).astype(np.uint8)
    ql = np.zeros((nb, 128), dtype=np.uint8); qh = np.zeros((nb, 64), dtype=np.uint8)
    for e in range(256):
        ql[:, qi[e]] |= (((q[:, e] & 0xF) << 4) if hi[e] else (q[:, e] & 0xF)).astype(np.uint8)
        qh[:, hq[e]] |= (((q[:, e] >> 4) & 3) << int(sh[e])).astype(np.uint8)
    return np.concatenate([ql, qh, sc.view(np.uint8), d.astype(np.float16).view(np.uint8).reshape(nb, 2)], axis=1).reshape(-1)


def dequantize_q6_k(raw, n):
    blk = np.frombuffer(raw, dtype=np.uint8, count=n // 256 * 210).reshape(-1, 210)
    qi, hi, hq, sh, si = _Q6.T
    ql, qh, sc = blk[:, :128].astype(np.int32), blk[:, 128:192].astype(np.int32), blk[:, 192:208].copy().view(np.int8).astype(np.float32)
    d = blk[:, 208:210].copy().view(np.float16).astype(np.float32).reshape(-1)
    nib = np.where(hi[None, :] != 0, ql[:, qi] >> 4, ql[:, qi] & 0xF)
    q = (nib | (((qh[:, hq] >> sh[None, :]) & 3) << 4)) - 32
    ds = (d[:, None] * sc[:, si]).astype(np.float32)
    return (ds * q.astype(np.float32)).astype(np.float32).reshape(-1)


def encode(arr, ty):
    a = np.ascontiguousarray(arr, dtype=np.float32)
    if ty == F32:
        return a.tobytes()
    if ty == F16:
        return a.astype(np.float16).tobytes()
    if ty == BF16:
        return f32_to_bf16_bits(a).tobytes()
    if ty == Q8_0:
        return quantize_q8_0(a).tobytes()
    if ty == Q4_K:
        return _quantize_k45(a, 4).tobytes()
    if ty == Q5_K:
        return _quantize_k45(a, 5).tobytes()
    if ty == Q6_K:
        return quantize_q6_k(a).tobytes()
    raise ValueError(ty)


def decode(raw, ty, n):
    if ty == F32:
        return np.frombuffer(raw, dtype="<f4", count=n).copy()
    if ty == F16:
        return np.frombuffer(raw, dtype="<f2", count=n).astype(np.float32)
    if ty == BF16:
        return bf16_bits_to_f32(np.frombuffer(raw, dtype="<u2", count=n))
    if ty == Q8_0:
        return dequantize_q8_0(raw, n)
    if ty == Q4_K:
        return _dequantize_k45(raw, n, 4)
    if ty == Q5_K:
        return _dequantize_k45(raw, n, 5)
    if ty == Q6_K:
        return dequantize_q6_k(raw, n)
    raise ValueError(ty)


def _s(b):
    return struct.pack("<Q", len(b)) + b


def write(path, tensors, meta=None, version=3, raw_types=None):
    """tensors: list of (name, ndarray [rows][cols] or [n], ggml type); meta: dict name -> int | float | str | list.
    raw_types: {name: type id} written into the tensor info instead of the real one (negative tests)."""
    kv = b""
    meta = dict(meta or {})
    for k, v in meta.items():
        kv += _s(k.encode())
        if isinstance(v, bool):
            kv += struct.pack("<IB", 7, int(v))
        elif isinstance(v, int):
            kv += struct.pack("<II", 4, v) if 0 <= v < 2 ** 32 else struct.pack("<Iq", 11, v)
        elif isinstance(v, float):
            kv += struct.pack("<If", 6, v)
        elif isinstance(v, str):
            kv += struct.pack("<I", 8) + _s(v.encode())
        elif isinstance(v, (list, tuple)):
            if all(isinstance(e, str) for e in v):
                kv += struct.pack("<IIQ", 9, 8, len(v)) + b"".join(_s(e.encode()) for e in v)
            else:
                kv += struct.pack("<IIQ", 9, 5, len(v)) + b"".join(struct.pack("<i", int(e)) for e in v)
        else:
            raise TypeError(k)
    infos, blobs, off = b"", [], 0
    for name, arr, ty in tensors:
        arr = np.asarray(arr)
        dims = list(arr.shape)[::-1]  # ggml: contiguous dimension first
        infos += _s(name.encode()) + struct.pack("<I", len(dims)) + b"".join(struct.pack("<Q", d) for d in dims)
        infos += struct.pack("<IQ", (raw_types or {}).get(name, ty), off)
        blob = encode(arr, ty)
        blob += b"\0" * ((-len(blob)) % ALIGN)
        blobs.append(blob)
        off += len(blob)
    head = b"GGUF" + struct.pack("<IQQ", version, len(tensors), len(meta)) + kv + infos
    head += b"\0" * ((-len(head)) % ALIGN)
    with open(path, "wb") as f:
        f.write(head)
        for b in blobs:
            f.write(b)


def read(path):
    """-> {name: (f32 array in numpy shape, ggml type)} — the restatement the C++ reader is checked against."""
    raw = open(path, "rb").read()
    assert raw[:4] == b"GGUF"
    ver, nt, nkv = struct.unpack_from("<IQQ", raw, 4)
    assert ver >= 2
    pos = 24

    def rs():
        nonlocal pos
        (n,) = struct.unpack_from("<Q", raw, pos)
        s = raw[pos + 8:pos + 8 + n]
        pos += 8 + n
        return s.decode()

    size = {0: 1, 1: 1, 2: 2, 3: 2, 4: 4, 5: 4, 6: 4, 7: 1, 10: 8, 11: 8, 12: 8}
    for _ in range(nkv):
        rs()
        (vt,) = struct.unpack_from("<I", raw, pos)
        pos += 4
        if vt == 8:
            rs()
        elif vt == 9:
            et, cnt = struct.unpack_from("<IQ", raw, pos)
            pos += 12
            if et == 8:
                for _ in range(cnt):
                    rs()
            else:
                pos += size[et] * cnt
        else:
            pos += size[vt]
    infos = []
    for _ in range(nt):
        name = rs()
        (nd,) = struct.unpack_from("<I", raw, pos)
        pos += 4
        dims = struct.unpack_from("<%dQ" % nd, raw, pos)
        pos += 8 * nd
        ty, off = struct.unpack_from("<IQ", raw, pos)
        pos += 12
        infos.append((name, dims, ty, off))
    data = (pos + ALIGN - 1) // ALIGN * ALIGN
    out = {}
    for name, dims, ty, off in infos:
        n = int(np.prod(dims))
        out[name] = (decode(raw[data + off:], ty, n).reshape(dims[::-1]), ty)
    return out
