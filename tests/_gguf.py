"""GGUF v3 writer + reader in numpy — test infrastructure (the CPU restatement of the file format for the loader row,
SURVEY.md §8f rank 2). Written from the public GGUF layout the reference's own reader follows
(/root/reference/src/assets_manager.rs:28-265): magic, version, counts, metadata KV, tensor infos, 32-byte aligned data.
Independent of csrc/q3_gguf.cpp: the tests write files with this module and read them back through the C ABI, and vice versa.
"""
import struct

import numpy as np

F32, F16, Q8_0, BF16 = 0, 1, 8, 30
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
    q = np.rint(x * inv[:, None]).astype(np.int8)
    out = np.zeros((x.shape[0], 34), dtype=np.uint8)
    out[:, :2] = d.astype(np.float16).view(np.uint8).reshape(-1, 2)
    out[:, 2:] = q.view(np.uint8)
    return out.reshape(-1)


def dequantize_q8_0(raw, n):
    blk = np.frombuffer(raw, dtype=np.uint8, count=n // 32 * 34).reshape(-1, 34)
    d = blk[:, :2].copy().view(np.float16).astype(np.float32).reshape(-1)
    q = blk[:, 2:].copy().view(np.int8).astype(np.float32)
    return (q * d[:, None]).reshape(-1)


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
