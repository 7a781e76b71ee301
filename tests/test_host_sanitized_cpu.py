"""The host-only readers of libq3tts (tokenizer.json, GGUF, NPY) under AddressSanitizer + UBSan, fed damaged files.

These parsers take files from disk, i.e. untrusted bytes; "never abort, never read out of bounds" is part of the C ABI's
error convention. GPU sanitizers are not available on the pool, so the two host sources are compiled here
with g++ -fsanitize=address,undefined into a side library and driven in a child process: every truncation point and a few
hundred byte flips of valid files must come back as an error code or a clean result, never a sanitizer report.
"""
import os
import shutil
import subprocess
import sys
import textwrap

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "qwen3-tts-rust_amd", "csrc")

DRIVER = textwrap.dedent('''
    import ctypes as C, os, sys
    import numpy as np
    lib = C.CDLL(sys.argv[1]); work = sys.argv[2]
    vp = C.c_void_p
    lib.q3tts_tokenizer_load.argtypes = [C.c_char_p, C.POINTER(vp), C.c_char_p, C.c_int32]
    lib.q3tts_tokenizer_free.argtypes = [vp]; lib.q3tts_tokenizer_free.restype = None
    lib.q3tts_tokenizer_encode.argtypes = [vp, C.c_char_p, C.c_int64, C.POINTER(C.c_uint32), C.c_int32, C.POINTER(C.c_int32), C.c_char_p, C.c_int32]
    lib.q3tts_tokenizer_decode.argtypes = [vp, C.POINTER(C.c_uint32), C.c_int32, C.c_char_p, C.c_int64, C.POINTER(C.c_int64), C.c_char_p, C.c_int32]
    lib.q3tts_k_gguf_read.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_float), C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int32)]
    err = C.create_string_buffer(256)
    rng = np.random.default_rng(0)

    def tok_try(path, texts):
        h = vp()
        rc = lib.q3tts_tokenizer_load(path.encode(), C.byref(h), err, len(err))
        if rc != 0:
            return 0
        ids = (C.c_uint32 * 4096)(); n = C.c_int32(0); out = C.create_string_buffer(1 << 16); nb = C.c_int64(0)
        for t in texts:
            if lib.q3tts_tokenizer_encode(h, t, len(t), ids, 4096, C.byref(n), err, len(err)) == 0:
                lib.q3tts_tokenizer_decode(h, ids, n.value, out, len(out), C.byref(nb), err, len(err))
        junk = (C.c_uint32 * 8)(0, 1, 2, 0xFFFFFFFF, 123456789, 5, 6, 7)
        lib.q3tts_tokenizer_decode(h, junk, 8, out, len(out), C.byref(nb), err, len(err))
        lib.q3tts_tokenizer_free(h)
        return 1

    good = open(os.path.join(work, "tokenizer.json"), "rb").read()
    texts = [b"", b"hello world's  test\\n\\n", "你好，世界 2024 😀".encode(), b"\\xff\\xfe bad", b"<|im_start|>x<|im_end|>", b"a" * 3000, bytes(rng.integers(0, 256, 500, dtype=np.uint8))]
    assert tok_try(os.path.join(work, "tokenizer.json"), texts) == 1
    tmp = os.path.join(work, "mut.json")
    loaded = 0
    cuts = sorted(set(list(range(0, 400, 7)) + list(range(400, len(good), max(1, len(good) // 150)))))
    for c in cuts:
        open(tmp, "wb").write(good[:c]); loaded += tok_try(tmp, texts[:3])
    for k in range(300):
        b = bytearray(good)
        for _ in range(int(rng.integers(1, 4))):
            b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
        open(tmp, "wb").write(bytes(b)); loaded += tok_try(tmp, texts[:4])
    print("tokenizer variants that still loaded:", loaded)

    def gguf_try(path, names):
        ne = C.c_int64(0); dims = (C.c_int64 * 4)(); ty = C.c_int32(0); ok = 0
        for nm in names:
            if lib.q3tts_k_gguf_read(path.encode(), nm, None, 0, C.byref(ne), dims, C.byref(ty)) == 0 and 0 <= ne.value <= (1 << 22):
                buf = (C.c_float * max(1, ne.value))()
                ok += lib.q3tts_k_gguf_read(path.encode(), nm, buf, ne.value, C.byref(ne), dims, C.byref(ty)) == 0
        return ok
    for fn, names in (("t.gguf", [b"a.f32", b"b.f16", b"c.q8", b"d.bf16", b"missing"]), ("v.npy", [b""])):
        good = open(os.path.join(work, fn), "rb").read()
        assert gguf_try(os.path.join(work, fn), names) >= 1
        tmp = os.path.join(work, "mut" + os.path.splitext(fn)[1])
        ok = 0
        for c in sorted(set(list(range(0, min(len(good), 600), 3)) + list(range(600, len(good), max(1, len(good) // 100))))):
            open(tmp, "wb").write(good[:c]); ok += gguf_try(tmp, names)
        for k in range(400):
            b = bytearray(good)
            for _ in range(int(rng.integers(1, 4))):
                b[int(rng.integers(0, min(len(b), 700)))] = int(rng.integers(0, 256))   # headers / metadata / tensor infos
            open(tmp, "wb").write(bytes(b)); ok += gguf_try(tmp, names)
        print(fn, "reads that still succeeded:", ok)
    print("DONE")
''')


def _asan_runtime():
    out = subprocess.run(["g++", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    return out if out and os.path.isabs(out) and os.path.exists(out) else None


@pytest.mark.skipif(shutil.which("g++") is None or _asan_runtime() is None, reason="g++ with libasan is needed")
def test_host_readers_survive_damaged_files_under_asan(tmp_path):
    import _gguf
    from test_tokenizer_cpu import _train
    lib = tmp_path / "libq3host_asan.so"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fPIC", "-shared", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-o", str(lib), os.path.join(CSRC, "q3_tokenizer.cpp"), os.path.join(CSRC, "q3_gguf.cpp"),
                           os.path.join(REPO, "tests", "asan_stub.cpp")])
    _train(420, False, tmp_path)
    rng = np.random.default_rng(1)
    _gguf.write(str(tmp_path / "t.gguf"), [("a.f32", rng.standard_normal((8, 64)).astype(np.float32), 0), ("b.f16", rng.standard_normal((4, 32)).astype(np.float32), 1),
                                           ("c.q8", rng.standard_normal((4, 64)).astype(np.float32), 8), ("d.bf16", rng.standard_normal((2, 32)).astype(np.float32), 30)],
                meta={"general.architecture": "qwen3", "general.alignment": 32, "some.array": [1, 2, 3], "names": ["x", "y"]})
    np.save(tmp_path / "v.npy", rng.standard_normal((5, 7)).astype(np.float32))
    drv = tmp_path / "driver.py"
    drv.write_text(DRIVER)
    env = dict(os.environ, LD_PRELOAD=_asan_runtime(), ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=86", UBSAN_OPTIONS="halt_on_error=1:exitcode=87")
    r = subprocess.run([sys.executable, str(drv), str(lib), str(tmp_path)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "DONE" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
