"""Runs tools/libprobe_bf16_mfma.so on crafted cases and stores inputs + outputs (gpurun_out/probe_bf16/cases.npz) for
offline analysis (library part: run(), bf16_bits()) of v_mfma_f32_16x16x32_bf16's accumulation arithmetic (tools/probe_bf16_analyze.py).
  python3 tools/probe_bf16_mfma.py
"""
import ctypes as C
import os

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = C.CDLL(os.path.join(R, "tools", "libprobe_bf16_mfma.so"))
lib.probe_bf16_mfma.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]


def bf16_bits(x):
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)


def run(A, B, Cm, chain=1):
    n = Cm.shape[0]
    D = np.zeros_like(Cm)
    rc = lib.probe_bf16_mfma(A.ctypes.data, B.ctypes.data, Cm.ctypes.data, D.ctypes.data, n, chain)
    assert rc == 0, rc
    return D


