"""ctypes loader for oracle/libq3oracle.so — the CPU restatement used ONLY as the checker.

Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the product package.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "qwen3-tts-rust_amd"))
from q3tts import _abi  # noqa: E402  (struct layouts only; the oracle restates the same fields)

ORACLE_DIR = os.path.join(REPO, "oracle")
ORACLE_LIB = os.path.join(ORACLE_DIR, "libq3oracle.so")

f32p, i32p, u32p = C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_uint32)
_lib = None


def build():
    subprocess.check_call(["make", "-C", ORACLE_DIR, "libq3oracle.so"], stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(ORACLE_LIB):
        build()
    L = C.CDLL(ORACLE_LIB)
    vp = C.c_void_p
    L.q3o_create.argtypes = [C.POINTER(_abi.ModelConfig), C.c_uint64, C.c_int32, C.c_int32]
    L.q3o_create.restype = vp
    L.q3o_destroy.argtypes = [vp]
    L.q3o_destroy.restype = None
    L.q3o_expf.argtypes = [C.c_float]
    L.q3o_expf.restype = C.c_float
    L.q3o_synth.argtypes = [C.c_uint64, C.c_uint32, C.c_uint64, C.c_float]
    L.q3o_synth.restype = C.c_float
    L.q3o_bf16.argtypes = [C.c_float]
    L.q3o_bf16.restype = C.c_uint16
    L.q3o_gemm_exact.argtypes = [f32p, C.c_int32, C.c_int32, C.POINTER(C.c_uint16), C.c_int32, f32p, C.c_float, f32p,
                                 C.c_int32, f32p, C.POINTER(C.c_uint64)]
    L.q3o_gemm_exact.restype = None
    L.q3o_rmsnorm.argtypes = [f32p, C.c_int32, f32p, C.c_float, f32p]
    L.q3o_rmsnorm.restype = None
    L.q3o_attention.argtypes = [f32p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, f32p, f32p, C.c_float,
                                C.c_float, i32p, f32p]
    L.q3o_attention.restype = None
    L.q3o_sample.argtypes = [f32p, C.c_int32, C.c_float, C.c_int32, C.c_float, C.c_float]
    L.q3o_sample.restype = C.c_int32
    L.q3o_rng_f32.argtypes = [C.c_uint64, C.c_int32, f32p]
    L.q3o_rng_f32.restype = None
    L.q3o_chacha_block.argtypes = [u32p, C.c_int32, u32p]
    L.q3o_chacha_block.restype = None
    L.q3o_qwen3_position.argtypes = [C.c_int32, C.c_int32, i32p]
    L.q3o_qwen3_position.restype = None
    L.q3o_build_prompt.argtypes = [vp, C.POINTER(_abi.PromptDesc), f32p, C.c_int32]
    L.q3o_build_prompt.restype = C.c_int32
    L.q3o_text_embedding.argtypes = [vp, C.c_int64, f32p]
    L.q3o_text_embedding.restype = None
    L.q3o_codec_embedding.argtypes = [vp, C.c_int32, C.c_int32, f32p]
    L.q3o_codec_embedding.restype = None
    L.q3o_project.argtypes = [vp, f32p, f32p]
    L.q3o_project.restype = None
    L.q3o_talker_prefill.argtypes = [vp, f32p, C.c_int32, f32p, f32p]
    L.q3o_talker_prefill.restype = None
    L.q3o_generate.argtypes = [vp, f32p, C.c_int32, C.c_float, C.c_int32, C.c_float, C.c_uint64, C.c_int32, C.c_int32,
                               C.c_int32, i32p, i32p]
    L.q3o_generate.restype = C.c_int32
    L.q3o_chunk_plan.argtypes = [C.c_int32, i32p, i32p, C.c_int32]
    L.q3o_chunk_plan.restype = C.c_int32
    if hasattr(L, "q3o_vocoder_create"):
        L.q3o_vocoder_create.argtypes = [C.POINTER(_abi.VocoderConfig), C.c_uint64, C.c_int32]
        L.q3o_vocoder_create.restype = vp
        L.q3o_vocoder_destroy.argtypes = [vp]
        L.q3o_vocoder_destroy.restype = None
        L.q3o_vocoder_reset.argtypes = [vp]
        L.q3o_vocoder_reset.restype = None
        L.q3o_vocoder_decode.argtypes = [vp, i32p, C.c_int32, C.c_int32, f32p, C.c_int32]
        L.q3o_vocoder_decode.restype = C.c_int32
    _lib = L
    return L


def ptr(a, typ):
    return a.ctypes.data_as(typ)


from q3tts.native import make_prompt_desc  # noqa: E402,F401  (struct packing helper of the product package)


class OracleModel:
    def __init__(self, model_cfg, seed=0, n_ctx=256, n_threads=4):
        self.cfg = model_cfg
        self.L = lib()
        self.h = self.L.q3o_create(C.byref(model_cfg), seed, n_ctx, n_threads)
        if not self.h:
            raise RuntimeError("q3o_create failed")

    def close(self):
        if self.h:
            self.L.q3o_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def build_prompt(self, desc):
        n = self.L.q3o_build_prompt(self.h, C.byref(desc), None, 0)
        out = np.zeros((n, self.cfg.d_embed), dtype=np.float32)
        self.L.q3o_build_prompt(self.h, C.byref(desc), ptr(out, f32p), n)
        return out

    def talker_prefill(self, embd):
        embd = np.ascontiguousarray(embd, dtype=np.float32)
        hid = np.zeros(self.cfg.t_d_model, dtype=np.float32)
        logits = np.zeros(self.cfg.t_vocab, dtype=np.float32)
        self.L.q3o_talker_prefill(self.h, ptr(embd, f32p), embd.shape[0], ptr(hid, f32p), ptr(logits, f32p))
        return hid, logits

    def generate(self, embd, temperature=0.0, top_k=40, top_p=0.9, seed=0, max_steps=16, min_frames=0, force_eos_at=-1):
        embd = np.ascontiguousarray(embd, dtype=np.float32)
        codes = np.zeros((max_steps, self.cfg.n_codebooks), dtype=np.int32)
        eos = C.c_int32(0)
        n = self.L.q3o_generate(self.h, ptr(embd, f32p), embd.shape[0], temperature, top_k, top_p, seed, max_steps,
                                min_frames, force_eos_at, ptr(codes, i32p), C.byref(eos))
        return codes[:n].copy(), bool(eos.value)
