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
    L.q3o_synth_fill.argtypes = [C.c_uint64, C.c_uint32, C.c_uint64, C.c_float, C.c_float, C.c_int32, f32p]
    L.q3o_synth_fill.restype = None
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
    L.q3o_mel_frames.argtypes = [C.c_int64]
    L.q3o_mel_frames.restype = C.c_int32
    L.q3o_mel.argtypes = [f32p, C.c_int64, f32p, f32p]
    L.q3o_mel.restype = C.c_int32
    L.q3o_audio_frames.argtypes = [C.POINTER(_abi.CloneConfig), C.c_int64]
    L.q3o_audio_frames.restype = C.c_int32
    L.q3o_speaker_encode.argtypes = [C.POINTER(_abi.CloneConfig), C.c_uint64, f32p, C.c_int32, f32p]
    L.q3o_speaker_encode.restype = C.c_int32
    L.q3o_audio_encode.argtypes = [C.POINTER(_abi.CloneConfig), C.c_uint64, f32p, C.c_int64, i32p, C.c_int32, f32p]
    L.q3o_audio_encode.restype = C.c_int32
    u16p = C.POINTER(C.c_uint16)
    L.q3o_mfma_bf16_dot32.argtypes = [u16p, u16p, C.c_float]
    L.q3o_mfma_bf16_dot32.restype = C.c_float
    L.q3o_mfma_bf16_dot32_ref.argtypes = [u16p, u16p, C.c_float]
    L.q3o_mfma_bf16_dot32_ref.restype = C.c_float
    L.q3o_bgemm.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_int32,
                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.q3o_bgemm.restype = None
    L.q3o_bgemm_q8.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_int32,
                               C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.q3o_bgemm_q8.restype = None
    L.q3o_quantize_q8_0.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    L.q3o_quantize_q8_0.restype = None
    L.q3o_bgemm_q8a8.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_int32,
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.q3o_bgemm_q8a8.restype = None
    L.q3o_set_talker_q8a8.argtypes = [vp]
    L.q3o_set_talker_q8a8.restype = None
    L.q3o_set_talker_q8.argtypes = [vp]
    L.q3o_set_talker_q8.restype = None
    L.q3o_project_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]
    L.q3o_project_rows.restype = None
    L.q3o_norm_inputs.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.q3o_norm_inputs.restype = None
    L.q3o_row_scale.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_float]
    L.q3o_row_scale.restype = C.c_float
    L.q3o_set_threads.argtypes = [C.c_int32]
    L.q3o_set_threads.restype = None
    L.q3o_set_arith.argtypes = [vp, C.c_int32]
    L.q3o_set_arith.restype = None
    L.q3o_norm_weight.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32]
    L.q3o_norm_weight.restype = f32p
    L.q3o_matrix.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
    L.q3o_matrix.restype = C.c_int32
    if hasattr(L, "q3o_vocoder_create"):
        L.q3o_vocoder_create.argtypes = [C.POINTER(_abi.VocoderConfig), C.c_uint64, C.c_int32]
        L.q3o_vocoder_create.restype = vp
        L.q3o_vocoder_destroy.argtypes = [vp]
        L.q3o_vocoder_destroy.restype = None
        L.q3o_vocoder_reset.argtypes = [vp]
        L.q3o_vocoder_reset.restype = None
        L.q3o_vocoder_set_arith.argtypes = [vp, C.c_int32]
        L.q3o_vocoder_set_arith.restype = None
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

    def set_talker_q8(self):
        """The Talker's matrices + lm_head as ggml Q8_0 blocks in the canonical Q8 order (device: q3tts_engine_config.talker_q8_0 = 1). One-way."""
        self.L.q3o_set_talker_q8(self.h)

    def set_talker_q8a8(self):
        """... and every Talker GEMM's activations as Q8_0 blocks as well: ggml's W8A8 (device: talker_q8_0 = 2). One-way."""
        self.L.q3o_set_talker_q8a8(self.h)

    def set_arith(self, mode):
        """0: the canonical bf16-MFMA order (default); 1: plain f32 of the same structure (family pinning)."""
        self.L.q3o_set_arith(self.h, mode)

    def matrix(self, talker, layer, which):
        """Natural-order f32 copy of one synthetic matrix: which 0 qkv, 1 o, 2 gate, 3 up, 4 down, 5 head."""
        n = self.L.q3o_matrix(self.h, int(talker), layer, which, None)
        c = self.cfg
        d, nq, F = (c.t_d_model, c.t_n_head * c.t_head_dim, c.t_d_ffn) if talker else (c.p_d_model, c.p_n_head * c.p_head_dim, c.p_d_ffn)
        K = {0: d, 1: nq, 2: d, 3: d, 4: F, 5: d}[which]
        out = np.zeros((n, K), dtype=np.float32)
        self.L.q3o_matrix(self.h, int(talker), layer, which, out.ctypes.data)
        return out

    def norm_weight(self, talker, layer, which, n):
        """which 0 attn_norm, 1 ffn_norm, 2 q_norm, 3 k_norm; layer < 0: the output norm."""
        return np.ctypeslib.as_array(self.L.q3o_norm_weight(self.h, int(talker), layer, which), shape=(n,)).copy()

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


def bgemm(xb, wb, ssp, d_norm, eps, epi, nw_next=None, y0=None):
    """oracle/q3_oracle_bf16.c q3o_bgemm on natural-order bf16 bit arrays; returns the outputs of the epilogue as a dict."""
    xb = np.ascontiguousarray(xb, dtype=np.uint16); wb = np.ascontiguousarray(wb, dtype=np.uint16)
    B, K = xb.shape
    N = wb.shape[0]
    y = np.zeros((B, N), dtype=np.float32) if y0 is None else np.ascontiguousarray(y0, dtype=np.float32).copy()
    yb = np.zeros((B, N // 2 if epi == 2 else N), dtype=np.uint16)
    sso = np.zeros((B, N // 16), dtype=np.float32)
    keys = np.zeros(B, dtype=np.uint64)
    sp = None if ssp is None else np.ascontiguousarray(ssp, dtype=np.float32)
    nw = None if nw_next is None else np.ascontiguousarray(nw_next, dtype=np.float32)
    lib().q3o_bgemm(xb.ctypes.data, B, K, wb.ctypes.data, N, None if sp is None else sp.ctypes.data, 0 if sp is None else sp.shape[1], d_norm, eps, epi,
                    None if nw is None else nw.ctypes.data, y.ctypes.data, yb.ctypes.data, sso.ctypes.data, keys.ctypes.data)
    return dict(y=y, yb=yb, ssp_out=sso, keys=keys)


def quantize_q8_0(x):
    """ggml's reference Q8_0 quantiser through the oracle's C restatement: x [..., K] f32 -> (q int8 same shape, d f16 bits [..., K/32])."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    q = np.zeros(x.shape, dtype=np.int8); d = np.zeros(x.shape[:-1] + (x.shape[-1] // 32,), dtype=np.uint16)
    lib().q3o_quantize_q8_0(x.ctypes.data, x.size, q.ctypes.data, d.ctypes.data)
    return q, d


def bgemm_q8a8(aq, ad, q, d16, ssp, d_norm, eps, epi, nw_next=None, y0=None):
    """oracle/q3_oracle_bf16.c q3o_bgemm_q8a8: ggml's Q8_0 x Q8_0 arithmetic (exact int32 block sums x f32(d_w) * f32(d_x)), the quantising epilogues."""
    aq = np.ascontiguousarray(aq, dtype=np.int8); ad = np.ascontiguousarray(ad, dtype=np.uint16)
    q = np.ascontiguousarray(q, dtype=np.int8); d16 = np.ascontiguousarray(d16, dtype=np.uint16)
    B, K = aq.shape
    N = q.shape[0]
    nout = N // 2 if epi == 2 else N
    y = np.zeros((B, N), dtype=np.float32) if y0 is None else np.ascontiguousarray(y0, dtype=np.float32).copy()
    yq = np.zeros((B, nout), dtype=np.int8); yd = np.zeros((B, nout // 32), dtype=np.uint16)
    sso = np.zeros((B, N // 16), dtype=np.float32)
    sp = None if ssp is None else np.ascontiguousarray(ssp, dtype=np.float32)
    nw = None if nw_next is None else np.ascontiguousarray(nw_next, dtype=np.float32)
    lib().q3o_bgemm_q8a8(aq.ctypes.data, ad.ctypes.data, B, K, q.ctypes.data, d16.ctypes.data, N, None if sp is None else sp.ctypes.data,
                         0 if sp is None else sp.shape[1], d_norm, eps, epi, None if nw is None else nw.ctypes.data, y.ctypes.data, yq.ctypes.data,
                         yd.ctypes.data, sso.ctypes.data)
    return dict(y=y, yq=yq, yd=yd, ssp_out=sso)


def bgemm_q8(xb, q, d16, ssp, d_norm, eps, epi, nw_next=None, y0=None):
    """oracle/q3_oracle_bf16.c q3o_bgemm_q8: the canonical Q8_0 order (per block P = MFMA from zero, t = fmaf(f32(d), P, t))."""
    xb = np.ascontiguousarray(xb, dtype=np.uint16); q = np.ascontiguousarray(q, dtype=np.int8); d16 = np.ascontiguousarray(d16, dtype=np.uint16)
    B, K = xb.shape
    N = q.shape[0]
    y = np.zeros((B, N), dtype=np.float32) if y0 is None else np.ascontiguousarray(y0, dtype=np.float32).copy()
    yb = np.zeros((B, N // 2 if epi == 2 else N), dtype=np.uint16)
    sso = np.zeros((B, N // 16), dtype=np.float32)
    keys = np.zeros(B, dtype=np.uint64)
    sp = None if ssp is None else np.ascontiguousarray(ssp, dtype=np.float32)
    nw = None if nw_next is None else np.ascontiguousarray(nw_next, dtype=np.float32)
    lib().q3o_bgemm_q8(xb.ctypes.data, B, K, q.ctypes.data, d16.ctypes.data, N, None if sp is None else sp.ctypes.data, 0 if sp is None else sp.shape[1],
                       d_norm, eps, epi, None if nw is None else nw.ctypes.data, y.ctypes.data, yb.ctypes.data, sso.ctypes.data, keys.ctypes.data)
    return dict(y=y, yb=yb, ssp_out=sso, keys=keys)


def project_rows(w, b, x):
    w = np.ascontiguousarray(w, dtype=np.float32); b = np.ascontiguousarray(b, dtype=np.float32); x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.zeros((x.shape[0], w.shape[0]), dtype=np.float32)
    lib().q3o_project_rows(w.ctypes.data, b.ctypes.data, w.shape[1], w.shape[0], x.ctypes.data, x.shape[0], y.ctypes.data)
    return y


def norm_inputs(x, nw):
    x = np.ascontiguousarray(x, dtype=np.float32); nw = np.ascontiguousarray(nw, dtype=np.float32)
    rows, d = x.shape
    xb = np.zeros((rows, d), dtype=np.uint16); ssp = np.zeros((rows, d // 16), dtype=np.float32)
    for r in range(rows):
        lib().q3o_norm_inputs(x[r].ctypes.data, d, nw.ctypes.data, xb[r].ctypes.data, ssp[r].ctypes.data)
    return xb, ssp


def row_scale(ssp_row, d, eps):
    sp = np.ascontiguousarray(ssp_row, dtype=np.float32)
    return float(lib().q3o_row_scale(sp.ctypes.data, sp.size, d, eps))


# ---- the synthetic model as files (loader parity): same tensor ids / scales as oracle/q3_oracle.c tfm_init, q3o_create
_G_TALKER, _G_PRED, _G_ASSET = 1, 2, 3
_W = dict(attn_norm=0, q=1, k=2, v=3, qnorm=4, knorm=5, o=6, ffn_norm=7, gate=8, up=9, down=10)
_L_MODEL = 255


def _tid(g, l, w):
    return (g << 16) | (l << 8) | w


def synth_tensor(seed, tid, shape, base, std, round_bf16):
    out = np.zeros(int(np.prod(shape)), dtype=np.float32)
    lib().q3o_synth_fill(seed, tid, out.size, base, std, 1 if round_bf16 else 0, ptr(out, f32p))
    return out.reshape(shape)


def synth_transformer_tensors(m, seed, talker):
    """llama.cpp qwen3 tensor names -> f32 arrays of the synthetic Talker / Predictor (matrices are bf16-exact)."""
    if talker:
        g, L, d, Hq, Hkv, hd, F, head_n = _G_TALKER, m.t_n_layer, m.t_d_model, m.t_n_head, m.t_n_kv_head, m.t_head_dim, m.t_d_ffn, m.t_vocab
    else:
        g, L, d, Hq, Hkv, hd, F = _G_PRED, m.p_n_layer, m.p_d_model, m.p_n_head, m.p_n_kv_head, m.p_head_dim, m.p_d_ffn
        head_n = (m.n_codebooks - 1) * m.codebook_size
    nq, nkv = Hq * hd, Hkv * hd
    t = {}
    for l in range(L):
        b = "blk.%d." % l
        t[b + "attn_norm.weight"] = synth_tensor(seed, _tid(g, l, _W["attn_norm"]), (d,), 1.0, 0.05, False)
        t[b + "ffn_norm.weight"] = synth_tensor(seed, _tid(g, l, _W["ffn_norm"]), (d,), 1.0, 0.05, False)
        t[b + "attn_q_norm.weight"] = synth_tensor(seed, _tid(g, l, _W["qnorm"]), (hd,), 1.0, 0.05, False)
        t[b + "attn_k_norm.weight"] = synth_tensor(seed, _tid(g, l, _W["knorm"]), (hd,), 1.0, 0.05, False)
        t[b + "attn_q.weight"] = synth_tensor(seed, _tid(g, l, _W["q"]), (nq, d), 0.0, 0.02, True)
        t[b + "attn_k.weight"] = synth_tensor(seed, _tid(g, l, _W["k"]), (nkv, d), 0.0, 0.02, True)
        t[b + "attn_v.weight"] = synth_tensor(seed, _tid(g, l, _W["v"]), (nkv, d), 0.0, 0.02, True)
        t[b + "attn_output.weight"] = synth_tensor(seed, _tid(g, l, _W["o"]), (d, nq), 0.0, 0.02, True)
        t[b + "ffn_gate.weight"] = synth_tensor(seed, _tid(g, l, _W["gate"]), (F, d), 0.0, 0.02, True)
        t[b + "ffn_up.weight"] = synth_tensor(seed, _tid(g, l, _W["up"]), (F, d), 0.0, 0.02, True)
        t[b + "ffn_down.weight"] = synth_tensor(seed, _tid(g, l, _W["down"]), (d, F), 0.0, 0.02, True)
    t["output_norm.weight"] = synth_tensor(seed, _tid(g, _L_MODEL, 0), (d,), 1.0, 0.05, False)
    t["output.weight"] = synth_tensor(seed, _tid(g, _L_MODEL, 1), (head_n, d), 0.0, 0.02, True)
    return t


def synth_asset_tensors(m, seed, with_text=True):
    """qwen3_assets.gguf tensor names (src/assets_manager.rs:212-241) -> f32 arrays of the synthetic assets."""
    d = m.d_embed
    t = {"proj.weight": synth_tensor(seed, _tid(_G_ASSET, 0, 1), (m.p_d_model, d), 0.0, 0.02, True),
         "proj.bias": synth_tensor(seed, _tid(_G_ASSET, 0, 2), (m.p_d_model,), 0.0, 0.02, False)}
    if with_text:
        t["text_embd"] = synth_tensor(seed, _tid(_G_ASSET, 0, 0), (m.text_vocab, d), 0.0, 0.05, True)
    for q in range(m.n_codebooks):
        rows = m.codec0_rows if q == 0 else m.codecq_rows
        t["codec_embd.%d" % q] = synth_tensor(seed, _tid(_G_ASSET, 1 + q, 0), (rows, d), 0.0, 0.05, True)
    return t


def write_model_dir(path, m, seed, matrix_type=30, assets="gguf", with_text=True, predictor_type=None):
    """The synthetic model as the reference's quant directory: two llama.cpp-style GGUFs + qwen3_assets.gguf (or NPY).
    predictor_type: tensor type of the Predictor's matrices when it differs from the Talker's (matrix_type)."""
    import _gguf as G
    os.makedirs(path, exist_ok=True)
    for talker, fname in ((True, "qwen3_tts_talker.gguf"), (False, "qwen3_tts_predictor.gguf")):
        tens = synth_transformer_tensors(m, seed, talker)
        mt = matrix_type if (talker or predictor_type is None) else predictor_type
        G.write(os.path.join(path, fname), [(k, v, mt if v.ndim == 2 else G.F32) for k, v in tens.items()],
                meta={"general.architecture": "qwen3", "general.alignment": 32, "qwen3.block_count": m.t_n_layer if talker else m.p_n_layer,
                      "tokenizer.ggml.tokens": ["<a>", "<b>"], "tokenizer.ggml.token_type": [1, 1]})
    at = synth_asset_tensors(m, seed, with_text)
    if assets == "gguf":
        G.write(os.path.join(path, "qwen3_assets.gguf"), [(k, v, G.F32) for k, v in at.items()], meta={"general.architecture": "qwen3-tts-assets"})
    else:  # the legacy NPY layout (src/assets_manager.rs:267-300)
        np.save(os.path.join(path, "proj_weight.npy"), at["proj.weight"])
        np.save(os.path.join(path, "proj_bias.npy"), at["proj.bias"])
        if with_text:
            np.save(os.path.join(path, "text_embedding_projected.npy"), at["text_embd"])
        for q in range(m.n_codebooks):
            np.save(os.path.join(path, "codec_embedding_%d.npy" % q), at["codec_embd.%d" % q])


def mel(audio, want_pre_log=False):
    """Oracle log-mel (oracle/q3_oracle_mel.c): [n_frames][128]."""
    a = np.ascontiguousarray(audio, dtype=np.float32)
    cap = max(1, int(lib().q3o_mel_frames(a.size)))
    out = np.zeros((cap, 128), dtype=np.float32)
    pre = np.zeros((cap, 128), dtype=np.float32)
    n = lib().q3o_mel(ptr(a, f32p) if a.size else None, a.size, ptr(out, f32p), ptr(pre, f32p))
    return (out[:n].copy(), pre[:n].copy()) if want_pre_log else out[:n].copy()


def speaker_encode(ccfg, seed, mel_rows):
    """Oracle speaker encoder (oracle/q3_oracle_clone.c) on a log-mel [T][128] -> [se_dim]."""
    m = np.ascontiguousarray(mel_rows, dtype=np.float32)
    out = np.zeros(ccfg.se_dim, dtype=np.float32)
    rc = lib().q3o_speaker_encode(C.byref(ccfg), seed, ptr(m, f32p), m.shape[0], ptr(out, f32p))
    assert rc == 0
    return out


def audio_encode(ccfg, seed, audio):
    """Oracle audio encoder: (codes [frames][ncb] int32, pre-quantiser rows [frames][ae_hidden])."""
    a = np.ascontiguousarray(audio, dtype=np.float32)
    cap = max(1, int(lib().q3o_audio_frames(C.byref(ccfg), a.size)))
    codes = np.zeros((cap, ccfg.ae_n_codebooks), dtype=np.int32)
    lat = np.zeros((cap, ccfg.ae_hidden), dtype=np.float32)
    n = lib().q3o_audio_encode(C.byref(ccfg), seed, ptr(a, f32p) if a.size else None, a.size, ptr(codes, i32p), cap, ptr(lat, f32p))
    assert n >= 0
    return codes[:n].copy(), lat[:n].copy()
