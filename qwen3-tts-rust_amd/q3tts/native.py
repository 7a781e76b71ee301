"""Thin object wrapper over the C ABI (include/q3tts.h). No arithmetic happens in Python."""
import ctypes as C
import os

import numpy as np

from . import _abi

f32p, i32p, u32p = C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_uint32)


def _ptr(a, typ):
    return a.ctypes.data_as(typ)


def make_prompt_desc(text_ids, spk_emb=None, lang_id=2055, spk_id=-1, instruct_ids=None, ref_codes=None,
                     ref_text_ids=None):
    """Returns (PromptDesc, keepalive list)."""
    keep = []
    d = _abi.PromptDesc()
    t = np.ascontiguousarray(text_ids, dtype=np.uint32)
    keep.append(t)
    d.text_ids, d.n_text = _ptr(t, u32p), len(t)
    if instruct_ids is not None:
        ins = np.ascontiguousarray(instruct_ids, dtype=np.uint32)
        keep.append(ins)
        d.instruct_ids, d.n_instruct = _ptr(ins, u32p), len(ins)
    d.lang_id, d.spk_id = (-1 if lang_id is None else lang_id), spk_id
    if spk_emb is not None:
        s = np.ascontiguousarray(spk_emb, dtype=np.float32)
        keep.append(s)
        d.spk_emb = _ptr(s, f32p)
    if ref_codes is not None:
        rc = np.ascontiguousarray(ref_codes, dtype=np.int32)
        keep.append(rc)
        d.ref_codes, d.n_ref_frames = _ptr(rc, i32p), rc.size // 16
        rt = np.ascontiguousarray(ref_text_ids if ref_text_ids is not None else [], dtype=np.uint32)
        keep.append(rt)
        d.ref_text_ids, d.n_ref_text = _ptr(rt, u32p), len(rt)
    return d, keep


class GenResult:
    def __init__(self, status, codes, pcm, hit_eos, first_chunk_ms, total_ms, sample_rate):
        self.status, self.codes, self.pcm, self.hit_eos = status, codes, pcm, hit_eos
        self.first_chunk_ms, self.total_ms, self.sample_rate = first_chunk_ms, total_ms, sample_rate

    @property
    def n_frames(self):
        return self.codes.shape[0]


class NativeEngine:
    """q3tts_engine handle. One engine per GPU; not re-entrant (same contract as `&mut self`)."""

    def __init__(self, cfg):
        self.lib = _abi.load_library()
        self.cfg = cfg
        self.h = C.c_void_p()
        rc = self.lib.q3tts_engine_create(C.byref(cfg), C.byref(self.h))
        if rc != 0:
            raise _abi.Q3Error(f"q3tts_engine_create failed ({rc}): {self.lib.q3tts_last_error(None).decode()}")

    def close(self):
        if getattr(self, "h", None):
            self.lib.q3tts_engine_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def _check(self, rc, what):
        if rc != 0:
            raise _abi.Q3Error(f"{what} failed ({rc}): {self.lib.q3tts_last_error(self.h).decode()}")

    def set_sampler(self, temperature, top_k, top_p, seed=None):
        self._check(self.lib.q3tts_set_sampler(self.h, temperature, top_k, top_p, 0 if seed is None else 1, seed or 0),
                    "q3tts_set_sampler")

    def set_max_steps(self, n):
        self._check(self.lib.q3tts_set_max_steps(self.h, n), "q3tts_set_max_steps")

    def build_prompt(self, desc):
        out, n = f32p(), C.c_int32()
        self._check(self.lib.q3tts_build_prompt(self.h, C.byref(desc), C.byref(out), C.byref(n)), "q3tts_build_prompt")
        d = self.cfg.model.d_embed
        arr = np.ctypeslib.as_array(out, shape=(n.value, d)).copy()
        self.lib.q3tts_free(out)
        return arr

    @staticmethod
    def make_request(embd=None, desc=None, temperature=0.0, top_k=40, top_p=0.9, seed=None, max_steps=0, min_frames=0,
                     force_eos_at=-1, want_pcm=0, use_engine_sampler=0):
        r = _abi.Request()
        keep = []
        if embd is not None:
            e = np.ascontiguousarray(embd, dtype=np.float32)
            keep.append(e)
            r.prompt_embd, r.n_tok = _ptr(e, f32p), e.shape[0]
        if desc is not None:
            keep.append(desc)
            r.prompt = C.pointer(desc)
        r.use_engine_sampler = use_engine_sampler
        r.temperature, r.top_k, r.top_p = temperature, top_k, top_p
        r.has_seed, r.seed = (0, 0) if seed is None else (1, seed)
        r.max_steps, r.min_frames, r.force_eos_at, r.want_pcm = max_steps, min_frames, force_eos_at, want_pcm
        return r, keep

    def _unpack(self, res):
        ncb = self.cfg.model.n_codebooks
        codes = np.ctypeslib.as_array(res.codes, shape=(res.n_frames, ncb)).copy() if res.n_frames > 0 else \
            np.zeros((0, ncb), dtype=np.int32)
        pcm = None
        if res.pcm:
            pcm = np.ctypeslib.as_array(res.pcm, shape=(res.n_samples,)).copy() if res.n_samples > 0 else \
                np.zeros(0, dtype=np.float32)
        out = GenResult(res.status, codes, pcm, bool(res.hit_eos), res.first_chunk_ms, res.total_ms, res.sample_rate)
        out.n_samples = int(res.n_samples)
        self.lib.q3tts_result_free(C.byref(res))
        return out

    def generate(self, **kw):
        r, keep = self.make_request(**kw)
        res = _abi.Result()
        self._check(self.lib.q3tts_generate(self.h, C.byref(r), C.byref(res)), "q3tts_generate")
        return self._unpack(res)

    def generate_batch(self, requests):
        """requests: list of kwargs dicts for make_request."""
        n = len(requests)
        arr = (_abi.Request * n)()
        keep = []
        for i, kw in enumerate(requests):
            r, k = self.make_request(**kw)
            arr[i] = r
            keep.append(k)
        res = (_abi.Result * n)()
        self._check(self.lib.q3tts_generate_batch(self.h, arr, n, res), "q3tts_generate_batch")
        return [self._unpack(res[i]) for i in range(n)]

    def set_device_pcm(self, enable=True):
        """Keep every batch's PCM on the device as well (row i of one buffer per generate_batch call): the multi-GPU gather reads it there."""
        self._check(self.lib.q3tts_set_device_pcm(self.h, 1 if enable else 0), "q3tts_set_device_pcm")

    def device_pcm(self):
        """(device pointer, stride in samples, rows) of the last batch's packed PCM; wrap with torch via __cuda_array_interface__."""
        base, stride, n = f32p(), C.c_int64(0), C.c_int32(0)
        self._check(self.lib.q3tts_get_device_pcm(self.h, C.byref(base), C.byref(stride), C.byref(n)), "q3tts_get_device_pcm")
        return (C.cast(base, C.c_void_p).value or 0), int(stride.value), int(n.value)

    def vocoder_bench(self, n_slots, chunks):
        ms = C.c_float(0)
        self._check(self.lib.q3tts_k_vocoder_bench(self.h, n_slots, chunks, C.byref(ms)), "q3tts_k_vocoder_bench")
        return ms.value

    def timings(self):
        t = _abi.Timings()
        self._check(self.lib.q3tts_get_timings(self.h, C.byref(t)), "q3tts_get_timings")
        return t

    def mel(self, audio):
        """24 kHz mono f32 -> [n_frames][128] log-mel (the clone path's front-end)."""
        a = np.ascontiguousarray(audio, dtype=np.float32)
        cap = int(self.lib.q3tts_mel_frames(a.size))
        out = np.zeros((max(cap, 1), 128), dtype=np.float32)
        n = C.c_int32(0)
        self._check(self.lib.q3tts_mel(self.h, _ptr(a, f32p) if a.size else None, a.size, _ptr(out, f32p), cap, C.byref(n)), "q3tts_mel")
        return out[:n.value].copy()

    # voice-clone encoders (q3_clone.hip) ----------------------------------------------------------
    def clone_init(self, ccfg):
        self._check(self.lib.q3tts_clone_init(self.h, C.byref(ccfg)), "q3tts_clone_init")
        self.clone_cfg = ccfg

    def clone_audio_frames(self, n_samples):
        return int(self.lib.q3tts_clone_audio_frames(self.h, int(n_samples)))

    def _clone_codebooks(self):
        c = getattr(self, "clone_cfg", None)
        return c.ae_n_codebooks if c is not None else 16

    def audio_encode(self, audio):
        """AudioEncoder::encode: 24 kHz mono f32 -> i64 codes [n_frames][n_codebooks]."""
        a = np.ascontiguousarray(audio, dtype=np.float32)
        cap = max(self.clone_audio_frames(a.size), 1)
        ncb = self._clone_codebooks()
        out = np.zeros((cap, ncb), dtype=np.int64)
        n = C.c_int32(0)
        self._check(self.lib.q3tts_clone_audio_encode(self.h, _ptr(a, f32p) if a.size else None, a.size,
                                                      out.ctypes.data_as(C.POINTER(C.c_int64)), cap, C.byref(n)), "q3tts_clone_audio_encode")
        return out[:n.value].copy()

    def audio_latent(self, audio):
        a = np.ascontiguousarray(audio, dtype=np.float32)
        cap = max(self.clone_audio_frames(a.size), 1)
        out = np.zeros((cap, self.clone_cfg.ae_hidden), dtype=np.float32)
        n = C.c_int32(0)
        self._check(self.lib.q3tts_k_audio_latent(self.h, _ptr(a, f32p) if a.size else None, a.size, _ptr(out, f32p), cap, C.byref(n)),
                    "q3tts_k_audio_latent")
        return out[:n.value].copy()

    def speaker_encode(self, audio):
        """SpeakerEncoder::encode: 24 kHz mono f32 -> [se_dim] (log-mel on the device, then the encoder)."""
        a = np.ascontiguousarray(audio, dtype=np.float32)
        c = getattr(self, "clone_cfg", None)
        out = np.zeros(c.se_dim if c is not None else self.cfg.model.d_embed, dtype=np.float32)
        self._check(self.lib.q3tts_clone_speaker_encode(self.h, _ptr(a, f32p) if a.size else None, a.size, _ptr(out, f32p)),
                    "q3tts_clone_speaker_encode")
        return out

    def speaker_from_mel(self, mel):
        m = np.ascontiguousarray(mel, dtype=np.float32)
        out = np.zeros(self.clone_cfg.se_dim, dtype=np.float32)
        self._check(self.lib.q3tts_k_speaker_from_mel(self.h, _ptr(m, f32p), m.shape[0], _ptr(out, f32p)), "q3tts_k_speaker_from_mel")
        return out

    def probe(self, enable):
        """Measurement mode (bench.py): eager frame steps with HIP events around one GEMM per frame — 2 (or True): the Talker's
        layer-0 gate/up (exact kernel); 1: the Predictor's pass-1 gate/up (bf16-MFMA kernel); 0 / False: off."""
        mode = 2 if enable is True else int(enable)
        self._check(self.lib.q3tts_k_probe(self.h, mode), "q3tts_k_probe")

    def alloc_upload_mismatches(self, nbytes):
        """Allocator contract hook (q3tts_k_alloc_upload): bytes that read back wrong after a null-stream upload right behind the allocation."""
        bad = C.c_int64(-1)
        self._check(self.lib.q3tts_k_alloc_upload(self.h, int(nbytes), C.byref(bad)), "q3tts_k_alloc_upload")
        return int(bad.value)

    def talker_prefill(self, embd):
        e = np.ascontiguousarray(embd, dtype=np.float32)
        hid = np.zeros(self.cfg.model.t_d_model, dtype=np.float32)
        lg = np.zeros(self.cfg.model.t_vocab, dtype=np.float32)
        self._check(self.lib.q3tts_k_talker_prefill(self.h, _ptr(e, f32p), e.shape[0], _ptr(hid, f32p), _ptr(lg, f32p)),
                    "q3tts_k_talker_prefill")
        return hid, lg

    def vocoder(self, codes, chunk_frames=0):
        c = np.ascontiguousarray(codes, dtype=np.int32)
        spf = 1
        v = self.cfg.vocoder
        for i in range(v.n_upsample):
            spf *= v.upsample_ratios[i]
        for i in range(v.n_dec_blocks):
            spf *= v.dec_rates[i]
        pcm = np.zeros(c.shape[0] * spf + 16, dtype=np.float32)
        ns = C.c_int32()
        self._check(self.lib.q3tts_k_vocoder(self.h, _ptr(c, i32p), c.shape[0], chunk_frames, _ptr(pcm, f32p), C.byref(ns)),
                    "q3tts_k_vocoder")
        return pcm[:ns.value].copy()


# ---- kernel-level hooks (host arrays in / out) -------------------------------------------------

class NativeNode:
    """q3tts_node_*: one engine + one host thread per listed GPU inside the library, requests sharded by index (i mod G), the PCM of a
    batch optionally gathered to the first device as i16 over RCCL (include/q3tts.h)."""

    def __init__(self, cfg, devices):
        self.lib = _abi.load_library()
        self.cfg = cfg
        self.devices = [int(d) for d in devices]
        arr = (C.c_int32 * len(self.devices))(*self.devices)
        self.h = C.c_void_p()
        rc = self.lib.q3tts_node_create(C.byref(cfg), arr, len(self.devices), C.byref(self.h))
        if rc != 0:
            raise _abi.Q3Error(f"q3tts_node_create failed ({rc}): {self.lib.q3tts_node_last_error(None).decode()}")

    def close(self):
        if self.h:
            self.lib.q3tts_node_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        self.close()

    def generate_batch(self, requests, gather_i16=False):
        """requests: kwargs dicts for NativeEngine.make_request. Returns GenResult list; with gather_i16 every result carries .pcm_i16
        (np.int16, gathered through device 0) instead of .pcm."""
        n = len(requests)
        arr = (_abi.Request * n)()
        keep = []
        for i, kw in enumerate(requests):
            r, k = NativeEngine.make_request(**kw)
            arr[i] = r
            keep.append(k)
        res = (_abi.Result * n)()
        i16 = (C.POINTER(C.c_int16) * n)() if gather_i16 else None
        rc = self.lib.q3tts_node_generate_batch(self.h, arr, n, res, i16)
        if rc != 0:
            self.last_failed_statuses = [int(res[i].status) for i in range(n)]
            for i in range(n):   # (the contract: every result is valid to free, whatever the return code; no i16 buffer is left allocated)
                self.lib.q3tts_result_free(C.byref(res[i]))
            assert not gather_i16 or not any(bool(i16[i]) for i in range(n))
            raise _abi.Q3Error(f"q3tts_node_generate_batch failed ({rc}): {self.lib.q3tts_node_last_error(self.h).decode()}")
        ncb = self.cfg.model.n_codebooks
        outs = []
        for i in range(n):
            r = res[i]
            codes = np.ctypeslib.as_array(r.codes, shape=(r.n_frames, ncb)).copy() if r.n_frames > 0 else np.zeros((0, ncb), dtype=np.int32)
            pcm = np.ctypeslib.as_array(r.pcm, shape=(r.n_samples,)).copy() if (r.pcm and r.n_samples > 0) else None
            o = GenResult(r.status, codes, pcm, bool(r.hit_eos), r.first_chunk_ms, r.total_ms, r.sample_rate)
            o.n_samples = int(r.n_samples)
            o.pcm_i16 = None
            if gather_i16 and i16[i]:
                o.pcm_i16 = np.ctypeslib.as_array(i16[i], shape=(r.n_samples,)).copy()
                self.lib.q3tts_free(C.cast(i16[i], C.c_void_p))
            self.lib.q3tts_result_free(C.byref(r))
            outs.append(o)
        return outs

    def timings(self):
        t = _abi.NodeTimings()
        self.lib.q3tts_node_get_timings(self.h, C.byref(t))
        return t


def node_shard(n_total, world, rank):
    """q3tts_node_shard (host only): the global request indices device `rank` of `world` owns, in order."""
    lib = _abi.load_library()
    k = lib.q3tts_node_shard(n_total, world, rank, None, 0)
    if k < 0:
        raise ValueError("bad shard arguments")
    idx = np.zeros(max(k, 1), dtype=np.int32)
    lib.q3tts_node_shard(n_total, world, rank, _ptr(idx, i32p), k)
    return idx[:k].tolist()


def k_gemm_exact(x, w_bf16, norm_w=None, eps=1e-6, bias=None, epilogue=0, y_in=None, device=0, iters=0):
    lib = _abi.load_library()
    x = np.ascontiguousarray(x, dtype=np.float32)
    w = np.ascontiguousarray(w_bf16, dtype=np.uint16)
    B, K = x.shape
    N = w.shape[0]
    ny = N // 2 if epilogue == 2 else N
    y = np.zeros((B, ny), dtype=np.float32) if y_in is None else np.ascontiguousarray(y_in, dtype=np.float32).copy()
    keys = np.zeros(B, dtype=np.uint64)
    ms = C.c_float(0)
    nw = None if norm_w is None else np.ascontiguousarray(norm_w, dtype=np.float32)
    bs = None if bias is None else np.ascontiguousarray(bias, dtype=np.float32)
    rc = lib.q3tts_k_gemm_exact(device, _ptr(x, f32p), B, K, w.ctypes.data_as(C.POINTER(C.c_uint16)), N,
                                None if nw is None else _ptr(nw, f32p), eps, None if bs is None else _ptr(bs, f32p),
                                epilogue, _ptr(y, f32p), keys.ctypes.data_as(C.POINTER(C.c_uint64)), iters, C.byref(ms))
    if rc != 0:
        raise _abi.Q3Error(f"q3tts_k_gemm_exact failed ({rc}): {lib.q3tts_last_error(None).decode()}")
    return y, keys, ms.value


def k_bgemm(xb, wb, ssp, d_norm, eps, epi, nw_next=None, y0=None, device=0, iters=0):
    """The decoder's GEMM through its launcher (q3tts_k_bgemm): natural-order bf16 bit arrays in, the epilogue's outputs out."""
    lib = _abi.load_library()
    xb = np.ascontiguousarray(xb, dtype=np.uint16); wb = np.ascontiguousarray(wb, dtype=np.uint16)
    B, K = xb.shape
    N = wb.shape[0]
    y = np.zeros((B, N), dtype=np.float32) if y0 is None else np.ascontiguousarray(y0, dtype=np.float32).copy()
    yb = np.zeros((B, N // 2 if epi == 2 else N), dtype=np.uint16)
    sso = np.zeros((B, N // 16), dtype=np.float32)
    keys = np.zeros(B, dtype=np.uint64)
    sp = None if ssp is None else np.ascontiguousarray(ssp, dtype=np.float32)
    nw = None if nw_next is None else np.ascontiguousarray(nw_next, dtype=np.float32)
    ms = C.c_float(0)
    rc = lib.q3tts_k_bgemm(device, xb.ctypes.data, B, K, wb.ctypes.data, N, None if sp is None else sp.ctypes.data, 0 if sp is None else sp.shape[1],
                           d_norm, eps, epi, None if nw is None else nw.ctypes.data, y.ctypes.data, yb.ctypes.data, sso.ctypes.data, keys.ctypes.data,
                           iters, C.byref(ms))
    if rc != 0:
        raise _abi.Q3Error(f"q3tts_k_bgemm failed ({rc}): {lib.q3tts_last_error(None).decode()}")
    return dict(y=y, yb=yb, ssp_out=sso, keys=keys, ms=ms.value)


def k_bgemm_q8(xb, q, d16, ssp, d_norm, eps, epi, nw_next=None, y0=None, device=0, iters=0):
    """The same launch with ggml Q8_0 weights kept in block form on the device (q3tts_k_bgemm_q8): q int8 [N][K], d16 f16 bits [N][K/32]."""
    lib = _abi.load_library()
    xb = np.ascontiguousarray(xb, dtype=np.uint16); q = np.ascontiguousarray(q, dtype=np.int8); d16 = np.ascontiguousarray(d16, dtype=np.uint16)
    B, K = xb.shape
    N = q.shape[0]
    y = np.zeros((B, N), dtype=np.float32) if y0 is None else np.ascontiguousarray(y0, dtype=np.float32).copy()
    yb = np.zeros((B, N // 2 if epi == 2 else N), dtype=np.uint16)
    sso = np.zeros((B, N // 16), dtype=np.float32)
    keys = np.zeros(B, dtype=np.uint64)
    sp = None if ssp is None else np.ascontiguousarray(ssp, dtype=np.float32)
    nw = None if nw_next is None else np.ascontiguousarray(nw_next, dtype=np.float32)
    ms = C.c_float(0)
    rc = lib.q3tts_k_bgemm_q8(device, xb.ctypes.data, B, K, q.ctypes.data, d16.ctypes.data, N, None if sp is None else sp.ctypes.data,
                              0 if sp is None else sp.shape[1], d_norm, eps, epi, None if nw is None else nw.ctypes.data, y.ctypes.data, yb.ctypes.data,
                              sso.ctypes.data, keys.ctypes.data, iters, C.byref(ms))
    if rc != 0:
        raise _abi.Q3Error(f"q3tts_k_bgemm_q8 failed ({rc}): {lib.q3tts_last_error(None).decode()}")
    return dict(y=y, yb=yb, ssp_out=sso, keys=keys, ms=ms.value)


def k_bgemm_q8a8(aq, ad, q, d16, ssp, d_norm, eps, epi, nw_next=None, y0=None, device=0, iters=0):
    """The decoder's GEMM in ggml's Q8_0 x Q8_0 arithmetic (q3tts_k_bgemm_q8a8): activations aq int8 [B][K] + ad f16 bits [B][K/32], weights q / d16."""
    lib = _abi.load_library()
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
    ms = C.c_float(0)
    rc = lib.q3tts_k_bgemm_q8a8(device, aq.ctypes.data, ad.ctypes.data, B, K, q.ctypes.data, d16.ctypes.data, N, None if sp is None else sp.ctypes.data,
                                0 if sp is None else sp.shape[1], d_norm, eps, epi, None if nw is None else nw.ctypes.data, y.ctypes.data, yq.ctypes.data,
                                yd.ctypes.data, sso.ctypes.data, iters, C.byref(ms))
    if rc != 0:
        raise _abi.Q3Error(f"q3tts_k_bgemm_q8a8 failed ({rc}): {lib.q3tts_last_error(None).decode()}")
    return dict(y=y, yq=yq, yd=yd, ssp_out=sso, ms=ms.value)


def k_bgemm_voc(xb, wb, epi, bias=None, col_scale=None, seg_rows=0, gap_rows=0, y0=None, want_yb=False, device=0):
    """The decoder's GEMM with the vocoder's epilogue extras (q3tts_k_bgemm_voc): returns dict(y=[B][N] f32 or None, yb=[B][N] bf16 bits or None)."""
    lib = _abi.load_library()
    xb = np.ascontiguousarray(xb, dtype=np.uint16); wb = np.ascontiguousarray(wb, dtype=np.uint16)
    B, K = xb.shape
    N = wb.shape[0]
    y = None if epi == 4 else (np.zeros((B, N), dtype=np.float32) if y0 is None else np.ascontiguousarray(y0, dtype=np.float32).copy())
    yb = np.zeros((B, N), dtype=np.uint16) if (epi == 4 or want_yb) else None
    b = None if bias is None else np.ascontiguousarray(bias, dtype=np.float32)
    cs = None if col_scale is None else np.ascontiguousarray(col_scale, dtype=np.float32)
    rc = lib.q3tts_k_bgemm_voc(device, xb.ctypes.data, B, K, wb.ctypes.data, N, epi, None if b is None else b.ctypes.data, 0 if b is None else b.size,
                               None if cs is None else cs.ctypes.data, seg_rows, gap_rows, None if y is None else y.ctypes.data,
                               None if yb is None else yb.ctypes.data, 1 if want_yb else 0)
    if rc != 0:
        raise _abi.Q3Error(f"q3tts_k_bgemm_voc failed ({rc}): {lib.q3tts_last_error(None).decode()}")
    return dict(y=y, yb=yb)


def k_project(x, w, bias, nw=None, device=0):
    lib = _abi.load_library()
    x = np.ascontiguousarray(x, dtype=np.float32); w = np.ascontiguousarray(w, dtype=np.float32); b = np.ascontiguousarray(bias, dtype=np.float32)
    rows, n_in = x.shape
    n_out = w.shape[0]
    y = np.zeros((rows, n_out), dtype=np.float32)
    xb = np.zeros((rows, n_out), dtype=np.uint16); ssp = np.zeros((rows, n_out // 16), dtype=np.float32)
    nwa = None if nw is None else np.ascontiguousarray(nw, dtype=np.float32)
    rc = lib.q3tts_k_project(device, x.ctypes.data, rows, n_in, w.ctypes.data, b.ctypes.data, n_out, None if nwa is None else nwa.ctypes.data,
                             y.ctypes.data, xb.ctypes.data, ssp.ctypes.data)
    if rc != 0:
        raise _abi.Q3Error(f"q3tts_k_project failed ({rc}): {lib.q3tts_last_error(None).decode()}")
    return y, xb, ssp


def k_norm_inputs(x, nw, device=0):
    lib = _abi.load_library()
    x = np.ascontiguousarray(x, dtype=np.float32); nw = np.ascontiguousarray(nw, dtype=np.float32)
    rows, d = x.shape
    xb = np.zeros((rows, d), dtype=np.uint16); ssp = np.zeros((rows, d // 16), dtype=np.float32)
    rc = lib.q3tts_k_norm_inputs(device, x.ctypes.data, rows, d, nw.ctypes.data, xb.ctypes.data, ssp.ctypes.data)
    if rc != 0:
        raise _abi.Q3Error(f"q3tts_k_norm_inputs failed ({rc}): {lib.q3tts_last_error(None).decode()}")
    return xb, ssp


def k_attention(qkv, pos0, n_head, n_kv_head, head_dim, q_norm_w, k_norm_w, eps, rope_theta, sections, device=0):
    lib = _abi.load_library()
    qkv = np.ascontiguousarray(qkv, dtype=np.float32)
    n = qkv.shape[0]
    out = np.zeros((n, n_head * head_dim), dtype=np.float32)
    qn = np.ascontiguousarray(q_norm_w, dtype=np.float32)
    kn = np.ascontiguousarray(k_norm_w, dtype=np.float32)
    sec = None if sections is None else np.ascontiguousarray(sections, dtype=np.int32)
    rc = lib.q3tts_k_attention(device, _ptr(qkv, f32p), n, pos0, n_head, n_kv_head, head_dim, _ptr(qn, f32p), _ptr(kn, f32p), eps,
                               rope_theta, None if sec is None else _ptr(sec, i32p), _ptr(out, f32p))
    if rc != 0:
        raise _abi.Q3Error(f"q3tts_k_attention failed ({rc}): {lib.q3tts_last_error(None).decode()}")
    return out


def k_sample(logits, limit, temperature, top_k, top_p, r=None, device=0):
    lib = _abi.load_library()
    lg = np.ascontiguousarray(logits, dtype=np.float32)
    n, ld = lg.shape
    out = np.zeros(n, dtype=np.int32)
    rr = None if r is None else np.ascontiguousarray(r, dtype=np.float32)
    rc = lib.q3tts_k_sample(device, _ptr(lg, f32p), n, ld, limit, temperature, top_k, top_p, None if rr is None else _ptr(rr, f32p),
                            _ptr(out, i32p))
    if rc != 0:
        raise _abi.Q3Error(f"q3tts_k_sample failed ({rc}): {lib.q3tts_last_error(None).decode()}")
    return out


def k_gguf_read(path, tensor=""):
    """One tensor of a GGUF file / the array of an .npy file as f32, through the engine's own reader (host only).
    Returns (array shaped like numpy would show it, ggml type)."""
    lib = _abi.load_library()
    n = C.c_int64(0); dims = (C.c_int64 * 4)(); ty = C.c_int32(0)
    rc = lib.q3tts_k_gguf_read(os.fsencode(path), tensor.encode(), None, 0, C.byref(n), dims, C.byref(ty))
    if rc != 0:
        raise _abi.Q3Error(f"q3tts_k_gguf_read: {lib.q3tts_last_error(None).decode()} (status {rc})")
    out = np.zeros(n.value, dtype=np.float32)
    rc = lib.q3tts_k_gguf_read(os.fsencode(path), tensor.encode(), _ptr(out, f32p), n.value, C.byref(n), dims, C.byref(ty))
    if rc != 0:
        raise _abi.Q3Error(f"q3tts_k_gguf_read: {lib.q3tts_last_error(None).decode()} (status {rc})")
    shape = [int(d) for d in dims if d]
    if not str(path).endswith(".npy"):
        shape = shape[::-1]  # ggml lists the contiguous dimension first
    return out.reshape(shape), int(ty.value)


def k_rng_f32(seed, n):
    lib = _abi.load_library()
    out = np.zeros(n, dtype=np.float32)
    lib.q3tts_k_rng_f32(seed, n, _ptr(out, f32p))
    return out


def stream_chunks(engine: "NativeEngine", **kw):
    """Generator over the streaming C ABI (q3tts_stream_begin / _poll / _end): yields (pcm_chunk, is_final); the final
    StopIteration value is the GenResult with all codes and PCM."""
    r, keep = engine.make_request(**kw)
    h = C.c_void_p()
    engine._check(engine.lib.q3tts_stream_begin(engine.h, C.byref(r), C.byref(h)), "q3tts_stream_begin")
    chunk, n, fin = f32p(), C.c_int32(), C.c_int32()
    try:
        while True:
            engine._check(engine.lib.q3tts_stream_poll(h, C.byref(chunk), C.byref(n), C.byref(fin)), "q3tts_stream_poll")
            if n.value > 0:
                yield np.ctypeslib.as_array(chunk, shape=(n.value,)).copy(), bool(fin.value)
            if fin.value:
                break
    finally:
        res = _abi.Result()
        engine._check(engine.lib.q3tts_stream_end(h, C.byref(res)), "q3tts_stream_end")
        engine.last_stream_result = engine._unpack(res)


class NativeTokenizer:
    """The C++ byte-level BPE reader (csrc/q3_tokenizer.cpp) behind Tokenizer::load / encode / decode of the reference
    (src/utils/tokenizer.rs). Host only."""

    def __init__(self, tokenizer_json_path):
        self.lib = _abi.load_library()
        self.h = C.c_void_p()
        err = C.create_string_buffer(512)
        rc = self.lib.q3tts_tokenizer_load(os.fsencode(str(tokenizer_json_path)), C.byref(self.h), err, len(err))
        if rc != 0:
            raise _abi.Q3Error(f"q3tts_tokenizer_load failed ({rc}): {err.value.decode('utf-8', 'replace')}")

    def close(self):
        if self.h:
            self.lib.q3tts_tokenizer_free(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def vocab_size(self):
        return int(self.lib.q3tts_tokenizer_vocab_size(self.h))

    def encode(self, text):
        raw = text.encode("utf-8") if isinstance(text, str) else bytes(text)
        cap = max(16, len(raw) + 8)
        ids = np.zeros(cap, dtype=np.uint32)
        n = C.c_int32(0)
        err = C.create_string_buffer(512)
        rc = self.lib.q3tts_tokenizer_encode(self.h, raw, len(raw), _ptr(ids, u32p), cap, C.byref(n), err, len(err))
        if rc != 0 and n.value > cap:  # NFC can lengthen the text (composition exclusions): *n_ids carries the size needed
            cap = n.value
            ids = np.zeros(cap, dtype=np.uint32)
            rc = self.lib.q3tts_tokenizer_encode(self.h, raw, len(raw), _ptr(ids, u32p), cap, C.byref(n), err, len(err))
        if rc != 0:
            raise _abi.Q3Error(f"q3tts_tokenizer_encode failed ({rc}): {err.value.decode('utf-8', 'replace')}")
        return ids[:n.value].copy()

    def decode(self, ids):
        a = np.ascontiguousarray(ids, dtype=np.uint32)
        cap = 64 + 64 * a.size
        out = C.create_string_buffer(cap)
        n = C.c_int64(0)
        err = C.create_string_buffer(512)
        rc = self.lib.q3tts_tokenizer_decode(self.h, _ptr(a, u32p), a.size, out, cap, C.byref(n), err, len(err))
        if rc != 0:
            raise _abi.Q3Error(f"q3tts_tokenizer_decode failed ({rc}): {err.value.decode('utf-8', 'replace')}")
        return out.raw[:n.value].decode("utf-8", "replace")
