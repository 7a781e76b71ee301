"""Host-side mirror of the reference crate's public API (names, argument meaning and error behaviour), over the C ABI.

Reference surface (src/lib.rs:11-20): TtsEngine, SamplerConfig, VoiceFile, AudioSample, PromptBuilder, cleanup().
The reference is Rust; Rust is not available in this image, so the host side above the C ABI is mirrored here in
Python for the tests and in `rust/` as (uncompiled) binding source — see INTEGRATION.md.
"""
import ctypes as C
import json
import os
import struct
import wave
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Union

import numpy as np

from . import _abi, native

LANG_ID_CHINESE = 2055  # hard-coded in the reference: src/tts/engine.rs:267,407,425


@dataclass
class SamplerConfig:
    """src/tts/engine.rs:14-45 — defaults 0.7 / 40 / 0.9 / None."""
    temperature: float = 0.7
    top_k: int = 40
    top_p: float = 0.9
    seed: Optional[int] = None


@dataclass
class VoiceFile:
    """src/utils/voice_file.rs:5-62 — serde JSON; `spk_emb` is an alias of `speaker_embedding`; unknown keys ignored."""
    ref_text: str = ""
    audio_codes: List[int] = field(default_factory=list)
    speaker_embedding: List[float] = field(default_factory=list)
    name: Optional[str] = None
    gender: Optional[str] = None
    age: Optional[str] = None
    description: Optional[str] = None

    @staticmethod
    def new(ref_text, audio_codes, speaker_embedding):
        return VoiceFile(ref_text, list(audio_codes), list(speaker_embedding))

    def with_metadata(self, name=None, gender=None, age=None, description=None):
        self.name, self.gender, self.age, self.description = name, gender, age, description
        return self

    @staticmethod
    def load(path):
        with open(path, "r", encoding="utf-8") as f:
            d = json.load(f)
        emb = d.get("speaker_embedding", d.get("spk_emb"))
        if emb is None:
            raise ValueError("missing field `speaker_embedding`")  # serde: the only non-default, non-Option field
        return VoiceFile(d.get("ref_text", ""), list(d.get("audio_codes", [])), list(emb), d.get("name"), d.get("gender"),
                         d.get("age"), d.get("description"))

    def save(self, path):
        with open(path, "w", encoding="utf-8") as f:
            json.dump({"ref_text": self.ref_text, "audio_codes": self.audio_codes, "speaker_embedding": self.speaker_embedding,
                       "name": self.name, "gender": self.gender, "age": self.age, "description": self.description}, f, indent=2)


@dataclass
class AudioSample:
    """src/utils/audio.rs:4-46 — mono 24 kHz f32 container; WAV i/o is 16-bit."""
    samples: np.ndarray
    sample_rate: int = 24000
    channels: int = 1

    @staticmethod
    def load_wav(path):
        with wave.open(str(path), "rb") as w:
            if w.getsampwidth() != 2:
                raise ValueError("load_wav reads 16-bit PCM only (src/utils/audio.rs:14-17)")
            raw = w.readframes(w.getnframes())
            data = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
            return AudioSample(data, w.getframerate(), w.getnchannels())

    def save_wav(self, path):
        amp = np.clip(np.asarray(self.samples, dtype=np.float32) * np.float32(32767.0), -32768.0, 32767.0)
        pcm = np.trunc(amp).astype("<i2")  # Rust `as i16` truncates toward zero (src/utils/audio.rs:35-37)
        with wave.open(str(path), "wb") as w:
            w.setnchannels(self.channels)
            w.setsampwidth(2)
            w.setframerate(self.sample_rate)
            w.writeframes(pcm.tobytes())

    def duration(self):
        return len(self.samples) / float(self.sample_rate)


# ref-audio cache `TTSC` v1 (src/utils/cache.rs:5-67): magic, u32 version, usize n + i64 codes, usize n + f32 emb
def save_cache(path, codes: Sequence[int], emb: Sequence[float]):
    with open(path, "wb") as f:
        f.write(b"TTSC" + struct.pack("<I", 1))
        f.write(struct.pack("<Q", len(codes)) + np.asarray(codes, dtype="<i8").tobytes())
        f.write(struct.pack("<Q", len(emb)) + np.asarray(emb, dtype="<f4").tobytes())


def load_cache(path):
    with open(path, "rb") as f:
        if f.read(4) != b"TTSC":
            raise ValueError("Invalid magic bytes")
        if struct.unpack("<I", f.read(4))[0] != 1:
            raise ValueError("Unsupported version")
        n = struct.unpack("<Q", f.read(8))[0]
        codes = np.frombuffer(f.read(8 * n), dtype="<i8").tolist()
        m = struct.unpack("<Q", f.read(8))[0]
        emb = np.frombuffer(f.read(4 * m), dtype="<f4").tolist()
    return codes, emb


def load_tokenizer(model_dir):
    """Tokenizer::load(model_dir) (src/utils/tokenizer.rs:9-15): model_dir/tokenizer/tokenizer.json through the C++ byte-level
    BPE reader of the library (q3tts_tokenizer_*); None when the file does not exist (token ids are then the input)."""
    tj = os.path.join(model_dir, "tokenizer", "tokenizer.json")
    return native.NativeTokenizer(tj) if os.path.exists(tj) else None


class TtsEngine:
    """src/tts/engine.rs:53-72 — the engine owns weights, contexts and speakers; one utterance at a time per call
    (`&mut self`), or a list through `generate_batch_with_voice` (continuous batching, an extension)."""

    def __init__(self, cfg, tokenizer=None):
        self._native = native.NativeEngine(cfg)
        self.cfg = cfg
        self.tokenizer = tokenizer
        self.speakers = {}
        self.max_steps = min(512, cfg.max_steps_cap)  # src/tts/engine.rs:152
        self.sampler_config = SamplerConfig()

    @classmethod
    def new(cls, model_dir: Optional[str] = None, quant: str = "none", config=None):
        """TtsEngine::new(model_dir, quant) (src/tts/engine.rs:84-169). There is no network here: with no weight
        container under model_dir the engine uses seeded synthetic weights of the configured shape."""
        cfg = config or _abi.default_config()
        if quant == "q8_0" and not cfg.talker_q8_0:
            cfg.talker_q8_0 = 2   # the gguf_q8_0 directory is multiplied as llama.cpp multiplies it: Q8_0 x Q8_0 (W8A8)
        if model_dir:
            quant_dir = {"q5_k_m": "gguf_q5_k_m", "q8_0": "gguf_q8_0"}.get(quant, "gguf")  # src/tts/engine.rs:91-95
            wdir = os.path.join(model_dir, quant_dir)
            if os.path.exists(os.path.join(wdir, "qwen3_tts_talker.gguf")):  # :121-122; absent -> synthetic weights (no downloader here)
                cfg.weights_path = wdir.encode()
        tok = load_tokenizer(model_dir) if model_dir else None
        eng = cls(cfg, tok)
        for d in ([os.path.join(model_dir, "preset_speakers")] if model_dir else []) + ["speakers"]:  # :156-166
            if os.path.isdir(d):
                eng.load_speakers(d)
                break
        return eng

    def close(self):
        self._native.close()

    def set_max_steps(self, steps: int):
        self.max_steps = steps

    def set_language(self, lang_id):
        """the reference hard-codes lang_id = 2055 (Chinese) at src/tts/engine.rs:267,407,425; None selects the
        no-language control block (NOTHINK variant, src/tts/prompt.rs:180-204)."""
        self.lang_id = lang_id

    def set_sampler_config(self, config: SamplerConfig):
        self.sampler_config = config

    def get_sampler_config(self) -> SamplerConfig:
        return self.sampler_config

    def load_speakers(self, speakers_dir):  # src/tts/engine.rs:187-208 (files that fail to parse are skipped)
        for fn in sorted(os.listdir(speakers_dir)):
            if fn.endswith(".json"):
                try:
                    self.speakers[os.path.splitext(fn)[0]] = VoiceFile.load(os.path.join(speakers_dir, fn))
                except Exception:
                    pass

    def get_speaker(self, id_or_name: str) -> VoiceFile:  # src/tts/engine.rs:211-231
        if id_or_name in self.speakers:
            return self.speakers[id_or_name]
        for v in self.speakers.values():
            if v.name == id_or_name:
                return v
        if "vivian" in self.speakers:
            return self.speakers["vivian"]
        if not self.speakers:
            raise RuntimeError("No speakers loaded in engine!")
        return next(iter(self.speakers.values()))

    def _encode(self, text: Union[str, Sequence[int]]):
        if isinstance(text, str):
            if self.tokenizer is None:
                raise _abi.Q3Error("no tokenizer.json available: pass token ids instead of text")
            return np.asarray(self.tokenizer.encode(text), dtype=np.uint32)  # src/utils/tokenizer.rs:17-25 (add_special_tokens = false)
        return np.asarray(text, dtype=np.uint32)

    def _lang(self):
        return getattr(self, "lang_id", LANG_ID_CHINESE)

    def _desc(self, text, voice: VoiceFile, instruct):
        ids = self._encode(text)
        ins = None if instruct is None else self._encode(instruct)
        emb = np.asarray(voice.speaker_embedding, dtype=np.float32)
        if emb.size != self.cfg.model.d_embed:
            raise _abi.Q3Error(f"speaker_embedding has {emb.size} values, expected {self.cfg.model.d_embed}")
        if len(voice.audio_codes) == 0:  # src/tts/engine.rs:398-412: x-vector-only prompt
            return native.make_prompt_desc(ids, spk_emb=emb, lang_id=self._lang(), instruct_ids=ins)
        ref_ids = self._encode(voice.ref_text)  # :414-427: ICL clone prompt
        return native.make_prompt_desc(ids, spk_emb=emb, lang_id=self._lang(), instruct_ids=ins,
                                       ref_codes=np.asarray(voice.audio_codes, dtype=np.int32), ref_text_ids=ref_ids)

    def generate_with_voice(self, text, voice: VoiceFile, instruct=None) -> AudioSample:
        """src/tts/engine.rs:390-435."""
        return self.generate_batch_with_voice([text], [voice], [instruct])[0]

    def generate_batch_with_voice(self, texts, voices, instructs=None, seeds=None):
        sc = self.sampler_config
        reqs, keep = [], []
        for i, (t, v) in enumerate(zip(texts, voices)):
            desc, k = self._desc(t, v, None if instructs is None else instructs[i])
            keep.append(k)
            seed = sc.seed if seeds is None else seeds[i]
            reqs.append(dict(desc=desc, temperature=sc.temperature, top_k=sc.top_k, top_p=sc.top_p, seed=seed, max_steps=self.max_steps,
                             want_pcm=1))
        outs = self._native.generate_batch(reqs)
        for o in outs:
            if o.status != 0:
                raise _abi.Q3Error(f"generation failed with status {o.status}")
        sr = self.cfg.vocoder.sample_rate
        return [AudioSample(o.pcm, sr, 1) for o in outs]

    def load_clone_encoders(self, clone_config=None):
        """The reference loads onnx/qwen3_tts_codec_encoder.onnx and onnx/qwen3_tts_speaker_encoder.onnx when they exist
        (src/tts/engine.rs:105-119). Those graphs are not available; this loads the family-structure encoders with seeded
        synthetic weights (q3tts_clone_init)."""
        if clone_config is None:
            clone_config = _abi.CloneConfig()
            self._native.lib.q3tts_clone_default_config(clone_config)
            clone_config.se_dim = self.cfg.model.d_embed
            clone_config.ae_n_codebooks = self.cfg.model.n_codebooks
            clone_config.ae_codebook_size = self.cfg.model.codebook_size
        self._native.clone_init(clone_config)

    @staticmethod
    def _read_wav_any(path):
        """create_voice_file's own WAV decoding (src/tts/engine.rs:339-373): f32 / i16 / i32, first channel, 24 kHz only."""
        with open(path, "rb") as f:
            raw = f.read()
        if raw[:4] != b"RIFF" or raw[8:12] != b"WAVE":
            raise _abi.Q3Error("WAV error: not a RIFF/WAVE file")
        pos, fmt, data = 12, None, None
        while pos + 8 <= len(raw):
            cid, size = raw[pos:pos + 4], struct.unpack("<I", raw[pos + 4:pos + 8])[0]
            body = raw[pos + 8:pos + 8 + size]
            if cid == b"fmt ":
                fmt = struct.unpack("<HHIIHH", body[:16])
                if fmt[0] == 0xFFFE and len(body) >= 26:  # WAVE_FORMAT_EXTENSIBLE: sub-format tag
                    fmt = (struct.unpack("<H", body[24:26])[0],) + fmt[1:]
            elif cid == b"data":
                data = body
            pos += 8 + size + (size & 1)
        if fmt is None or data is None:
            raise _abi.Q3Error("WAV error: missing fmt or data chunk")
        tag, channels, rate, _, _, bits = fmt
        if rate != 24000:
            raise _abi.Q3Error(f"Expected 24000Hz audio, found {rate}Hz")
        if tag == 3 and bits == 32:
            a = np.frombuffer(data[:len(data) // 4 * 4], dtype="<f4").astype(np.float32)
        elif tag == 1 and bits == 16:
            a = np.frombuffer(data[:len(data) // 2 * 2], dtype="<i2").astype(np.float32) / np.float32(32768.0)
        elif tag == 1 and bits == 32:
            a = (np.frombuffer(data[:len(data) // 4 * 4], dtype="<i4").astype(np.float32) / np.float32(2147483648.0)).astype(np.float32)
        else:
            raise _abi.Q3Error(f"Unsupported WAV format: {'Float' if tag == 3 else 'Int'} {bits} bits")
        return a[::channels].copy() if channels > 1 else a

    def create_voice_file(self, audio_path, ref_text):  # src/tts/engine.rs:324-387
        if not hasattr(self._native, "clone_cfg"):
            raise _abi.Q3Error("AudioEncoder not loaded. Please ensure models/onnx/qwen3_tts_codec_encoder.onnx exists.")
        audio = self._read_wav_any(audio_path)
        codes = self._native.audio_encode(audio)            # "Extracting audio codes..."
        emb = self._native.speaker_encode(audio)            # "Extracting speaker embedding..."
        return VoiceFile.new(ref_text, codes.reshape(-1).tolist(), emb.tolist())

    def _process_reference(self, audio_path):  # src/tts/engine.rs:275-302 (TTSC cache beside the audio file)
        cache_path = os.path.splitext(str(audio_path))[0] + ".cache"
        if os.path.exists(cache_path):
            try:
                return load_cache(cache_path)
            except (ValueError, struct.error, OSError):
                pass
        audio = AudioSample.load_wav(audio_path)
        if not hasattr(self._native, "clone_cfg"):
            raise _abi.Q3Error("AudioEncoder not loaded (required for processing raw audio)")
        codes = self._native.audio_encode(audio.samples).reshape(-1).tolist()
        emb = self._native.speaker_encode(audio.samples).tolist()
        try:
            save_cache(cache_path, codes, emb)
        except OSError:
            pass
        return codes, emb

    def generate(self, text, ref_audio_path, ref_text, instruct=None):  # src/tts/engine.rs:243-272
        codes, emb = self._process_reference(ref_audio_path)
        return self.generate_with_voice(text, VoiceFile.new(ref_text, codes, emb), instruct)


def cleanup():
    """qwen3_tts::cleanup() (src/lib.rs:18-20): llama_backend_free in the reference; nothing global to free here."""
    return None
