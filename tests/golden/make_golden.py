"""Generates tests/golden/oracle_tiny.json — golden SELF-vectors of the CPU restatement (oracle/), tiny config, seed 0.

The reference (cgisky1980/Qwen3-TTS-Rust) holds no tests, golden vectors or fixtures for this path and its arithmetic
cannot be built or run here (Rust crate over llama.cpp / onnxruntime binaries that are not in the repository), so
these vectors pin the restatement against regressions; they are NOT outputs of the reference. Run: python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import _oracle as O  # noqa: E402
from q3tts import _abi  # noqa: E402

cfg = _abi.tiny_config()
m = O.OracleModel(cfg.model, seed=0, n_ctx=256, n_threads=4)
L = O.lib()
text_ids = [int(x) for x in np.random.default_rng(1234).integers(0, 151643, size=20)]
spk = ((np.arange(cfg.model.d_embed) % 13 - 6) * 0.03125).astype(np.float32)
desc, keep = O.make_prompt_desc(text_ids, spk_emb=spk)
pe = m.build_prompt(desc)
hid, lg = m.talker_prefill(pe)
max_steps = 10
greedy, _ = m.generate(pe, temperature=0.0, max_steps=max_steps)
sampled, _ = m.generate(pe, temperature=0.7, top_k=40, top_p=0.9, seed=1234, max_steps=max_steps)
r = np.zeros(16, dtype=np.float32)
L.q3o_rng_f32(42, 16, O.ptr(r, O.f32p))
out = {
    "note": "golden self-vectors of oracle/ (tiny config, synth seed 0); not reference outputs",
    "text_ids": text_ids, "max_steps": max_steps,
    "rng_seed42": r.view(np.uint32).tolist(),
    "prompt_bits_sum": int(pe.view(np.uint32).astype(np.uint64).sum()),
    "prefill_logits_bits_head": lg.view(np.uint32)[:8].tolist(),
    "greedy_codes": greedy.tolist(), "sampled_codes_seed1234": sampled.tolist(),
}
with open(os.path.join(HERE, "oracle_tiny.json"), "w") as f:
    json.dump(out, f)
print("wrote", os.path.join(HERE, "oracle_tiny.json"))
