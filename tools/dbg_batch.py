"""Debug helper: continuous batching over 4 slots with 7 mixed-length requests vs the oracle (prints the first mismatch)."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "qwen3-tts-rust_amd")); sys.path.insert(0, os.path.join(REPO, "tests"))
import _oracle as O
from q3tts import _abi, native
cfg = _abi.tiny_config(max_batch=4, n_ctx=256, with_vocoder=0)
eng = native.NativeEngine(cfg)
om = O.OracleModel(cfg.model, seed=0, n_ctx=256, n_threads=4)
spk = ((np.arange(cfg.model.d_embed) % 13 - 6) * 0.03125).astype(np.float32)
reqs, refs = [], []
for i in range(7):
    desc, keep = O.make_prompt_desc(np.arange(50 * i, 50 * i + 5 + 3 * i), spk_emb=spk)
    pe = om.build_prompt(desc)
    kw = dict(temperature=0.7, top_k=40, top_p=0.9, seed=1000 + i, max_steps=16, min_frames=3 + i, force_eos_at=3 + i)
    refs.append(om.generate(pe, **kw)[0]); reqs.append(dict(embd=pe, **kw))
outs = eng.generate_batch(reqs)
for i, (o, r) in enumerate(zip(outs, refs)):
    same = o.codes.shape == r.shape and np.array_equal(o.codes, r)
    first = None
    if not same and o.codes.shape == r.shape:
        bad = np.argwhere(o.codes != r); first = tuple(bad[0])
    print(i, "status", o.status, "frames", o.codes.shape[0], r.shape[0], "OK" if same else f"MISMATCH first at {first}")
