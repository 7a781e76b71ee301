"""The decoder oracle (oracle/q3_oracle.c) against the model family's own code.

The reference runs the Talker / Predictor inside llama.cpp (src/models/llama/mod.rs:442-451), which is not in its repository, so
nothing of the reference pins the block arithmetic (PARITY UNPINNED). What can be pinned is that the oracle restates the family
STRUCTURE correctly — pre-norm residual blocks, fused QKV row order, per-head QK-RMSNorm before RoPE, NeoX pairing (i, i + hd/2),
GQA head mapping h -> h / (Hq / Hkv), SwiGLU, final norm, lm_head: this container's transformers package holds that structure as
plain PyTorch (Qwen3ForCausalLM; the Talker's interleaved M-RoPE with t = h = w = pos, the only layout the reference produces,
src/tts/engine.rs:306-314, IS plain RoPE). The test loads the oracle's seeded synthetic weights into it and compares the prefill
of a prompt: the oracle in plain f32 (q3o_set_arith 1: same structure, sums in double) to 2e-5 of the largest logit, and the
canonical bf16-MFMA arithmetic (what the device computes bit for bit) within the stated bf16 distance. The Predictor runs the
same block code (tfm_layers) on plain positions, so one test covers both. CPU only.
"""
import numpy as np
import pytest

torch = pytest.importorskip("torch")


def _family_model(O, om, m, talker):
    from transformers.models.qwen3.configuration_qwen3 import Qwen3Config
    from transformers.models.qwen3.modeling_qwen3 import Qwen3ForCausalLM
    if talker:
        L, d, Hq, Hkv, hd, F, V, theta = m.t_n_layer, m.t_d_model, m.t_n_head, m.t_n_kv_head, m.t_head_dim, m.t_d_ffn, m.t_vocab, m.t_rope_theta
    else:
        L, d, Hq, Hkv, hd, F, theta = m.p_n_layer, m.p_d_model, m.p_n_head, m.p_n_kv_head, m.p_head_dim, m.p_d_ffn, m.p_rope_theta
        V = (m.n_codebooks - 1) * m.codebook_size
    cfg = Qwen3Config(vocab_size=V, hidden_size=d, intermediate_size=F, num_hidden_layers=L, num_attention_heads=Hq, num_key_value_heads=Hkv,
                      head_dim=hd, hidden_act="silu", max_position_embeddings=512, rms_norm_eps=m.rms_eps, rope_theta=theta, attention_bias=False,
                      tie_word_embeddings=False, use_sliding_window=False, attention_dropout=0.0, attn_implementation="eager")
    cfg.rope_parameters = {"rope_type": "default", "rope_theta": float(theta)}
    net = Qwen3ForCausalLM(cfg).eval().float()
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    nq, nkv = Hq * hd, Hkv * hd
    with torch.no_grad():
        for l, blk in enumerate(net.model.layers):
            qkv = om.matrix(talker, l, 0)
            blk.self_attn.q_proj.weight.copy_(T(qkv[:nq])); blk.self_attn.k_proj.weight.copy_(T(qkv[nq:nq + nkv])); blk.self_attn.v_proj.weight.copy_(T(qkv[nq + nkv:]))
            blk.self_attn.o_proj.weight.copy_(T(om.matrix(talker, l, 1)))
            blk.mlp.gate_proj.weight.copy_(T(om.matrix(talker, l, 2))); blk.mlp.up_proj.weight.copy_(T(om.matrix(talker, l, 3)))
            blk.mlp.down_proj.weight.copy_(T(om.matrix(talker, l, 4)))
            blk.input_layernorm.weight.copy_(T(om.norm_weight(talker, l, 0, d))); blk.post_attention_layernorm.weight.copy_(T(om.norm_weight(talker, l, 1, d)))
            blk.self_attn.q_norm.weight.copy_(T(om.norm_weight(talker, l, 2, hd))); blk.self_attn.k_norm.weight.copy_(T(om.norm_weight(talker, l, 3, hd)))
        net.model.norm.weight.copy_(T(om.norm_weight(talker, -1, 0, d)))
        net.lm_head.weight.copy_(T(om.matrix(talker, 0, 5)))
    return net


def test_talker_oracle_equals_family_qwen3(oracle):
    O = oracle
    from q3tts import _abi
    cfg = _abi.tiny_config()
    m = cfg.model
    om = O.OracleModel(m, seed=0, n_ctx=256, n_threads=4)
    try:
        net = _family_model(O, om, m, True)
        spk = ((np.arange(m.d_embed) % 13 - 6) * 0.03125).astype(np.float32)
        desc, keep = O.make_prompt_desc(np.arange(300, 331), spk_emb=spk)   # 42 prompt rows: more positions than one RoPE period of the fast pairs
        pe = om.build_prompt(desc)
        with torch.no_grad():
            out = net(inputs_embeds=torch.from_numpy(pe)[None], output_hidden_states=False)
            ref_logits = out.logits[0, -1].numpy()
            ref_hidden = net.model(inputs_embeds=torch.from_numpy(pe)[None]).last_hidden_state[0, -1].numpy()
        om.set_arith(1)
        hid, lg = om.talker_prefill(pe)
        om.set_arith(0)
        scale = float(np.abs(ref_logits).max())
        assert scale > 0.1
        e_f32 = float(np.abs(lg - ref_logits).max()) / scale
        assert e_f32 <= 2e-5, e_f32
        assert float(np.abs(hid - ref_hidden).max()) <= 2e-5 * float(np.abs(ref_hidden).max())
        # the canonical arithmetic (bf16 GEMM operands: weights, normalised rows, attention and SwiGLU outputs; bf16 K/V) against
        # the same family code: the stated distance of the device path from f32 on this model
        hid_c, lg_c = om.talker_prefill(pe)
        e_c = float(np.abs(lg_c - ref_logits).max()) / scale
        print(f"talker prefill logits vs transformers Qwen3: plain f32 restatement {e_f32:.2e}, canonical bf16-MFMA order {e_c:.2e} (of the largest logit)")
        assert e_c <= 3e-2, e_c
    finally:
        om.close()
