"""The vocoder oracle (oracle/q3_oracle_vocoder.c) against the model family's own code.

The reference's vocoder is an ONNX graph outside its repository (src/models/onnx.rs:342-459 pins only the I/O contract and the
state shapes), so nothing of the reference pins the arithmetic (PARITY UNPINNED). What can be pinned is that the oracle restates
the family STRUCTURE correctly: transformers' Qwen3OmniMoeCode2Wav (modeling_qwen3_omni_moe.py:3180-3696) holds it as plain
PyTorch. The test loads the oracle's seeded synthetic weights into that module and compares, stage by stage, with the oracle in
plain f32 (q3o_vocoder_set_arith 1):
  * the sliding-window transformer with LayerScale (pre_transformer) on the oracle's latent rows;
  * the two (ConvTranspose k = r, s = r + ConvNeXt) up-sampling stages;
  * the decoder (Conv k7 -> 4 x {SnakeBeta, ConvTranspose k = 2r s = r, 3 residual units dil 1/3/9} -> SnakeBeta -> Conv k7).
Two stated differences, both outside the compared tensors: the family module embeds codes by a mean over one table while the TTS
decoder graph sums 16 codebooks and applies a causal pre-conv (src/models/onnx.rs:477-479: 512-channel pre-conv state) — so the
comparison starts at the transformer's input; and the family's decoder-block ConvTranspose (k = 2r) trims r samples on BOTH sides (its output
looks one latent step ahead: the reference's look-ahead buffer, V4) where the streaming restatement trims r on the right only
(strictly causal) — the same samples r positions later, so the PCM is compared at the accumulated shift, away from the start.
Also reported: how far the bf16-operand arithmetic (what the device computes) is from f32 on this model. CPU only.
"""
import ctypes as C

import numpy as np
import pytest

torch = pytest.importorskip("torch")


def test_vocoder_oracle_equals_family_code2wav(oracle):
    O = oracle
    from q3tts import _abi
    from transformers.models.qwen3_omni_moe.configuration_qwen3_omni_moe import Qwen3OmniMoeCode2WavConfig
    from transformers.models.qwen3_omni_moe.modeling_qwen3_omni_moe import Qwen3OmniMoeCode2Wav
    vc = _abi.tiny_config().vocoder
    d, H, F, W, dd = vc.latent_dim, vc.n_head * vc.head_dim, vc.d_ffn, vc.sliding_window, vc.decoder_dim
    rates = [vc.dec_rates[i] for i in range(vc.n_dec_blocks)]
    ups = [vc.upsample_ratios[i] for i in range(vc.n_upsample)]
    cfg = Qwen3OmniMoeCode2WavConfig(codebook_size=vc.codebook_size, hidden_size=d, max_position_embeddings=8000, num_attention_heads=vc.n_head,
                                     num_key_value_heads=vc.n_head, attention_bias=False, sliding_window=W, intermediate_size=F, hidden_act="silu",
                                     layer_scale_initial_scale=vc.layer_scale_init, rms_norm_eps=vc.rms_eps, num_hidden_layers=vc.n_layer,
                                     num_quantizers=vc.n_codebooks, upsample_rates=rates, upsampling_ratios=ups, decoder_dim=dd,
                                     rope_parameters={"rope_type": "default", "rope_theta": float(vc.rope_theta)}, attn_implementation="eager")
    net = Qwen3OmniMoeCode2Wav(cfg).eval().float()
    L = O.lib()
    L.q3o_vocoder_mat.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.c_int32, C.c_float, C.c_void_p]
    L.q3o_vocoder_mat.restype = None
    L.q3o_vocoder_vec.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int64, C.c_float, C.c_float, C.c_void_p]
    L.q3o_vocoder_vec.restype = None
    L.q3o_vocoder_stage.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
    L.q3o_vocoder_stage.restype = C.c_int32
    v = L.q3o_vocoder_create(C.byref(vc), 0, 4)

    def mat(comp, which, rows, cols, fan_in, gain=1.0):
        out = np.zeros((rows, cols), dtype=np.float32)
        L.q3o_vocoder_mat(v, comp, which, rows, cols, fan_in, gain, out.ctypes.data)
        return out

    def vec(comp, which, n, base, std):
        out = np.zeros(n, dtype=np.float32)
        L.q3o_vocoder_vec(v, comp, which, n, base, std, out.ctypes.data)
        return out
    T_ = lambda a: torch.from_numpy(np.ascontiguousarray(a))

    def conv_w(comp, ww, ntap, cin, nout, gain=1.0):  # oracle [tap][n][ci] -> Conv1d [n][ci][tap]
        return mat(comp, ww, ntap * nout, cin, ntap * cin, gain).reshape(ntap, nout, cin).transpose(1, 2, 0)
    try:
        with torch.no_grad():
            for l, blk in enumerate(net.pre_transformer.layers):
                comp = 40 + l
                blk.input_layernorm.weight.copy_(T_(vec(comp, 2, d, 1.0, 0.05))); blk.post_attention_layernorm.weight.copy_(T_(vec(comp, 8, d, 1.0, 0.05)))
                blk.self_attn_layer_scale.scale.copy_(T_(vec(comp, 7, d, vc.layer_scale_init, 0.1 * vc.layer_scale_init)))
                blk.mlp_layer_scale.scale.copy_(T_(vec(comp, 12, d, vc.layer_scale_init, 0.1 * vc.layer_scale_init)))
                blk.self_attn.q_proj.weight.copy_(T_(mat(comp, 3, H, d, d))); blk.self_attn.k_proj.weight.copy_(T_(mat(comp, 4, H, d, d)))
                blk.self_attn.v_proj.weight.copy_(T_(mat(comp, 5, H, d, d))); blk.self_attn.o_proj.weight.copy_(T_(mat(comp, 6, d, H, H)))
                blk.mlp.gate_proj.weight.copy_(T_(mat(comp, 9, F, d, d))); blk.mlp.up_proj.weight.copy_(T_(mat(comp, 10, F, d, d)))
                blk.mlp.down_proj.weight.copy_(T_(mat(comp, 11, d, F, F)))
            net.pre_transformer.norm.weight.copy_(T_(vec(60, 0, d, 1.0, 0.05)))
            for u, (ct, cnx) in enumerate(net.upsample):
                comp, r = 64 + u, ups[u]
                Wc = mat(comp, 0, r * d, d, d).reshape(r, d, d)                      # [j][o][i]
                ct.conv.weight.copy_(T_(Wc.transpose(2, 1, 0)))                      # ConvTranspose1d [in][out][k]
                ct.conv.bias.copy_(T_(vec(comp, 1, d, 0.0, 0.02)))
                cnx.dwconv.conv.weight.copy_(T_(vec(comp, 13, 7 * d, 0.0, 0.3).reshape(7, d).T[:, None, :]))
                cnx.dwconv.conv.bias.copy_(T_(vec(comp, 14, d, 0.0, 0.02)))
                cnx.norm.weight.copy_(T_(vec(comp, 15, d, 1.0, 0.05))); cnx.norm.bias.copy_(T_(vec(comp, 16, d, 0.0, 0.02)))
                cnx.pwconv1.weight.copy_(T_(mat(comp, 17, 4 * d, d, d))); cnx.pwconv1.bias.copy_(T_(vec(comp, 18, 4 * d, 0.0, 0.02)))
                cnx.pwconv2.weight.copy_(T_(mat(comp, 19, d, 4 * d, 4 * d))); cnx.pwconv2.bias.copy_(T_(vec(comp, 20, d, 0.0, 0.02)))
                cnx.gamma.copy_(T_(vec(comp, 21, d, 0.1, 0.01)))
            dec = net.decoder
            dec[0].conv.weight.copy_(T_(conv_w(72, 0, 7, d, dd))); dec[0].conv.bias.copy_(T_(vec(72, 1, dd, 0.0, 0.02)))
            ch = dd
            for b in range(vc.n_dec_blocks):
                blk, comp, r, co = dec[1 + b].block, 80 + 4 * b, rates[b], ch // 2
                blk[0].alpha.copy_(T_(vec(comp, 22, ch, 0.0, 0.1))); blk[0].beta.copy_(T_(vec(comp, 23, ch, 0.0, 0.1)))
                Wt = mat(comp, 0, 2 * r * co, ch, 2 * ch).reshape(2, r, co, ch)     # [tap][j][o][i]; tap 1 = the current latent step
                wt = np.zeros((ch, co, 2 * r), dtype=np.float32)
                wt[:, :, :r] = Wt[1].transpose(2, 1, 0); wt[:, :, r:] = Wt[0].transpose(2, 1, 0)
                blk[1].conv.weight.copy_(T_(wt)); blk[1].conv.bias.copy_(T_(vec(comp, 1, co, 0.0, 0.02)))
                for u_, dil in enumerate((1, 3, 9)):
                    ru, rc = blk[2 + u_], comp + 1 + u_
                    ru.act1.alpha.copy_(T_(vec(rc, 22, co, 0.0, 0.1))); ru.act1.beta.copy_(T_(vec(rc, 23, co, 0.0, 0.1)))
                    ru.conv1.conv.weight.copy_(T_(conv_w(rc, 0, 7, co, co, 0.5))); ru.conv1.conv.bias.copy_(T_(vec(rc, 1, co, 0.0, 0.02)))
                    ru.act2.alpha.copy_(T_(vec(rc, 26, co, 0.0, 0.1))); ru.act2.beta.copy_(T_(vec(rc, 27, co, 0.0, 0.1)))
                    ru.conv2.conv.weight.copy_(T_(conv_w(rc, 24, 1, co, co, 0.5))); ru.conv2.conv.bias.copy_(T_(vec(rc, 25, co, 0.0, 0.02)))
                ch = co
            dec[-2].alpha.copy_(T_(vec(120, 22, ch, 0.0, 0.1))); dec[-2].beta.copy_(T_(vec(120, 23, ch, 0.0, 0.1)))
            dec[-1].conv.weight.copy_(T_(conv_w(120, 0, 7, ch, 1, 0.1))); dec[-1].conv.bias.copy_(T_(vec(120, 1, 1, 0.0, 0.02)))

        n_frames = 10
        codes = np.random.default_rng(7).integers(0, vc.codebook_size, size=(n_frames, 16)).astype(np.int32)
        up_total = int(np.prod(ups)); spf = up_total * int(np.prod(rates))

        def stage(k, shape):
            out = np.zeros(shape, dtype=np.float32)
            L.q3o_vocoder_stage(v, codes.ctypes.data, n_frames, k, out.ctypes.data)
            return out
        L.q3o_vocoder_set_arith(v, 1)
        try:
            s1, s2 = stage(1, (n_frames, d)), stage(2, (n_frames, d))
            s3, s4 = stage(3, (n_frames * up_total, d)), stage(4, (n_frames * spf,))
        finally:
            L.q3o_vocoder_set_arith(v, 0)
        s4_bf16 = stage(4, (n_frames * spf,))
        rel = lambda a, b: float(np.abs(a - b).max() / max(1e-9, np.abs(b).max()))
        with torch.no_grad():
            h2 = net.pre_transformer(inputs_embeds=T_(s1)[None]).last_hidden_state[0].numpy()
            assert rel(s2, h2) <= 2e-5, rel(s2, h2)
            hid = T_(s2).T[None]
            for blocks in net.upsample:
                for blk in blocks:
                    hid = blk(hid)
            assert rel(s3, hid[0].T.numpy()) <= 2e-5, rel(s3, hid[0].T.numpy())
            wav = T_(s3).T[None]
            for blk in net.decoder:
                wav = blk(wav)
            wav = wav[0, 0].numpy()
        # the family's decoder blocks each drop the first r samples at their own rate (see the module docstring): total shift
        shift, mult = 0, 1
        for r in reversed(rates):
            shift += r * mult; mult *= r
        assert wav.shape[0] == n_frames * spf - shift
        start = 7000   # beyond the receptive field of the boundary difference (3 x (6 + 18 + 54) samples per block, scaled by the later rates)
        a, b = s4[start + shift: shift + wav.shape[0]], wav[start:]
        assert a.size > 4000 and rel(a, b) <= 1e-4, rel(a, b)
        e_bf16 = float(np.sqrt(np.mean((s4_bf16 - s4) ** 2)))
        print(f"vocoder oracle vs transformers Code2Wav (f32): transformer {rel(s2, h2):.1e}, up-sampling {rel(s3, hid[0].T.numpy()):.1e}, decoder PCM {rel(a, b):.1e}; "
              f"bf16-operand arithmetic vs f32 on this model: PCM RMS {e_bf16:.2e} (signal RMS {float(np.sqrt(np.mean(s4 ** 2))):.2f})")
    finally:
        L.q3o_vocoder_destroy(v)


def test_full_shape_pcm_error_budget_of_the_bf16_operand_rounding(oracle):
    """Where the full-shape PCM error comes from. The device rounds GEMM / convolution inputs to bf16 (the MFMA's
    operand type) and sits 2.4-2.6e-3 RMS from the f32 arithmetic the reference's ORT CPU vocoder uses (src/models/onnx.rs:390-405); the
    oracle's bf16-input mode reproduces that distance (tests/test_parity_gpu.py prints both), so the budget is taken here, on the CPU: one
    stage group at a time rounds its inputs to bf16 while the others stay f32. Result: the groups contribute 0.4-1.2e-3 EACH and add up
    in quadrature to the total; the HBM-bound last stages (blocks 2, 3 and the output convolution) hold < 1e-3 of it, so wider operands
    there cannot bring the total under 2e-3 — the full-shape tolerance is 3.5e-3 (PCM_RMS_TOL_FULL), not a per-stage defect."""
    import ctypes as C
    import os
    from q3tts import _abi
    cfg = _abi.full_config_py()
    L = oracle.lib()
    L.q3o_vocoder_set_arith_mask.argtypes = [C.c_void_p, C.c_uint32]
    v = L.q3o_vocoder_create(C.byref(cfg.vocoder), 0, min(8, os.cpu_count() or 4))
    codes = np.random.default_rng(4).integers(0, cfg.vocoder.codebook_size, size=(4, 16)).astype(np.int32)

    def run():
        L.q3o_vocoder_reset(v)
        pcm = np.zeros(4 * 1920 + 64, dtype=np.float32)
        n = L.q3o_vocoder_decode(v, oracle.ptr(codes, oracle.i32p), 4, 1, oracle.ptr(pcm, oracle.f32p), pcm.size)
        return pcm[:n].copy()

    try:
        L.q3o_vocoder_set_arith(v, 1); ref = run(); L.q3o_vocoder_set_arith(v, 0)
        L.q3o_vocoder_set_arith_mask(v, 0); total = float(np.sqrt(np.mean((run() - ref) ** 2)))
        names = ["pre-conv + transformer", "up-sampling stages", "decoder input conv", "block 0 (768 ch)", "block 1 (384 ch)", "block 2 (192 ch)", "block 3 (96 ch)", "output conv"]
        errs = []
        for g, nm in enumerate(names):
            L.q3o_vocoder_set_arith_mask(v, 0xFF & ~(1 << g))   # only group g rounds its inputs to bf16
            errs.append(float(np.sqrt(np.mean((run() - ref) ** 2))))
            print(f"  only {nm:24s} with bf16 inputs: PCM RMS error {errs[-1]:.2e}")
        rss = float(np.sqrt(np.sum(np.square(errs))))
        print(f"  all groups: {total:.2e}; root-sum-square of the single groups {rss:.2e}; signal RMS {float(np.sqrt(np.mean(ref ** 2))):.2f}")
    finally:
        L.q3o_vocoder_set_arith_mask(v, 0)
        L.q3o_vocoder_destroy(v)
    assert abs(rss - total) <= 0.15 * total          # independent roundings: the groups add in quadrature
    assert max(errs) <= 1.6e-3 and min(errs) >= 2e-4   # no stage dominates, none is free
    assert float(np.sqrt(np.sum(np.square(errs[5:])))) <= 1.2e-3   # the HBM-bound tail holds the smaller part
    assert 2.0e-3 <= total <= 3.0e-3
