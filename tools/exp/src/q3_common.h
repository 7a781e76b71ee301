// q3_common.h — spec primitives shared by host and device code of libq3tts (gfx950 only).
// Everything numerically significant is written with explicit fmaf / rintf and compiled with
// -ffp-contract=off so that the canonical summation orders of DESIGN.md §4 hold on the device.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#define Q3_HD __host__ __device__ __forceinline__

typedef float f32x4 __attribute__((ext_vector_type(4)));

Q3_HD uint32_t q3_f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
Q3_HD float q3_u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

// exp with a fixed operation sequence
Q3_HD float q3_expf(float x) {
    if (x < -87.0f) return 0.0f;
    float n = rintf(x * 1.44269504088896341f);
    float r = fmaf(n, -0.693145751953125f, x);
    r = fmaf(n, -1.42860682030941723212e-6f, r);
    float p = 1.0f / 720.0f;
    p = fmaf(p, r, 1.0f / 120.0f);
    p = fmaf(p, r, 1.0f / 24.0f);
    p = fmaf(p, r, 1.0f / 6.0f);
    p = fmaf(p, r, 0.5f);
    p = fmaf(p, r, 1.0f);
    p = fmaf(p, r, 1.0f);
    int e = (int)n + 127;
    return p * q3_u2f((uint32_t)e << 23);
}

Q3_HD uint16_t q3_bf16(float x) {  // RNE, finite inputs
    uint32_t u = q3_f2u(x);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
Q3_HD float q3_bf16f(uint16_t h) { return q3_u2f((uint32_t)h << 16); }
Q3_HD float q3_round_bf16(float x) { return q3_bf16f(q3_bf16(x)); }

Q3_HD uint64_t q3_mix64(uint64_t z) {
    z ^= z >> 30; z *= 0xbf58476d1ce4e5b9ULL;
    z ^= z >> 27; z *= 0x94d049bb133111ebULL;
    z ^= z >> 31; return z;
}
// counter-based synthetic value: Irwin-Hall(4) of 16-bit uniforms * scale
Q3_HD float q3_synth(uint64_t seed, uint32_t tensor, uint64_t idx, float scale) {
    uint64_t h = q3_mix64(q3_mix64(seed ^ ((uint64_t)tensor * 0x9E3779B97F4A7C15ULL)) + idx);
    int32_t s = (int32_t)(h & 0xffff) + (int32_t)((h >> 16) & 0xffff) + (int32_t)((h >> 32) & 0xffff) +
                (int32_t)(h >> 48) - 131070;
    return (float)s * scale;
}
#define Q3_IH4_STD 37837.227f
#define Q3_TID(g, l, w) (((uint32_t)(g) << 16) | ((uint32_t)(l) << 8) | (uint32_t)(w))
enum { Q3G_TALKER = 1, Q3G_PRED = 2, Q3G_ASSET = 3, Q3G_VOC = 4 };
enum { Q3W_ATTN_NORM = 0, Q3W_Q, Q3W_K, Q3W_V, Q3W_QNORM, Q3W_KNORM, Q3W_O, Q3W_FFN_NORM, Q3W_GATE, Q3W_UP, Q3W_DOWN };
enum { Q3WM_OUT_NORM = 0, Q3WM_HEAD = 1 };
enum { Q3WA_TEXT = 0, Q3WA_PROJ_W = 1, Q3WA_PROJ_B = 2 };
#define Q3_L_MODEL 255

// argmax key: larger logit wins, ties -> smaller index, NaN never wins (key 0)
Q3_HD uint64_t q3_argmax_key(float v, uint32_t n) {
    if (v != v) return 0;
    if (v == 0.0f) v = 0.0f;
    uint32_t u = q3_f2u(v);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ((uint64_t)u << 32) | (uint64_t)(0xFFFFFFFFu - n);
}
Q3_HD int32_t q3_argmax_idx(uint64_t key) { return key == 0 ? 0 : (int32_t)(0xFFFFFFFFu - (uint32_t)(key & 0xFFFFFFFFu)); }
Q3_HD float q3_key_value(uint64_t key) {  // inverse of the order-preserving map (for sorted candidates)
    uint32_t u = (uint32_t)(key >> 32);
    u = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    return q3_u2f(u);
}

Q3_HD float q3_swiglu(float g, float u) { return (g / (1.0f + q3_expf(-g))) * u; }

// per-slot decode state, resident on the device and mirrored to pinned host memory for polling
struct Q3Slot {
    int32_t active;       // 1 while the utterance is generating
    int32_t cur_pos;      // Talker position of the next token (src/tts/engine.rs:466,641)
    int32_t n_frames;     // frames kept so far
    int32_t max_steps;
    int32_t min_frames, force_eos_at;
    int32_t hit_eos;
    int32_t top_k;
    float temperature, top_p;
    int32_t rng_base;     // offset of this utterance's f32 draws in the rng buffer
    int32_t code0;        // code sampled in the current frame
    int32_t steps;        // loop iterations executed (== n_frames unless EOS)
    int32_t pad_[3];
};
