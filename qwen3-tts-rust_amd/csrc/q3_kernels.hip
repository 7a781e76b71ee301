// q3_kernels.hip — hand-written gfx950 (CDNA4) kernels of the Qwen3-TTS codec-token decoder.
//
// Numerics contract: every reduction has ONE canonical order that the CPU oracle restates.
//  * GEMM: v_mfma_f32_16x16x4_f32 (f32 in / f32 accumulate) is bit-for-bit a k-ordered fmaf chain
//    (measured: tools/probe_mfma.hip, 0/256 mismatches at K=2048), so the 8-slice / (t, kq) order below is exact.
//  * norms / softmax: 64-lane butterflies (xor 32,16,8,4,2,1) over per-lane sequential partials.
// Built with -ffp-contract=off: every fused multiply-add is an explicit fmaf.
#include "q3_kernels.h"

#define WAVE 64

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// once-read weight stream: non-temporal 16-byte load (MI355X_MICROARCH.md, row nt-weights)
__device__ __forceinline__ uint4 ntload16(const uint4* p) {
    const u32x4 v = __builtin_nontemporal_load((const u32x4*)p);
    return make_uint4(v.x, v.y, v.z, v.w);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = v + __shfl_xor(v, m);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m));
    return v;
}

// (the exact GEMM lives in q3_gemm.hip)

// ---------------------------------------------------------------------------------------------------
// weight / table initialisation
// ---------------------------------------------------------------------------------------------------
__global__ void k_fill_tiled(Q3Fill f, int nb0, int nb_count) {
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int kblocks = f.K >> 5;
    const size_t total = (size_t)nb_count * kblocks * 64;
    if (gid >= total) return;
    const int lane = (int)(gid & 63);
    const size_t t = gid >> 6;
    const int kb = (int)(t % kblocks), nb = nb0 + (int)(t / kblocks);
    const int n = nb * 16 + (lane & 15), k0 = kb * 32 + (lane >> 4) * 4;  // element e <-> k = k0 + (e/4)*16 + (e%4)
    uint32_t tid; int lr; const uint16_t* src;
    if (f.mode == 0) { tid = f.tid_a; lr = n - f.row0; src = f.src_a; }
    else {
        const int tile = n >> 4, c = n & 15;
        if (c < 8) { tid = f.tid_a; lr = tile * 8 + c; src = f.src_a; }
        else { tid = f.tid_b; lr = tile * 8 + c - 8; src = f.src_b; }
    }
    uint16_t h[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const size_t idx = (size_t)lr * f.K + k0 + (e >> 2) * 16 + (e & 3);
        h[e] = src ? src[idx] : q3_bf16(q3_synth(f.seed, tid, idx, f.scale));
    }
    uint4 v;
    v.x = (uint32_t)h[0] | ((uint32_t)h[1] << 16); v.y = (uint32_t)h[2] | ((uint32_t)h[3] << 16);
    v.z = (uint32_t)h[4] | ((uint32_t)h[5] << 16); v.w = (uint32_t)h[6] | ((uint32_t)h[7] << 16);
    f.dst[((size_t)nb * kblocks + kb) * 64 + lane] = v;
}
void q3_launch_fill_tiled(const Q3Fill& f, hipStream_t s) {
    int nb0, nbc;
    if (f.mode == 0) { nb0 = f.row0 / 16; nbc = f.rows / 16; } else { nb0 = 0; nbc = f.N / 16; }
    const size_t total = (size_t)nbc * (f.K >> 5) * 64;
    hipLaunchKernelGGL(k_fill_tiled, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, f, nb0, nbc);
}
// The same matrices as ggml Q8_0 blocks in the tiled Q8 layout (q3_kernels.h): one thread per (column tile, k-block pair, lane) = 16 output
// bytes. A thread needs the scale of both of its blocks, i.e. the largest magnitude of all 32 weights of each: it re-reads them (8 x
// redundant: load-time only). Quantiser = ggml's quantize_row_q8_0_ref: d = amax / 127, id = d ? 1 / d : 0, q = roundf(x * id), d kept as f16.
__global__ void k_fill_tiled_q8(Q3Fill f, int nb0, int nb_count) {
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int kpairs = f.K >> 6, kblocks = f.K >> 5;
    const size_t total = (size_t)nb_count * kpairs * 64;
    if (gid >= total) return;
    const int lane = (int)(gid & 63);
    const size_t t = gid >> 6;
    const int kp = (int)(t % kpairs), nb = nb0 + (int)(t / kpairs);
    const int n = nb * 16 + (lane & 15), kq = lane >> 4;
    uint32_t tid; int lr; const uint16_t* src; const uint8_t* src8;
    if (f.mode == 0) { tid = f.tid_a; lr = n - f.row0; src = f.src_a; src8 = f.src8_a; }
    else {
        const int tile = n >> 4, c = n & 15;
        if (c < 8) { tid = f.tid_a; lr = tile * 8 + c; src = f.src_a; src8 = f.src8_a; }
        else { tid = f.tid_b; lr = tile * 8 + c - 8; src = f.src_b; src8 = f.src8_b; }
    }
    uint32_t out[4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int kb = 2 * kp + h;
        int8_t q[8]; uint16_t d16;
        if (src8) {  // a block_q8_0 as stored: f16 d, 32 x int8
            const uint8_t* blk = src8 + ((size_t)lr * kblocks + kb) * 34;
            d16 = (uint16_t)blk[0] | ((uint16_t)blk[1] << 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) q[e] = (int8_t)blk[2 + (e >> 2) * 16 + kq * 4 + (e & 3)];
        } else {
            float amax = 0.0f, mine[8];
            for (int k = 0; k < 32; ++k) {
                const size_t idx = (size_t)lr * f.K + (size_t)kb * 32 + k;
                const float v = src ? q3_bf16f(src[idx]) : q3_round_bf16(q3_synth(f.seed, tid, idx, f.scale));
                amax = fmaxf(amax, fabsf(v));
                const int kk = k & 15;
                if ((kk >> 2) == kq) mine[((k >> 4) << 2) + (kk & 3)] = v;
            }
            const float d = amax / 127.0f, id = d != 0.0f ? 1.0f / d : 0.0f;
            d16 = __builtin_bit_cast(unsigned short, (_Float16)d);
#pragma unroll
            for (int e = 0; e < 8; ++e) q[e] = (int8_t)(int)roundf(mine[e] * id);
        }
        out[2 * h] = (uint32_t)(uint8_t)q[0] | ((uint32_t)(uint8_t)q[1] << 8) | ((uint32_t)(uint8_t)q[2] << 16) | ((uint32_t)(uint8_t)q[3] << 24);
        out[2 * h + 1] = (uint32_t)(uint8_t)q[4] | ((uint32_t)(uint8_t)q[5] << 8) | ((uint32_t)(uint8_t)q[6] << 16) | ((uint32_t)(uint8_t)q[7] << 24);
        if (kq == 0) f.dst_scale[(size_t)n * kblocks + kb] = d16;
    }
    f.dst[((size_t)nb * kpairs + kp) * 64 + lane] = make_uint4(out[0], out[1], out[2], out[3]);
}
void q3_launch_fill_tiled_q8(const Q3Fill& f, hipStream_t s) {
    int nb0, nbc;
    if (f.mode == 0) { nb0 = f.row0 / 16; nbc = f.rows / 16; } else { nb0 = 0; nbc = f.N / 16; }
    const size_t total = (size_t)nbc * (f.K >> 6) * 64;
    hipLaunchKernelGGL(k_fill_tiled_q8, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, f, nb0, nbc);
}
__global__ void k_fill_f32(float* dst, size_t n, uint64_t seed, uint32_t tid, float base, float scale, int rb) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        float v = base + q3_synth(seed, tid, i, scale);
        dst[i] = rb ? q3_round_bf16(v) : v;
    }
}
void q3_launch_fill_f32(float* dst, size_t n, uint64_t seed, uint32_t tid, float base, float scale, int rb, hipStream_t s) {
    size_t blocks = (n + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(k_fill_f32, dim3((unsigned)blocks), dim3(256), 0, s, dst, n, seed, tid, base, scale, rb);
}

// ---------------------------------------------------------------------------------------------------
// q/k RMSNorm + RoPE + KV append. One wave per (row, head); hd == 128 (lanes 0..31 own 4 elements each).
// K cache layout: per (slot, kv head) blocks of 64 keys, [block][hd/8 chunks][64 keys][8] bf16,
// so that the score kernel reads 1 KiB contiguous per wave-load with one key per lane. V is row-major [t][hd].
// ---------------------------------------------------------------------------------------------------
// RMSNorm(hd) + RoPE of one head by one wave: lanes [0, hd/4) own 4 consecutive elements each; result in o[4]
__device__ __forceinline__ void prep_head(const float* src, const float* nw, float eps, const float* cs, const float* sn, int hd,
                                          int lane, float o[4]) {
    const int nl = hd >> 2, hl = nl >> 1;
    float4 v = (float4){0.f, 0.f, 0.f, 0.f};
    if (lane < nl) v = ((const float4*)src)[lane];
    float acc = 0.0f;
    acc = fmaf(v.x, v.x, acc); acc = fmaf(v.y, v.y, acc); acc = fmaf(v.z, v.z, acc); acc = fmaf(v.w, v.w, acc);
    acc = wave_sum(acc);
    const float rinv = 1.0f / sqrtf(acc / (float)hd + eps);
    float4 w4 = (float4){0.f, 0.f, 0.f, 0.f};
    if (lane < nl) w4 = ((const float4*)nw)[lane];
    const float y[4] = {(v.x * rinv) * w4.x, (v.y * rinv) * w4.y, (v.z * rinv) * w4.z, (v.w * rinv) * w4.w};
    // RoPE (NeoX pairing i <-> i + hd/2): lanes [0, nl/2) hold the first halves, partner lane = lane ^ (nl/2)
    const int i0 = 4 * (lane & (hl - 1));
    float4 c4 = (float4){1.f, 1.f, 1.f, 1.f}, s4 = (float4){0.f, 0.f, 0.f, 0.f};
    if (lane < nl) { c4 = *(const float4*)(cs + i0); s4 = *(const float4*)(sn + i0); }
    const float cc[4] = {c4.x, c4.y, c4.z, c4.w}, ss[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float other = __shfl_xor(y[e], hl);
        if (lane < hl) o[e] = fmaf(-other, ss[e], y[e] * cc[e]);   // x_i*c - x_{i+half}*s
        else o[e] = fmaf(other, ss[e], y[e] * cc[e]);              // x_{i+half}*c + x_i*s
    }
}
// bf16 K (key-interleaved blocks) and V (row-major) append of one kv head by lanes [0, hd/4)
__device__ __forceinline__ void kv_append(uint16_t* kc, uint16_t* vc, size_t hb, int hd, int pos, int lane, const float o[4], const float4 vv) {
    const int blk = pos >> 6, kl = pos & 63, c = lane >> 1, e0 = 4 * (lane & 1);
    uint2 kk;
    kk.x = (uint32_t)q3_bf16(o[0]) | ((uint32_t)q3_bf16(o[1]) << 16);
    kk.y = (uint32_t)q3_bf16(o[2]) | ((uint32_t)q3_bf16(o[3]) << 16);
    *(uint2*)(kc + hb * hd + ((size_t)(blk * (hd >> 3) + c) * 64 + kl) * 8 + e0) = kk;
    uint2 vk;
    vk.x = (uint32_t)q3_bf16(vv.x) | ((uint32_t)q3_bf16(vv.y) << 16);
    vk.y = (uint32_t)q3_bf16(vv.z) | ((uint32_t)q3_bf16(vv.w) << 16);
    *(uint2*)(vc + (hb + pos) * hd + 4 * lane) = vk;
}

__global__ __launch_bounds__(64) void k_qk_prep(Q3QkPrep a) {
    const int row = blockIdx.x, hx = blockIdx.y, lane = threadIdx.x;
    int pos, slot;
    q3_row_map(row, a.row_pos, a.row_slot, a.slot_mod, a.pos_const, &pos, &slot);
    if (pos < 0) return;
    const int hd = a.hd, half = hd >> 1, nl = hd >> 2;
    const bool isq = hx < a.Hq;
    const int g = hx - a.Hq;
    float* src = a.qkv + (size_t)row * a.ld + (size_t)(isq ? hx : a.Hq + g) * hd;
    float o[4];
    prep_head(src, isq ? a.qnw : a.knw, a.eps, a.cs + (size_t)pos * half, a.sn + (size_t)pos * half, hd, lane, o);
    if (lane >= nl) return;
    if (isq) {
        ((float4*)src)[lane] = (float4){o[0], o[1], o[2], o[3]};
    } else {
        const float4 vv = ((const float4*)(a.qkv + (size_t)row * a.ld + (size_t)(a.Hq + a.Hkv + g) * hd))[lane];
        kv_append(a.kc, a.vc, ((size_t)slot * a.Hkv + g) * a.n_ctx, hd, pos, lane, o, vv);
    }
}
void q3_launch_qk_prep(const Q3QkPrep& a, hipStream_t s) {
    hipLaunchKernelGGL(k_qk_prep, dim3(a.rows, a.Hq + a.Hkv), dim3(64), 0, s, a);
}

// ---------------------------------------------------------------------------------------------------
// Attention over the cache, canonical order of DESIGN.md §4.4. Workgroup (kv head g, row): 4 waves per query
// head of the GQA group. Scores: one key per lane (256 virtual lanes = 4 waves), dot over d ascending.
// PV: 16 key-partials (u = t mod 16: wave u/4, lane group u%4), 16 lanes x 8 dims per key.
// ---------------------------------------------------------------------------------------------------
// FUSED (one row per slot, e.g. every decode step): the workgroup first does the q/k RMSNorm + RoPE + KV append of
// its own row (k_qk_prep's work) and serves the newest key/value from LDS, saving one launch per layer.
template <int R, bool FUSED>
__global__ __launch_bounds__(R * 256) void k_attend(Q3Attend a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int g = blockIdx.x, row = blockIdx.y;
    Q3_STAMP(a, 0);
    int pos, slot;
    q3_row_map(row, a.row_pos, a.row_slot, a.slot_mod, a.pos_const, &pos, &slot);
    if (pos < 0) return;
    const int T = pos + 1, Tcap = a.n_ctx, hd = a.hd, nch = hd >> 3;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, hh = wave >> 2, sw = wave & 3;
    float* p_all = smem;                      // [R][Tcap]
    float* qh = p_all + (size_t)R * Tcap;     // [R][hd]
    float* ow = qh + R * hd;                  // [R][4][hd]
    float* lw = ow + R * 4 * hd;              // [R][4]
    float* mw = lw + R * 4;                   // [R][4]
    float* kh = mw + R * 4;                   // [hd] newest key (bf16-rounded), FUSED only
    float* vh = kh + hd;                      // [hd] newest value
    const size_t hb = ((size_t)slot * a.Hkv + g) * a.n_ctx;
    if constexpr (FUSED) {
        const Q3QkPrep& pr = a.prep;
        const int half = hd >> 1, nl = hd >> 2;
        const float* rowp = a.qkv + (size_t)row * a.ld;
        float o[4];
        if (sw == 0) {  // first wave of each query head: q-norm + RoPE -> LDS
            prep_head(rowp + (size_t)(g * R + hh) * hd, pr.qnw, pr.eps, pr.cs + (size_t)pos * half, pr.sn + (size_t)pos * half, hd, lane, o);
            if (lane < nl) *(float4*)(qh + hh * hd + 4 * lane) = (float4){o[0], o[1], o[2], o[3]};
        }
        if (wave == 1 || (R * 4 == 1)) {  // a second wave: k-norm + RoPE + append, v append
            prep_head(rowp + (size_t)(a.Hq + g) * hd, pr.knw, pr.eps, pr.cs + (size_t)pos * half, pr.sn + (size_t)pos * half, hd, lane, o);
            if (lane < nl) {
                const float4 vv = ((const float4*)(rowp + (size_t)(a.Hq + a.Hkv + g) * hd))[lane];
                kv_append(pr.kc, pr.vc, hb, hd, pos, lane, o, vv);
                // the newest key in the cache's own packed form (bf16 pairs, chunk c = 16 bytes at kh + 4 c words): the lane that owns it
                // then runs the same chain as every cached key (an LDS-fed float chain of its own cost ~1.5 us of this kernel)
                uint2 kk;
                kk.x = (uint32_t)q3_bf16(o[0]) | ((uint32_t)q3_bf16(o[1]) << 16); kk.y = (uint32_t)q3_bf16(o[2]) | ((uint32_t)q3_bf16(o[3]) << 16);
                *(uint2*)((uint32_t*)kh + 2 * lane) = kk;
                *(float4*)(vh + 4 * lane) = (float4){q3_round_bf16(vv.x), q3_round_bf16(vv.y), q3_round_bf16(vv.z), q3_round_bf16(vv.w)};
            }
        }
    } else {
        for (int i = tid; i < R * hd; i += R * 256) qh[i] = a.qkv[(size_t)row * a.ld + (size_t)g * R * hd + i];
    }
    __syncthreads();
    Q3_STAMP(a, 1);
    const uint16_t* kb = a.kc + hb * hd;
    const uint16_t* vb = a.vc + hb * hd;
    const float scale = 1.0f / sqrtf((float)hd);
    float* p = p_all + (size_t)hh * Tcap;
    const float* q = qh + hh * hd;
    float mloc = -INFINITY;
    for (int blk = sw; blk * 64 < T; blk += 4) {
        const int t = blk * 64 + lane;
        const uint4* kp = (const uint4*)(kb + (size_t)blk * 64 * hd) + lane;
        float s = 0.0f;
        if (nch == 16) {  // hd = 128: all 16 key chunks in flight at once (a runtime-bound loop issues load, use, load, use ...:
                          // measured 10 k of the kernel's 20 k cycles at T <= 17, tools/exp/stamp_attend.py)
            uint4 kv[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) kv[c] = kp[c * 64];
            if (FUSED && t == pos) {
#pragma unroll
                for (int c = 0; c < 16; ++c) kv[c] = *(const uint4*)((const uint32_t*)kh + 4 * c);
            }
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const float4 qa = *(const float4*)(q + c * 8), qb = *(const float4*)(q + c * 8 + 4);
                s = fmaf(qa.x, q3_u2f(kv[c].x << 16), s); s = fmaf(qa.y, q3_u2f(kv[c].x & 0xffff0000u), s);
                s = fmaf(qa.z, q3_u2f(kv[c].y << 16), s); s = fmaf(qa.w, q3_u2f(kv[c].y & 0xffff0000u), s);
                s = fmaf(qb.x, q3_u2f(kv[c].z << 16), s); s = fmaf(qb.y, q3_u2f(kv[c].z & 0xffff0000u), s);
                s = fmaf(qb.z, q3_u2f(kv[c].w << 16), s); s = fmaf(qb.w, q3_u2f(kv[c].w & 0xffff0000u), s);
            }
        } else
        for (int c = 0; c < nch; ++c) {
            const uint4 kv = (FUSED && t == pos) ? *(const uint4*)((const uint32_t*)kh + 4 * c) : kp[c * 64];
            const float* qc = q + c * 8;
            s = fmaf(qc[0], q3_u2f(kv.x << 16), s); s = fmaf(qc[1], q3_u2f(kv.x & 0xffff0000u), s);
            s = fmaf(qc[2], q3_u2f(kv.y << 16), s); s = fmaf(qc[3], q3_u2f(kv.y & 0xffff0000u), s);
            s = fmaf(qc[4], q3_u2f(kv.z << 16), s); s = fmaf(qc[5], q3_u2f(kv.z & 0xffff0000u), s);
            s = fmaf(qc[6], q3_u2f(kv.w << 16), s); s = fmaf(qc[7], q3_u2f(kv.w & 0xffff0000u), s);
        }
        s = s * scale;
        if (t < T) { p[t] = s; mloc = fmaxf(mloc, s); }
    }
    mloc = wave_max(mloc);
    if (lane == 0) mw[hh * 4 + sw] = mloc;
    __syncthreads();
    const float m = fmaxf(fmaxf(mw[hh * 4], mw[hh * 4 + 1]), fmaxf(mw[hh * 4 + 2], mw[hh * 4 + 3]));
    float lsum = 0.0f;
    for (int blk = sw; blk * 64 < T; blk += 4) {
        const int t = blk * 64 + lane;
        if (t < T) { const float e = q3_expf(p[t] - m); p[t] = e; lsum += e; }
    }
    lsum = wave_sum(lsum);
    if (lane == 0) lw[hh * 4 + sw] = lsum;
    __syncthreads();
    Q3_STAMP(a, 2);
    const int kg = lane >> 4, dl = lane & 15;
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = 0.0f;
    // four value rows per trip, loaded together (a one-row loop issues load, use, load, use ... : one memory round trip per
    // 16 cached tokens); the chain of a lane still sees its keys in ascending order
    for (int t0 = 4 * sw + kg; t0 < T; t0 += 64) {
        uint4 vv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) vv[u] = *(const uint4*)(vb + (size_t)min(t0 + 16 * u, T - 1) * hd + dl * 8);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int t = t0 + 16 * u;
            if (t < T) {
                const float pt = p[t];
                if (FUSED && t == pos) {  // newest value: from LDS
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = fmaf(pt, vh[dl * 8 + e], o[e]);
                } else {
                    o[0] = fmaf(pt, q3_u2f(vv[u].x << 16), o[0]); o[1] = fmaf(pt, q3_u2f(vv[u].x & 0xffff0000u), o[1]);
                    o[2] = fmaf(pt, q3_u2f(vv[u].y << 16), o[2]); o[3] = fmaf(pt, q3_u2f(vv[u].y & 0xffff0000u), o[3]);
                    o[4] = fmaf(pt, q3_u2f(vv[u].z << 16), o[4]); o[5] = fmaf(pt, q3_u2f(vv[u].z & 0xffff0000u), o[5]);
                    o[6] = fmaf(pt, q3_u2f(vv[u].w << 16), o[6]); o[7] = fmaf(pt, q3_u2f(vv[u].w & 0xffff0000u), o[7]);
                }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        o[e] = o[e] + __shfl_xor(o[e], 16);
        o[e] = o[e] + __shfl_xor(o[e], 32);
    }
    if (kg == 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) ow[(hh * 4 + sw) * hd + dl * 8 + e] = o[e];
    }
    __syncthreads();
    Q3_STAMP(a, 3);
    for (int i = tid; i < R * hd; i += R * 256) {
        const int h2 = i / hd, d = i - h2 * hd;
        const float r0 = ow[(h2 * 4 + 0) * hd + d], r1 = ow[(h2 * 4 + 1) * hd + d], r2 = ow[(h2 * 4 + 2) * hd + d],
                    r3 = ow[(h2 * 4 + 3) * hd + d];
        const float ov = ((r0 + r1) + r2) + r3;
        const float l = ((lw[h2 * 4] + lw[h2 * 4 + 1]) + lw[h2 * 4 + 2]) + lw[h2 * 4 + 3];
        const size_t oi = (size_t)row * a.ldo + (size_t)(g * R + h2) * hd + d;
        if (a.out_bf16 == 2) q3_q8_out32(ov / l, row, (g * R + h2) * hd + d, (a.Hq * hd) >> 6, a.out_rt16, (int8_t*)a.out, a.out_scale);  // W8A8: Q8_0 blocks (hd % 32 == 0: a half wave = one block)
        else if (a.out_bf16) ((uint16_t*)a.out)[q3_atile_off(row, (g * R + h2) * hd + d, (a.Hq * hd) >> 5)] = q3_bf16(ov / l);  // the O projection's A-tiled operand
        else a.out[oi] = ov / l;
    }
#ifdef Q3_STAMPS
    Q3_STAMP(a, 4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    Q3_STAMP(a, 5);
#endif
}
// ---------------------------------------------------------------------------------------------------------------------
// Decode attention of a GQA PAIR (two query heads per KV head, hd = 128, one row per slot: every Talker step). Same canonical order as
// k_attend<2, true>, bit for bit, with the work laid out around what the in-kernel timestamps showed (tools/chain_stamps.hip: scores
// 5.2 us and value pass 3.4 us of a 12 us kernel at 150 cached keys, against ~6.3 us for its K / V bytes at the HBM rate):
//  * FOUR waves per (slot, KV head) instead of eight: wave sw owns the key blocks sw, sw + 4, ... for BOTH heads — a key block (and a
//    value row) is loaded once and used twice; the two heads' d-ascending fmaf chains are independent and interleave in the pipeline
//  * every wave requests its first key block and its first four value rows before the q / k / v preparation (waves 0, 1: the two query
//    heads; wave 2: k; wave 3: v), so the preparation runs under the memory latency instead of in front of it
//  * the newest key / value reach their lane through LDS in the cache's packed form: one code path for cached and newest keys
// ---------------------------------------------------------------------------------------------------------------------
// (Q3_LDS_BARRIER: q3_kernels.h)
__global__ __launch_bounds__(256, 2) void k_attend_gqa2(Q3Attend a) {  // (<= 256 registers: two workgroups per CU; unconstrained the compiler took 405)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int g = blockIdx.x, row = blockIdx.y;
    Q3_STAMP(a, 0);
    int pos, slot;
    q3_row_map(row, a.row_pos, a.row_slot, a.slot_mod, a.pos_const, &pos, &slot);
    if (pos < 0) return;
    const int T = pos + 1, Tcap = a.n_ctx, hd = 128, half = 64, nl = 32;
    const int tid = threadIdx.x, sw = tid >> 6, lane = tid & 63, kg = lane >> 4, dl = lane & 15;
    float* p0 = smem;                          // [Tcap] scores / probabilities of head 0
    float* p1 = p0 + Tcap;                     // [Tcap] head 1
    float* qh = p1 + Tcap;                     // [2][hd]
    float* ow = qh + 2 * hd;                   // [2][4][hd]
    float* lw = ow + 2 * 4 * hd;               // [2][4]
    float* mw = lw + 8;                        // [2][4]
    uint32_t* knew = (uint32_t*)(mw + 8);      // [64] newest key, bf16 pairs, chunk c = 16 bytes at knew + 4 c
    uint32_t* vnew = knew + 64;                // [64] newest value, bf16 pairs
    const Q3QkPrep& pr = a.prep;
    const size_t hb = ((size_t)slot * a.Hkv + g) * a.n_ctx;
    const uint16_t* kb = a.kc + hb * hd;
    const uint16_t* vb = a.vc + hb * hd;
    // Load order = the order of use (vmcnt retires in order): the row's own operands (L2, short latency) first, then the wave's first
    // key block (one key per lane, 16 KiB) and the value rows of its first FOUR trips (8 dims per lane; every cached key of a context of
    // <= 256) — all from HBM, all in flight while the preparation runs. (Cache loads first made the preparation wait ~5 us for them.)
    const float* rowp = a.qkv + (size_t)row * a.ld;
    const float* src = rowp + (size_t)(sw < 2 ? g * 2 + sw : (sw == 2 ? a.Hq + g : a.Hq + a.Hkv + g)) * hd;
    float4 x4 = (float4){0.f, 0.f, 0.f, 0.f}, w4 = x4, c4 = (float4){1.f, 1.f, 1.f, 1.f}, s4 = x4;
    if (lane < nl) {
        x4 = ((const float4*)src)[lane];
        if (sw < 3) {
            w4 = ((const float4*)(sw < 2 ? pr.qnw : pr.knw))[lane];
            c4 = *(const float4*)(pr.cs + (size_t)pos * half + 4 * (lane & 15)); s4 = *(const float4*)(pr.sn + (size_t)pos * half + 4 * (lane & 15));
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    {   // preparation of this row: q heads (waves 0, 1), k (wave 2: norm + RoPE + append), v (wave 3: append)
        if (sw < 3) {  // prep_head's arithmetic
            float acc = 0.0f;
            acc = fmaf(x4.x, x4.x, acc); acc = fmaf(x4.y, x4.y, acc); acc = fmaf(x4.z, x4.z, acc); acc = fmaf(x4.w, x4.w, acc);
            acc = wave_sum(acc);
            const float rinv = 1.0f / sqrtf(acc / (float)hd + pr.eps);
            const float y[4] = {(x4.x * rinv) * w4.x, (x4.y * rinv) * w4.y, (x4.z * rinv) * w4.z, (x4.w * rinv) * w4.w};
            const float cc[4] = {c4.x, c4.y, c4.z, c4.w}, ss[4] = {s4.x, s4.y, s4.z, s4.w};
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float other = __shfl_xor(y[e], 16);
                o[e] = (lane < 16) ? fmaf(-other, ss[e], y[e] * cc[e]) : fmaf(other, ss[e], y[e] * cc[e]);
            }
            if (lane < nl) {
                if (sw < 2) *(float4*)(qh + sw * hd + 4 * lane) = (float4){o[0], o[1], o[2], o[3]};
                else {
                    uint2 kk;
                    kk.x = (uint32_t)q3_bf16(o[0]) | ((uint32_t)q3_bf16(o[1]) << 16); kk.y = (uint32_t)q3_bf16(o[2]) | ((uint32_t)q3_bf16(o[3]) << 16);
                    const int blk = pos >> 6, kl = pos & 63, c = lane >> 1, e0 = 4 * (lane & 1);
                    *(uint2*)(pr.kc + hb * hd + ((size_t)(blk * (hd >> 3) + c) * 64 + kl) * 8 + e0) = kk;
                    *(uint2*)(knew + 2 * lane) = kk;
                }
            }
        } else if (lane < nl) {
            uint2 vk;
            vk.x = (uint32_t)q3_bf16(x4.x) | ((uint32_t)q3_bf16(x4.y) << 16); vk.y = (uint32_t)q3_bf16(x4.z) | ((uint32_t)q3_bf16(x4.w) << 16);
            *(uint2*)(pr.vc + (hb + pos) * hd + 4 * lane) = vk;
            *(uint2*)(vnew + 2 * lane) = vk;
        }
    }
    // The cache operands are requested only now: the vector L1 returns loads in issue order across the waves of a CU, so HBM misses issued
    // ahead of the preparation's (L2-hit) operands held every wave's preparation back by the HBM latency (first barrier at 4.9 us instead of 1.9).
    // vv[0..1]: value rows of the first two trips; vv[2..3] (keys 128..255) follow once the key block's registers are free, under the softmax phases
    __builtin_amdgcn_sched_barrier(0);
    uint4 kv[16], vv[4][4];
    {
        const uint4* kp = (const uint4*)(kb + (size_t)min(sw, (T - 1) >> 6) * 64 * hd) + lane;
#pragma unroll
        for (int c = 0; c < 16; ++c) kv[c] = kp[c * 64];
#pragma unroll
        for (int tr = 0; tr < 2; ++tr)
#pragma unroll
            for (int u = 0; u < 4; ++u) vv[tr][u] = *(const uint4*)(vb + (size_t)min(64 * tr + 4 * sw + kg + 16 * u, T - 1) * hd + dl * 8);
    }
    __builtin_amdgcn_sched_barrier(0);
    Q3_LDS_BARRIER();
    Q3_STAMP(a, 1);
    const float scale = 1.0f / sqrtf((float)hd);
    float ml0 = -INFINITY, ml1 = -INFINITY;
    for (int blk = sw; blk * 64 < T; blk += 4) {
        const int t = blk * 64 + lane;
        if (blk != sw) {  // (contexts beyond 256 keys: the following blocks of this wave)
            const uint4* kp = (const uint4*)(kb + (size_t)blk * 64 * hd) + lane;
#pragma unroll
            for (int c = 0; c < 16; ++c) kv[c] = kp[c * 64];
        }
        if (t == pos) {
#pragma unroll
            for (int c = 0; c < 16; ++c) kv[c] = *(const uint4*)(knew + 4 * c);
        }
        float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const float4 qa = *(const float4*)(qh + c * 8), qb = *(const float4*)(qh + c * 8 + 4);
            const float4 ra = *(const float4*)(qh + hd + c * 8), rb = *(const float4*)(qh + hd + c * 8 + 4);
            const float k0 = q3_u2f(kv[c].x << 16), k1 = q3_u2f(kv[c].x & 0xffff0000u), k2 = q3_u2f(kv[c].y << 16), k3 = q3_u2f(kv[c].y & 0xffff0000u);
            const float k4 = q3_u2f(kv[c].z << 16), k5 = q3_u2f(kv[c].z & 0xffff0000u), k6 = q3_u2f(kv[c].w << 16), k7 = q3_u2f(kv[c].w & 0xffff0000u);
            s0 = fmaf(qa.x, k0, s0); s1 = fmaf(ra.x, k0, s1); s0 = fmaf(qa.y, k1, s0); s1 = fmaf(ra.y, k1, s1);
            s0 = fmaf(qa.z, k2, s0); s1 = fmaf(ra.z, k2, s1); s0 = fmaf(qa.w, k3, s0); s1 = fmaf(ra.w, k3, s1);
            s0 = fmaf(qb.x, k4, s0); s1 = fmaf(rb.x, k4, s1); s0 = fmaf(qb.y, k5, s0); s1 = fmaf(rb.y, k5, s1);
            s0 = fmaf(qb.z, k6, s0); s1 = fmaf(rb.z, k6, s1); s0 = fmaf(qb.w, k7, s0); s1 = fmaf(rb.w, k7, s1);
        }
        s0 = s0 * scale; s1 = s1 * scale;
        if (t < T) { p0[t] = s0; p1[t] = s1; ml0 = fmaxf(ml0, s0); ml1 = fmaxf(ml1, s1); }
    }
#pragma unroll
    for (int tr = 2; tr < 4; ++tr)
#pragma unroll
        for (int u = 0; u < 4; ++u) vv[tr][u] = *(const uint4*)(vb + (size_t)min(64 * tr + 4 * sw + kg + 16 * u, T - 1) * hd + dl * 8);
    ml0 = wave_max(ml0); ml1 = wave_max(ml1);
    if (lane == 0) { mw[sw] = ml0; mw[4 + sw] = ml1; }
    Q3_LDS_BARRIER();
    const float m0 = fmaxf(fmaxf(mw[0], mw[1]), fmaxf(mw[2], mw[3])), m1 = fmaxf(fmaxf(mw[4], mw[5]), fmaxf(mw[6], mw[7]));
    float ls0 = 0.0f, ls1 = 0.0f;
    for (int blk = sw; blk * 64 < T; blk += 4) {
        const int t = blk * 64 + lane;
        if (t < T) {
            const float e0 = q3_expf(p0[t] - m0), e1 = q3_expf(p1[t] - m1);
            p0[t] = e0; p1[t] = e1; ls0 += e0; ls1 += e1;
        }
    }
    ls0 = wave_sum(ls0); ls1 = wave_sum(ls1);
    if (lane == 0) { lw[sw] = ls0; lw[4 + sw] = ls1; }
    Q3_LDS_BARRIER();
    Q3_STAMP(a, 2);
    float o0[8], o1[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { o0[e] = 0.0f; o1[e] = 0.0f; }
    const uint4 vn = *(const uint4*)(vnew + 4 * dl);
#pragma unroll
    for (int tr = 0; tr < 4; ++tr) {  // the first four trips: value rows already in registers
        const int t0 = 64 * tr + 4 * sw + kg;
        if (64 * tr >= T) break;  // (uniform)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int t = t0 + 16 * u;
            if (t < T) {
                const float pa = p0[t], pb = p1[t];
                const uint4 w = (t == pos) ? vn : vv[tr][u];
                const float v0 = q3_u2f(w.x << 16), v1 = q3_u2f(w.x & 0xffff0000u), v2 = q3_u2f(w.y << 16), v3 = q3_u2f(w.y & 0xffff0000u);
                const float v4 = q3_u2f(w.z << 16), v5 = q3_u2f(w.z & 0xffff0000u), v6 = q3_u2f(w.w << 16), v7 = q3_u2f(w.w & 0xffff0000u);
                o0[0] = fmaf(pa, v0, o0[0]); o0[1] = fmaf(pa, v1, o0[1]); o0[2] = fmaf(pa, v2, o0[2]); o0[3] = fmaf(pa, v3, o0[3]);
                o0[4] = fmaf(pa, v4, o0[4]); o0[5] = fmaf(pa, v5, o0[5]); o0[6] = fmaf(pa, v6, o0[6]); o0[7] = fmaf(pa, v7, o0[7]);
                o1[0] = fmaf(pb, v0, o1[0]); o1[1] = fmaf(pb, v1, o1[1]); o1[2] = fmaf(pb, v2, o1[2]); o1[3] = fmaf(pb, v3, o1[3]);
                o1[4] = fmaf(pb, v4, o1[4]); o1[5] = fmaf(pb, v5, o1[5]); o1[6] = fmaf(pb, v6, o1[6]); o1[7] = fmaf(pb, v7, o1[7]);
            }
        }
    }
    for (int t0 = 256 + 4 * sw + kg; t0 < T; t0 += 64) {  // contexts beyond 256 keys: four value rows loaded together per trip (ascending keys per lane, as above)
        uint4 vl[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) vl[u] = *(const uint4*)(vb + (size_t)min(t0 + 16 * u, T - 1) * hd + dl * 8);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int t = t0 + 16 * u;
            if (t < T) {
                const float pa = p0[t], pb = p1[t];
                const uint4 w = (t == pos) ? vn : vl[u];
                const float v0 = q3_u2f(w.x << 16), v1 = q3_u2f(w.x & 0xffff0000u), v2 = q3_u2f(w.y << 16), v3 = q3_u2f(w.y & 0xffff0000u);
                const float v4 = q3_u2f(w.z << 16), v5 = q3_u2f(w.z & 0xffff0000u), v6 = q3_u2f(w.w << 16), v7 = q3_u2f(w.w & 0xffff0000u);
                o0[0] = fmaf(pa, v0, o0[0]); o0[1] = fmaf(pa, v1, o0[1]); o0[2] = fmaf(pa, v2, o0[2]); o0[3] = fmaf(pa, v3, o0[3]);
                o0[4] = fmaf(pa, v4, o0[4]); o0[5] = fmaf(pa, v5, o0[5]); o0[6] = fmaf(pa, v6, o0[6]); o0[7] = fmaf(pa, v7, o0[7]);
                o1[0] = fmaf(pb, v0, o1[0]); o1[1] = fmaf(pb, v1, o1[1]); o1[2] = fmaf(pb, v2, o1[2]); o1[3] = fmaf(pb, v3, o1[3]);
                o1[4] = fmaf(pb, v4, o1[4]); o1[5] = fmaf(pb, v5, o1[5]); o1[6] = fmaf(pb, v6, o1[6]); o1[7] = fmaf(pb, v7, o1[7]);
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        o0[e] = o0[e] + __shfl_xor(o0[e], 16); o0[e] = o0[e] + __shfl_xor(o0[e], 32);
        o1[e] = o1[e] + __shfl_xor(o1[e], 16); o1[e] = o1[e] + __shfl_xor(o1[e], 32);
    }
    if (kg == 0) {
        *(float4*)(ow + (0 * 4 + sw) * hd + dl * 8) = (float4){o0[0], o0[1], o0[2], o0[3]}; *(float4*)(ow + (0 * 4 + sw) * hd + dl * 8 + 4) = (float4){o0[4], o0[5], o0[6], o0[7]};
        *(float4*)(ow + (1 * 4 + sw) * hd + dl * 8) = (float4){o1[0], o1[1], o1[2], o1[3]}; *(float4*)(ow + (1 * 4 + sw) * hd + dl * 8 + 4) = (float4){o1[4], o1[5], o1[6], o1[7]};
    }
    Q3_LDS_BARRIER();
    Q3_STAMP(a, 3);
    if (tid < 128) {  // two consecutive dims of one head per thread: one 4-byte store of the bf16 pair
        const int h2 = tid >> 6, d0 = 2 * (tid & 63);
        const float l = ((lw[h2 * 4] + lw[h2 * 4 + 1]) + lw[h2 * 4 + 2]) + lw[h2 * 4 + 3];
        float ov[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int d = d0 + q;
            const float r0 = ow[(h2 * 4 + 0) * hd + d], r1 = ow[(h2 * 4 + 1) * hd + d], r2 = ow[(h2 * 4 + 2) * hd + d], r3 = ow[(h2 * 4 + 3) * hd + d];
            ov[q] = (((r0 + r1) + r2) + r3) / l;
        }
        const int hq = g * 2 + h2;
        if (a.out_bf16 == 2) q3_q8_out2x16(ov[0], ov[1], row, hq * hd + d0, (a.Hq * hd) >> 6, a.out_rt16, (int8_t*)a.out, a.out_scale);  // W8A8: Q8_0 blocks (16 lanes x 2 dims)
        else if (a.out_bf16) *(uint32_t*)((uint16_t*)a.out + q3_atile_off(row, hq * hd + d0, (a.Hq * hd) >> 5)) = (uint32_t)q3_bf16(ov[0]) | ((uint32_t)q3_bf16(ov[1]) << 16);
        else *(float2*)(a.out + (size_t)row * a.ldo + (size_t)hq * hd + d0) = make_float2(ov[0], ov[1]);
    }
#ifdef Q3_STAMPS
    Q3_STAMP(a, 4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    Q3_STAMP(a, 5);
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// The same attention for short caches (n_ctx <= 64: the Predictor's <= 17 keys per frame), one row per slot. Every output follows the
// canonical order of k_attend bit for bit. Built around the kernel's dependent chain, measured with in-kernel timestamps
// (tools/chain_stamps.hip: the round-2 kernel spent 3.2 us before its first barrier on three dependent load round trips, 2.0 us on two
// divergent LDS-fed score chains and 1.7 us on the value pass with 64 cross-lane shuffles):
//  * workgroup = R query-head waves + ONE wave for k (lanes 0-31: RMSNorm + RoPE + append) and v (lanes 32-63: append); every global
//    operand of a wave — row segments, norm weights, RoPE entries, the cached keys (one key per lane, 16 chunks) and the first 16 cached
//    value rows (2 dims per lane) — is requested before anything is computed: one round trip
//  * the newest key reaches lane `pos` of the query waves through LDS in the cache's own packed layout, so cached and newest keys run the
//    SAME d-ascending fmaf chain (one code path); q is read back from LDS as broadcast 16-byte reads
//  * value pass without shuffles: a lane owns 2 output dims and keeps all 16 key partials u = t mod 16 of them in registers; p_t comes
//    from v_readlane (a scalar); r_w = (o_4w + o_4w+1) + (o_4w+2 + o_4w+3), o = ((r0 + r1) + r2) + r3 are plain adds in the lane
//  * l = the 64-lane butterfly of wave 0 of k_attend (+0 +0 +0 for the three absent waves is exact)
// ---------------------------------------------------------------------------------------------------------------------
template <int R>
__global__ __launch_bounds__((R + 1) * 64) void k_attend_small(Q3Attend a) {
    __shared__ __attribute__((aligned(16))) float qh[R][128];      // q after norm + RoPE (f32)
    __shared__ __attribute__((aligned(16))) uint32_t knew[64];     // newest key, bf16 pairs in the cache's chunk order: chunk c = 16 bytes at knew + 4 c
    __shared__ __attribute__((aligned(16))) uint32_t vnew[64];     // newest value, bf16 pairs: dims 2 i, 2 i + 1 in word i
    const int g = blockIdx.x, row = blockIdx.y, wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    Q3_STAMP(a, 0);
    int pos, slot;
    q3_row_map(row, a.row_pos, a.row_slot, a.slot_mod, a.pos_const, &pos, &slot);
    if (pos < 0) return;
    const int T = pos + 1, hd = 128, half = 64, nl = 32;
    const Q3QkPrep& pr = a.prep;
    const size_t hb = ((size_t)slot * a.Hkv + g) * a.n_ctx;
    const uint16_t* kb = a.kc + hb * hd;
    const uint16_t* vb = a.vc + hb * hd;
    const float* rowp = a.qkv + (size_t)row * a.ld;
    const float* csp = pr.cs + (size_t)pos * half + 4 * (lane & 15);
    const float* snp = pr.sn + (size_t)pos * half + 4 * (lane & 15);
    if (wv == R) {
        // ---- k (lanes 0..31) and v (lanes 32..63) of this row: norm + RoPE, bf16, append, and the LDS copies the query waves read
        const bool isk = lane < nl;
        const float4 x4 = isk ? ((const float4*)(rowp + (size_t)(a.Hq + g) * hd))[lane] : ((const float4*)(rowp + (size_t)(a.Hq + a.Hkv + g) * hd))[lane - nl];
        float4 w4 = (float4){0.f, 0.f, 0.f, 0.f}, c4 = (float4){1.f, 1.f, 1.f, 1.f}, s4 = (float4){0.f, 0.f, 0.f, 0.f};
        if (isk) { w4 = ((const float4*)pr.knw)[lane]; c4 = *(const float4*)csp; s4 = *(const float4*)snp; }
        // prep_head's arithmetic, with the operands above (lanes >= 32 contribute +0 to the sum of squares, as there)
        const float4 v = isk ? x4 : (float4){0.f, 0.f, 0.f, 0.f};
        float acc = 0.0f;
        acc = fmaf(v.x, v.x, acc); acc = fmaf(v.y, v.y, acc); acc = fmaf(v.z, v.z, acc); acc = fmaf(v.w, v.w, acc);
        acc = wave_sum(acc);
        const float rinv = 1.0f / sqrtf(acc / (float)hd + pr.eps);
        const float y[4] = {(v.x * rinv) * w4.x, (v.y * rinv) * w4.y, (v.z * rinv) * w4.z, (v.w * rinv) * w4.w};
        const float cc[4] = {c4.x, c4.y, c4.z, c4.w}, ss[4] = {s4.x, s4.y, s4.z, s4.w};
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float other = __shfl_xor(y[e], 16);
            o[e] = (lane < 16) ? fmaf(-other, ss[e], y[e] * cc[e]) : fmaf(other, ss[e], y[e] * cc[e]);
        }
        if (isk) {
            uint2 kk;
            kk.x = (uint32_t)q3_bf16(o[0]) | ((uint32_t)q3_bf16(o[1]) << 16); kk.y = (uint32_t)q3_bf16(o[2]) | ((uint32_t)q3_bf16(o[3]) << 16);
            const int blk = pos >> 6, kl = pos & 63, c = lane >> 1, e0 = 4 * (lane & 1);
            *(uint2*)(pr.kc + hb * hd + ((size_t)(blk * (hd >> 3) + c) * 64 + kl) * 8 + e0) = kk;
            *(uint2*)(knew + 2 * lane) = kk;   // elements 4 lane .. 4 lane + 3 = chunk lane / 2, half lane & 1
        } else {
            const int j = lane - nl;
            uint2 vk;
            vk.x = (uint32_t)q3_bf16(x4.x) | ((uint32_t)q3_bf16(x4.y) << 16); vk.y = (uint32_t)q3_bf16(x4.z) | ((uint32_t)q3_bf16(x4.w) << 16);
            *(uint2*)(pr.vc + (hb + pos) * hd + 4 * j) = vk;
            *(uint2*)(vnew + 2 * j) = vk;
        }
        Q3_STAMP(a, 1);
        __syncthreads();
        return;
    }
    // ---- query head wv: operands first
    const int hq = g * R + wv;
    uint4 kv[16];
    uint32_t vv[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) kv[c] = lane < pos ? ((const uint4*)kb)[c * 64 + lane] : make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int u = 0; u < 16; ++u) vv[u] = u < pos ? *(const uint32_t*)(vb + (size_t)u * hd + 2 * lane) : 0u;
    float4 x4 = (float4){0.f, 0.f, 0.f, 0.f}, w4 = x4, c4 = (float4){1.f, 1.f, 1.f, 1.f}, s4 = x4;
    if (lane < nl) { x4 = ((const float4*)(rowp + (size_t)hq * hd))[lane]; w4 = ((const float4*)pr.qnw)[lane]; c4 = *(const float4*)csp; s4 = *(const float4*)snp; }
    {
        float acc = 0.0f;
        acc = fmaf(x4.x, x4.x, acc); acc = fmaf(x4.y, x4.y, acc); acc = fmaf(x4.z, x4.z, acc); acc = fmaf(x4.w, x4.w, acc);
        acc = wave_sum(acc);
        const float rinv = 1.0f / sqrtf(acc / (float)hd + pr.eps);
        const float y[4] = {(x4.x * rinv) * w4.x, (x4.y * rinv) * w4.y, (x4.z * rinv) * w4.z, (x4.w * rinv) * w4.w};
        const float cc[4] = {c4.x, c4.y, c4.z, c4.w}, ss[4] = {s4.x, s4.y, s4.z, s4.w};
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float other = __shfl_xor(y[e], 16);
            o[e] = (lane < 16) ? fmaf(-other, ss[e], y[e] * cc[e]) : fmaf(other, ss[e], y[e] * cc[e]);
        }
        if (lane < nl) *(float4*)(qh[wv] + 4 * lane) = (float4){o[0], o[1], o[2], o[3]};
    }
    Q3_STAMP(a, 1);
    __syncthreads();
    Q3_STAMP(a, 2);
    if (lane == pos) {  // the newest key, in the packed form the cached keys arrive in
#pragma unroll
        for (int c = 0; c < 16; ++c) kv[c] = *(const uint4*)(knew + 4 * c);
    }
    const float* q = qh[wv];
    float sc = 0.0f;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const float4 qa = *(const float4*)(q + c * 8), qb = *(const float4*)(q + c * 8 + 4);
        sc = fmaf(qa.x, q3_u2f(kv[c].x << 16), sc); sc = fmaf(qa.y, q3_u2f(kv[c].x & 0xffff0000u), sc);
        sc = fmaf(qa.z, q3_u2f(kv[c].y << 16), sc); sc = fmaf(qa.w, q3_u2f(kv[c].y & 0xffff0000u), sc);
        sc = fmaf(qb.x, q3_u2f(kv[c].z << 16), sc); sc = fmaf(qb.y, q3_u2f(kv[c].z & 0xffff0000u), sc);
        sc = fmaf(qb.z, q3_u2f(kv[c].w << 16), sc); sc = fmaf(qb.w, q3_u2f(kv[c].w & 0xffff0000u), sc);
    }
    sc = sc * (1.0f / sqrtf((float)hd));
    const float m = wave_max(lane < T ? sc : -INFINITY);
    const float e = lane < T ? q3_expf(sc - m) : 0.0f;
    float l = wave_sum(e);
    l = ((l + 0.0f) + 0.0f) + 0.0f;
#ifdef Q3_STAMPS
    asm volatile("" :: "v"(l)); Q3_STAMP(a, 3);
#endif
    // value pass: this lane's dims d0 = 2 lane, d0 + 1; partial u holds the keys t = u, u + 16, ... in ascending order
    float o0[16], o1[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) { o0[u] = 0.0f; o1[u] = 0.0f; }
    const uint32_t vn = vnew[lane];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        if (16 * c >= T) break;  // (uniform)
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int t = 16 * c + u;
            if (t >= T) break;  // (uniform)
            uint32_t w;
            if (t == pos) w = vn;
            else if (c == 0) w = vv[u];
            else w = *(const uint32_t*)(vb + (size_t)t * hd + 2 * lane);  // (caches beyond 16 keys: not the shipped Predictor)
            const float pt = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(e), t));  // (t is uniform: a scalar broadcast, no LDS)
            o0[u] = fmaf(pt, q3_u2f(w << 16), o0[u]);
            o1[u] = fmaf(pt, q3_u2f(w & 0xffff0000u), o1[u]);
        }
    }
    float r0[4], r1[4];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        r0[w] = (o0[4 * w] + o0[4 * w + 1]) + (o0[4 * w + 2] + o0[4 * w + 3]);
        r1[w] = (o1[4 * w] + o1[4 * w + 1]) + (o1[4 * w + 2] + o1[4 * w + 3]);
    }
    const float ov0 = (((r0[0] + r0[1]) + r0[2]) + r0[3]) / l, ov1 = (((r1[0] + r1[1]) + r1[2]) + r1[3]) / l;
    const int d0 = 2 * lane;
    if (a.out_bf16) *(uint32_t*)((uint16_t*)a.out + q3_atile_off(row, hq * hd + d0, (a.Hq * hd) >> 5)) = (uint32_t)q3_bf16(ov0) | ((uint32_t)q3_bf16(ov1) << 16);
    else *(float2*)(a.out + (size_t)row * a.ldo + (size_t)hq * hd + d0) = make_float2(ov0, ov1);
#ifdef Q3_STAMPS
    Q3_STAMP(a, 4);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    Q3_STAMP(a, 5);
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// Pass A of the Predictor: TWO rows per slot in one launch — row b at position 0 (the projected hidden row), row slot_mod + b at
// position 1 (the first code row) — and nothing cached yet. One workgroup per (KV group, slot): its four waves are (row, query head);
// k / v of both rows are prepared into LDS (and appended to the cache for the later passes), so no wave reads the cache and the
// separate k_qk_prep launch is not needed. Scores, softmax and PV follow k_attend_small's expressions term by term with every key
// served from LDS (the bf16-rounded values the cache holds): same chains, same order, same bits.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_attend_pair(Q3Attend a) {
    __shared__ __attribute__((aligned(16))) float kh[2][128], vh[2][128], qh[2][2][128], ps[2][2][64];
    const int g = blockIdx.x, b = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int rr = wave >> 1, hh = wave & 1, R = 2, hd = 128;
    const int row = rr * a.slot_mod + b, slot = b, pos = rr;
    const Q3QkPrep& pr = a.prep;
    const size_t hb = ((size_t)slot * a.Hkv + g) * a.n_ctx;
    const float* rowp = a.qkv + (size_t)row * a.ld;
    const int half = hd >> 1, nl = hd >> 2;
    float qo[4], o4[4];
    prep_head(rowp + (size_t)(g * R + hh) * hd, pr.qnw, pr.eps, pr.cs + (size_t)pos * half, pr.sn + (size_t)pos * half, hd, lane, qo);
    if (lane < nl) *(float4*)(qh[rr][hh] + 4 * lane) = (float4){qo[0], qo[1], qo[2], qo[3]};
    if (hh == 0) {  // k of this row: norm + RoPE + append
        prep_head(rowp + (size_t)(a.Hq + g) * hd, pr.knw, pr.eps, pr.cs + (size_t)pos * half, pr.sn + (size_t)pos * half, hd, lane, o4);
        if (lane < nl) {
            const int blk = pos >> 6, kl = pos & 63, c = lane >> 1, e0 = 4 * (lane & 1);
            uint2 kk;
            kk.x = (uint32_t)q3_bf16(o4[0]) | ((uint32_t)q3_bf16(o4[1]) << 16); kk.y = (uint32_t)q3_bf16(o4[2]) | ((uint32_t)q3_bf16(o4[3]) << 16);
            *(uint2*)(pr.kc + hb * hd + ((size_t)(blk * (hd >> 3) + c) * 64 + kl) * 8 + e0) = kk;
            *(float4*)(kh[rr] + 4 * lane) = (float4){q3_round_bf16(o4[0]), q3_round_bf16(o4[1]), q3_round_bf16(o4[2]), q3_round_bf16(o4[3])};
        }
    } else if (lane < nl) {  // v of this row: append
        const float4 v4 = ((const float4*)(rowp + (size_t)(a.Hq + a.Hkv + g) * hd))[lane];
        *(float4*)(vh[rr] + 4 * lane) = (float4){q3_round_bf16(v4.x), q3_round_bf16(v4.y), q3_round_bf16(v4.z), q3_round_bf16(v4.w)};
        uint2 vk;
        vk.x = (uint32_t)q3_bf16(v4.x) | ((uint32_t)q3_bf16(v4.y) << 16); vk.y = (uint32_t)q3_bf16(v4.z) | ((uint32_t)q3_bf16(v4.w) << 16);
        *(uint2*)(pr.vc + (hb + pos) * hd + 4 * lane) = vk;
    }
    __syncthreads();
    const int T = pos + 1;
    const float* q = qh[rr][hh];
    const float scale = 1.0f / sqrtf((float)hd);
    float sc = 0.0f;
    if (lane < T) {  // key `lane`: the d-ascending chain
        const float* kk = kh[lane];
        for (int d = 0; d < hd; d += 4) {
            const float4 qa = *(const float4*)(q + d), ka = *(const float4*)(kk + d);
            sc = fmaf(qa.x, ka.x, sc); sc = fmaf(qa.y, ka.y, sc); sc = fmaf(qa.z, ka.z, sc); sc = fmaf(qa.w, ka.w, sc);
        }
    }
    sc = sc * scale;
    const float m = wave_max(lane < T ? sc : -INFINITY);
    const float e = lane < T ? q3_expf(sc - m) : 0.0f;
    ps[rr][hh][lane] = e;
    float l = wave_sum(e);
    l = ((l + 0.0f) + 0.0f) + 0.0f;
    const int kg = lane >> 4, dl = lane & 15;
    float out8[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) out8[i] = 0.0f;
#pragma unroll
    for (int uu = 0; uu < 4; ++uu) {
        float o[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) o[i] = 0.0f;
        for (int t = 4 * uu + kg; t < T; t += 16) {
            const float pt = ps[rr][hh][t];
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = fmaf(pt, vh[t][dl * 8 + i], o[i]);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float r = o[i] + __shfl_xor(o[i], 16);
            r = r + __shfl_xor(r, 32);
            out8[i] = uu == 0 ? r : out8[i] + r;
        }
    }
    if (kg == 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int d = dl * 8 + i;
            const float ov = out8[i] / l;
            if (a.out_bf16) ((uint16_t*)a.out)[q3_atile_off(row, (g * R + hh) * hd + d, (a.Hq * hd) >> 5)] = q3_bf16(ov);
            else a.out[(size_t)row * a.ldo + (size_t)(g * R + hh) * hd + d] = ov;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Prefill attention of a whole prompt run: one workgroup per (KV head, slot) keeps the run's keys (in the cache's packed block form) and
// values in LDS and its 8 waves each take (row, query head) tasks — k_attend<2, false> starts one 8-wave workgroup per (row, KV head) that
// fetches the same keys again (23 500 workgroups for 64 prompts: 220 us per layer). hd = 128, two query heads per KV head, runs of
// n <= 128 rows at positions 0 .. n - 1 (what admit_group builds). The canonical order is k_attend's, element for element:
//   score t: the d-ascending fmaf chain of lane t % 64 over block t / 64, times the scale; maximum over all t;
//   weights: q3_expf(score - max); their sum: per key-block class sw = block % 4 the lanes' sums in block order, the 64-lane butterfly,
//            then ((l0 + l1) + l2) + l3;
//   value pass: 16 partials per (row, head, 8 dims) — class u = t % 16 lives in lane group kg = u % 4 of class sw = u / 4, keys ascending —
//            combined (kg0 + kg1) + (kg2 + kg3) by the two shuffles, then ((r0 + r1) + r2) + r3 over sw; output = sum / l.
// A wave runs the four sw classes one after the other where k_attend runs them on four waves: the same sums in the same order.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void k_attend_prefill(Q3Attend a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int g = blockIdx.x, sg = blockIdx.y, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int row0 = a.seg[3 * sg], n = a.seg[3 * sg + 1], slot = a.seg[3 * sg + 2];
    constexpr int hd = 128;
    const int nblk = (n + 63) >> 6;
    uint4* kl = (uint4*)smem;                                  // [nblk][16 chunks][64 lanes]: a key block as the cache stores it
    uint4* vl = kl + (size_t)nblk * 1024;                       // [n][16]: value rows
    float* scr = (float*)(vl + (size_t)n * 16) + wave * 256;    // per wave: weights p[128] | query q[128]
    float* p = scr; float* q = scr + 128;
    const size_t hb = ((size_t)slot * a.Hkv + g) * a.n_ctx;
    {
        const uint4* kb = (const uint4*)(a.kc + hb * hd);
        const uint4* vb = (const uint4*)(a.vc + hb * hd);
        for (int i = tid; i < nblk * 1024; i += 512) kl[i] = kb[i];
        for (int i = tid; i < n * 16; i += 512) vl[i] = vb[i];
    }
    __syncthreads();
    const float scale = 1.0f / sqrtf((float)hd);
    const int kg = lane >> 4, dl = lane & 15;
    for (int task = wave; task < 2 * n; task += 8) {
        const int r = task >> 1, hh = task & 1, row = row0 + r, T = r + 1;
        {
            const float2 qv = *(const float2*)(a.qkv + (size_t)row * a.ld + (size_t)(g * 2 + hh) * hd + 2 * lane);
            *(float2*)(q + 2 * lane) = qv;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (one wave: its LDS operations complete in order; the compiler must not reorder around this)
        float mloc = -INFINITY;
        for (int blk = 0; blk * 64 < T; ++blk) {
            const int t = blk * 64 + lane;
            const uint4* kp = kl + (size_t)blk * 1024 + lane;
            float s = 0.0f;
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const uint4 kv = kp[c * 64];
                const float4 qa = *(const float4*)(q + c * 8), qb = *(const float4*)(q + c * 8 + 4);
                s = fmaf(qa.x, q3_u2f(kv.x << 16), s); s = fmaf(qa.y, q3_u2f(kv.x & 0xffff0000u), s);
                s = fmaf(qa.z, q3_u2f(kv.y << 16), s); s = fmaf(qa.w, q3_u2f(kv.y & 0xffff0000u), s);
                s = fmaf(qb.x, q3_u2f(kv.z << 16), s); s = fmaf(qb.y, q3_u2f(kv.z & 0xffff0000u), s);
                s = fmaf(qb.z, q3_u2f(kv.w << 16), s); s = fmaf(qb.w, q3_u2f(kv.w & 0xffff0000u), s);
            }
            s = s * scale;
            if (t < T) { p[t] = s; mloc = fmaxf(mloc, s); }
        }
        const float m = wave_max(mloc);
        float lw[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int sw = 0; sw < 4; ++sw) {
            float lsum = 0.0f;
            for (int blk = sw; blk * 64 < T; blk += 4) {
                const int t = blk * 64 + lane;
                if (t < T) { const float e = q3_expf(p[t] - m); p[t] = e; lsum += e; }
            }
            lw[sw] = wave_sum(lsum);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        float rsw[4][8];
#pragma unroll
        for (int sw = 0; sw < 4; ++sw) {
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = 0.0f;
            for (int t0 = 4 * sw + kg; t0 < T; t0 += 64) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int t = t0 + 16 * u;
                    if (t < T) {
                        const uint4 vv = vl[(size_t)t * 16 + dl];
                        const float pt = p[t];
                        o[0] = fmaf(pt, q3_u2f(vv.x << 16), o[0]); o[1] = fmaf(pt, q3_u2f(vv.x & 0xffff0000u), o[1]);
                        o[2] = fmaf(pt, q3_u2f(vv.y << 16), o[2]); o[3] = fmaf(pt, q3_u2f(vv.y & 0xffff0000u), o[3]);
                        o[4] = fmaf(pt, q3_u2f(vv.z << 16), o[4]); o[5] = fmaf(pt, q3_u2f(vv.z & 0xffff0000u), o[5]);
                        o[6] = fmaf(pt, q3_u2f(vv.w << 16), o[6]); o[7] = fmaf(pt, q3_u2f(vv.w & 0xffff0000u), o[7]);
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                o[e] = o[e] + __shfl_xor(o[e], 16);
                o[e] = o[e] + __shfl_xor(o[e], 32);
                rsw[sw][e] = o[e];
            }
        }
        const float l = ((lw[0] + lw[1]) + lw[2]) + lw[3];
        if (a.out_bf16 == 2) {  // W8A8: the head's output as Q8_0 blocks — a lane owns 8 consecutive dims, lanes dl ^ 1, dl ^ 2 the rest of its block
            float ov8[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) ov8[e] = (((rsw[0][e] + rsw[1][e]) + rsw[2][e]) + rsw[3][e]) / l;
            q3_q8_out8x4(ov8, row, (g * 2 + hh) * hd + dl * 8, (a.Hq * hd) >> 6, a.out_rt16, (int8_t*)a.out, a.out_scale, kg == 0);
        } else if (kg == 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float ov = ((rsw[0][e] + rsw[1][e]) + rsw[2][e]) + rsw[3][e];
                const int d = dl * 8 + e, col = (g * 2 + hh) * hd + d;
                if (a.out_bf16) ((uint16_t*)a.out)[q3_atile_off(row, col, (a.Hq * hd) >> 5)] = q3_bf16(ov / l);
                else a.out[(size_t)row * a.ldo + col] = ov / l;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the next task overwrites p / q
    }
}

// Which kernel serves the Talker's decode attention / the prefill of whole prompts. The variants produce the same bits
// (tests/test_parity_gpu.py compares them in one process through q3tts_k_attend_policy); the environment variables Q3TTS_ATT_OLD /
// Q3TTS_ATT_PREFILL_OLD give the initial values once per process (A/B runs).
//   decode:  0 = k_attend_gqa2 (default), 1 = k_attend<2, true>
//   prefill: 0 = k_attend_prefill when the launch has >= 128 (run, KV head) workgroups (default), 1 = never (k_attend<2, false>), 2 = whenever eligible
static int g_att_decode = 0, g_att_prefill = 0;
static std::once_flag g_att_once;
static void att_policy_init() {
    std::call_once(g_att_once, []() {
        const char* ev = getenv("Q3TTS_ATT_OLD"); g_att_decode = (ev && atoi(ev)) ? 1 : 0;
        ev = getenv("Q3TTS_ATT_PREFILL_OLD"); g_att_prefill = (ev && atoi(ev)) ? 1 : 0;
    });
}
void q3_attend_policy(int decode, int prefill) { att_policy_init(); g_att_decode = decode; g_att_prefill = prefill; }

void q3_launch_attend(const Q3Attend& a, hipStream_t s) {
    att_policy_init();
    const int R = a.Hq / a.Hkv;
    const size_t lds = ((size_t)R * a.n_ctx + R * a.hd + R * 4 * a.hd + R * 8 + 2 * a.hd) * sizeof(float);
    dim3 grid(a.Hkv, a.rows);
    static Q3PerDevice pd;  // the score buffer is R * n_ctx floats: above 64 KiB the dynamic LDS size has to be allowed per kernel (and per device)
    if (lds > 65536)
        pd.ensure(lds, [&]() {
            hipFuncSetAttribute((const void*)k_attend<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipFuncSetAttribute((const void*)k_attend<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipFuncSetAttribute((const void*)k_attend<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipFuncSetAttribute((const void*)k_attend<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipFuncSetAttribute((const void*)k_attend<4, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        });
    if (a.fused == 2) {  // two rows per slot, empty cache (the Predictor's pass A): see k_attend_pair
        hipLaunchKernelGGL(k_attend_pair, dim3(a.Hkv, a.slot_mod), dim3(256), 0, s, a);
        return;
    }
    if (a.fused && a.n_ctx <= 64 && a.hd == 128 && (R == 1 || R == 2)) {  // short caches (the Predictor): one wave per query head
        if (R == 2) hipLaunchKernelGGL((k_attend_small<2>), grid, dim3(192), 0, s, a);
        else hipLaunchKernelGGL((k_attend_small<1>), grid, dim3(128), 0, s, a);
        return;
    }
    if (a.fused && R == 2 && a.hd == 128) {  // the Talker's decode step: one workgroup of four waves per (slot, KV head), both query heads
        if (g_att_decode == 0) {  // (1: k_attend<2, true> below — same bits: test_attention_kernel_variants_agree)
            const size_t lds2 = ((size_t)2 * a.n_ctx + 2 * a.hd + 8 * a.hd + 16 + 128) * sizeof(float);
            static Q3PerDevice pd2;
            if (lds2 > 65536) pd2.ensure(lds2, [&]() { hipFuncSetAttribute((const void*)k_attend_gqa2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2); });
            hipLaunchKernelGGL(k_attend_gqa2, grid, dim3(256), lds2, s, a);
            return;
        }
    }
    if (a.fused) {
        if (R == 2) hipLaunchKernelGGL((k_attend<2, true>), grid, dim3(512), lds, s, a);
        else hipLaunchKernelGGL((k_attend<4, true>), grid, dim3(1024), lds, s, a);
        return;
    }
    // whole prompt runs (admit_group): keys and values once per run — when there are enough runs to occupy the chip (one workgroup per run and
    // KV head walks its rows 8 at a time: a single prompt of 31 rows took 45 us per layer on 8 workgroups against 9 us on k_attend's 248)
    if (R == 2 && a.hd == 128 && a.seg && a.seg_max_n <= 128 && g_att_prefill != 1 && (a.n_seg * a.Hkv >= 128 || g_att_prefill == 2)) {
        {
            const int nblk = (a.seg_max_n + 63) / 64;
            const size_t lds3 = (size_t)nblk * 16384 + (size_t)a.seg_max_n * 256 + 8 * 256 * sizeof(float);
            static Q3PerDevice pd3;
            if (lds3 > 65536) pd3.ensure(lds3, [&]() { hipFuncSetAttribute((const void*)k_attend_prefill, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3); });
            hipLaunchKernelGGL(k_attend_prefill, dim3(a.Hkv, a.n_seg), dim3(512), lds3, s, a);
            return;
        }
    }
    if (R == 1) hipLaunchKernelGGL((k_attend<1, false>), grid, dim3(256), lds, s, a);
    else if (R == 2) hipLaunchKernelGGL((k_attend<2, false>), grid, dim3(512), lds, s, a);
    else hipLaunchKernelGGL((k_attend<4, false>), grid, dim3(1024), lds, s, a);
}

// ---------------------------------------------------------------------------------------------------
// Sampler (H4: src/models/llama/mod.rs:666-772). 256 threads; keys sorted by (logit desc, index asc) which is
// what the reference's stable descending sort of an index-ordered list produces.
// ---------------------------------------------------------------------------------------------------
#define SAMP_MAX 4096
__device__ int sample_row(const float* logits, int limit, float temperature, int top_k_i, float top_p, float r,
                          unsigned long long* keys, float* probs) {
    const int tid = threadIdx.x;
    __shared__ unsigned long long wbest[4];
    if (temperature <= 0.0f) {  // :690-701
        unsigned long long best = 0;
        for (int i = tid; i < limit; i += 256) { const unsigned long long k = q3_argmax_key(logits[i], (uint32_t)i); best = k > best ? k : best; }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) { const unsigned long long o = __shfl_xor(best, m); best = o > best ? o : best; }
        if ((tid & 63) == 0) wbest[tid >> 6] = best;
        __syncthreads();
        unsigned long long b = wbest[0];
        for (int w = 1; w < 4; ++w) b = wbest[w] > b ? wbest[w] : b;
        __syncthreads();
        return q3_argmax_idx(b);
    }
    bool select = top_k_i > 0 && top_k_i <= 256 && top_k_i < limit;
    if (select) {
        // top-k without k block-wide reductions (each one is a chain of cross-lane steps: 40 rounds cost ~50 us). Only the k largest
        // keys are ever read below (:711-713); keys are unique (the index is part of the key) except 0 = "not a candidate".
        //  A  every thread's 16 candidates and their maximum; the k-th largest of the 256 thread maxima, T, is a lower bound of the
        //     k-th largest key (the k thread maxima >= T are k distinct keys >= T), found by counting: rank = #{maxima > mine}
        //  B  the candidates >= T (the top k are among them; typically ~2k of the 2160) are appended to a short list
        //  C  the list is ranked by counting (#{larger}) — ranks are a permutation because keys are unique — and rank r < k lands in keys[r]
        __shared__ unsigned long long lm[256];
        __shared__ unsigned long long thr_s;
        __shared__ int n_sel;
        unsigned long long* sel = (unsigned long long*)probs;  // SAMP_MAX floats = SAMP_MAX / 2 keys
        unsigned long long loc[SAMP_MAX / 256];
        unsigned long long lmax = 0;
#pragma unroll
        for (int j = 0; j < SAMP_MAX / 256; ++j) { const int i = tid + j * 256; loc[j] = i < limit ? q3_argmax_key(logits[i], (uint32_t)i) : 0ull; lmax = loc[j] > lmax ? loc[j] : lmax; }
        lm[tid] = lmax;
        if (tid == 0) { thr_s = 0ull; n_sel = 0; }
        __syncthreads();
        int cnt = 0;
        for (int j = 0; j < 256; ++j) cnt += lm[j] > lmax ? 1 : 0;
        if (cnt == top_k_i - 1 && lmax != 0ull) thr_s = lmax;  // (stays 0 when fewer than k threads hold a candidate: then everything is listed)
        __syncthreads();
        const unsigned long long T = thr_s;
#pragma unroll
        for (int j = 0; j < SAMP_MAX / 256; ++j)
            if (loc[j] != 0ull && loc[j] >= T) { const int pos = atomicAdd(&n_sel, 1); if (pos < SAMP_MAX / 2) sel[pos] = loc[j]; }
        __syncthreads();
        const int n = n_sel;
        if (n > SAMP_MAX / 2) select = false;  // (uniform) pathological input: the list does not fit, take the full sort below
        else {
            for (int i = tid; i < top_k_i; i += 256) keys[i] = 0ull;
            __syncthreads();
            for (int i = tid; i < n; i += 256) {
                const unsigned long long mine = sel[i];
                int rank = 0;
                for (int j = 0; j < n; ++j) rank += sel[j] > mine ? 1 : 0;
                if (rank < top_k_i) keys[rank] = mine;
            }
        }
        __syncthreads();
    }
    int NP = 64;
    while (NP < limit) NP <<= 1;
    if (!select) {
    for (int i = tid; i < NP; i += 256) keys[i] = i < limit ? q3_argmax_key(logits[i], (uint32_t)i) : 0ull;
    __syncthreads();
    }
    for (int k = 2; !select && k <= NP; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < NP; i += 256) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const unsigned long long x = keys[i], y = keys[ixj];
                    const bool desc = (i & k) == 0;
                    if (desc ? (x < y) : (x > y)) { keys[i] = y; keys[ixj] = x; }
                }
            }
            __syncthreads();
        }
    __shared__ int result;
    int n_all = limit;
    {
        const size_t top_k = (size_t)(long long)top_k_i;                 // `top_k as usize` :646
        if (top_k > 0 && top_k < (size_t)n_all) n_all = (int)top_k;      // :711-713
    }
    {   // :717-723, element-wise: evaluated by all threads; the sums below stay sequential in the reference's order
        const float max_logit = n_all > 0 ? q3_key_value(keys[0]) : 0.0f;    // :716
        for (int i = tid; i < n_all; i += 256) probs[i] = q3_expf((q3_key_value(keys[i]) - max_logit) / temperature);
    }
    __syncthreads();
    // :726-764. The reference's sums are sequential f32 adds and stay so (thread 0, in order); the divisions between them are
    // element-wise and run on all threads — one thread doing everything spent ~15 us of this kernel walking LDS five times.
    __shared__ float bc_s; __shared__ int cut_s;
    int n = n_all;
    if (tid == 0) { float sum = 0.0f; for (int i = 0; i < n; ++i) sum += probs[i]; bc_s = sum; }
    __syncthreads();
    { const float sum = bc_s; if (sum > 0.0f) for (int i = tid; i < n; i += 256) probs[i] /= sum; }     // :726-731
    __syncthreads();
    if (top_p < 1.0f) {                                                  // :734-753
        if (tid == 0) {
            float cum = 0.0f; int cutoff = n;
            for (int i = 0; i < n; ++i) { cum += probs[i]; if (cum >= top_p) { cutoff = i + 1; break; } }
            float ns = 0.0f;
            for (int i = 0; i < cutoff; ++i) ns += probs[i];
            cut_s = cutoff; bc_s = ns;
        }
        __syncthreads();
        n = cut_s;
        { const float ns = bc_s; if (ns > 0.0f) for (int i = tid; i < n; i += 256) probs[i] /= ns; }
        __syncthreads();
    }
    if (tid == 0) {
        float cum = 0.0f; int res = -1;                                  // :756-764
        for (int i = 0; i < n; ++i) { cum += probs[i]; if (r < cum) { res = q3_argmax_idx(keys[i]); break; } }
        if (res < 0) res = n > 0 ? q3_argmax_idx(keys[0]) : 0;           // :767-770
        result = res;
    }
    __syncthreads();
    const int res = result;
    __syncthreads();
    return res;
}

// H4 / H5 for row b; returns the sampled code, or -1 when the row produces no frame (inactive slot, step bound, EOS)
__device__ int sample_frame(const Q3Sample& a, int b, unsigned long long* keys, float* probs) {
    const int tid = threadIdx.x, slot = a.row_slot[b];
    Q3Slot* sl = a.slots + slot;
    if (!sl->active) return -1;
    const int step = sl->n_frames;
    if (step >= sl->max_steps) {  // loop bound: src/tts/engine.rs:545
        __syncthreads();
        if (tid == 0) sl->active = 0;
        return -1;
    }
    float* lg = a.logits + (size_t)b * a.ld;
    int code0;
    if (sl->force_eos_at >= 0 && step == sl->force_eos_at) code0 = a.eos;
    else {
        if (step < sl->min_frames && a.eos < a.limit) { if (tid == 0) lg[a.eos] = -INFINITY; }
        __syncthreads();
        const float temperature = sl->temperature;
        const float r = temperature > 0.0f ? a.rng[sl->rng_base + step] : 0.0f;
        code0 = sample_row(lg, a.limit, temperature, sl->top_k, sl->top_p, r, keys, probs);
    }
    __syncthreads();
    if (tid == 0) {
        if (code0 == a.eos) { sl->hit_eos = 1; sl->active = 0; }  // :558-561
        else { a.codes[((size_t)slot * a.max_steps_cap + step) * a.ncb] = code0; sl->code0 = code0; }
    }
    return code0 == a.eos ? -1 : code0;
}
__global__ __launch_bounds__(256) void k_sample(Q3Sample a) {
    __shared__ unsigned long long keys[SAMP_MAX];
    __shared__ float probs[SAMP_MAX];
    sample_frame(a, blockIdx.x, keys, probs);
}
void q3_launch_sample(const Q3Sample& a, hipStream_t s) { hipLaunchKernelGGL(k_sample, dim3(a.B), dim3(256), 0, s, a); }

__global__ __launch_bounds__(256) void k_sample_rows(const float* logits, int ld, int limit, float temperature, int top_k,
                                                     float top_p, const float* r, int* out) {
    __shared__ unsigned long long keys[SAMP_MAX];
    __shared__ float probs[SAMP_MAX];
    const int b = blockIdx.x;
    const int id = sample_row(logits + (size_t)b * ld, limit, temperature, top_k, top_p, r ? r[b] : 0.0f, keys, probs);
    if (threadIdx.x == 0) out[b] = id;
}
void q3_launch_sample_rows(const float* logits, int n, int ld, int limit, float temperature, int top_k, float top_p,
                           const float* r, int* out, hipStream_t s) {
    hipLaunchKernelGGL(k_sample_rows, dim3(n), dim3(256), 0, s, logits, ld, limit, temperature, top_k, top_p, r, out);
}

// ---------------------------------------------------------------------------------------------------
// Predictor glue (H6/H7: src/tts/engine.rs:565-631)
// ---------------------------------------------------------------------------------------------------
__device__ void pred_input_row(const Q3PredInput& a, int b, int code0) {
    __shared__ float rinv_s;
    const int tid = threadIdx.x, d = a.d;
    const float* x = a.xT + (size_t)b * d;
    const bool wantX = a.X != nullptr;  // (uniform) the frame step normalises inside the projection tile instead
    if (wantX && tid < 64) {
        float acc = 0.0f;
        for (int c = tid; c < (d >> 2); c += 64) {
            const float4 v = ((const float4*)x)[c];
            acc = fmaf(v.x, v.x, acc); acc = fmaf(v.y, v.y, acc); acc = fmaf(v.z, v.z, acc); acc = fmaf(v.w, v.w, acc);
        }
        acc = wave_sum(acc);
        if (tid == 0) rinv_s = 1.0f / sqrtf(acc / (float)d + a.eps);
    }
    __syncthreads();
    const float rinv = wantX ? rinv_s : 0.0f;
    const bool ok = code0 >= 0 && code0 < a.codec0_rows;  // OOB rows embed as zeros: src/assets_manager.rs:419-437
    const float* e = a.codec0 + (size_t)(ok ? code0 : 0) * d;
    const float* pr = ok ? a.pproj0 + (size_t)code0 * a.dp : a.proj_b;  // proj(0) = bias
    // every operand of the row is requested before anything is stored (a loop of load, store, load, store ... paid one memory round
    // trip per 256 elements: 12 in a row, ~10 us of this kernel)
    constexpr int NI = 8, NP = 4;  // d <= 8 * 256, dp <= 4 * 256 (checked at engine creation for the shipped shapes; larger: extra trips)
    for (int i0 = 0; i0 < d; i0 += NI * 256) {
        float xv[NI], nv[NI], ev[NI];
#pragma unroll
        for (int u = 0; u < NI; ++u) {
            const int i = min(i0 + tid + u * 256, d - 1);
            xv[u] = wantX ? x[i] : 0.0f; nv[u] = wantX ? a.out_norm[i] : 0.0f; ev[u] = ok ? e[i] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < NI; ++u) {
            const int i = i0 + tid + u * 256;
            if (i < d) { if (wantX) a.X[(size_t)b * d + i] = (xv[u] * rinv) * nv[u]; a.fb[(size_t)b * d + i] = 0.0f + ev[u]; }
        }
    }
    const int r1 = a.B + b;  // pass A rows: [0, B) the projected hidden rows, [B, 2B) the code rows
    for (int i0 = 0; i0 < a.dp; i0 += NP * 256) {  // (dp % 256 == 0: whole waves, 16 consecutive lanes per norm tile)
        float pv[NP], wv[NP];
#pragma unroll
        for (int u = 0; u < NP; ++u) { const int i = min(i0 + tid + u * 256, a.dp - 1); pv[u] = pr[i]; wv[u] = a.nw[i]; }
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            const int i = i0 + tid + u * 256;
            if (i < a.dp) {  // (uniform per wave: dp % 256 == 0)
                a.px[(size_t)r1 * a.dp + i] = pv[u];
                q3_norm_out(pv[u], wv[u], a.xb + q3_atile_off(r1, i, a.dp >> 5), a.ssp + (size_t)r1 * (a.dp >> 4) + (i >> 4), (i & 15) == 0);
            }
        }
    }
}
__global__ __launch_bounds__(256) void k_pred_input(Q3PredInput a) {
    const Q3Slot* sl = a.slots + a.row_slot[blockIdx.x];
    if (!sl->active) return;
    pred_input_row(a, blockIdx.x, sl->code0);
}
void q3_launch_pred_input(const Q3PredInput& a, hipStream_t s) { hipLaunchKernelGGL(k_pred_input, dim3(a.B), dim3(256), 0, s, a); }

// ---------------------------------------------------------------------------------------------------------------------
// H6 — Assets::project (/root/reference/src/assets_manager.rs:383-399) in the reference's OWN arithmetic: f32 weights, the
// accumulator starts from the bias and takes `sum += h * w` (one f32 multiply, one f32 add: the build has -ffp-contract=off)
// over the inputs in ascending order. One thread per output element (a 2048-long dependent chain: latency-bound by design);
// 16 consecutive lanes = 16 consecutive outputs of one row, so the same kernel can emit the Predictor's norm inputs.
// tile = 16 rows x 16 outputs; the operands of 64 inputs at a time are staged through LDS by coalesced loads (a thread reading
// its own weight row straight from memory touches 16 cache lines per wave-load). The row stride of 68 floats keeps the 16 weight
// rows of a 128-bit LDS read on distinct banks.
// p.norm_w != nullptr: x holds RAW rows and the tile applies the RMSNorm itself while staging, x' = (x * rinv) * norm_w with rinv from
// the canonical 64-lane chain (the same operations pred_input_row used to store as X), so that the projection
// does not wait for another kernel's normalised copy.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int PJ_KC = 64, PJ_LD = PJ_KC + 4, PJ_LDS_FLOATS = 4 * 16 * PJ_LD;
__device__ void project_tile(const Q3Project& p, int bx, int by, float* lds) {
    constexpr int KC = PJ_KC, LD = PJ_LD;
    float* ws = lds; float* xs = lds + 2 * 16 * LD;  // [2][16 * LD] each
    __shared__ float rinv16[16];
    const int tid = threadIdx.x, oc = tid & 15, rr = tid >> 4;
    const int o = bx * 16 + oc, row = by * 16 + rr;
    // staging role: thread t loads 4 consecutive inputs (t & 15) of weight row / activation row (t >> 4)
    const float* wsrc = p.w + (size_t)(bx * 16 + rr) * p.n_in + 4 * oc;
    const float* xsrc = p.x + (size_t)min(by * 16 + rr, p.rows - 1) * p.ldx + 4 * oc;
    const float* nsrc = p.norm_w ? p.norm_w + 4 * oc : nullptr;
    float sum = p.bias[o];
    const int nch = p.n_in / KC;
    float4 wv = *(const float4*)wsrc, xv = *(const float4*)xsrc, nv = nsrc ? *(const float4*)nsrc : float4{1.0f, 1.0f, 1.0f, 1.0f};
    float rinv = 1.0f;
    if (nsrc) {  // (uniform) wave w owns rows 4w .. 4w + 3 of the tile: four chains side by side, lane c takes the float4 chunks c, c + 64, ...
        const int wave = tid >> 6, lane = tid & 63;
        const float4* xr[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) xr[u] = (const float4*)(p.x + (size_t)min(by * 16 + wave * 4 + u, p.rows - 1) * p.ldx);
        float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int c = lane; c < (p.n_in >> 2); c += 64) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = xr[u][c];
#pragma unroll
            for (int u = 0; u < 4; ++u) { acc[u] = fmaf(v[u].x, v[u].x, acc[u]); acc[u] = fmaf(v[u].y, v[u].y, acc[u]); acc[u] = fmaf(v[u].z, v[u].z, acc[u]); acc[u] = fmaf(v[u].w, v[u].w, acc[u]); }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { const float t = wave_sum(acc[u]); if (lane == 0) rinv16[wave * 4 + u] = 1.0f / sqrtf(t / (float)p.n_in + p.eps); }
        __syncthreads();
        rinv = rinv16[rr];
    }
    for (int c = 0; c < nch; ++c) {
        float* wl = ws + (c & 1) * 16 * LD; float* xl = xs + (c & 1) * 16 * LD;
        if (nsrc) { xv.x = (xv.x * rinv) * nv.x; xv.y = (xv.y * rinv) * nv.y; xv.z = (xv.z * rinv) * nv.z; xv.w = (xv.w * rinv) * nv.w; }
        *(float4*)(wl + rr * LD + 4 * oc) = wv; *(float4*)(xl + rr * LD + 4 * oc) = xv;
        if (c + 1 < nch) {
            wv = *(const float4*)(wsrc + (c + 1) * KC); xv = *(const float4*)(xsrc + (c + 1) * KC);
            if (nsrc) nv = *(const float4*)(nsrc + (c + 1) * KC);
        }
        __syncthreads();  // (two buffers: the stores of chunk c + 2 come after the barrier of chunk c + 1, which every reader of chunk c has passed)
        const float* wr = wl + oc * LD; const float* xr = xl + rr * LD;
#pragma unroll
        for (int k = 0; k < KC; k += 4) {
            const float4 a = *(const float4*)(xr + k), w4 = *(const float4*)(wr + k);
            sum += a.x * w4.x; sum += a.y * w4.y; sum += a.z * w4.z; sum += a.w * w4.w;
        }
    }
    const bool live = row < p.rows;
    if (live) p.y[(size_t)row * p.ldy + o] = sum;
    if (p.nw) {
        uint16_t hb = q3_bf16(sum * p.nw[o]);
        float sq = sum * sum;
        sq = sq + __shfl_xor(sq, 1); sq = sq + __shfl_xor(sq, 2); sq = sq + __shfl_xor(sq, 4); sq = sq + __shfl_xor(sq, 8);
        if (live) {
            p.xb[q3_atile_off(row, o, p.n_out >> 5)] = hb;
            if (oc == 0) p.ssp[(size_t)row * p.ld_ssp + (o >> 4)] = sq;
        }
    }
}
__global__ __launch_bounds__(256) void k_project(Q3Project p) {
    __shared__ __attribute__((aligned(16))) float lds[PJ_LDS_FLOATS];
    project_tile(p, blockIdx.x, blockIdx.y, lds);
}
static bool project_ok(const Q3Project& p) { return p.rows >= 1 && p.n_out % 16 == 0 && p.n_in % 64 == 0 && p.ldx % 4 == 0; }
int q3_launch_project(const Q3Project& p, hipStream_t s) {
    if (!project_ok(p)) return -1;
    hipLaunchKernelGGL(k_project, dim3(p.n_out / 16, (p.rows + 15) / 16), dim3(256), 0, s, p);
    return 0;
}

// the frame's first launch, two kinds of workgroup side by side (they touch disjoint data, so neither waits for the other):
//  [0, B)   H4/H5 (sample, EOS, bookkeeping) and, for rows that go on, the code row of the Predictor's pass A and the feedback start
//  [B, ...) H6 for the hidden rows, normalised in the tile (project_tile with norm_w)
static_assert(PJ_LDS_FLOATS * sizeof(float) <= SAMP_MAX * sizeof(unsigned long long), "the projection tile stages through the sampler's key array");
__global__ __launch_bounds__(256) void k_sample_input(Q3Sample a, Q3PredInput p, Q3Project pj) {
    __shared__ __attribute__((aligned(16))) unsigned long long keys[SAMP_MAX];
    __shared__ float probs[SAMP_MAX];
    if ((int)blockIdx.x >= a.B) {  // (uniform over the workgroup)
        const int t = blockIdx.x - a.B, nx = pj.n_out >> 4;
        project_tile(pj, t % nx, t / nx, (float*)keys);
        return;
    }
    const int code0 = sample_frame(a, blockIdx.x, keys, probs);  // (uniform over the workgroup)
    if (code0 < 0) return;
    pred_input_row(p, blockIdx.x, code0);
}
int q3_launch_sample_input(const Q3Sample& a, const Q3PredInput& p, const Q3Project& pj, hipStream_t s) {
    if (!project_ok(pj) || !pj.norm_w) return -1;
    hipLaunchKernelGGL(k_sample_input, dim3(a.B + (pj.n_out / 16) * ((pj.rows + 15) / 16)), dim3(256), 0, s, a, p, pj);
    return 0;
}

__global__ __launch_bounds__(256) void k_pred_next(Q3PredNext a) {
    const int b = blockIdx.x, tid = threadIdx.x, d = a.d;
    // the per-tile keys do not depend on the slot: requested before the slot state is looked at (one round trip less on the chain)
    unsigned long long kk = tid < a.n_key_parts ? a.keys[(size_t)b * a.n_key_parts + tid] : 0ull;
    const int slot = a.row_slot[b];
    Q3Slot* sl = a.slots + slot;
    const bool last = a.q == a.ncb - 1;
    if (!sl->active) {
        if (last && tid == 0) a.row_pos_t[b] = -1;
        return;
    }
    // code_q = argmax of the head's logits: the maximum of the per-tile keys the head GEMM left (ties -> smaller index, NaN never wins)
    __shared__ unsigned long long kmax_s[4];
    for (int t = tid + 256; t < a.n_key_parts; t += 256) { const unsigned long long o = a.keys[(size_t)b * a.n_key_parts + t]; kk = o > kk ? o : kk; }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) { const unsigned long long o = __shfl_xor(kk, m); kk = o > kk ? o : kk; }
    if ((tid & 63) == 0) kmax_s[tid >> 6] = kk;
    __syncthreads();
    kk = kmax_s[0];
#pragma unroll
    for (int w = 1; w < 4; ++w) kk = kmax_s[w] > kk ? kmax_s[w] : kk;
    const int code = q3_argmax_idx(kk);
    const bool ok = code >= 0 && code < a.rows_q;
    const float* e = a.codec_q + (size_t)(ok ? code : 0) * d;
    const int frame = sl->n_frames;
    if (tid == 0) a.codes[((size_t)slot * a.max_steps_cap + frame) * a.ncb + a.q] = code;
    // operands of the whole row first, stores after (see pred_input_row): the table rows and the running feedback sum in one round trip
    constexpr int NI = 8, NP = 4;
    const float* pr = ok ? a.pproj_q + (size_t)code * a.dp : a.proj_b;
    float pv[NP], wv[NP];
    if (!last) {
#pragma unroll
        for (int u = 0; u < NP; ++u) { const int i = min(tid + u * 256, a.dp - 1); pv[u] = pr[i]; wv[u] = a.nw[i]; }
    }
    for (int i0 = 0; i0 < d; i0 += NI * 256) {
        float ev[NI], fv[NI], tp[NI], nv[NI];
#pragma unroll
        for (int u = 0; u < NI; ++u) {
            const int i = min(i0 + tid + u * 256, d - 1);
            ev[u] = ok ? e[i] : 0.0f; fv[u] = a.fb[(size_t)b * d + i];
            if (last) { tp[u] = a.tts_pad[i]; nv[u] = a.nw[i]; }
        }
#pragma unroll
        for (int u = 0; u < NI; ++u) {
            const int i = i0 + tid + u * 256;
            if (i >= d) continue;  // (uniform per wave: d % 256 == 0)
            float f = fv[u] + ev[u];
            if (!last) a.fb[(size_t)b * d + i] = f;
            else {  // the Talker's next input row and its norm inputs for layer 0
                f = f + tp[u]; a.xT[(size_t)b * d + i] = f;
                if (a.xscale) {  // W8A8 Talker: the row as Q8_0 blocks (a half wave = 32 consecutive columns = one block)
                    q3_q8_out32(f * nv[u], b, i, d >> 6, a.x_rt16, (int8_t*)a.xb, a.xscale);
                    float sq = f * f;
                    sq = sq + __shfl_xor(sq, 1); sq = sq + __shfl_xor(sq, 2); sq = sq + __shfl_xor(sq, 4); sq = sq + __shfl_xor(sq, 8);
                    if ((i & 15) == 0) a.ssp[(size_t)b * (d >> 4) + (i >> 4)] = sq;
                } else
                q3_norm_out(f, nv[u], a.xb + q3_atile_off(b, i, d >> 5), a.ssp + (size_t)b * (d >> 4) + (i >> 4), (i & 15) == 0);
            }
        }
    }
    if (!last) {
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            const int i = tid + u * 256;
            if (i < a.dp) {
                a.px[(size_t)b * a.dp + i] = pv[u];
                q3_norm_out(pv[u], wv[u], a.xb + q3_atile_off(b, i, a.dp >> 5), a.ssp + (size_t)b * (a.dp >> 4) + (i >> 4), (i & 15) == 0);
            }
        }
        for (int i = tid + NP * 256; i < a.dp; i += 256) {  // (dp > 1024: not a shipped shape)
            const float v = pr[i];
            a.px[(size_t)b * a.dp + i] = v;
            q3_norm_out(v, a.nw[i], a.xb + q3_atile_off(b, i, a.dp >> 5), a.ssp + (size_t)b * (a.dp >> 4) + (i >> 4), (i & 15) == 0);
        }
    }
    if (last) {
        __syncthreads();
        if (tid == 0) { a.row_pos_t[b] = sl->cur_pos; sl->cur_pos = sl->cur_pos + 1; sl->n_frames = frame + 1; }
    }
}
void q3_launch_pred_next(const Q3PredNext& a, hipStream_t s) { hipLaunchKernelGGL(k_pred_next, dim3(a.B), dim3(256), 0, s, a); }

// ---------------------------------------------------------------------------------------------------
// Prompt builder (H1: src/tts/prompt.rs:141-277)
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float prompt_elem(int kind, int id, int i, const float* text, int text_vocab,
                                             const float* const* codec, int codec0_rows, int codecq_rows, int ncb,
                                             const float* spk, int d) {
    if (kind == 1) {  // src/assets_manager.rs:444-460
        if (id >= 0 && id < text_vocab) return text[(size_t)id * d + i];
        return fmodf((float)((unsigned long long)id * 17ull + (unsigned long long)i), 2.0f) - 1.0f;
    }
    if (kind >= 2) {  // src/assets_manager.rs:419-437
        const int q = kind - 2;
        const int rows = q == 0 ? codec0_rows : codecq_rows;
        if (id < 0) id = 0;
        if (q < ncb && id < rows) return codec[q][(size_t)id * d + i];
        return 0.0f;
    }
    if (kind == -1) return spk[i];
    return 0.0f;
}
__global__ void k_prompt_rows(const Q3PromptRow* rows, const float* text, int text_vocab, const float* const* codec,
                              int codec0_rows, int codecq_rows, int ncb, const float* spk, int d, float* out) {
    const Q3PromptRow r = rows[blockIdx.x];
    for (int i = threadIdx.x; i < d; i += blockDim.x) {
        const float a = prompt_elem(r.kindA, r.idA, i, text, text_vocab, codec, codec0_rows, codecq_rows, ncb, spk, d);
        float v = a;
        if (r.kindB != 0) v = a + prompt_elem(r.kindB, r.idB, i, text, text_vocab, codec, codec0_rows, codecq_rows, ncb, spk, d);
        out[(size_t)blockIdx.x * d + i] = v;
    }
}
void q3_launch_prompt_rows(const Q3PromptRow* rows, int n, const float* text, int text_vocab, const float* const* codec,
                           int codec0_rows, int codecq_rows, int ncb, const float* spk, int d, float* out, hipStream_t s) {
    hipLaunchKernelGGL(k_prompt_rows, dim3(n), dim3(256), 0, s, rows, text, text_vocab, codec, codec0_rows, codecq_rows, ncb, spk, d, out);
}
__global__ void k_prompt_ref_frames(const int* codes, const float* marker, const float* const* codec, int codec0_rows,
                                    int codecq_rows, int ncb, int d, float* out) {
    const int f = blockIdx.x;
    for (int i = threadIdx.x; i < d; i += blockDim.x) {
        float sum = 0.0f;
        for (int q = 0; q < 16; ++q)
            sum += prompt_elem(2 + q, codes[f * 16 + q], i, nullptr, 0, codec, codec0_rows, codecq_rows, ncb, nullptr, d);
        out[(size_t)f * d + i] = marker[i] + sum;
    }
}
void q3_launch_prompt_ref_frames(const int* codes, int n_frames, const float* marker, const float* const* codec,
                                 int codec0_rows, int codecq_rows, int ncb, int d, float* out, hipStream_t s) {
    hipLaunchKernelGGL(k_prompt_ref_frames, dim3(n_frames), dim3(256), 0, s, codes, marker, codec, codec0_rows, codecq_rows, ncb, d, out);
}

__global__ void k_copy_rows(float* dst, int ldd, const float* src, int lds, int cols) {
    const int r = blockIdx.x;
    for (int i = threadIdx.x; i < cols; i += blockDim.x) dst[(size_t)r * ldd + i] = src[(size_t)r * lds + i];
}
// dst[r] = src[perm[r]] (row compaction of the decode rows)
__global__ void k_gather_rows(float* dst, const float* src, const int* perm, int cols) {
    const int r = blockIdx.y;
    const float* sp = src + (size_t)perm[r] * cols;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < cols; i += gridDim.x * blockDim.x) dst[(size_t)r * cols + i] = sp[i];
}
void q3_launch_gather_rows(float* dst, const float* src, const int* perm, int rows, int cols, hipStream_t s) {
    hipLaunchKernelGGL(k_gather_rows, dim3((cols + 255) / 256, rows), dim3(256), 0, s, dst, src, perm, cols);
}
void q3_launch_copy_rows(float* dst, int ldd, const float* src, int lds, int rows, int cols, hipStream_t s) {
    hipLaunchKernelGGL(k_copy_rows, dim3(rows), dim3(256), 0, s, dst, ldd, src, lds, cols);
}

// One wave per row, 4 rows per workgroup. CPL = float4 chunks per lane (d = 256*CPL) held in registers so the row is
// read once: all loads issue first, then the canonical per-lane fmaf chain (chunks lane, lane+64, ...), the butterfly
// and one float4 store per chunk. CPL = 0: generic two-pass loop.
template <int CPL>
__global__ __launch_bounds__(256) void k_rmsnorm_rows(const float* x, int ldx, const float* w, float eps, int d, int rows, float* out, int ldo) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= rows) return;
    const float4* xr = (const float4*)(x + (size_t)r * ldx);
    float4* orow = (float4*)(out + (size_t)r * ldo);
    const float4* w4 = (const float4*)w;
    if constexpr (CPL > 0) {
        float4 v[CPL], g[CPL];
#pragma unroll
        for (int i = 0; i < CPL; ++i) { v[i] = xr[lane + 64 * i]; g[i] = w4[lane + 64 * i]; }
        float acc = 0.0f;
#pragma unroll
        for (int i = 0; i < CPL; ++i) { acc = fmaf(v[i].x, v[i].x, acc); acc = fmaf(v[i].y, v[i].y, acc); acc = fmaf(v[i].z, v[i].z, acc); acc = fmaf(v[i].w, v[i].w, acc); }
        acc = wave_sum(acc);
        const float rinv = 1.0f / sqrtf(acc / (float)d + eps);
#pragma unroll
        for (int i = 0; i < CPL; ++i)
            orow[lane + 64 * i] = make_float4((v[i].x * rinv) * g[i].x, (v[i].y * rinv) * g[i].y, (v[i].z * rinv) * g[i].z, (v[i].w * rinv) * g[i].w);
    } else {
        float acc = 0.0f;
        for (int c = lane; c < (d >> 2); c += 64) {
            const float4 v = xr[c];
            acc = fmaf(v.x, v.x, acc); acc = fmaf(v.y, v.y, acc); acc = fmaf(v.z, v.z, acc); acc = fmaf(v.w, v.w, acc);
        }
        acc = wave_sum(acc);
        const float rinv = 1.0f / sqrtf(acc / (float)d + eps);
        for (int c = lane; c < (d >> 2); c += 64) {
            const float4 v = xr[c], g = w4[c];
            orow[c] = make_float4((v.x * rinv) * g.x, (v.y * rinv) * g.y, (v.z * rinv) * g.z, (v.w * rinv) * g.w);
        }
    }
}
void q3_launch_rmsnorm_rows(const float* x, int ldx, const float* w, float eps, int d, int rows, float* out, int ldo, hipStream_t s) {
    dim3 grid((rows + 3) / 4);
    if (d == 2048) hipLaunchKernelGGL((k_rmsnorm_rows<8>), grid, dim3(256), 0, s, x, ldx, w, eps, d, rows, out, ldo);
    else if (d == 1024) hipLaunchKernelGGL((k_rmsnorm_rows<4>), grid, dim3(256), 0, s, x, ldx, w, eps, d, rows, out, ldo);
    else if (d == 512) hipLaunchKernelGGL((k_rmsnorm_rows<2>), grid, dim3(256), 0, s, x, ldx, w, eps, d, rows, out, ldo);
    else hipLaunchKernelGGL((k_rmsnorm_rows<0>), grid, dim3(256), 0, s, x, ldx, w, eps, d, rows, out, ldo);
}
