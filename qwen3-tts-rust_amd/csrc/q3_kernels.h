// q3_kernels.h — launchers of the hand-written gfx950 kernels of the codec-token decoder.
#pragma once
#include "q3_common.h"

enum { Q3_EPI_STORE = 0, Q3_EPI_RESID = 1, Q3_EPI_SWIGLU = 2, Q3_EPI_ARGMAX = 3, Q3_EPI_GELU = 4 };

// Experiment builds (-DQ3_STAMPS, tools/chain_stamps.hip): every workgroup's first thread records up to 8 constant-rate (100 MHz)
// timestamps at dbg[(linear workgroup id) * 8 + i]. Compiled out of the product library.
#ifdef Q3_STAMPS
#define Q3_STAMP_FIELD unsigned long long* dbg;
#define Q3_STAMP(p, i) do { if ((p).dbg && threadIdx.x == 0) (p).dbg[((size_t)blockIdx.x + (size_t)gridDim.x * blockIdx.y) * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define Q3_STAMP_FIELD
#define Q3_STAMP(p, i) do { } while (0)
#endif

// Kernel attributes (dynamic LDS above 64 KiB) belong to a function ON A DEVICE: a process that drives several GPUs (q3tts_node_*)
// has to set them once per device, not once per process. need(dev_value) returns true when the current device has not yet been given
// a value >= dev_value; the caller then sets the attribute and calls done(dev_value). One lock per launch site, uncontended.
#include <mutex>
struct Q3PerDevice {
    std::mutex mu; size_t have[64] = {0};
    template <class F> void ensure(size_t want, F set_attr) {
        int dev = 0; if (hipGetDevice(&dev) != hipSuccess) dev = 0;
        std::lock_guard<std::mutex> lk(mu);
        if (have[dev & 63] < want) { set_attr(); have[dev & 63] = want; }
    }
};

// Exact GEMM y[B][N] = x[B][K] * W[N][K]^T in the canonical order of DESIGN.md §4.1.
// W is bf16 in the tiled HBM layout of DESIGN.md §2.1: tile (nb = n/16, kb = k/32) is 1 KiB,
// lane l = (kq = l>>4, n = l&15) owns the 8 weights W[nb*16+n][kb*32+kq*8 .. +8].
struct Q3Gemm {
    const float* x; int ldx; int B;
    const uint4* w; int K, N;
    const float* norm_w; float eps;   // RMSNorm fused into the prologue when norm_w != nullptr
    const float* bias;                // STORE only
    float* y; int ldy;                // SWIGLU writes [B][N/2]
    unsigned long long* keys; int key_stride;  // ARGMAX: atomicMax(keys[row*key_stride])
    int epi;
#ifdef Q3_STAMPS
    unsigned long long* dbg;          // experiment builds only (tools/exp): s_memtime stamps of workgroup 0 / wave 0
#endif
};
void q3_launch_gemm(const Q3Gemm& g, hipStream_t s);
// The decoder's GEMM (q3_bgemm.hip): bf16 rows x tiled bf16 weights on v_mfma_f32_16x16x32_bf16 in the canonical
// order RAW = (((s_0 + s_1) + ...) + s_7) over 8 K-slices. K % 256 == 0, N % 16 == 0.
//   STORE   y[B][N] f32 = s_r * RAW (s_r from the producer's tile partials ssp; no scale when ssp == nullptr)
//   RESID   y[B][N] f32 += RAW (+= col_scale[column] * RAW when col_scale is given: the vocoder's LayerScale); with nw_next also yb = bf16(y * nw_next) (A-tiled) and ssp_out[B][N/16] (the consumer's norm inputs)
//   SWIGLU  yb = bf16(swiglu(s_r * RAW_gate, s_r * RAW_up)) (A-tiled, N/2 columns); each 16-column weight tile = 8 gate + 8 up columns
//   GELU    yb = bf16(gelu_erf(RAW + bias)) (A-tiled, N columns): the vocoder's ConvNeXt pointwise pair
//   ARGMAX  keys[row * key_stride + column tile] = the largest key(s_r * RAW, column) of that 16-column tile (key_stride >= N/16)
// bf16 activation rows live in the SAME fragment-tiled layout as the weights ("A-tiled"): tile (rt = row/16, kb = k/32) is 1 KiB,
// lane l = (kq = l>>4, r = l&15) owns the 16 bytes holding A[rt*16 + r][kb*32 + {4kq..4kq+3, 16+4kq..16+4kq+3}], so one wave-load of
// an A fragment is one contiguous KiB (a row-major load would touch 16 cache lines for 32 bytes each). Buffers hold
// ceil(rows/16)*16 rows. q3_atile_off(row, k, K/32) = element offset of A[row][k].
Q3_HD size_t q3_atile_off(int row, int k, int kblocks) {
    const int c = k & 31;
    return ((((size_t)(row >> 4) * kblocks + (k >> 5)) * 64 + ((c & 15) >> 2) * 16 + (row & 15)) << 3) + (c & 3) + ((c & 16) >> 2);
}
// Q8_0 ACTIVATIONS (W8A8: q3_bgemm8.hip). The int8 quants of a row operand live in the weights' tiled Q8 form with rows in
// place of columns: tile PAIR (rt = row/16, kp = k/64) is 1 KiB, lane (kq, r) owns 16 bytes — the 8 int8 of k-block 2 kp that its MFMA
// fragment holds (k = 4kq..4kq+3, 16+4kq..16+4kq+3), then the 8 of block 2 kp + 1. q3_q8_off = BYTE offset of element (row, k); kpairs = K/64.
Q3_HD size_t q3_q8_off(int row, int k, int kpairs) {
    const int c = k & 31;
    return ((((size_t)(row >> 4) * kpairs + (k >> 6)) * 64 + ((c & 15) >> 2) * 16 + (row & 15)) << 4) + ((k >> 5) & 1) * 8 + (c & 3) + ((c & 16) >> 2);
}
// ... and their f16 block scales as [K/64][row tiles][4 row quads][2 blocks][4 rows]: the 8 scales an MFMA lane needs for one (row tile, block
// pair) — rows 4 kq .. 4 kq + 3 of the tile, both blocks — are one 16-byte load. Index (in f16 elements) of the scale of (row, k-block kb);
// rt16 = row tiles of the buffer (rows padded to 16).
Q3_HD size_t q3_q8_scale_idx(int row, int kb, int rt16) {
    return ((((size_t)(kb >> 1) * rt16 + (row >> 4)) * 4 + ((row & 15) >> 2)) << 3) + (kb & 1) * 4 + (row & 3);
}
struct Q3BGemm {
    const uint16_t* a; int a_row0; int B;       // A-tiled bf16 rows [a_row0, a_row0 + B) of a buffer with K columns
    const uint4* w; int K, N;                   // tiled bf16
    int w_once;                                 // 1: the weights are read once per long interval (Talker decode): non-temporal loads when one workgroup owns a tile
    const float* ssp; int ld_ssp; int ntiles; int d_norm; float eps;  // row scale: s_r = 1 / sqrtf(SS(ssp[row][0..ntiles)) / d_norm + eps)
    int epi;
    float* y; int ldy;
    uint16_t* yb;                               // A-tiled bf16 output (RESID: N columns; SWIGLU: N/2 columns), rows as y
    const float* nw_next; float* ssp_out; int ld_ssp_out;
    const float* col_scale;                     // RESID only, optional
    // vocoder extras (all optional, zero = off): a bias added to RAW first (column % bias_n); f32 rows that live in per-slot segments
    // (row m at y + (m / seg_rows) * seg_stride + (m % seg_rows) * ldy); RESID without nw_next but with yb: yb = bf16(y) (A-tiled)
    const float* bias; int bias_n;
    int seg_rows; size_t seg_stride;
    unsigned long long* keys; int key_stride;   // ARGMAX: per-tile maxima, [B][key_stride]
    // Q8_0 weights (the Talker with q3tts_engine_config.talker_q8_0): w then holds ggml block_q8_0 quants in the tiled
    // Q8 layout — tile PAIR (nb = n/16, kp = k/64) is 1 KiB, lane l = (kq, n) owns 16 bytes: the 8 int8 of k-block 2 kp that its bf16
    // fragment would hold (k = 4kq..4kq+3, 16+4kq..16+4kq+3), then the 8 of k-block 2 kp + 1 — and wscale the f16 block scales
    // [N][K/32] (physical column order). RAW = the canonical Q8 order: per block P = MFMA from zero, t = fmaf(f32(d), P, t). K % 512 == 0.
    const uint16_t* wscale;
    // W8A8 (q3_launch_bgemm8; the Talker with talker_q8_0 = 2): `a` then holds the rows' int8 quants (q3_q8_off) and ascale their block scales
    // (q3_q8_scale_idx, a_rt16 row tiles); RESID / SWIGLU write the consumer's operand the same way: yb = int8 quants, yscale / y_rt16.
    const uint16_t* ascale; int a_rt16; uint16_t* yscale; int y_rt16;
    Q3_STAMP_FIELD
};
int q3_launch_bgemm(const Q3BGemm& g, hipStream_t s);
int q3_launch_bgemm8(const Q3BGemm& g, hipStream_t s);   // the same launch in ggml's Q8_0 x Q8_0 arithmetic (STORE / RESID / SWIGLU)
void q3_bgemm8_pick(const Q3BGemm& g, int* rt, int* nt);
void q3_bgemm_pick(const Q3BGemm& g, int* rt, int* nt, int* d, int* ntw, int* big);  // the instance q3_launch_bgemm takes for g (no launch)
void q3_bgemm_force(int rt, int nt);  // tuning only: force a tile instance (0, 0: back to the cost model)
void q3_bgemm_prepare();  // kernel attributes + the Q3TTS_BG_BIG policy (read once); call once outside stream capture
void q3_bgemm_big_policy(int policy);  // 1 / -1 / 0: k_bgemm_big always / never / when it fills the chip (tests, A/B runs)
// producer side of the split RMSNorm for plain f32 rows: xb = bf16(x * nw), ssp[row][t] = sum of squares of columns 16t..16t+15
void q3_launch_norm_inputs(const float* x, int ldx, int rows, int d, const float* nw, uint16_t* xb, int xb_row0, float* ssp, int ld_ssp, hipStream_t s);
// the same for a W8A8 consumer: v = x * nw as Q8_0 blocks (int8 quants at xq, f16 scales at xscale, rt16 row tiles) + ssp
void q3_launch_norm_inputs_q8(const float* x, int ldx, int rows, int d, const float* nw, int8_t* xq, uint16_t* xscale, int rt16, float* ssp, int ld_ssp, hipStream_t s);
// H6 (src/assets_manager.rs:383-399) in the reference's own f32 sequence: y[row][o] = bias[o]; for i: y += x[row][i] * w[o][i].
// nw != nullptr: also the norm inputs of y (xb, ssp) for the Predictor's first layer.
struct Q3Project {
    const float* x; int ldx; int rows;
    const float* w; const float* bias; int n_in, n_out;   // w f32 row-major [n_out][n_in]
    float* y; int ldy;
    const float* nw; uint16_t* xb; float* ssp; int ld_ssp;   // xb A-tiled (n_out columns), rows as y
    const float* norm_w; float eps;   // != nullptr: x holds raw rows, the kernel projects rmsnorm(x) * norm_w (normalised while staging)
};
int q3_launch_project(const Q3Project& p, hipStream_t s);
// Workgroup barrier that orders LDS traffic only. __syncthreads() also fences global memory, and on gfx9-family parts loads and stores
// share one counter: with global loads (or LDS-DMA) in flight it becomes s_waitcnt vmcnt(0) — the barrier waits for the wave's slowest
// outstanding load and a software pipeline of loads collapses to a depth of one (k_attend_gqa2: first barrier 4.9 us after the start;
// the vocoder's LDS-DMA ring: every K step waited for the stage issued one step earlier). Use it only where nothing read after the barrier
// was written to GLOBAL memory by another wave of the workgroup; data brought by LDS-DMA needs its own s_waitcnt vmcnt(N) before it.
#define Q3_LDS_BARRIER() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); } while (0)
#ifdef __HIPCC__
__device__ __forceinline__ void q3_row_map(int row, const int* row_pos, const int* row_slot, int slot_mod, int pos_const, int* pos, int* slot) {
    if (slot_mod > 0) { *slot = row % slot_mod; *pos = pos_const + row / slot_mod; }
    else { *pos = row_pos[row]; *slot = *pos < 0 ? 0 : row_slot[row]; }
}
// one element of a norm-input row: 16 consecutive lanes own one tile (all of them must be active); xb_elem = the A-tiled slot
__device__ __forceinline__ void q3_norm_out(float v, float nwv, uint16_t* xb_elem, float* ssp_tile, bool tile_leader) {
    *xb_elem = q3_bf16(v * nwv);
    float sq = v * v;
    sq = sq + __shfl_xor(sq, 1); sq = sq + __shfl_xor(sq, 2); sq = sq + __shfl_xor(sq, 4); sq = sq + __shfl_xor(sq, 8);
    if (tile_leader) *ssp_tile = sq;
}
#endif

#ifdef __HIPCC__
// Q8_0 activation producers (W8A8): ggml's quantiser on a block of 32 consecutive columns of one row, from the f32 values.
// q3_q8_out32: lane <-> column — the 32 lanes of a half wave hold v for columns k .. k + 31 (k % 32 == lane % 32), all active.
__device__ __forceinline__ void q3_q8_out32(float v, int row, int k, int kpairs, int rt16, int8_t* q, uint16_t* sc) {
    float amax = fabsf(v);
#pragma unroll
    for (int m = 1; m <= 16; m <<= 1) amax = fmaxf(amax, __shfl_xor(amax, m));
    const float d = amax / 127.0f, id = d != 0.0f ? 1.0f / d : 0.0f;
    q[q3_q8_off(row, k, kpairs)] = (int8_t)(int)roundf(v * id);
    if ((k & 31) == 0) sc[q3_q8_scale_idx(row, k >> 5, rt16)] = __builtin_bit_cast(unsigned short, (_Float16)d);
}
// q3_q8_out2x16: a lane holds columns k0, k0 + 1 (k0 even), 16 consecutive lanes the block
__device__ __forceinline__ void q3_q8_out2x16(float v0, float v1, int row, int k0, int kpairs, int rt16, int8_t* q, uint16_t* sc) {
    float amax = fmaxf(fabsf(v0), fabsf(v1));
#pragma unroll
    for (int m = 1; m <= 8; m <<= 1) amax = fmaxf(amax, __shfl_xor(amax, m));
    const float d = amax / 127.0f, id = d != 0.0f ? 1.0f / d : 0.0f;
    q[q3_q8_off(row, k0, kpairs)] = (int8_t)(int)roundf(v0 * id);
    q[q3_q8_off(row, k0 + 1, kpairs)] = (int8_t)(int)roundf(v1 * id);
    if ((k0 & 31) == 0) sc[q3_q8_scale_idx(row, k0 >> 5, rt16)] = __builtin_bit_cast(unsigned short, (_Float16)d);
}
// q3_q8_out8x4: a lane holds 8 consecutive columns k0 .. k0 + 7 (k0 % 8 == 0), four neighbouring lanes (lane ^ 1, lane ^ 2) the block
__device__ __forceinline__ void q3_q8_out8x4(const float* v, int row, int k0, int kpairs, int rt16, int8_t* q, uint16_t* sc, bool store) {
    float amax = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; ++i) amax = fmaxf(amax, fabsf(v[i]));
    amax = fmaxf(amax, __shfl_xor(amax, 1)); amax = fmaxf(amax, __shfl_xor(amax, 2));
    const float d = amax / 127.0f, id = d != 0.0f ? 1.0f / d : 0.0f;
    if (store) {
#pragma unroll
        for (int i = 0; i < 8; ++i) q[q3_q8_off(row, k0 + i, kpairs)] = (int8_t)(int)roundf(v[i] * id);
        if ((k0 & 31) == 0) sc[q3_q8_scale_idx(row, k0 >> 5, rt16)] = __builtin_bit_cast(unsigned short, (_Float16)d);
    }
}
#endif

// weight tiling: dst tiled [N/16][K/32][64 lanes][8], source either synthetic or a row-major bf16 staging buffer
struct Q3Fill {
    uint4* dst; int N, K;            // physical (fused) matrix
    int mode;                        // 0: rows [row0, row0+rows) come from one logical tensor; 1: gate/up interleave
    int row0, rows;                  // mode 0: physical row range filled by this call
    uint32_t tid_a, tid_b;           // synthetic tensor ids (mode 1: a = gate, b = up)
    const uint16_t* src_a; const uint16_t* src_b;  // row-major [rows][K] bf16 or nullptr -> synthetic
    uint64_t seed; float scale;
    // Q8_0 destination (q3_launch_fill_tiled_q8): dst = the tiled Q8 layout above, dst_scale = f16 [N][K/32]. Sources: raw ggml block_q8_0
    // rows (src8_a / src8_b: [rows][K/32] blocks of 34 bytes, kept as stored), else the bf16 / synthetic sources above, quantised with
    // ggml's reference rule (d = amax / 127, q = roundf(x / d))
    uint16_t* dst_scale; const uint8_t* src8_a; const uint8_t* src8_b;
};
void q3_launch_fill_tiled(const Q3Fill& f, hipStream_t s);
void q3_launch_fill_tiled_q8(const Q3Fill& f, hipStream_t s);
void q3_launch_fill_f32(float* dst, size_t n, uint64_t seed, uint32_t tid, float base, float scale, int round_bf16, hipStream_t s);

// q/k RMSNorm + RoPE (q in place) + bf16 K/V append. One wave per (row, head).
struct Q3QkPrep {
    float* qkv; int ld; int rows;
    int Hq, Hkv, hd;
    const float* qnw; const float* knw; float eps;
    const float* cs; const float* sn;      // [n_ctx][hd/2]
    uint16_t* kc; uint16_t* vc; int n_ctx;  // layer base; per (slot, kv head): n_ctx*hd elements
    const int* row_pos; const int* row_slot;
    int slot_mod, pos_const;  // slot_mod > 0: slot = row % slot_mod, pos = pos_const + row / slot_mod (no loads: the Predictor's per-frame cache is indexed by row)
};
void q3_launch_qk_prep(const Q3QkPrep& a, hipStream_t s);

// decode / prefill attention over the cache (canonical order DESIGN.md §4.4). One workgroup per (kv head, row).
struct Q3Attend {
    const float* qkv; int ld; int rows;
    float* out; int ldo;
    int out_bf16;     // 1: out is an A-tiled bf16 buffer with Hq*hd columns (the O projection's operand); 0: f32 rows [row][ldo] (test hook)
                      // 2: W8A8 — out holds the rows as Q8_0 blocks (int8 quants, q3_q8_off) and out_scale / out_rt16 their f16 scales
    uint16_t* out_scale; int out_rt16;
    int Hq, Hkv, hd;
    const uint16_t* kc; const uint16_t* vc; int n_ctx;
    const int* row_pos; const int* row_slot;
    int slot_mod, pos_const;  // as in Q3QkPrep
    int fused;        // 1: every slot has exactly one row in this launch -> q/k prep + KV append done in-kernel (R >= 2)
                      // 2: rows b (position 0) and slot_mod + b (position 1) of every slot, nothing cached yet (R == 2, hd == 128): k_attend_pair
    Q3QkPrep prep;    // used when fused
    // prefill (fused == 0): the launch's rows as per-slot runs of consecutive positions 0 .. n - 1 — seg[i] = {first row, n, slot}, device memory.
    // With it (and hd = 128, two query heads per KV head, every n <= 128) one workgroup serves a whole run from LDS: k_attend_prefill
    const int* seg; int n_seg; int seg_max_n;
    Q3_STAMP_FIELD
};
void q3_launch_attend(const Q3Attend& a, hipStream_t s);
void q3_attend_policy(int decode, int prefill);  // test hook (q3tts_k_attend_policy): which kernel variant serves decode / prefill attention (same bits)

// Talker sampler + frame bookkeeping (H4/H5). One workgroup per slot.
struct Q3Sample {
    float* logits; int ld; int limit; int eos;
    Q3Slot* slots; int B;             // slots: ALL slots of the engine; row b works for slot row_slot[b]
    const int* row_slot;
    const float* rng;                 // per-slot draws: rng[slot.rng_base + step]
    int* codes; int max_steps_cap; int ncb;
};
void q3_launch_sample(const Q3Sample& a, hipStream_t s);
// stand-alone sampler for the test hook: n rows, explicit draws
void q3_launch_sample_rows(const float* logits, int n, int ld, int limit, float temperature, int top_k, float top_p,
                           const float* r, int* out, hipStream_t s);

// predictor input of pass A (rows [0, B): the projected hidden rows, rows [B, 2B): the code rows):
// X[b] = rmsnorm(xT[b]) (X == nullptr: left to the projection tile, Q3Project.norm_w); px[B + b] = proj(codec0[code0]) taken
// from the pre-projected table (row-independent exact GEMM: the table row IS what projecting on the fly gives);
// fb[b] = 0 + codec0[code0]
struct Q3PredInput {
    const float* xT; const float* out_norm; float eps; int d;
    const float* codec0; int codec0_rows;
    const float* pproj0; const float* proj_b; int dp;  // proj(codec0) table [codec0_rows][dp]; bias = proj(0)
    const Q3Slot* slots; const int* row_slot; float* X; float* px; float* fb; int B;
    const float* nw; uint16_t* xb; float* ssp;         // norm inputs of px row B + b (the code row; A-tiled xb, ssp ld dp/16)
};
void q3_launch_pred_input(const Q3PredInput& a, hipStream_t s);
// the frame's first kernel: sampler + code rows (B workgroups) and the H6 tiles of the hidden rows (pj, with norm_w) side by side in one launch
int q3_launch_sample_input(const Q3Sample& a, const Q3PredInput& p, const Q3Project& pj, hipStream_t s);

// after pass q-1: code_q from the argmax key, record it, fb += codec_q[code_q]; q<ncb-1: px[b] = projected emb;
// last: fb += tts_pad -> xT[b], row_pos_t[b] = cur_pos++, n_frames++
struct Q3PredNext {
    const unsigned long long* keys; int n_key_parts; int q; int ncb;  // keys[b][n_key_parts]: the head GEMM's per-tile maxima of pass q - 1
    const float* codec_q; int rows_q; int d;
    Q3Slot* slots; const int* row_slot; int B;
    int* codes; int max_steps_cap;
    float* fb; const float* tts_pad; float* xT; int* row_pos_t;
    const float* pproj_q; const float* proj_b; int dp; float* px;  // q < ncb-1: px[b] = proj(codec_q[code]) from the table
    const float* nw; uint16_t* xb; float* ssp;  // norm inputs of the row just written: px[b] (Predictor layer 0) or, last, xT[b] (Talker layer 0)
    uint16_t* xscale; int x_rt16;               // last pass with a W8A8 Talker: xb then takes the row as Q8_0 blocks (int8 quants + these f16 scales)
    Q3_STAMP_FIELD
};
void q3_launch_pred_next(const Q3PredNext& a, hipStream_t s);

// prompt builder (H1): out[row] = tabA[idA] (+ tabB[idB]) with the reference's OOB rules
struct Q3PromptRow { int32_t kindA, idA, kindB, idB; };  // kind: 0 none, 1 text, 2.. codec table (kind-2), -1 spk_emb, -2 zero
void q3_launch_prompt_rows(const Q3PromptRow* rows, int n, const float* text, int text_vocab, const float* const* codec,
                           int codec0_rows, int codecq_rows, int ncb, const float* spk, int d, float* out, hipStream_t s);
// clone prompt frame rows: out[row] = marker + sum_q codec_q[codes[row*16+q]] (src/tts/prompt.rs:79-96)
void q3_launch_prompt_ref_frames(const int* codes, int n_frames, const float* marker, const float* const* codec,
                                 int codec0_rows, int codecq_rows, int ncb, int d, float* out, hipStream_t s);

void q3_launch_copy_rows(float* dst, int ldd, const float* src, int lds, int rows, int cols, hipStream_t s);

// canonical RMSNorm of rows (test hook / hidden read-back)
void q3_launch_gather_rows(float* dst, const float* src, const int* perm, int rows, int cols, hipStream_t s);
void q3_launch_rmsnorm_rows(const float* x, int ldx, const float* w, float eps, int d, int rows, float* out, int ldo, hipStream_t s);
