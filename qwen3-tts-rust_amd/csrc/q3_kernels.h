// q3_kernels.h — launchers of the hand-written gfx950 kernels of the codec-token decoder.
#pragma once
#include "q3_common.h"

enum { Q3_EPI_STORE = 0, Q3_EPI_RESID = 1, Q3_EPI_SWIGLU = 2, Q3_EPI_ARGMAX = 3 };

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
// Predictor gate/up in the canonical bf16-MFMA order (q3_gemm_bf16.hip): fused RMSNorm prologue on f32 rows, SwiGLU
// epilogue; weights in the same tiled gate/up-interleaved layout. _ok(): the shapes it takes (K = d_model in {512, 1024}).
bool q3_gemm_bf16_norm_swiglu_ok(int K, int N);
bool q3_gemm_bf16_norm_ok(int K, int N);
int q3_launch_gemm_bf16_norm_store(const float* x, int ldx, int B, const uint4* w, int K, int N, const float* nw, float eps, float* y, int ldy,
                                   hipStream_t s);
int q3_launch_gemm_bf16_norm_swiglu(const float* x, int ldx, int B, const uint4* w, int K, int N, const float* nw, float eps, float* y, int ldy,
                                    hipStream_t s, int y_bf16 = 0);
// y[B][N] += canonical bf16 GEMM of bf16 rows x (residual epilogue): the Predictor's O and down projections
bool q3_gemm_bf16_plain_ok(int K);
int q3_launch_gemm_bf16_resid(const uint16_t* x, int ldx, int B, const uint4* w, int K, int N, float* y, int ldy, hipStream_t s);

// weight tiling: dst tiled [N/16][K/32][64 lanes][8], source either synthetic or a row-major bf16 staging buffer
struct Q3Fill {
    uint4* dst; int N, K;            // physical (fused) matrix
    int mode;                        // 0: rows [row0, row0+rows) come from one logical tensor; 1: gate/up interleave
    int row0, rows;                  // mode 0: physical row range filled by this call
    uint32_t tid_a, tid_b;           // synthetic tensor ids (mode 1: a = gate, b = up)
    const uint16_t* src_a; const uint16_t* src_b;  // row-major [rows][K] bf16 or nullptr -> synthetic
    uint64_t seed; float scale;
};
void q3_launch_fill_tiled(const Q3Fill& f, hipStream_t s);
void q3_launch_fill_f32(float* dst, size_t n, uint64_t seed, uint32_t tid, float base, float scale, int round_bf16, hipStream_t s);

// q/k RMSNorm + RoPE (q in place) + bf16 K/V append. One wave per (row, head).
struct Q3QkPrep {
    float* qkv; int ld; int rows;
    int Hq, Hkv, hd;
    const float* qnw; const float* knw; float eps;
    const float* cs; const float* sn;      // [n_ctx][hd/2]
    uint16_t* kc; uint16_t* vc; int n_ctx;  // layer base; per (slot, kv head): n_ctx*hd elements
    const int* row_pos; const int* row_slot;
};
void q3_launch_qk_prep(const Q3QkPrep& a, hipStream_t s);

// decode / prefill attention over the cache (canonical order DESIGN.md §4.4). One workgroup per (kv head, row).
struct Q3Attend {
    const float* qkv; int ld; int rows;
    float* out; int ldo;
    int out_bf16;     // 1: out is a bf16 buffer (uint16 bits, ldo in elements): the consumer is a bf16-MFMA GEMM (Predictor)
    int Hq, Hkv, hd;
    const uint16_t* kc; const uint16_t* vc; int n_ctx;
    const int* row_pos; const int* row_slot;
    int fused;        // 1: every slot has exactly one row in this launch -> q/k prep + KV append done in-kernel (R >= 2)
    Q3QkPrep prep;    // used when fused
};
void q3_launch_attend(const Q3Attend& a, hipStream_t s);

// Talker sampler + frame bookkeeping (H4/H5). One workgroup per slot.
struct Q3Sample {
    float* logits; int ld; int limit; int eos;
    Q3Slot* slots; int B;             // slots: ALL slots of the engine; row b works for slot row_slot[b]
    const int* row_slot;
    const float* rng;                 // per-slot draws: rng[slot.rng_base + step]
    int* codes; int max_steps_cap; int ncb;
    unsigned long long* keys;         // [B][ncb] argmax keys, zeroed here for the frame
};
void q3_launch_sample(const Q3Sample& a, hipStream_t s);
// stand-alone sampler for the test hook: n rows, explicit draws
void q3_launch_sample_rows(const float* logits, int n, int ld, int limit, float temperature, int top_k, float top_p,
                           const float* r, int* out, hipStream_t s);

// predictor input of pass A: X[b] = rmsnorm(xT[b]) (projected by a GEMM into px[2b]); px[2b+1] = proj(codec0[code0]) taken
// from the pre-projected table (row-independent exact GEMM: the table row IS what projecting on the fly gives);
// fb[b] = 0 + codec0[code0]
struct Q3PredInput {
    const float* xT; const float* out_norm; float eps; int d;
    const float* codec0; int codec0_rows;
    const float* pproj0; const float* proj_b; int dp;  // proj(codec0) table [codec0_rows][dp]; bias = proj(0)
    const Q3Slot* slots; const int* row_slot; float* X; float* px; float* fb; int B;
};
void q3_launch_pred_input(const Q3PredInput& a, hipStream_t s);

// after pass q-1: code_q from the argmax key, record it, fb += codec_q[code_q]; q<ncb-1: px[b] = projected emb;
// last: fb += tts_pad -> xT[b], row_pos_t[b] = cur_pos++, n_frames++
struct Q3PredNext {
    const unsigned long long* keys; int q; int ncb;
    const float* codec_q; int rows_q; int d;
    Q3Slot* slots; const int* row_slot; int B;
    int* codes; int max_steps_cap;
    float* fb; const float* tts_pad; float* xT; int* row_pos_t;
    const float* pproj_q; const float* proj_b; int dp; float* px;  // q < ncb-1: px[b] = proj(codec_q[code]) from the table
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
