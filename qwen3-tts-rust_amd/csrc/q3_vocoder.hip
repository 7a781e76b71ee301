// q3_vocoder.hip — streaming neural-codec vocoder (V1-V6 of SURVEY.md §8a) on gfx950.
//
// Replaces the onnxruntime session behind AudioDecoder::decode (/root/reference/src/models/onnx.rs:342-459): codes
// [N][16] (+ is_last) -> 24 kHz PCM, with the streaming state (conv histories, sliding-window KV ring) resident on
// the device per utterance slot instead of being deep-copied through the host on every call
// (src/models/onnx.rs:369-384,410-455). Structure: restated in oracle/q3_oracle_vocoder.c (same model family as
// transformers' qwen3_omni_moe Code2Wav); every dimension comes from q3tts_vocoder_config.
//
// Layout: activations are channels-last f32 [slot][hist + T][C]; every convolution (k taps, dilation d, transposed or
// not) is a multi-tap GEMM  out[t][n] = bias[n] + sum_tap X[t - (ntap-1-tap)*d][:] . W[tap][n][:]  on
// v_mfma_f32_16x16x32_bf16 (bf16 operands, f32 accumulate) with fused epilogues. All convolutions are causal, so
// chunked streaming equals one-shot decoding; each conv input keeps its last (k-1)*d rows per slot.
#include <algorithm>
#include <cmath>
#include <map>
#include <set>
#include <vector>

#include "q3_engine.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define VOC_MAX_NS 64   // slots per batched call
#define VOC_FCAP 4      // frames per slot per call (the reference's 4-frame chunk: src/tts/engine.rs:509-512)

enum { VC_CODEBOOK = 0, VC_PRE = 32, VC_TFM = 40, VC_FINAL_NORM = 60, VC_UP = 64, VC_DEC_IN = 72, VC_BLK = 80, VC_OUT = 120 };
enum { VW_W = 0, VW_B = 1, VW_IN_NORM = 2, VW_Q = 3, VW_K = 4, VW_V = 5, VW_O = 6, VW_LS_ATTN = 7, VW_POST_NORM = 8, VW_GATE = 9,
       VW_UP = 10, VW_DOWN = 11, VW_LS_MLP = 12, VW_DW_W = 13, VW_DW_B = 14, VW_LN_W = 15, VW_LN_B = 16, VW_PW1 = 17, VW_PW1_B = 18,
       VW_PW2 = 19, VW_PW2_B = 20, VW_GAMMA = 21, VW_ALPHA = 22, VW_BETA = 23, VW_W2 = 24, VW_B2 = 25, VW_ALPHA2 = 26, VW_BETA2 = 27 };
#define VTID(l, w) Q3_TID(Q3G_VOC, l, w)

struct VCall {  // passed by value to the kernels of one batched call
    int ns, nf;
    int slot[VOC_MAX_NS];
    int pos[VOC_MAX_NS];      // frames already decoded for the slot
};

struct VConv { int ntap = 1, dil = 1, cin = 0, nout = 0, bias_n = 0; uint16_t* w = nullptr; float* b = nullptr; };
// work buffer of a conv input: [VOC_MAX_NS][H + Tcap][C] (slot-major) + per-slot history [B][H][C]
struct VBuf { float* p = nullptr; float* hist = nullptr; int H = 0, C = 0, Tcap = 0; int bf16 = 0;  // bf16: elements are bf16 (GEMM-only inputs)
              size_t stride() const { return (size_t)(H + Tcap) * C; } };  // in elements

struct VLayer { float *in_norm, *post_norm, *ls_attn, *ls_mlp; VConv q, k, v, o, gate, up, down;
                uint4 *qkv_t = nullptr, *o_t = nullptr, *gu_t = nullptr, *down_t = nullptr;  // the four projections in the decoder GEMM's tiled layout (tfm_bg)
                VConv qkv, gu; };  // fused launches: qkv = rows of q | k | v; gu = 16-row groups of gate and up alternating
struct VUp { VConv ct, pw1, pw2; float *dw_w, *dw_b, *ln_w, *ln_b, *gamma; int r; VBuf dw_in;
             uint4 *ct_t = nullptr, *pw1_t = nullptr, *pw2_t = nullptr; };  // tiled copies for k_bgemm (up_bg)
struct VRes { float *ea, *ib, *ea2, *ib2; VConv c1, c2; VBuf c1_in; };
struct VBlk { float *ea, *ib; VConv ct; VRes res[3]; int r, cin, cout; VBuf ct_in; };

struct Q3Voc {
    q3tts_vocoder_config c;
    int spf = 1, B = 0, RW = 0;
    std::vector<float*> cb; const float** cb_dev = nullptr;
    VConv pre; VBuf pre_in;
    std::vector<VLayer> L; float* final_norm = nullptr;
    std::vector<VUp> U;
    VConv dec_in; VBuf dec_in_in;
    std::vector<VBlk> Bk;
    float *oea = nullptr, *oib = nullptr, *out_w = nullptr, *out_b = nullptr; VBuf out_in; int out_c = 0;
    float *kring = nullptr, *vring = nullptr;  // [n_layer][B][RW][HH]
    float* rope = nullptr; int rope_rows = 0;    // [position][hd/2][cos, sin], evaluated in double on the host like the oracle's
    struct ZeroEnt { char* base; unsigned long long bytes; };  // per-slot history blocks: base + slot * bytes
    ZeroEnt* zero_tab = nullptr; int n_zero = 0;  // q3_voc_reset: one launch instead of one memset per buffer
    bool tfm_bg = false;                         // the transformer's projections run on k_bgemm (every K a multiple of 256)
    bool up_bg = false; uint16_t* upb = nullptr; // so do the up-sampling stages' ConvTranspose / pointwise GEMMs; upb: a stage's output as A-tiled bf16 for the next
    float *x = nullptr, *xn = nullptr, *xnb = nullptr, *qkv = nullptr, *att = nullptr, *g = nullptr;  // transformer scratch [M][.] (xnb, att, g: bf16)
    float *t1 = nullptr, *t2 = nullptr;  // generic scratch (largest stage)
    float* pcm = nullptr; size_t pcm_stride = 0;         // [B][max_steps_cap * spf]
    std::vector<int> frames_done, last_flag;
    VCall* call_dev = nullptr;                   // the running call's slot / position table (kernels read it; voc_call uploads it in stream order)
    VCall* call_host = nullptr; hipEvent_t call_ev[16] = {}; unsigned call_i = 0;  // pinned staging ring of the uploads + "copy done" events
    std::map<unsigned long long, hipGraphExec_t> call_graphs;  // one captured call per (slots, frames, launch mode, switches): voc_call
    std::set<unsigned long long> call_seen;
    std::vector<void*> allocs;
};

// "Polite" launches: while the decoder is running beside it, every long-lived vocoder workgroup is ONE per CU (>= 81 KiB of LDS
// declared: a second cannot join it) with 4 waves of <= 208 VGPRs, so that 79 KiB of LDS and >= 304 VGPRs per SIMD stay free for the decoder's
// 8-wave workgroups — which otherwise wait, launch after launch, for a vocoder workgroup to retire on every CU (tools/coresidency_bench.hip:
// 3.7 us per launch alone, 24-81 us beside workgroups of 30-80 us that leave no room, 3.6-4.0 beside one that does). The vocoder itself is
// slower that way, which costs nothing while it hides behind the decoder; a call that has the GPU to itself (a draining batch's tail, the
// stand-alone hooks) and small calls (short-lived workgroups anyway) are launched greedily. The engine sets the mode per call
// (q3_voc_decode_batch); Q3TTS_VOC_POLITE=0 / 1 forces never / always. Same bits either way (a launch parameter and a tile choice).
static thread_local bool g_voc_polite_now = false;   // (one host thread drives an engine: q3tts_node_* runs one per device)
static int voc_polite_env() { const char* ev = getenv("Q3TTS_VOC_POLITE"); return ev ? (atoi(ev) ? 1 : 0) : -1; }  // (read per launch: the tests compare both modes in one process)
static bool voc_polite() { const int ev = voc_polite_env(); return ev >= 0 ? ev == 1 : g_voc_polite_now; }
static size_t voc_lds_floor(size_t lds) { return voc_polite() ? std::max(lds, (size_t)81 * 1024) : lds; }
// ------------------------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------------------------
struct VGemm {
    const float* x; size_t x_stride; int x_off;   // row (s, t) tap 0 shift 0 at x + s*x_stride + x_off + t*cin
    int T, M;                                     // rows per slot, total rows = ns*T
    VConv c;
    float* y; size_t y_stride; int y_off;         // out row (s,t) at y + s*y_stride + y_off + t*nout (t*nout/2 for epi 4)
    const float* scale;                           // epilogue 1: y += scale[n % scale_n] * (acc + bias)
    int scale_n;
    int epi;                                      // 0 store, 1 y += scale*(.), 2 y += (.), 3 gelu, 4 swiglu (16-column tiles alternate gate / up)
    int store;                                    // 0: the primary output is not written (only y2 is wanted)
    int a_bf16;                                   // x holds bf16 (the value a f32 source would be rounded to anyway); strides in elements
    int y_bf16;                                   // epi 4 only: write the SwiGLU result as bf16 (it only ever feeds a GEMM)
    float* y2; size_t y2_stride; int y2_off;      // optional second output: SnakeBeta(v) with the NEXT layer's parameters,
    const float *ea, *ib; int snake_n;            //   written straight into that layer's conv-input work buffer
    int y2_bf16;                                  //   ... which holds bf16 when it only ever feeds GEMMs
    int chunked;                                  // K-step order of the wide 7-tap convolutions (vconv_chunked): channel chunks of 32, taps inside
};
// K step -> (tap, first channel). Default: taps ascending, 32-wide channel steps inside a tap. chunked: 32-channel chunks ascending, inside a
// chunk the taps ascending — the order of k_vconv_tap, which keeps a chunk of the input rows in LDS for all taps.
__device__ __forceinline__ void vstep(const VGemm& g, int step, int kpt, int& tap, int& k0) {
    if (g.chunked) { const int ch = step / g.c.ntap; tap = step - ch * g.c.ntap; k0 = ch << 5; }
    else { tap = step / kpt; k0 = (step - tap * kpt) << 5; }
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }

// shared epilogue of one output element (row (s, t), column n)
__device__ __forceinline__ void vepi(const VGemm& g, float v, int s, int t, int n, float yold) {
    const int nout = g.c.nout;
    if (g.c.b) v += g.c.b[n % g.c.bias_n];
    float* yp = g.y + (size_t)s * g.y_stride + g.y_off + (size_t)t * nout + n;
    if (g.epi == 1) v = yold + g.scale[n % g.scale_n] * v;
    else if (g.epi == 2) v = yold + v;
    else if (g.epi == 3) v = gelu_erf(v);
    if (g.store) *yp = v;
    if (g.y2) {
        const int c = n % g.snake_n;
        const float sn = __sinf(v * g.ea[c]);
        const float sv = v + g.ib[c] * (sn * sn);
        const size_t o2 = (size_t)s * g.y2_stride + g.y2_off + (size_t)t * nout + n;
        if (g.y2_bf16) ((__bf16*)g.y2)[o2] = (__bf16)sv; else g.y2[o2] = sv;
    }
}

// Small-M GEMM (M <= 512 rows: the 12.5 Hz transformer, the first up-sampling stages, the drain phase of a batch):
// workgroup tile 64 x 32, wave = 16 rows x 32 cols (2 MFMA tiles), no LDS and no barriers. The few rows cannot hide
// memory latency with MFMA work, so fragments ride an 8-deep register ring (A: 32 B of f32 per lane -> bf16x8,
// B: 16 B of bf16 per lane per column tile) and the grid is cut fine enough to put a workgroup on every CU.
#define VS_PF 8
template <int NJ, bool ABF>  // NJ column tiles per wave: 2 (64 x 32 workgroup tile) or 1 (64 x 16: twice the workgroups for
                              // narrow N); ABF: the A operand is stored as bf16 (half the bytes, no conversion)
__global__ __launch_bounds__(256) void k_vgemm_small(VGemm g) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int m0 = blockIdx.y * 64 + wave * 16, n0 = blockIdx.x * (NJ * 16);
    if (m0 >= g.M) return;
    const int lr = lane & 15, kq = lane >> 4;
    const int cin = g.c.cin, nout = g.c.nout, kpt = cin >> 5, steps = g.c.ntap * kpt;
    const float* xrow; const uint16_t* xrow16;
    {
        int m = m0 + lr; if (m >= g.M) m = g.M - 1;
        const int s = m / g.T, t = m - s * g.T;
        const size_t eo = (size_t)s * g.x_stride + g.x_off + (size_t)t * cin + kq * 8;
        xrow = g.x + eo; xrow16 = (const uint16_t*)g.x + eo;
    }
    const uint16_t* wrow[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) { int n = n0 + j * 16 + lr; if (n >= nout) n = nout - 1; wrow[j] = g.c.w + (size_t)n * cin + kq * 8; }
    float4 ra[VS_PF][ABF ? 1 : 2]; uint4 rb[VS_PF][NJ];
    f32x4 acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // (no branch inside the ring: loads past the end re-read the last step and their A fragment is zeroed, so the
    //  compiler can count outstanding loads exactly: vmcnt(28) at each consumption instead of vmcnt(0))
#define VS_ISSUE(slot_, step_)                                                                        \
    do {                                                                                              \
        const int st__ = min((step_), steps - 1);                                                     \
        int tap__, k0__; vstep(g, st__, kpt, tap__, k0__);                                            \
        const long sh__ = (long)(g.c.ntap - 1 - tap__) * g.c.dil * cin - k0__;                          \
        if (ABF) ra[slot_][0] = *(const float4*)(xrow16 - sh__);                                      \
        else { ra[slot_][0] = *(const float4*)(xrow - sh__); ra[slot_][ABF ? 0 : 1] = *(const float4*)(xrow - sh__ + 4); } \
        const size_t wo__ = (size_t)tap__ * nout * cin + k0__;                                        \
        _Pragma("unroll") for (int jj__ = 0; jj__ < NJ; ++jj__) rb[slot_][jj__] = *(const uint4*)(wrow[jj__] + wo__); \
    } while (0)
#pragma unroll
    for (int j = 0; j < VS_PF; ++j) VS_ISSUE(j, j);
    __builtin_amdgcn_sched_barrier(0);
    for (int s0 = 0; s0 < steps; s0 += VS_PF) {
#pragma unroll
        for (int j = 0; j < VS_PF; ++j) {
            const int step = s0 + j;
            const float live = step < steps ? 1.0f : 0.0f;
            bf16x8 a;
            if (ABF) {  // 8 bf16 as they are; a step past the end contributes zeros
                float4 v0 = ra[j][0];
                if (step >= steps) v0 = make_float4(0.f, 0.f, 0.f, 0.f);
                a = *(const bf16x8*)&v0;
            } else {
                const float4 v0 = ra[j][0], v1 = ra[j][ABF ? 0 : 1];
                a[0] = (__bf16)(v0.x * live); a[1] = (__bf16)(v0.y * live); a[2] = (__bf16)(v0.z * live); a[3] = (__bf16)(v0.w * live);
                a[4] = (__bf16)(v1.x * live); a[5] = (__bf16)(v1.y * live); a[6] = (__bf16)(v1.z * live); a[7] = (__bf16)(v1.w * live);
            }
            bf16x8 bfr[NJ];
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj) { const uint4 u = rb[j][jj]; bfr[jj] = *(const bf16x8*)&u; }
            __builtin_amdgcn_sched_barrier(0);
            VS_ISSUE(j, step + VS_PF);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj) acc[jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bfr[jj], acc[jj], 0, 0, 0);
        }
    }
#undef VS_ISSUE
    // D layout: lane holds rows 4*(lane>>4)+e, column lane&15
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int m = m0 + 4 * kq + e;
        if (m >= g.M) continue;
        const int s = m / g.T, t = m - s * g.T;
        if (NJ == 2 && g.epi == 4) {  // column tile 0 = gate, tile 1 = the matching up columns (interleaved weight rows)
            const int n = blockIdx.x * 16 + lr;
            if (n0 + 16 + lr < nout) {
                const float gt = acc[0][e], up = acc[NJ - 1][e];
                const float sv = (gt / (1.0f + expf(-gt))) * up;
                const size_t yo = (size_t)s * g.y_stride + g.y_off + (size_t)t * (nout >> 1) + n;
                if (g.y_bf16) ((__bf16*)g.y)[yo] = (__bf16)sv; else g.y[yo] = sv;
            }
        } else {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int n = n0 + j * 16 + lr;
                if (n < nout) {
                    const float yold = (g.epi == 1 || g.epi == 2) ? g.y[(size_t)s * g.y_stride + g.y_off + (size_t)t * nout + n] : 0.0f;
                    vepi(g, acc[j][e], s, t, n, yold);
                }
            }
        }
    }
}

// LDS-tiled GEMM for the big-M convolutions (the decoder blocks: thousands of rows per slot): workgroup tile
// 128 x (NJ*32), K step 32, 2 x 2 waves of 64 x (NJ*16) (4 x NJ MFMA tiles); NJ = 3 serves the 96- and 192-channel
// blocks without padding waste. A (f32 -> bf16) and W tiles go through registers into double-buffered LDS; the
// registers run two K steps ahead of the LDS copy (three ahead of the MFMAs), rows are padded to 48 bf16 (96 B) so
// that the 16 lanes of a ds_read_b128 group hit 16 distinct 4-bank slots (any row stride = 32 mod 64 bytes does). Conv taps are just extra K steps with a
// shifted row pointer.
#define VR_SW(r) ((4 - (((r) >> 2) & 3)) & 3)  // chunk XOR of the unpadded 64-byte-row tiles (derivation: k_vgemm_ring)
#define VT_LD 48  // 96-byte rows: conflict-free for ds_read_b128's real 16-lane groups ({0-3, 12-15, 20-27}, ...); 80-byte rows were 2-way
template <bool ABF> struct VStage;  // one K step of staging registers (only the fields a variant uses: a union-style struct spills)
template <> struct VStage<false> { float4 a0, a1, a2, a3; uint4 b0, b1; float live; };
template <> struct VStage<true> { uint4 ab0, ab1; uint4 b0, b1; float live; };
template <int NJ, bool ABF>
__global__ __launch_bounds__(256) void k_vgemm_lds(VGemm g) {
    constexpr int BN = NJ * 32;
    __shared__ __attribute__((aligned(16))) __bf16 As[2][128 * VT_LD];
    __shared__ __attribute__((aligned(16))) __bf16 Bs[2][BN * VT_LD];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1, lr = lane & 15, kq = lane >> 4;
    const int m0 = blockIdx.y * 128, n0 = blockIdx.x * BN;
    const int cin = g.c.cin, nout = g.c.nout, kpt = cin >> 5, steps = g.c.ntap * kpt;
    const int ldr = tid >> 1, half = tid & 1;  // loader: row/col ldr, 16 k-elements at half*16
    const bool bload = ldr < BN;
    const float* xrow; const uint16_t* xrow16;
    {
        int m = m0 + ldr; if (m >= g.M) m = g.M - 1;
        const int s = m / g.T, t = m - s * g.T;
        const size_t eo = (size_t)s * g.x_stride + g.x_off + (size_t)t * cin + half * 16;
        xrow = g.x + eo; xrow16 = (const uint16_t*)g.x + eo;
    }
    const uint16_t* wrow;
    { int n = n0 + ldr; if (n >= nout) n = nout - 1; wrow = g.c.w + (size_t)n * cin + half * 16; }
    auto gload = [&](VStage<ABF>& r, int step_) {  // steps past the end re-read the last tile; their A part is zeroed
        const int step = min(step_, steps - 1);
        int tap, k0; vstep(g, step, kpt, tap, k0);
        const long sh = (long)(g.c.ntap - 1 - tap) * g.c.dil * cin - k0;
        if constexpr (ABF) { const uint4* p = (const uint4*)(xrow16 - sh); r.ab0 = p[0]; r.ab1 = p[1]; }
        else { const float* p = xrow - sh; r.a0 = ((const float4*)p)[0]; r.a1 = ((const float4*)p)[1]; r.a2 = ((const float4*)p)[2]; r.a3 = ((const float4*)p)[3]; }
        if (bload) { const uint16_t* q = wrow + (size_t)tap * nout * cin + k0; r.b0 = ((const uint4*)q)[0]; r.b1 = ((const uint4*)q)[1]; }
        r.live = step_ < steps ? 1.0f : 0.0f;
    };
    auto sstore = [&](const VStage<ABF>& r, int buf) {
        const float lv = r.live;
        __bf16* ap = &As[buf][ldr * VT_LD + half * 16];
        if constexpr (ABF) {
            uint4 z0 = r.ab0, z1 = r.ab1;
            if (lv == 0.0f) { z0 = make_uint4(0, 0, 0, 0); z1 = z0; }
            *(uint4*)ap = z0; *(uint4*)(ap + 8) = z1;
        } else {
            bf16x8 lo, hi;
            lo[0] = (__bf16)(r.a0.x * lv); lo[1] = (__bf16)(r.a0.y * lv); lo[2] = (__bf16)(r.a0.z * lv); lo[3] = (__bf16)(r.a0.w * lv);
            lo[4] = (__bf16)(r.a1.x * lv); lo[5] = (__bf16)(r.a1.y * lv); lo[6] = (__bf16)(r.a1.z * lv); lo[7] = (__bf16)(r.a1.w * lv);
            hi[0] = (__bf16)(r.a2.x * lv); hi[1] = (__bf16)(r.a2.y * lv); hi[2] = (__bf16)(r.a2.z * lv); hi[3] = (__bf16)(r.a2.w * lv);
            hi[4] = (__bf16)(r.a3.x * lv); hi[5] = (__bf16)(r.a3.y * lv); hi[6] = (__bf16)(r.a3.z * lv); hi[7] = (__bf16)(r.a3.w * lv);
            *(bf16x8*)ap = lo; *(bf16x8*)(ap + 8) = hi;
        }
        if (bload) { uint4* bp = (uint4*)&Bs[buf][ldr * VT_LD + half * 16]; bp[0] = r.b0; bp[1] = r.b1; }
    };
    f32x4 acc[4][NJ];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#define VT_COMPUTE(buf_)                                                                                              \
    do {                                                                                                              \
        bf16x8 a__[4], b__[NJ];                                                                                       \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) a__[i] = *(const bf16x8*)&As[buf_][(wm * 64 + i * 16 + lr) * VT_LD + kq * 8];     \
        _Pragma("unroll") for (int j = 0; j < NJ; ++j) b__[j] = *(const bf16x8*)&Bs[buf_][(wn * NJ * 16 + j * 16 + lr) * VT_LD + kq * 8]; \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                                 \
            _Pragma("unroll") for (int j = 0; j < NJ; ++j)                                                            \
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a__[i], b__[j], acc[i][j], 0, 0, 0);              \
    } while (0)
    // at the top of step k: LDS[k & 1] holds tile k, R0/R1 (alternating) hold tiles k+1 and k+2
    VStage<ABF> R0, R1;
    gload(R0, 0); sstore(R0, 0);
    gload(R0, 1);
    gload(R1, 2);
    __syncthreads();
    for (int step = 0; step < steps; step += 2) {  // an odd tail runs one zeroed tile: the loop body has no branch
        sstore(R0, 1);       // tile step+1 -> LDS[1] (last read before the previous barrier)
        gload(R0, step + 3);
        VT_COMPUTE(0);
        Q3_LDS_BARRIER();
        sstore(R1, 0);       // tile step+2 -> LDS[0]
        gload(R1, step + 4);
        VT_COMPUTE(1);
        Q3_LDS_BARRIER();
    }
    const bool rmw = g.epi == 1 || g.epi == 2;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float yv[4][NJ];  // residual operands of this row tile: one batch of loads, not a round trip per element
        if (rmw) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int m = min(m0 + wm * 64 + i * 16 + 4 * kq + e, g.M - 1);
                const int s = m / g.T, t = m - s * g.T;
                const float* yp = g.y + (size_t)s * g.y_stride + g.y_off + (size_t)t * nout;
#pragma unroll
                for (int j = 0; j < NJ; ++j) yv[e][j] = yp[min(n0 + wn * NJ * 16 + j * 16 + lr, nout - 1)];
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int m = m0 + wm * 64 + i * 16 + 4 * kq + e;
            if (m >= g.M) continue;
            const int s = m / g.T, t = m - s * g.T;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int n = n0 + wn * NJ * 16 + j * 16 + lr;
                if (n < nout) vepi(g, acc[i][j][e], s, t, n, rmw ? yv[e][j] : 0.0f);
            }
        }
    }
}

// Second half of the LDS-staged epilogues (k_vgemm_ring, k_vconv_tap): the f32 tile Ot[ROWS][BN + 4] goes out 4 consecutive columns of a row
// per item. A thread keeps ONE column group for all its rows (threads past the last whole row of a pass idle: 16 of 256 at BN = 96), so the
// per-column operands — bias, layer scale, the consumer's SnakeBeta parameters — are loaded once per thread, and a row's place in the output
// (slot, row of the slot: an integer division) is worked out once per row into rinfo[] = slot << 20 | row, -1 for a dead row, by the caller.
// Loaded / divided per item (4 + 4 + 4 scalar loads behind an integer modulo each, re-issued for every item because the stores in between may
// alias them) this loop was 75 % of a workgroup's life in the memory-bound layers: in-kernel stamps of the 96-column transposed convolution,
// 6.7 of 26 us up to the tile in LDS, 20 us here. LITE: epilogues 0 / 2 only (no GELU / layer-scale code in the instruction stream).
// Per element the arithmetic is vepi's.
template <int BN, int ROWS, int NTHR, bool LITE>
__device__ __forceinline__ void vepi_tile(const VGemm& g, const float* Ot, const int* rinfo, int n0) {
    constexpr int LDO = BN + 4, C4 = BN / 4, RPP = NTHR / C4, PER = (ROWS + RPP - 1) / RPP, PB = 8;
    const int tid = threadIdx.x, cg = tid % C4, r0 = tid / C4, nout = g.c.nout, n = n0 + cg * 4;
    const bool act = r0 < RPP && n < nout;  // (nout is a multiple of 4: checked by the launcher)
    const int epi = g.epi;
    const bool rmw = epi == 2 || (!LITE && epi == 1), has_b = g.c.b != nullptr, has_y2 = g.y2 != nullptr;
    float b4[4], sc4[4], ea4[4], ib4[4];
    {   // column n + q of a period-P operand: (n mod P) + q when P is a multiple of 4 (n is); no division at all when P covers every column
        const int nc = min(n, nout - 4);
        auto base = [&](int P) { return P >= nout ? nc : nc % P; };
        auto at = [&](const float* p, int P, int b0, int q) { return (P & 3) == 0 ? p[b0 + q] : p[(nc + q) % P]; };
        const int bb = has_b ? base(g.c.bias_n) : 0, bs = (!LITE && epi == 1) ? base(g.scale_n) : 0, bk = has_y2 ? base(g.snake_n) : 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            b4[q] = has_b ? at(g.c.b, g.c.bias_n, bb, q) : 0.0f;
            sc4[q] = (!LITE && epi == 1) ? at(g.scale, g.scale_n, bs, q) : 1.0f;
            ea4[q] = has_y2 ? at(g.ea, g.snake_n, bk, q) : 0.0f;
            ib4[q] = has_y2 ? at(g.ib, g.snake_n, bk, q) : 0.0f;
        }
    }
    const float* ybase = g.y + g.y_off + n;
    for (int u0 = 0; u0 < PER; u0 += PB) {
        float4 yo[PB]; int info[PB];
#pragma unroll
        for (int u = 0; u < PB; ++u) {
            const int row = r0 + (u0 + u) * RPP;
            info[u] = (act && u0 + u < PER && row < ROWS) ? rinfo[row] : -1;
            if (rmw && info[u] >= 0) yo[u] = *(const float4*)(ybase + (size_t)(info[u] >> 20) * g.y_stride + (size_t)(info[u] & 0xFFFFF) * nout);
        }
#pragma unroll
        for (int u = 0; u < PB; ++u) {
            if (info[u] < 0) continue;
            const int row = r0 + (u0 + u) * RPP, sl = info[u] >> 20, t = info[u] & 0xFFFFF;
            const float4 a4 = *(const float4*)&Ot[(size_t)row * LDO + cg * 4];
            float v[4] = {a4.x, a4.y, a4.z, a4.w};
            const float yv[4] = {rmw ? yo[u].x : 0.f, rmw ? yo[u].y : 0.f, rmw ? yo[u].z : 0.f, rmw ? yo[u].w : 0.f};
            float sv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (has_b) v[q] += b4[q];
                if (!LITE && epi == 1) v[q] = yv[q] + sc4[q] * v[q];
                else if (epi == 2) v[q] = yv[q] + v[q];
                else if (!LITE && epi == 3) v[q] = gelu_erf(v[q]);
                if (has_y2) {
                    const float sn = __sinf(v[q] * ea4[q]);
                    sv[q] = v[q] + ib4[q] * (sn * sn);
                }
            }
            if (g.store) *(float4*)(g.y + g.y_off + n + (size_t)sl * g.y_stride + (size_t)t * nout) = make_float4(v[0], v[1], v[2], v[3]);
            if (has_y2) {
                const size_t o2 = (size_t)sl * g.y2_stride + g.y2_off + (size_t)t * nout + n;
                if (g.y2_bf16) {
                    __bf16 hh[4] = {(__bf16)sv[0], (__bf16)sv[1], (__bf16)sv[2], (__bf16)sv[3]};
                    *(uint2*)((__bf16*)g.y2 + o2) = *(const uint2*)hh;
                } else *(float4*)(g.y2 + o2) = make_float4(sv[0], sv[1], sv[2], sv[3]);
            }
        }
    }
}

// The same GEMM for bf16 A operands with the operand tiles brought in by LDS-DMA (global_load_lds_dwordx4) into a ring of VR_NS stages:
// k_vgemm_lds keeps two K steps of lookahead in registers, which left it bound by the global-load latency (~0.7 us per 32-wide K step with
// one workgroup on a CU, 8 % MFMA issue rate on the decoder's input convolution); here three stages (48 KiB per workgroup, two workgroups
// per CU) are in flight while one is consumed and no operand passes through registers on its way to LDS. Tile 128 x (NJ*32), 2 x 2 waves of
// 64 x (NJ*16), one barrier per K step. A stage holds rows of 64 bytes (32 bf16) unpadded; the 16-byte chunk c of row r sits at chunk
// position c ^ VR_SW(r), VR_SW(r) = (4 - (r >> 2)) & 3. ds_read_b128 is served in four 16-lane groups that are NOT contiguous
// (MI355X_MICROARCH.md, LDS: {0-3, 12-15, 20-27}, ...): with lane = 16 kq + (row & 15) a group reads rows 0-3 and 12-15 at chunk kq and
// rows 4-11 at chunk kq ^ 1, and this XOR puts those sixteen 16-byte slots on sixteen different slots of the 256-byte bank row
// ((r >> 2) & 3 alone leaves every group 2-way conflicted). A DMA load writes lane l's 16 bytes at (wave-uniform base) + 16 l, so
// lane l FETCHES the chunk that belongs there: row l >> 2 of its 16-row group, chunk (l & 3) ^ VR_SW(row).
// Accumulation order per output element is unchanged (32-wide K steps ascending over taps, then channels).
// VR_NS stages (4: three in flight)
#ifdef Q3_VOC_STAMPS  // experiment builds (tools/r3_voc_stamps.sh): s_memrealtime (100 MHz) stamps of three workgroups of the NJ = 3 instance
__device__ unsigned long long g_ring_stamps[4][8];
#define VG_STAMP(i_) do { if (NJ == 3 && vg_wg >= 0 && threadIdx.x == 0) g_ring_stamps[vg_wg][i_] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define VG_STAMP(i_) do { } while (0)
#endif
template <int NJ, int VR_NS>
__global__ __launch_bounds__(256) void k_vgemm_ring(VGemm g) {
#ifdef Q3_VOC_STAMPS
    const int vg_wg = blockIdx.x == 1 && blockIdx.y == 100 ? 0 : (blockIdx.x == 0 && blockIdx.y == 700 ? 1 : (blockIdx.x == 2 && blockIdx.y == 1200 ? 2 : -1));
#endif
    VG_STAMP(0);
    constexpr int BN = NJ * 32, LA = 2, LB = (BN / 16 + 3) / 4, L = LA + LB;
    constexpr int STAGE = (128 + BN) * 64;  // bytes
    extern __shared__ __attribute__((aligned(16))) char ring[];  // [VR_NS][A 128 x 64 B | B BN x 64 B]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wm = wave >> 1, wn = wave & 1, lr = lane & 15, kq = lane >> 4;
    // Workgroups go to the 8 XCDs round-robin by linear id. Tile t = (column tile fastest) is given to XCD t / (tiles / 8): every XCD then
    // owns a contiguous band of row tiles with all their column tiles, so an A band is fetched into ONE L2 (not into as many as it has
    // column tiles) and the workgroups running together on an XCD walk the same few weight tiles.
    int bx = blockIdx.x, by = blockIdx.y;
    {
        const int total = gridDim.x * gridDim.y;
        if ((total & 7) == 0) {
            const int id = blockIdx.x + gridDim.x * blockIdx.y, t = (id & 7) * (total >> 3) + (id >> 3);
            bx = t % gridDim.x; by = t / gridDim.x;
        }
    }
    const int m0 = by * 128, n0 = bx * BN;
    const int cin = g.c.cin, nout = g.c.nout, kpt = cin >> 5, steps = g.c.ntap * kpt;
    // loader role: 16-row group gA (A) / gB (B), row lrow of the group, chunk fetched = (lane & 3) ^ ((row >> 2) & 3)
    const int lrow = lane >> 2;
    const uint16_t* arow[LA]; int achunk[LA];
#pragma unroll
    for (int i = 0; i < LA; ++i) {
        const int row = (wave * LA + i) * 16 + lrow;  // row of the 128-row tile
        int m = m0 + row; if (m >= g.M) m = g.M - 1;
        const int s = m / g.T, t = m - s * g.T;
        achunk[i] = ((lane & 3) ^ VR_SW(row)) * 8;
        arow[i] = (const uint16_t*)g.x + (size_t)s * g.x_stride + g.x_off + (size_t)t * cin + achunk[i];
    }
    const uint16_t* brow[LB]; int bgrp[LB];
#pragma unroll
    for (int i = 0; i < LB; ++i) {
        bgrp[i] = min(wave * LB + i, BN / 16 - 1);  // (a clamped duplicate re-writes the same bytes)
        const int row = bgrp[i] * 16 + lrow;
        int n = n0 + row; if (n >= nout) n = nout - 1;
        brow[i] = g.c.w + (size_t)n * cin + ((lane & 3) ^ VR_SW(row)) * 8;
    }
    auto issue = [&](int step) {
        int tap, k0; vstep(g, step, kpt, tap, k0);
        const long sh = (long)(g.c.ntap - 1 - tap) * g.c.dil * cin - k0;
        char* st = ring + (size_t)(step % VR_NS) * STAGE;
#pragma unroll
        for (int i = 0; i < LA; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(arow[i] - sh),
                                             (__attribute__((address_space(3))) void*)(st + (wave * LA + i) * 1024), 16, 0, 0);
        const size_t wo = (size_t)tap * nout * cin + k0;
#pragma unroll
        for (int i = 0; i < LB; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(brow[i] + wo),
                                             (__attribute__((address_space(3))) void*)(st + 128 * 64 + bgrp[i] * 1024), 16, 0, 0);
    };
    f32x4 acc[4][NJ];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // fragment addresses inside a stage: row R, chunk kq -> R * 64 + ((kq ^ VR_SW(R)) * 16)
    int aoff[4], boff[NJ];
#pragma unroll
    for (int i = 0; i < 4; ++i) { const int R = wm * 64 + i * 16 + lr; aoff[i] = R * 64 + ((kq ^ VR_SW(R)) << 4); }
#pragma unroll
    for (int j = 0; j < NJ; ++j) { const int R = wn * NJ * 16 + j * 16 + lr; boff[j] = 128 * 64 + R * 64 + ((kq ^ VR_SW(R)) << 4); }
#pragma unroll
    for (int p = 0; p < VR_NS - 1; ++p)
        if (p < steps) issue(p);
    VG_STAMP(1);
    for (int step = 0; step < steps; ++step) {
        // this wave's loads of stage `step` have landed when at most the later stages' loads are outstanding
        if (step + VR_NS - 2 < steps) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((VR_NS - 2) * L) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        Q3_LDS_BARRIER();  // every wave's part of the stage is in LDS, and every wave is done reading stage step - 1
        if (step == 0) VG_STAMP(2);
        if (step + VR_NS - 1 < steps) issue(step + VR_NS - 1);  // into the buffer stage step - 1 used
        const char* st = ring + (size_t)(step % VR_NS) * STAGE;
        bf16x8 a_[4], b_[NJ];
#pragma unroll
        for (int i = 0; i < 4; ++i) a_[i] = *(const bf16x8*)(st + aoff[i]);
#pragma unroll
        for (int j = 0; j < NJ; ++j) b_[j] = *(const bf16x8*)(st + boff[j]);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_[i], b_[j], acc[i][j], 0, 0, 0);
    }
    // Epilogue through LDS: the accumulators go to an f32 tile [128][BN + 4] (the ring is idle; the launch sizes the LDS for the larger
    // of the two), then every thread finishes 4 consecutive columns of a row per trip with 16-byte loads / stores of y (and 8-byte stores
    // of the bf16 snake output). The D layout itself gives 64-byte runs of 4-byte stores: on the memory-bound layers (1x1 convolutions,
    // the transposed convolutions) issuing those was a large part of the kernel. Per element the arithmetic is vepi's.
    constexpr int LDO = BN + 4;
    float* Ot = (float*)ring;
    VG_STAMP(3);
    __syncthreads();  // every wave is done with the last stage
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) Ot[(size_t)(wm * 64 + i * 16 + 4 * kq + e) * LDO + wn * NJ * 16 + j * 16 + lr] = acc[i][j][e];
    int* rinfo = (int*)(Ot + 128 * LDO);
    if (tid < 128) { const int m = m0 + tid, sl = m / g.T; rinfo[tid] = m < g.M ? (sl << 20) | (m - sl * g.T) : -1; }
    __syncthreads();
    VG_STAMP(4);
    if (g.epi == 0 || g.epi == 2) vepi_tile<BN, 128, 256, true>(g, Ot, rinfo, n0);
    else vepi_tile<BN, 128, 256, false>(g, Ot, rinfo, n0);
    VG_STAMP(5);
#ifdef Q3_VOC_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    VG_STAMP(6);
#endif
}
template <int NJ, int NS>
static void launch_vgemm_ring_t(hipStream_t s, const VGemm& g, dim3 grid) {
    constexpr size_t lds_ring = (size_t)NS * (128 + NJ * 32) * 64, lds_out = (size_t)128 * (NJ * 32 + 4) * 4 + 128 * 4, lds = lds_ring > lds_out ? lds_ring : lds_out;
    static Q3PerDevice pd;
    const size_t ldsp = voc_lds_floor(lds);
    pd.ensure(ldsp, [&]() { hipFuncSetAttribute((const void*)k_vgemm_ring<NJ, NS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsp); });
    hipLaunchKernelGGL((k_vgemm_ring<NJ, NS>), grid, dim3(256), ldsp, s, g);
}
template <int NJ>
static void launch_vgemm_ring(hipStream_t s, const VGemm& g, dim3 grid) {
    launch_vgemm_ring_t<NJ, 4>(s, g, grid);  // (six stages for the one-workgroup-per-CU launches beside the decoder: 107 against 104 us — depth is not their limit)
}

// The wide 7-tap convolutions of the decoder blocks (768 / 384 channels: 40 % of a batched call) on a tile that keeps its INPUT ROWS in LDS
// across the taps. k_vgemm_ring treats a tap as 32-wide K steps with a shifted row pointer, so every step brings 128 input rows AND 128
// weight rows through the L2 -> LDS path: 64 flop per byte, and that path (66-73 GB/s per CU measured, MI355X_MICROARCH.md "gather into
// LDS") together with its ~2 us latency under load — not the MFMA (22 % busy), not the LDS — set its time. Here a workgroup of 8 waves owns
// TWO 128-row sub-tiles x 128 columns:
//  * a 32-channel chunk of each sub-tile's rows plus its 6 * dil halo rows lands in LDS once (two image buffers, LDS-DMA) and serves all
//    seven taps — a tap is a row offset into the same image;
//  * the weights of one (chunk, tap) — 128 columns x 32 channels — are an 8 KiB stage of a twelve-stage LDS-DMA ring shared by both
//    sub-tiles: ten stages (80 KiB) in flight, because a stage takes ~2 us to arrive and an MFMA-bound step lasts 0.2 us.
// Per 2 x 128 x 128 x 32 x 7 MACs that is 7 x 8 KiB of weights + 2 x (128 + halo) x 64 B of rows: ~190 flop per byte at dil = 9.
// One barrier per PAIR of steps (32 MFMAs per wave between barriers). Rows are 64 bytes unpadded; the 16-byte chunk c of row r sits at
// chunk position c ^ ((r >> 1) & 3): conflict-free for ds_read_b128's 16-lane groups at ANY row offset (the taps shift the rows by
// tap * dil; searched exhaustively over the offsets).
// K order: vstep(chunked) — the register-staged kernels follow the same order for these shapes, so the choice of kernel never changes a bit.
// A sub-tile never straddles two slots (its halo is the slot's own history): sub-tile q = (slot q / tps, rows (q % tps) * 128 ...).
// Schedule: 14 steps = 2 chunks = 7 pairs per trip of the main loop. Chunk c (even) reads image buffer 0, chunk c + 1 buffer 1. After the
// barrier of pair j a wave issues, in this order: [3 row pieces of image c + 1 at j = 0 (buffer 1 was last read in the previous trip's pair 6);
// 3 row pieces of image c + 2 at j = 4 (buffer 0 was last read in pair 3)], then its 2 weight pieces of pair p + 5. LDS-DMA completes in
// issue order (vmcnt), so before pair j's barrier a wave waits until at most N(j) of its operations are outstanding:
//   the stages of pair p were issued five pairs ago, younger are the issues of the last four pairs: 4 x 2 + 3 per image in them;
//   pair 3 reads image c + 1 (issued at pair 0) and pair 0 image c + 2 (issued at pair 4 of the trip before): younger are 3 x 2.
// Two forms: NSUB = 2 (8 waves, 12 stages, 5 pairs in flight: the call has the GPU to itself) and NSUB = 1 (4 waves on ONE 128-row sub-tile,
// 8 stages, 3 pairs in flight, 82-90 KiB of LDS: the "polite" form beside the decoder — one workgroup per CU by its LDS, 72 KiB and 336 VGPRs
// per SIMD left for the decoder's workgroups). Per pair a wave issues [3 row pieces at pairs 0 and 4] then 2 BP weight pieces (BP = 8 / waves).
template <int N> __device__ __forceinline__ void vc_waitcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// operations a wave may leave outstanding before pair j's barrier: the issues of the last LA - 1 pairs (this pair's stages went out LA pairs ago,
// last in their group); pairs 3 and 0 also start an image issued three pairs earlier, FIRST in its group: everything after those pieces
constexpr int vc_cnt(int q, int BP) { return 2 * BP + ((((q % 7) + 7) % 7 == 0 || ((q % 7) + 7) % 7 == 4) ? 3 : 0); }
constexpr int vc_wait(int j, int LA, int BP) {
    int n = 0;
    for (int q = j - LA + 1; q <= j - 1; ++q) n += vc_cnt(q, BP);
    const int special = 6 * BP;  // the stages of the image's own group and of the two groups after it
    return ((j == 0 || j == 3) && special < n) ? special : n;
}
template <int NTAP, int NSUB>
__global__ __launch_bounds__(256 * NSUB) void k_vconv_tap(VGemm g) {
    static_assert(NTAP == 7, "the pair schedule below is written for seven taps");
    constexpr int BN = 128, NW = 4 * NSUB, BP = 8 / NW, NS = NSUB == 2 ? 12 : 8, LA = NS / 2 - 1;
    extern __shared__ __attribute__((aligned(16))) char tlds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 15, kq = lane >> 4;
    const int sub = wave >> 2, wm = (wave >> 1) & 1, wn = wave & 1;
    int bx = blockIdx.x, by = blockIdx.y;
    {   // XCD-aware tile order (see k_vgemm_ring)
        const int total = gridDim.x * gridDim.y;
        if ((total & 7) == 0) {
            const int id = blockIdx.x + gridDim.x * blockIdx.y, t = (id & 7) * (total >> 3) + (id >> 3);
            bx = t % gridDim.x; by = t / gridDim.x;
        }
    }
    const int cin = g.c.cin, nout = g.c.nout, dil = g.c.dil, halo = (NTAP - 1) * dil, n0 = bx * BN;
    const int RA = (128 + halo + 15) & ~15, pps = RA >> 4;           // rows / 16-row pieces of one sub-tile's image
    const int abytes = NSUB * RA * 64;                               // one image buffer (all sub-tiles)
    char* const Bring = tlds + 2 * abytes;                           // [NS][128 columns][64 B]
    const int tps = (g.T + 127) >> 7, nsub = (g.M / g.T) * tps, nchunk = cin >> 5, steps = nchunk * NTAP;
    // loader roles: a piece is 16 rows x 64 B (one wave-instruction); lane l fetches row l >> 2 of the piece, the 16-byte chunk that belongs
    // at position l & 3 of that row: (l & 3) ^ ((row >> 1) & 3) = (l & 3) ^ ((l >> 3) & 3) (pieces start at multiples of 16 rows).
    // Rows: piece pi = wave + NW k (k < 3) of the NSUB * pps <= 3 NW pieces.
    const int lrow = lane >> 2, lchunk = ((lane & 3) ^ ((lane >> 3) & 3)) * 8;
    const uint16_t* ap[3]; int adst[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int pi = min(wave + NW * k, NSUB * pps - 1);           // (a clamped duplicate re-writes the same bytes)
        const int sb = pi / pps, r16 = pi - sb * pps;
        const int q = min(by * NSUB + sb, nsub - 1), sl = q / tps, t0 = (q - sl * tps) << 7;
        const int t = min(t0 - halo + r16 * 16 + lrow, g.T - 1);     // rows past the call's last row are never part of a stored result
        ap[k] = (const uint16_t*)g.x + (size_t)sl * g.x_stride + g.x_off + (long)t * cin + lchunk;
        adst[k] = (sb * RA + r16 * 16) * 64;
    }
    // weights: wave w brings columns 16 BP w .. 16 BP (w + 1) - 1 of a stage (BP pieces)
    const uint16_t* bp[BP];
#pragma unroll
    for (int h = 0; h < BP; ++h) bp[h] = g.c.w + (size_t)min(n0 + (wave * BP + h) * 16 + lrow, nout - 1) * cin + lchunk;
    auto issue_a = [&](int chunk) {
        const int c = min(chunk, nchunk - 1);                        // past the end: the last chunk again, into the idle buffer
#pragma unroll
        for (int k = 0; k < 3; ++k)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ap[k] + c * 32),
                                             (__attribute__((address_space(3))) void*)(tlds + (chunk & 1) * abytes + adst[k]), 16, 0, 0);
    };
    auto issue_b = [&](int step) {
        const int st = min(step, steps - 1), ch = st / NTAP, tap = st - ch * NTAP;
#pragma unroll
        for (int h = 0; h < BP; ++h)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bp[h] + (size_t)tap * nout * cin + ch * 32),
                                             (__attribute__((address_space(3))) void*)(Bring + (step % NS) * (BN * 64) + (wave * BP + h) * 16 * 64), 16, 0, 0);
    };
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int arow0 = sub * RA + wm * 64 + lr;                       // image row of this lane's fragment row at tap 0, i = 0
    int boff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int R = wn * 64 + j * 16 + lr; boff[j] = R * 64 + ((kq ^ ((R >> 1) & 3)) << 4); }
    issue_a(0);
#pragma unroll
    for (int p = 0; p < LA; ++p) { issue_b(2 * p); issue_b(2 * p + 1); }
    for (int c0 = 0; c0 < nchunk; c0 += 2) {                         // (cin % 64 == 0: an even number of chunks)
#pragma unroll
        for (int j = 0; j < 7; ++j) {                                // pair j of the trip: steps 14 * (c0 / 2) + 2 j, + 1
            if (j == 0) vc_waitcnt<vc_wait(0, LA, BP)>();
            else if (j == 1) vc_waitcnt<vc_wait(1, LA, BP)>();
            else if (j == 2) vc_waitcnt<vc_wait(2, LA, BP)>();
            else if (j == 3) vc_waitcnt<vc_wait(3, LA, BP)>();
            else if (j == 4) vc_waitcnt<vc_wait(4, LA, BP)>();
            else if (j == 5) vc_waitcnt<vc_wait(5, LA, BP)>();
            else vc_waitcnt<vc_wait(6, LA, BP)>();
            Q3_LDS_BARRIER();  // this pair's stages (and the image it starts) are in LDS for every wave; everyone is done with the previous pair
            const int s0 = c0 * NTAP + 2 * j;
            if (j == 0) issue_a(c0 + 1);
            if (j == 4) issue_a(c0 + 2);
            issue_b(s0 + 2 * LA); issue_b(s0 + 2 * LA + 1);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int sl = 2 * j + h, cc = sl >= NTAP ? 1 : 0, tap = sl - cc * NTAP;   // step within the trip -> (chunk parity, tap)
                const char* img = tlds + cc * abytes;
                const char* st = Bring + ((s0 + h) % NS) * (BN * 64);
                bf16x8 a_[4], b_[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) { const int R = arow0 + i * 16 + tap * dil; a_[i] = *(const bf16x8*)(img + R * 64 + ((kq ^ ((R >> 1) & 3)) << 4)); }
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) b_[jj] = *(const bf16x8*)(st + boff[jj]);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj) acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_[i], b_[jj], acc[i][jj], 0, 0, 0);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the clamped loads past the end still write LDS
    // Epilogue through LDS, as in k_vgemm_ring: f32 tile [128 NSUB][BN + 4], then 4 consecutive columns of a row per item.
    constexpr int LDO = BN + 4, ROWS = 128 * NSUB;
    float* Ot = (float*)tlds;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) Ot[(size_t)(sub * 128 + wm * 64 + i * 16 + 4 * kq + e) * LDO + wn * 64 + j * 16 + lr] = acc[i][j][e];
    int* rinfo = (int*)(Ot + ROWS * LDO);
    if (tid < ROWS) {
        const int q = by * NSUB + (tid >> 7), sl = q / tps, t = ((q - sl * tps) << 7) + (tid & 127);
        rinfo[tid] = (q < nsub && t < g.T) ? (sl << 20) | t : -1;
    }
    __syncthreads();
    if (g.epi == 0 || g.epi == 2) vepi_tile<BN, ROWS, 64 * NW, true>(g, Ot, rinfo, n0);
    else vepi_tile<BN, ROWS, 64 * NW, false>(g, Ot, rinfo, n0);
}
template <int NSUB>
static void launch_vconv_tap(hipStream_t s, const VGemm& g) {
    constexpr int NS = NSUB == 2 ? 12 : 8;
    const int halo = (g.c.ntap - 1) * g.c.dil, RA = (128 + halo + 15) & ~15;
    const size_t lds_main = (size_t)2 * NSUB * RA * 64 + (size_t)NS * 128 * 64, lds_out = (size_t)128 * NSUB * (128 + 4) * 4 + 128 * NSUB * 4;
    const size_t lds = voc_lds_floor(std::max(lds_main, lds_out));
    static Q3PerDevice pd;
    pd.ensure(1, []() { hipFuncSetAttribute((const void*)k_vconv_tap<7, NSUB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024); });
    const int tps = (g.T + 127) / 128, nsub = (g.M / g.T) * tps;
    hipLaunchKernelGGL((k_vconv_tap<7, NSUB>), dim3(g.c.nout / 128, (nsub + NSUB - 1) / NSUB), dim3(256 * NSUB), lds, s, g);
}

// Fused residual unit of the narrow decoder blocks (C <= 192 channels, where everything is HBM-bound):
//   o += conv1x1(snake2(conv7_dil(xin)))  and  next_in = snake_next(o)
// in ONE pass: the workgroup stages its R + 6*dil input rows in LDS as bf16 once (the seven taps read LDS, not HBM),
// streams the weight chunks through a double-buffered LDS ring, keeps the intermediate (snake2 output) as a bf16 tile
// in LDS for the 1x1 convolution and finishes with the residual read-modify-write. HBM traffic per unit: read xin,
// read/write o, write next_in: the un-fused chain moved the activations three more times. Arithmetic per output element
// is the same as k_vgemm_lds + vepi (32-wide K steps in order, bf16 operands, f32 accumulate).
struct VResUnit {
    const float* xin; size_t xin_stride;  // bf16 work buffer [H + T][C] per slot (stride in elements), H = 6*dil history rows in front
    int T, dil;
    const uint16_t *w1, *w2; const float *b1, *b2;  // conv k7 [7][C][C], conv k1 [C][C]
    const float *ea2, *ib2;               // snake between the convolutions
    float* o; size_t o_stride; int store_o;
    float* y2; size_t y2_stride; int y2_off; const float *ea3, *ib3;  // snake of the consumer, written (bf16) into its work buffer
};
#if defined(Q3_STAMPS) || defined(Q3_VOC_STAMPS)
__device__ unsigned long long g_voc_stamps[2][4][16];  // experiment builds: s_memrealtime (100 MHz) stamps of three workgroups' wave 0, [192-channel unit, others]
#define VR_STAMP(i_) do { if (stamp_wg >= 0 && threadIdx.x == 0) g_voc_stamps[NT == 12 ? 0 : 1][stamp_wg][i_] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define VR_STAMP(i_) do { } while (0)
#endif
// (the step barrier stays __syncthreads(): with the LDS-only barrier the 96-channel unit measured 183 us instead of 137, the 192-channel one
//  unchanged — the weight loads it lets run ahead share the CU's in-order return path with the other workgroups' HBM phases)
#define VR_BARRIER() __syncthreads()
// NWV waves: 4, or 8 for the 192-channel unit — two 64-row halves behind ONE weight stream: with two 4-wave workgroups of 64 rows on a CU the
// weight chunks (12 KiB per step each) ran at 49 GB/s per CU through the L2 path that tops out at 66-73 (stamps: 0.49 us per step, 40 % MFMA)
template <int NT, int MT, int NWV>
__global__ __launch_bounds__(64 * NWV) void k_voc_resunit(VResUnit g) {
#if defined(Q3_STAMPS) || defined(Q3_VOC_STAMPS)
    const int stamp_wg = (blockIdx.y == 5 && (blockIdx.x == 3 || blockIdx.x == 20)) ? (blockIdx.x == 3 ? 0 : 1) : ((blockIdx.y == 40 && blockIdx.x == 11) ? 2 : -1);
#endif
    VR_STAMP(0);
    constexpr int C = NT * 16, KS = C / 32, R = 64 * MT, LDA = C + 16, S1 = 7 * KS, S = 8 * KS;  // row stride 2 C + 32 bytes = 32 mod 64: conflict-free fragment reads
    constexpr int NTHR = 64 * NWV, BROWS = NTHR / 4, BPASS = (C + BROWS - 1) / BROWS;  // weight-chunk loader passes: BROWS rows x 64 B per pass
    extern __shared__ __attribute__((aligned(16))) __bf16 lds[];
    const int halo = 6 * g.dil;
    __bf16* At = lds;                                  // [R + halo][LDA]
    __bf16* Zt = lds;                                  // [R][LDA]: reuses the input tile once conv1 has consumed it
    __bf16* Bs = At + (size_t)(R + halo) * LDA;         // [2][C][32], 16-byte chunk c of row n at chunk position c ^ VR_SW(n) (see k_vgemm_ring)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 15, kq = lane >> 4;
    const int sidx = blockIdx.y, t0 = blockIdx.x * R, T = g.T;
    // weight chunk pipeline: chunk c < S1 is (tap, ks) of conv1, chunk S1 + ks is conv2. Four register sets hold the chunks k + 1 .. k + 4
    // while step k computes from LDS: with one set (a chunk requested one step before its LDS copy) every step waited ~0.35 us for its
    // weights to arrive from L2 and the kernel sat at 16 % of the MFMA rate whatever the LDS layout.
    const int bn = tid >> 2, bpart = tid & 3;
    // (named scalars: an indexed array lands in scratch here)
#define VR_DECL(X_) uint4 X_##0, X_##1, X_##2; X_##0 = X_##1 = X_##2 = make_uint4(0, 0, 0, 0)
    VR_DECL(ra); VR_DECL(rb); VR_DECL(rc); VR_DECL(rd);
#define VR_GLOADB(X_, step_)                                                                                            \
    do {                                                                                                                \
        const int st__ = min((step_), S - 1);                                                                           \
        const uint16_t* base__ = (st__ < S1 ? g.w1 + (size_t)(st__ / KS) * C * C + (st__ % KS) * 32 : g.w2 + (st__ - S1) * 32) + bpart * 8; \
        X_##0 = *(const uint4*)(base__ + (size_t)min(bn, C - 1) * C);                                                    \
        if (BPASS > 1) X_##1 = *(const uint4*)(base__ + (size_t)min(bn + BROWS, C - 1) * C);                                \
        if (BPASS > 2) X_##2 = *(const uint4*)(base__ + (size_t)min(bn + 2 * BROWS, C - 1) * C);                               \
    } while (0)
#define VR_SSTOREB(X_, buf_)                                                                                            \
    do {                                                                                                                \
        __bf16* d__ = &Bs[((size_t)(buf_) * C + bn) * 32 + ((bpart ^ VR_SW(bn)) << 3)];  /* VR_SW(bn + 64 k) = VR_SW(bn) */     \
        if (bn < C) *(uint4*)d__ = X_##0;                                                                               \
        if (BPASS > 1 && bn + BROWS < C) *(uint4*)(d__ + BROWS * 32) = X_##1;                                                 \
        if (BPASS > 2 && bn + 2 * BROWS < C) *(uint4*)(d__ + 2 * BROWS * 32) = X_##2;                                               \
    } while (0)
    VR_GLOADB(ra, 0);
    VR_STAMP(1);
    // stage the input rows (bf16 work buffer), rows past T are zero: 16 bytes per lane, up to 12 loads in flight per thread (the whole
    // tile in one round trip; 8-byte loads 8 at a time took three and were 23 % of the workgroup's time)
    {
        const uint16_t* xp = (const uint16_t*)g.xin + (size_t)sidx * g.xin_stride + (size_t)t0 * C;  // buffer row t0 = output row t0 - halo
        const int nrow = R + halo, c8 = C / 8, total = nrow * c8;
        for (int base = tid; base < total; base += 12 * NTHR) {
            uint4 v[12];
#pragma unroll
            for (int u = 0; u < 12; ++u) {
                const int i = min(base + u * NTHR, total - 1), r = i / c8, c = (i - r * c8) * 8;
                v[u] = *(const uint4*)(xp + (size_t)min(r, T + halo - 1 - t0) * C + c);
            }
#pragma unroll
            for (int u = 0; u < 12; ++u) {
                const int i = base + u * NTHR;
                if (i < total) {
                    const int r = i / c8, c = (i - r * c8) * 8;
                    const bool live = t0 + r < T + halo;
                    *(uint4*)(At + (size_t)r * LDA + c) = live ? v[u] : make_uint4(0, 0, 0, 0);
                }
            }
        }
    }
    VR_SSTOREB(ra, 0);
    // at the top of step k: LDS[k & 1] = chunk k, set (k + j) % LK = chunk k + j, j = 1..LK. LK = 4 sets, 3 for the 192-channel block whose
    // sets are three registers wide (a fourth would push the wave past 256 registers and halve the occupancy)
    constexpr int LK = BPASS > 2 ? 3 : 4;
    VR_GLOADB(rb, 1); VR_GLOADB(rc, 2);
    if (LK == 4) { VR_GLOADB(rd, 3); VR_GLOADB(ra, 4); } else { VR_GLOADB(ra, 3); }
    __syncthreads();
    VR_STAMP(2);
    // wave tiling: a wave owns MTW row tiles x NTW column tiles. The wide blocks (one row tile per wave, >= 8 column tiles) split the
    // columns over wave pairs instead — 2 row tiles x NT/2 column tiles per wave: 2 + NT/2 fragment reads per K step instead of 1 + NT
    // for the same MFMAs (at 192 channels the LDS reads, 13 KiB per wave-step for 12 MFMAs, were what bounded the loop)
    constexpr int WN = ((MT == 1 || NWV == 8) && NT >= 8) ? 2 : 1, MTW = (R / 16) / (NWV / WN), NTW = NT / WN;
    f32x4 acc[MTW][NTW];
#pragma unroll
    for (int i = 0; i < MTW; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int wrow0 = (wave / WN) * MTW * 16, ncol0 = (wave % WN) * NTW * 16;
    // between the convolutions: snake2(conv1 + bias) -> bf16 tile (D layout: lane holds rows 4*kq+e, column lr of every tile)
#define VR_BETWEEN()                                                                                                    \
    do {                                                                                                                \
        _Pragma("unroll") for (int i = 0; i < MTW; ++i)                                                                  \
            _Pragma("unroll") for (int j = 0; j < NTW; ++j) {                                                            \
                const int n = ncol0 + j * 16 + lr;                                                                      \
                const float bb = g.b1[n], ea = g.ea2[n], ib = g.ib2[n];                                                 \
                _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                         \
                    const float v = acc[i][j][e] + bb;                                                                  \
                    const float sn = __sinf(v * ea);                                                                    \
                    Zt[(size_t)(wrow0 + i * 16 + 4 * kq + e) * LDA + n] = (__bf16)(v + ib * (sn * sn));                 \
                    acc[i][j][e] = 0.0f;                                                                                \
                }                                                                                                       \
            }                                                                                                           \
        VR_BARRIER();                                                                                                \
    } while (0)
#define VR_STEP(step_, X_)  /* X_ = the set holding chunk step_ + 1 */                                                  \
    do {                                                                                                                \
        const int sp__ = (step_), buf__ = sp__ & 1, tap__ = sp__ / KS, koff__ = (sp__ - tap__ * KS) * 32;               \
        const int rowoff__ = sp__ < S1 ? tap__ * g.dil : 0;  /* conv2 (chunks S1..) reads the snake tile in place */    \
        bf16x8 a__[MTW], b__[NTW];                                                                                      \
        _Pragma("unroll") for (int i = 0; i < MTW; ++i)                                                                  \
            a__[i] = *(const bf16x8*)&At[(size_t)(wrow0 + i * 16 + lr + rowoff__) * LDA + koff__ + kq * 8];             \
        _Pragma("unroll") for (int j = 0; j < NTW; ++j)                                                                  \
            b__[j] = *(const bf16x8*)&Bs[((size_t)buf__ * C + ncol0 + j * 16 + lr) * 32 + ((kq ^ VR_SW(lr)) << 3)];     \
        _Pragma("unroll") for (int i = 0; i < MTW; ++i)                                                                  \
            _Pragma("unroll") for (int j = 0; j < NTW; ++j)                                                              \
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a__[i], b__[j], acc[i][j], 0, 0, 0);                \
        VR_SSTOREB(X_, buf__ ^ 1);                                                                                      \
        VR_GLOADB(X_, sp__ + 1 + LK);                                                                                        \
        VR_BARRIER();                                                                                                \
    } while (0)
    static_assert(S % LK == 0, "the step loop is unrolled by the number of register sets");
    for (int step0 = 0; step0 < S; step0 += LK) {  // S = 8 KS; S1 = 7 KS falls on sub-step S1 % LK of its trip
#define VR_SUB(u_, X_)                                                                                                  \
        if ((S1 % LK) == (u_) && step0 + (u_) == S1) {                                                                  \
            VR_STAMP(3); VR_BETWEEN(); VR_STAMP(4);                                                                     \
        }                                                                                                               \
        VR_STEP(step0 + (u_), X_)
        if (LK == 4) { VR_SUB(0, rb); VR_SUB(1, rc); VR_SUB(2, rd); VR_SUB(3, ra); }
        else { VR_SUB(0, rb); VR_SUB(1, rc); VR_SUB(2, ra); }
    }
#undef VR_SUB
#undef VR_BETWEEN
    VR_STAMP(5);
#undef VR_STEP
#undef VR_GLOADB
#undef VR_SSTOREB
#undef VR_DECL
    // Epilogue through LDS: conv2 + bias goes to an f32 tile [R][C + 4] (the input tile and the weight ring are dead: the last step ended
    // with a barrier), then every thread finishes 4 consecutive channels of a row at a time: o (16-byte load) + tile -> o (16-byte store)
    // and snake_next(o) -> bf16 x 4 (8-byte store). Straight from the D layout it was 96 four- and two-byte stores and 48 four-byte
    // loads per thread, 64-byte runs each: 38 % of the workgroup's time went into issuing them.
    constexpr int LDO = C + 4;
    float* Ot = (float*)lds;
#pragma unroll
    for (int i = 0; i < MTW; ++i)
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const int n = ncol0 + j * 16 + lr;
            const float bb = g.b2[n];
#pragma unroll
            for (int e = 0; e < 4; ++e) Ot[(size_t)(wrow0 + i * 16 + 4 * kq + e) * LDO + n] = acc[i][j][e] + bb;
        }
    __syncthreads();
    {   // a thread keeps one group of 4 channels for all its rows (16 of 256 threads idle at 96 / 192 channels): the consumer's SnakeBeta
        // parameters are loaded once, not per item
        constexpr int c4n = C / 4, RPP = NTHR / c4n, PER = (R + RPP - 1) / RPP;
        float* ob = g.o + (size_t)sidx * g.o_stride;
        __bf16* yb = (__bf16*)g.y2 + (size_t)sidx * g.y2_stride + g.y2_off;
        const int cg = tid % c4n, r0 = tid / c4n, c = cg * 4;
        const bool act = r0 < RPP;
        const float4 ea = *(const float4*)(g.ea3 + c), ib = *(const float4*)(g.ib3 + c);
        constexpr int PB = PER < 13 ? PER : 13;  // residual loads in flight per thread: all of them for every instantiated shape
        for (int p0 = 0; p0 < PER; p0 += PB) {
            float4 ov[PB];
#pragma unroll
            for (int u = 0; u < PB; ++u) {
                const int row = min(r0 + (p0 + u) * RPP, R - 1);
                ov[u] = *(const float4*)(ob + (size_t)min(t0 + row, T - 1) * C + c);
            }
#pragma unroll
            for (int u = 0; u < PB; ++u) {
                const int row = r0 + (p0 + u) * RPP, t = t0 + row;
                if (!act || p0 + u >= PER || row >= R || t >= T) continue;
                const float4 a = *(const float4*)&Ot[(size_t)row * LDO + c];
                float4 v; v.x = ov[u].x + a.x; v.y = ov[u].y + a.y; v.z = ov[u].z + a.z; v.w = ov[u].w + a.w;
                if (g.store_o) *(float4*)(ob + (size_t)t * C + c) = v;
                float sn; __bf16 h[4];
                sn = __sinf(v.x * ea.x); h[0] = (__bf16)(v.x + ib.x * (sn * sn));
                sn = __sinf(v.y * ea.y); h[1] = (__bf16)(v.y + ib.y * (sn * sn));
                sn = __sinf(v.z * ea.z); h[2] = (__bf16)(v.z + ib.z * (sn * sn));
                sn = __sinf(v.w * ea.w); h[3] = (__bf16)(v.w + ib.w * (sn * sn));
                *(uint2*)(yb + (size_t)t * C + c) = *(const uint2*)h;
            }
        }
    }
    VR_STAMP(6);
}

__global__ void k_voc_embed(const VCall* __restrict__ clp, const int* codes, int max_steps_cap, int ncb_model, const float* const* cb, int ncb, int cbs, int cd,
                            float* out, size_t out_stride, int out_off) {
    const VCall& cl = *clp;
    const int s = blockIdx.y, t = blockIdx.x;
    const int slot = cl.slot[s], frame = min(cl.pos[s] + t, max_steps_cap - 1);  // (padding frames may point past the last row)
    const int* cp = codes + ((size_t)slot * max_steps_cap + frame) * ncb_model;
    // codes and table bases first, then every row element in ONE round of loads, then the sum in codebook order (a loop of code -> table
    // pointer -> element paid three dependent round trips per codebook: 32 us for 16 MB)
    constexpr int QB = 16;
    for (int q0 = 0; q0 < ncb; q0 += QB) {
        const float* row[QB];
#pragma unroll
        for (int u = 0; u < QB; ++u) {
            const int q = min(q0 + u, ncb - 1);
            int code = cp[q];
            code = code < 0 ? 0 : (code >= cbs ? cbs - 1 : code);  // clamp [0, 2047]: src/tts/engine.rs:515-519
            row[u] = cb[q] + (size_t)code * cd;
        }
        for (int i = threadIdx.x; i < cd; i += blockDim.x) {
            float v[QB];
#pragma unroll
            for (int u = 0; u < QB; ++u) v[u] = row[u][i];
            float* op = out + (size_t)s * out_stride + out_off + (size_t)t * cd + i;
            float acc = q0 ? *op : 0.0f;
#pragma unroll
            for (int u = 0; u < QB; ++u) if (q0 + u < ncb) acc += v[u];
            *op = acc;
        }
    }
}

// f32 rows of a slot -> the bf16 they would be rounded to by the consuming GEMM, into its work buffer (n a multiple of 4)
__global__ void k_voc_rows_bf16(const float* src, size_t src_stride, uint16_t* dst, size_t dst_stride, int n) {
    const int s = blockIdx.y, i = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    const float4 v = *(const float4*)(src + (size_t)s * src_stride + i);
    __bf16 h[4] = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
    *(uint2*)(dst + (size_t)s * dst_stride + i) = *(const uint2*)h;
}

// history rows: work[s][0:H] <- hist[slot]  (load)   /   hist[slot] <- work[s][T : T+H]  (save)
template <class E>
__global__ void k_voc_hist(const VCall* __restrict__ clp, E* work, size_t stride, E* hist, int H, int C, int T, int save) {
    const VCall& cl = *clp;
    const int s = blockIdx.y, slot = cl.slot[s];
    const size_t n = (size_t)H * C;
    E* w = work + (size_t)s * stride + (save ? (size_t)T * C : 0);
    E* h = hist + (size_t)slot * n;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        if (save) h[i] = w[i]; else w[i] = h[i];
    }
}

// All history loads of a call in ONE launch at its start (work[s][0:H] <- hist[slot]) and all saves in ONE at its end
// (hist[slot] <- work[s][T : T+H]): 42 launches of ~5 us per batched call became 4. grid (chunks, ns, entries); blocks are dwords.
struct VHistTab {
    int n;
    struct Ent { char* work; char* hist; unsigned long long stride_b, block_b, tail_b; } e[24];  // bytes: per-slot stride, H*C block, offset T*C
};
__global__ void k_voc_hist_all(const VCall* __restrict__ clp, VHistTab tab, int save) {
    const VCall& cl = *clp;
    const VHistTab::Ent z = tab.e[blockIdx.z];
    const int s = blockIdx.y, slot = cl.slot[s];
    uint32_t* w = (uint32_t*)(z.work + (size_t)s * z.stride_b + (save ? z.tail_b : 0));
    uint32_t* h = (uint32_t*)(z.hist + (size_t)slot * z.block_b);
    const size_t n = z.block_b / 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        if (save) h[i] = w[i]; else w[i] = h[i];
    }
}

// out_bf16: 0 f32 rows, 1 bf16 rows, 2 bf16 in the decoder GEMM's A-tiled layout (q3_atile_off)
__global__ __launch_bounds__(64) void k_voc_rmsnorm(const float* x, const float* w, float eps, int d, float* y, int out_bf16) {
    const int r = blockIdx.x, lane = threadIdx.x;
    const float* xr = x + (size_t)r * d;
    float ss = 0.0f;
    for (int i0 = lane; i0 < d; i0 += 8 * 64) {  // 8 loads in flight per trip, same ascending per-lane chain
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = xr[min(i0 + u * 64, d - 1)];
#pragma unroll
        for (int u = 0; u < 8; ++u) if (i0 + u * 64 < d) ss += v[u] * v[u];
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) ss += __shfl_xor(ss, m);
    const float rinv = 1.0f / sqrtf(ss / (float)d + eps);
    for (int i0 = lane; i0 < d; i0 += 8 * 64) {
        float v[8], wv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int i = min(i0 + u * 64, d - 1); v[u] = xr[i]; wv[u] = w[i]; }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (i0 + u * 64 < d) {
                const float o = (v[u] * rinv) * wv[u];
                if (out_bf16 == 2) ((__bf16*)y)[q3_atile_off(r, i0 + u * 64, d >> 5)] = (__bf16)o;
                else if (out_bf16) ((__bf16*)y)[(size_t)r * d + i0 + u * 64] = (__bf16)o;
                else y[(size_t)r * d + i0 + u * 64] = o;
            }
    }
}

// RoPE + ring append + sliding-window attention; one workgroup per (slot, head), one wave per new token. All tokens append their
// K / V rows first; the ring holds W + VOC_FCAP rows, so no token of the call overwrites a row another token of the call still reads.
// Per token the arithmetic is what the one-wave kernel of round 1 did (same chains, same order); cos / sin come from the host table.
__global__ __launch_bounds__(64 * VOC_FCAP) void k_voc_attn(const VCall* __restrict__ clp, const float* qkv, float* kring, float* vring, const float* rope,
                                                            int H, int hd, int RW, int W, float* att, int tiled) {
    __shared__ float sc_s[VOC_FCAP][512];
    __shared__ float qs_s[VOC_FCAP][128];
    const VCall& cl = *clp;
    const int s = blockIdx.y, h = blockIdx.x, t = threadIdx.x >> 6, lane = threadIdx.x & 63, HH = H * hd, half = hd >> 1;
    const int slot = cl.slot[s], T = cl.nf;  // blockDim.x = 64 * T
    float* kr = kring + (size_t)slot * RW * HH + h * hd;
    float* vr = vring + (size_t)slot * RW * HH + h * hd;
    float* sc = sc_s[t]; float* qs = qs_s[t];
    const float scale = 1.0f / sqrtf((float)hd);
    const int pos = cl.pos[s] + t, m = s * T + t;
    {
        const float* qp = qkv + (size_t)m * 3 * HH + h * hd;  // q | k | v of this row inside the fused [M][3*HH] buffer
        const float* kp0 = qp + HH; const float* vp = qp + 2 * HH;
        if (lane < half) {
            const float2 csn = ((const float2*)rope)[(size_t)pos * half + lane];
            const float cs = csn.x, sn = csn.y;
            float a = qp[lane], b = qp[lane + half];
            qs[lane] = a * cs - b * sn; qs[lane + half] = b * cs + a * sn;
            a = kp0[lane]; b = kp0[lane + half];
            float* kd = kr + (size_t)(pos % RW) * HH;
            kd[lane] = a * cs - b * sn; kd[lane + half] = b * cs + a * sn;
        }
        for (int i = lane; i < hd; i += 64) vr[(size_t)(pos % RW) * HH + i] = vp[i];
    }
    __syncthreads();
    const int j0 = pos - W + 1 > 0 ? pos - W + 1 : 0, nk = pos - j0 + 1;
    for (int j = lane; j < nk; j += 64) {
        const float4* kp = (const float4*)(kr + (size_t)((j0 + j) % RW) * HH);
        float a = 0.0f;
        for (int i0 = 0; i0 < hd; i0 += 64) {  // 16 x 16 B of the key row in flight per trip (a 64-wide head: the whole row), same ascending chain
            float4 kk[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) kk[u] = kp[min(i0 / 4 + u, hd / 4 - 1)];
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (i0 + 4 * u < hd) {
                    const float* qp2 = qs + i0 + 4 * u;
                    a += qp2[0] * kk[u].x; a += qp2[1] * kk[u].y; a += qp2[2] * kk[u].z; a += qp2[3] * kk[u].w;
                }
        }
        sc[j] = a * scale;
    }
    __syncthreads();
    // softmax weights once per key (lane j), max and sum by wave reductions; then P.V with the weights from LDS
    float mx = -INFINITY;
    for (int j = lane; j < nk; j += 64) mx = fmaxf(mx, sc[j]);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float l = 0.0f;
    for (int j = lane; j < nk; j += 64) { const float pj = expf(sc[j] - mx); sc[j] = pj; l += pj; }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) l += __shfl_xor(l, o);
    __syncthreads();
    for (int i = lane; i < hd; i += 64) {
        float o = 0.0f;
        for (int jb = 0; jb < nk; jb += 24) {  // 24 value rows in flight per trip (a 72-row window: three trips), same ascending chain
            float vv[24];
#pragma unroll
            for (int u = 0; u < 24; ++u) vv[u] = vr[(size_t)((j0 + min(jb + u, nk - 1)) % RW) * HH + i];
#pragma unroll
            for (int u = 0; u < 24; ++u)
                if (jb + u < nk) o += sc[jb + u] * vv[u];
        }
        // only ever the A operand of the output projection
        const size_t off = tiled ? q3_atile_off(m, h * hd + i, HH >> 5) : (size_t)m * HH + h * hd + i;
        ((__bf16*)att)[off] = (__bf16)(o / l);
    }
}

// The same attention with the (slot, head)'s key / value window staged in LDS ONCE for the call's tokens: the rows arrive by coalesced
// 16-byte loads (k_voc_attn's score pass has every lane walk its own key row: 64 cache lines per load instruction, and the four tokens of a
// slot read the same window four times), keys in rows of hd + 1 floats (lane j reads row j: conflict-free). Same chains in the same order as
// k_voc_attn — bit-identical (the tests compare chunkings and launch modes bit for bit); hd <= 64.
__global__ __launch_bounds__(64 * VOC_FCAP) void k_voc_attn_lds(const VCall* __restrict__ clp, const float* qkv, float* kring, float* vring, const float* rope,
                                                                int H, int hd, int RW, int W, float* att, int tiled) {
    extern __shared__ float al[];   // K rows [NR][hd + 1] | V rows [NR][hd], NR = W + T - 1
    __shared__ float sc_s[VOC_FCAP][512];
    __shared__ float qs_s[VOC_FCAP][128];
    const VCall& cl = *clp;
    const int s = blockIdx.y, h = blockIdx.x, t = threadIdx.x >> 6, lane = threadIdx.x & 63, HH = H * hd, half = hd >> 1, tid = threadIdx.x;
    const int slot = cl.slot[s], T = cl.nf;  // blockDim.x = 64 * T
    float* kr = kring + (size_t)slot * RW * HH + h * hd;
    float* vr = vring + (size_t)slot * RW * HH + h * hd;
    float* sc = sc_s[t]; float* qs = qs_s[t];
    const float scale = 1.0f / sqrtf((float)hd);
    const int pos0 = cl.pos[s], pos = pos0 + t, m = s * T + t;
    const int jmin = pos0 - W + 1 > 0 ? pos0 - W + 1 : 0, NR = W + T - 1, KS = hd + 1;
    float* kl = al; float* vl = al + (size_t)NR * KS;
    {   // the rows cached by earlier calls: [jmin, pos0) — 16 bytes per thread per trip
        const int nold = pos0 - jmin, q4 = hd >> 2;
        for (int i = tid; i < nold * q4; i += 64 * T) {
            const int r = i / q4, c = (i - r * q4) * 4;
            const size_t ro = (size_t)((jmin + r) % RW) * HH + c;
            const float4 kk = *(const float4*)(kr + ro), vv = *(const float4*)(vr + ro);
            float* kd = kl + (size_t)r * KS + c; kd[0] = kk.x; kd[1] = kk.y; kd[2] = kk.z; kd[3] = kk.w;
            *(float4*)(vl + (size_t)r * hd + c) = vv;
        }
        // this wave's token: RoPE, append to the ring (later calls) and to the staged window
        const float* qp = qkv + (size_t)m * 3 * HH + h * hd;
        const float* kp0 = qp + HH; const float* vp = qp + 2 * HH;
        const int lr = pos - jmin;
        if (lane < half) {
            const float2 csn = ((const float2*)rope)[(size_t)pos * half + lane];
            const float cs = csn.x, sn = csn.y;
            float a = qp[lane], b = qp[lane + half];
            qs[lane] = a * cs - b * sn; qs[lane + half] = b * cs + a * sn;
            a = kp0[lane]; b = kp0[lane + half];
            const float k0 = a * cs - b * sn, k1 = b * cs + a * sn;
            float* kd = kr + (size_t)(pos % RW) * HH;
            kd[lane] = k0; kd[lane + half] = k1;
            kl[(size_t)lr * KS + lane] = k0; kl[(size_t)lr * KS + lane + half] = k1;
        }
        for (int i = lane; i < hd; i += 64) { const float vv = vp[i]; vr[(size_t)(pos % RW) * HH + i] = vv; vl[(size_t)lr * hd + i] = vv; }
    }
    __syncthreads();
    const int j0 = pos - W + 1 > 0 ? pos - W + 1 : 0, nk = pos - j0 + 1, r0 = j0 - jmin;
    for (int j = lane; j < nk; j += 64) {
        const float* kp = kl + (size_t)(r0 + j) * KS;
        float a = 0.0f;
        for (int i = 0; i < hd; i += 4) { a += qs[i] * kp[i]; a += qs[i + 1] * kp[i + 1]; a += qs[i + 2] * kp[i + 2]; a += qs[i + 3] * kp[i + 3]; }
        sc[j] = a * scale;
    }
    __syncthreads();
    float mx = -INFINITY;
    for (int j = lane; j < nk; j += 64) mx = fmaxf(mx, sc[j]);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float l = 0.0f;
    for (int j = lane; j < nk; j += 64) { const float pj = expf(sc[j] - mx); sc[j] = pj; l += pj; }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) l += __shfl_xor(l, o);
    __syncthreads();
    for (int i = lane; i < hd; i += 64) {
        float o = 0.0f;
        for (int j = 0; j < nk; ++j) o += sc[j] * vl[(size_t)(r0 + j) * hd + i];
        const size_t off = tiled ? q3_atile_off(m, h * hd + i, HH >> 5) : (size_t)m * HH + h * hd + i;
        ((__bf16*)att)[off] = (__bf16)(o / l);
    }
}

// ConvNeXt front: depthwise causal conv k7 + LayerNorm(eps 1e-6) per position; one wave per (slot, t)
__global__ __launch_bounds__(64) void k_voc_dw_ln(const float* x, size_t x_stride, int H, int T, int C, const float* dw_w, const float* dw_b,
                                                  const float* ln_w, const float* ln_b, float* y, int tiled) {
    extern __shared__ float row[];
    const int s = blockIdx.y, t = blockIdx.x, lane = threadIdx.x;
    const float* xp = x + (size_t)s * x_stride + (size_t)(H + t) * C;
    float sum = 0.0f;
    for (int i = lane; i < C; i += 64) {
        float a = dw_b[i];
#pragma unroll
        for (int tap = 0; tap < 7; ++tap) a += xp[(long)(tap - 6) * C + i] * dw_w[(size_t)tap * C + i];
        row[i] = a; sum += a;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) sum += __shfl_xor(sum, m);
    const float mean = sum / (float)C;
    float var = 0.0f;
    for (int i = lane; i < C; i += 64) { const float z = row[i] - mean; var += z * z; }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) var += __shfl_xor(var, m);
    const float rinv = 1.0f / sqrtf(var / (float)C + 1e-6f);
    float* yp = y + ((size_t)s * T + t) * C;
    for (int i = lane; i < C; i += 64) {
        const float o = ((row[i] - mean) * rinv) * ln_w[i] + ln_b[i];
        if (tiled) ((__bf16*)y)[q3_atile_off(s * T + t, i, C >> 5)] = (__bf16)o;  // GEMM-only: the bf16 it would be rounded to, A-tiled
        else yp[i] = o;
    }
}

// V6: Conv k7 C -> 1 over the snaked input (with history) + clamp -> PCM of the slot. A block produces 64 samples:
// the 70-row input window is staged in LDS with coalesced loads (rows padded to C+1 floats: the per-thread row
// stride then walks all banks), weights in LDS too; same summation order as before (per tap, channels ascending).
__global__ __launch_bounds__(256) void k_voc_out(const VCall* __restrict__ clp, const float* x, size_t x_stride, int H, int T, int C, const float* w, const float* b,
                                                 float* pcm, size_t pcm_stride, int spf) {
    extern __shared__ float sm[];  // win[70][C+1] | wl[7*C]
    const VCall& cl = *clp;
    const int s = blockIdx.y, slot = cl.slot[s], t0 = blockIdx.x * 64, tid = threadIdx.x, CP = C + 1;
    float* win = sm; float* wl = sm + 70 * CP;
    const uint16_t* xp = (const uint16_t*)x + (size_t)s * x_stride + (size_t)(H + t0 - 6) * C;  // window row 0 = t0 - 6 (bf16 buffer)
    const int nrow = min(70, T - t0 + 6);
    for (int i0 = tid; i0 < nrow * C; i0 += 8 * 256) {  // 8 loads in flight per trip
        uint16_t v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = xp[min(i0 + u * 256, nrow * C - 1)];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * 256;
            if (i < nrow * C) { const int r = i / C, c = i - r * C; win[r * CP + c] = q3_u2f((uint32_t)v[u] << 16); }
        }
    }
    for (int i = tid; i < 7 * C; i += 256) wl[i] = w[i];
    __syncthreads();
    // 4 threads per output sample split the channels; partial sums are combined in a fixed order
    const int o = tid >> 2, part = tid & 3, t = t0 + o;
    float acc[7];
#pragma unroll
    for (int tap = 0; tap < 7; ++tap) {
        float a = 0.0f;
        if (t < T) for (int i = part; i < C; i += 4) a += win[(o + tap) * CP + i] * wl[tap * C + i];
        a += __shfl_xor(a, 1); a += __shfl_xor(a, 2);
        acc[tap] = a;
    }
    if (t < T && part == 0) {
        float r = 0.0f;
#pragma unroll
        for (int tap = 0; tap < 7; ++tap) r += acc[tap];
        r += b[0];
        pcm[(size_t)slot * pcm_stride + (size_t)cl.pos[s] * spf + t] = fminf(1.0f, fmaxf(-1.0f, r));
    }
}

// V6 for channel counts that are multiples of 8 (every shipped shape): the window is staged as the bf16 it is stored as, with 16-byte
// loads in one round trip, and every thread owns 8 channels x 8 consecutive samples: 14 window rows (one ds_read_b128 each) and its 56
// weights in registers feed 448 FMAs — 0.03 LDS reads per MAC. k_voc_out reads both operands of every MAC from LDS and stages the window
// two bytes at a time in four round trips: 114 us for 94 MB of input (0.8 TB/s) at the full shape.
// Order per sample: a slice's chain runs over the taps, inside a tap over its 8 channels (fmaf); the C / 8 slices are then added in
// ascending order, then the bias. (Its own fixed order — independent of the tile position, so chunked == one-shot holds bit for bit.)
__global__ __launch_bounds__(256) void k_voc_out8(const VCall* __restrict__ clp, const uint16_t* x, size_t x_stride, int H, int T, int C, const float* w, const float* b,
                                                  float* pcm, size_t pcm_stride, int spf, int G) {
    extern __shared__ __attribute__((aligned(16))) char sm8[];  // window [G * 8 + 6][C bf16 + 16 B] | partial sums [C / 8][G * 8]
    const int NS = C >> 3, R = G * 8, nrow = R + 6, ld = C * 2 + 16;
    float* part = (float*)(sm8 + (size_t)nrow * ld);
    const VCall& cl = *clp;
    const int s = blockIdx.y, slot = cl.slot[s], t0 = blockIdx.x * R, tid = threadIdx.x;
    const uint16_t* xp = x + (size_t)s * x_stride + (size_t)(H + t0 - 6) * C;  // window row 0 = t0 - 6
    const int lim = min(nrow, T - t0 + 6), total = nrow * NS;
    for (int base = tid; base < total; base += 8 * 256) {
        uint4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int i = min(base + u * 256, total - 1), r = i / NS, c = i - r * NS; v[u] = *(const uint4*)(xp + (size_t)min(r, lim - 1) * C + c * 8); }
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int i = base + u * 256; if (i < total) { const int r = i / NS, c = i - r * NS; *(uint4*)(sm8 + (size_t)r * ld + c * 16) = v[u]; } }
    }
    const int g = tid / NS, k = tid - g * NS;
    float wr[7][8];
    if (g < G) {
#pragma unroll
        for (int tap = 0; tap < 7; ++tap) {
            const float4 w0 = *(const float4*)(w + tap * C + k * 8), w1 = *(const float4*)(w + tap * C + k * 8 + 4);
            wr[tap][0] = w0.x; wr[tap][1] = w0.y; wr[tap][2] = w0.z; wr[tap][3] = w0.w; wr[tap][4] = w1.x; wr[tap][5] = w1.y; wr[tap][6] = w1.z; wr[tap][7] = w1.w;
        }
    }
    __syncthreads();
    if (g < G) {
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 14; ++j) {
            const uint4 q = *(const uint4*)(sm8 + (size_t)(g * 8 + j) * ld + k * 16);
            const float xv[8] = {q3_u2f(q.x << 16), q3_u2f(q.x & 0xFFFF0000u), q3_u2f(q.y << 16), q3_u2f(q.y & 0xFFFF0000u),
                                 q3_u2f(q.z << 16), q3_u2f(q.z & 0xFFFF0000u), q3_u2f(q.w << 16), q3_u2f(q.w & 0xFFFF0000u)};
#pragma unroll
            for (int tap = 0; tap < 7; ++tap) {
                const int o = j - tap;  // sample g * 8 + o reads window row o + tap
                if (o >= 0 && o < 8) {
#pragma unroll
                    for (int c = 0; c < 8; ++c) acc[o] = fmaf(xv[c], wr[tap][c], acc[o]);
                }
            }
        }
#pragma unroll
        for (int o = 0; o < 8; ++o) part[k * R + g * 8 + o] = acc[o];
    }
    __syncthreads();
    if (tid < R && t0 + tid < T) {
        float r = 0.0f;
        for (int kk = 0; kk < NS; ++kk) r += part[kk * R + tid];
        r += b[0];
        pcm[(size_t)slot * pcm_stride + (size_t)cl.pos[s] * spf + t0 + tid] = fminf(1.0f, fmaxf(-1.0f, r));
    }
}

// ------------------------------------------------------------------------------------------------------------------
// host
// ------------------------------------------------------------------------------------------------------------------
template <class T>
static int valloc(q3tts_engine* e, Q3Voc* v, T** p, size_t n) {
    void* q = nullptr;
    const int rc = q3_dev_alloc_zeroed(e, &q, n * sizeof(T) + 256);  // the zero fill has completed on return (q3_engine.hip)
    if (rc != Q3TTS_OK) return rc;
    v->allocs.push_back(q);
    *p = (T*)q;
    return Q3TTS_OK;
}
#define VTRY(x) do { int rc__ = (x); if (rc__ != Q3TTS_OK) return rc__; } while (0)

static int gen_vec(q3tts_engine* e, Q3Voc* v, float** p, uint32_t tid, size_t n, float base, float std) {
    VTRY(valloc(e, v, p, n));
    q3_launch_fill_f32(*p, n, e->cfg.synth_seed, tid, base, std / Q3_IH4_STD, 0, e->stream);
    return Q3TTS_OK;
}
__global__ void k_fill_bf16(uint16_t* dst, size_t n, uint64_t seed, uint32_t tid, float scale) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = q3_bf16(q3_synth(seed, tid, i, scale));
}
static int gen_conv(q3tts_engine* e, Q3Voc* v, VConv* c, int comp, int ww, int wb, int ntap, int dil, int cin, int nout, int bias_n, float gain) {
    c->ntap = ntap; c->dil = dil; c->cin = cin; c->nout = nout; c->bias_n = bias_n;
    const size_t n = (size_t)ntap * nout * cin;
    VTRY(valloc(e, v, &c->w, n));
    const float scale = (gain / sqrtf((float)(ntap * cin))) / Q3_IH4_STD;
    hipLaunchKernelGGL(k_fill_bf16, dim3((unsigned)std::min<size_t>((n + 255) / 256, 65536)), dim3(256), 0, e->stream, c->w, n,
                       e->cfg.synth_seed, VTID(comp, ww), scale);
    c->b = nullptr;
    if (bias_n) VTRY(gen_vec(e, v, &c->b, VTID(comp, wb), bias_n, 0.0f, 0.02f));
    return Q3TTS_OK;
}
// SnakeBeta parameters: exp() evaluated in double on the host (same as the oracle)
static int gen_snake(q3tts_engine* e, Q3Voc* v, uint32_t ta, uint32_t tb, int C, float** ea, float** ib) {
    std::vector<float> a(C), b(C);
    const float scale = 0.1f / Q3_IH4_STD;
    for (int i = 0; i < C; ++i) {
        const float al = 0.0f + q3_synth(e->cfg.synth_seed, ta, i, scale), be = 0.0f + q3_synth(e->cfg.synth_seed, tb, i, scale);
        a[i] = (float)exp((double)al); b[i] = (float)(1.0 / (exp((double)be) + 1e-9));
    }
    VTRY(valloc(e, v, ea, (size_t)C)); VTRY(valloc(e, v, ib, (size_t)C));
    Q3_HIP(e, hipMemcpy(*ea, a.data(), (size_t)C * 4, hipMemcpyHostToDevice));  // (a / b are locals: synchronous copies)
    Q3_HIP(e, hipMemcpy(*ib, b.data(), (size_t)C * 4, hipMemcpyHostToDevice));
    return Q3TTS_OK;
}
static int mk_buf(q3tts_engine* e, Q3Voc* v, VBuf* b, int H, int C, int Tcap, int bf16 = 0) {
    b->H = H; b->C = C; b->Tcap = Tcap; b->bf16 = bf16;
    VTRY(valloc(e, v, &b->p, (size_t)VOC_MAX_NS * b->stride()));
    VTRY(valloc(e, v, &b->hist, (size_t)v->B * std::max(1, H) * C));
    return Q3TTS_OK;
}

int q3_voc_samples_per_frame(const q3tts_engine* e) { return e->voc ? e->voc->spf : 0; }

int q3_voc_create(q3tts_engine* e) {
    const q3tts_vocoder_config& c = e->cfg.vocoder;
#define REQ(cond) do { if (!(cond)) return q3_set_err(e, Q3TTS_ERR_INVALID, "vocoder config check failed: " #cond); } while (0)
    REQ(c.n_codebooks >= 1 && c.n_codebooks <= 16 && c.n_codebooks <= e->cfg.model.n_codebooks);
    REQ(c.codebook_dim % 32 == 0 && c.latent_dim % 32 == 0 && c.d_ffn % 32 == 0 && (c.n_head * c.head_dim) % 32 == 0);
    REQ(c.d_ffn % 16 == 0);
    REQ(c.head_dim <= 128 && c.head_dim % 4 == 0 && c.sliding_window >= 1 && c.sliding_window + VOC_FCAP <= 512);
    REQ(c.n_upsample >= 0 && c.n_upsample <= Q3TTS_MAX_UPSAMPLE && c.n_dec_blocks >= 1 && c.n_dec_blocks <= Q3TTS_MAX_DEC_BLOCKS);
    REQ(c.pre_conv_kernel >= 1 && c.lookahead_frames >= 0);
    { int ch = c.decoder_dim; for (int b = 0; b < c.n_dec_blocks; ++b) { REQ(ch % 64 == 0); ch /= 2; } REQ(ch >= 1); }
#undef REQ
    Q3Voc* v = new Q3Voc();
    e->voc = v;
    v->c = c; v->B = e->B;
    const int d = c.latent_dim, HH = c.n_head * c.head_dim;
    v->RW = c.sliding_window + VOC_FCAP;
    v->cb.resize(c.n_codebooks);
    for (int q = 0; q < c.n_codebooks; ++q) {
        VTRY(valloc(e, v, &v->cb[q], (size_t)c.codebook_size * c.codebook_dim));
        q3_launch_fill_f32(v->cb[q], (size_t)c.codebook_size * c.codebook_dim, e->cfg.synth_seed, VTID(VC_CODEBOOK + q, VW_W), 0.0f,
                           (1.0f / sqrtf(16.0f)) / Q3_IH4_STD, 1, e->stream);
    }
    { float** cd = nullptr; VTRY(valloc(e, v, &cd, (size_t)16)); v->cb_dev = (const float**)cd;
      Q3_HIP(e, hipMemcpy((void*)cd, v->cb.data(), sizeof(float*) * c.n_codebooks, hipMemcpyHostToDevice)); }
    VTRY(gen_conv(e, v, &v->pre, VC_PRE, VW_W, VW_B, c.pre_conv_kernel, 1, c.codebook_dim, d, d, 1.0f));
    VTRY(mk_buf(e, v, &v->pre_in, c.pre_conv_kernel - 1, c.codebook_dim, VOC_FCAP));
    v->L.resize(c.n_layer);
    for (int l = 0; l < c.n_layer; ++l) {
        VLayer& y = v->L[l]; const int comp = VC_TFM + l;
        VTRY(gen_vec(e, v, &y.in_norm, VTID(comp, VW_IN_NORM), d, 1.0f, 0.05f));
        VTRY(gen_vec(e, v, &y.post_norm, VTID(comp, VW_POST_NORM), d, 1.0f, 0.05f));
        VTRY(gen_vec(e, v, &y.ls_attn, VTID(comp, VW_LS_ATTN), d, c.layer_scale_init, 0.1f * c.layer_scale_init));
        VTRY(gen_vec(e, v, &y.ls_mlp, VTID(comp, VW_LS_MLP), d, c.layer_scale_init, 0.1f * c.layer_scale_init));
        VTRY(gen_conv(e, v, &y.q, comp, VW_Q, 0, 1, 1, d, HH, 0, 1.0f)); VTRY(gen_conv(e, v, &y.k, comp, VW_K, 0, 1, 1, d, HH, 0, 1.0f));
        VTRY(gen_conv(e, v, &y.v, comp, VW_V, 0, 1, 1, d, HH, 0, 1.0f)); VTRY(gen_conv(e, v, &y.o, comp, VW_O, 0, 1, 1, HH, d, 0, 1.0f));
        VTRY(gen_conv(e, v, &y.gate, comp, VW_GATE, 0, 1, 1, d, c.d_ffn, 0, 1.0f)); VTRY(gen_conv(e, v, &y.up, comp, VW_UP, 0, 1, 1, d, c.d_ffn, 0, 1.0f));
        VTRY(gen_conv(e, v, &y.down, comp, VW_DOWN, 0, 1, 1, c.d_ffn, d, 0, 1.0f));
        // fused copies (same bf16 values, rearranged rows)
        y.qkv = y.q; y.qkv.nout = 3 * HH;
        VTRY(valloc(e, v, &y.qkv.w, (size_t)3 * HH * d));
        Q3_HIP(e, hipMemcpyAsync(y.qkv.w, y.q.w, (size_t)HH * d * 2, hipMemcpyDeviceToDevice, e->stream));
        Q3_HIP(e, hipMemcpyAsync(y.qkv.w + (size_t)HH * d, y.k.w, (size_t)HH * d * 2, hipMemcpyDeviceToDevice, e->stream));
        Q3_HIP(e, hipMemcpyAsync(y.qkv.w + (size_t)2 * HH * d, y.v.w, (size_t)HH * d * 2, hipMemcpyDeviceToDevice, e->stream));
        y.gu = y.gate; y.gu.nout = 2 * c.d_ffn;
        VTRY(valloc(e, v, &y.gu.w, (size_t)2 * c.d_ffn * d));
        Q3_HIP(e, hipMemcpy2DAsync(y.gu.w, (size_t)32 * d * 2, y.gate.w, (size_t)16 * d * 2, (size_t)16 * d * 2, c.d_ffn / 16, hipMemcpyDeviceToDevice, e->stream));
        Q3_HIP(e, hipMemcpy2DAsync(y.gu.w + (size_t)16 * d, (size_t)32 * d * 2, y.up.w, (size_t)16 * d * 2, (size_t)16 * d * 2, c.d_ffn / 16, hipMemcpyDeviceToDevice, e->stream));
    }
    // every K a multiple of 256 (the shipped 1024 / 1024 / 3072): the four projections of a layer run on the decoder's k_bgemm, from
    // copies of the same bf16 values in its tiled layout (gate | up interleaved 8 + 8 columns per tile); other shapes keep k_vgemm_small
    v->tfm_bg = d % 256 == 0 && HH % 256 == 0 && c.d_ffn % 256 == 0 && (3 * HH) % 16 == 0 && c.d_ffn % 32 == 0;
    if (v->tfm_bg)
        for (int l = 0; l < c.n_layer; ++l) {
            VLayer& y = v->L[l];
            VTRY(valloc(e, v, &y.qkv_t, (size_t)3 * HH * d / 8)); VTRY(valloc(e, v, &y.o_t, (size_t)d * HH / 8));
            VTRY(valloc(e, v, &y.gu_t, (size_t)2 * c.d_ffn * d / 8)); VTRY(valloc(e, v, &y.down_t, (size_t)d * c.d_ffn / 8));
            Q3Fill f{};
            f.mode = 0; f.row0 = 0;
            f.dst = y.qkv_t; f.N = 3 * HH; f.K = d; f.rows = 3 * HH; f.src_a = y.qkv.w; q3_launch_fill_tiled(f, e->stream);
            f.dst = y.o_t; f.N = d; f.K = HH; f.rows = d; f.src_a = y.o.w; q3_launch_fill_tiled(f, e->stream);
            f.dst = y.down_t; f.N = d; f.K = c.d_ffn; f.rows = d; f.src_a = y.down.w; q3_launch_fill_tiled(f, e->stream);
            f.mode = 1; f.dst = y.gu_t; f.N = 2 * c.d_ffn; f.K = d; f.src_a = y.gate.w; f.src_b = y.up.w; q3_launch_fill_tiled(f, e->stream);
        }
    {   // RoPE table: the oracle's expressions (double pow / cos / sin, rounded to f32), for every position a frame can take
        v->rope_rows = e->cfg.max_steps_cap + VOC_FCAP;
        const int half = c.head_dim / 2;
        std::vector<float> tab((size_t)v->rope_rows * half * 2);
        for (int i = 0; i < half; ++i) {
            const double inv = pow((double)c.rope_theta, -2.0 * (double)i / (double)c.head_dim);
            for (int p = 0; p < v->rope_rows; ++p) {
                const double ang = (double)p * inv;
                tab[((size_t)p * half + i) * 2] = (float)cos(ang); tab[((size_t)p * half + i) * 2 + 1] = (float)sin(ang);
            }
        }
        VTRY(valloc(e, v, &v->rope, tab.size()));
        Q3_HIP(e, hipMemcpy(v->rope, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
    }
    VTRY(gen_vec(e, v, &v->final_norm, VTID(VC_FINAL_NORM, VW_W), d, 1.0f, 0.05f));
    int rows = VOC_FCAP;  // rows per slot at the current stage
    v->spf = 1;
    v->U.resize(c.n_upsample);
    for (int s = 0; s < c.n_upsample; ++s) {
        VUp& p = v->U[s]; const int comp = VC_UP + s, r = c.upsample_ratios[s]; p.r = r; v->spf *= r;
        VTRY(gen_conv(e, v, &p.ct, comp, VW_W, VW_B, 1, 1, d, r * d, d, 1.0f));
        rows *= r;
        VTRY(mk_buf(e, v, &p.dw_in, 6, d, rows));
        VTRY(gen_vec(e, v, &p.dw_w, VTID(comp, VW_DW_W), (size_t)7 * d, 0.0f, 0.3f)); VTRY(gen_vec(e, v, &p.dw_b, VTID(comp, VW_DW_B), d, 0.0f, 0.02f));
        VTRY(gen_vec(e, v, &p.ln_w, VTID(comp, VW_LN_W), d, 1.0f, 0.05f)); VTRY(gen_vec(e, v, &p.ln_b, VTID(comp, VW_LN_B), d, 0.0f, 0.02f));
        VTRY(gen_conv(e, v, &p.pw1, comp, VW_PW1, VW_PW1_B, 1, 1, d, 4 * d, 4 * d, 1.0f));
        VTRY(gen_conv(e, v, &p.pw2, comp, VW_PW2, VW_PW2_B, 1, 1, 4 * d, d, d, 1.0f));
        if (d % 256 == 0) {  // tiled copies for k_bgemm (K = d and 4 d, N = r d, 4 d, d: all multiples of 32)
            v->up_bg = true;
            VTRY(valloc(e, v, &p.ct_t, (size_t)r * d * d / 8)); VTRY(valloc(e, v, &p.pw1_t, (size_t)4 * d * d / 8)); VTRY(valloc(e, v, &p.pw2_t, (size_t)4 * d * d / 8));
            Q3Fill f{}; f.mode = 0; f.row0 = 0;
            f.dst = p.ct_t; f.N = r * d; f.K = d; f.rows = r * d; f.src_a = p.ct.w; q3_launch_fill_tiled(f, e->stream);
            f.dst = p.pw1_t; f.N = 4 * d; f.K = d; f.rows = 4 * d; f.src_a = p.pw1.w; q3_launch_fill_tiled(f, e->stream);
            f.dst = p.pw2_t; f.N = d; f.K = 4 * d; f.rows = d; f.src_a = p.pw2.w; q3_launch_fill_tiled(f, e->stream);
        }
        VTRY(gen_vec(e, v, &p.gamma, VTID(comp, VW_GAMMA), d, 0.1f, 0.01f));
    }
    VTRY(gen_conv(e, v, &v->dec_in, VC_DEC_IN, VW_W, VW_B, 7, 1, d, c.decoder_dim, c.decoder_dim, 1.0f));
    VTRY(mk_buf(e, v, &v->dec_in_in, 6, d, rows, 1));  // bf16: it only ever feeds the decoder's input convolution
    size_t scratch = (size_t)rows * std::max(4 * d, c.decoder_dim);
    v->Bk.resize(c.n_dec_blocks);
    int ch = c.decoder_dim;
    for (int b = 0; b < c.n_dec_blocks; ++b) {
        VBlk& k = v->Bk[b]; const int comp = VC_BLK + 4 * b, r = c.dec_rates[b]; k.r = r; k.cin = ch; k.cout = ch / 2; v->spf *= r;
        VTRY(gen_snake(e, v, VTID(comp, VW_ALPHA), VTID(comp, VW_BETA), ch, &k.ea, &k.ib));
        VTRY(gen_conv(e, v, &k.ct, comp, VW_W, VW_B, 2, 1, ch, r * k.cout, k.cout, 1.0f));
        VTRY(mk_buf(e, v, &k.ct_in, 1, ch, rows, 1));
        rows *= r;
        scratch = std::max(scratch, (size_t)rows * k.cout);
        const int dil[3] = {1, 3, 9};
        for (int u = 0; u < 3; ++u) {
            VRes& s = k.res[u]; const int rc = comp + 1 + u;
            VTRY(gen_snake(e, v, VTID(rc, VW_ALPHA), VTID(rc, VW_BETA), k.cout, &s.ea, &s.ib));
            VTRY(gen_conv(e, v, &s.c1, rc, VW_W, VW_B, 7, dil[u], k.cout, k.cout, k.cout, 0.5f));
            VTRY(mk_buf(e, v, &s.c1_in, 6 * dil[u], k.cout, rows, 1));
            VTRY(gen_snake(e, v, VTID(rc, VW_ALPHA2), VTID(rc, VW_BETA2), k.cout, &s.ea2, &s.ib2));
            VTRY(gen_conv(e, v, &s.c2, rc, VW_W2, VW_B2, 1, 1, k.cout, k.cout, k.cout, 0.5f));
        }
        ch = k.cout;
    }
    v->out_c = ch;
    VTRY(gen_snake(e, v, VTID(VC_OUT, VW_ALPHA), VTID(VC_OUT, VW_BETA), ch, &v->oea, &v->oib));
    VTRY(valloc(e, v, &v->out_w, (size_t)7 * ch)); VTRY(gen_vec(e, v, &v->out_b, VTID(VC_OUT, VW_B), 1, 0.0f, 0.02f));
    q3_launch_fill_f32(v->out_w, (size_t)7 * ch, e->cfg.synth_seed, VTID(VC_OUT, VW_W), 0.0f, (0.1f / sqrtf((float)(7 * ch))) / Q3_IH4_STD, 1, e->stream);
    VTRY(mk_buf(e, v, &v->out_in, 6, ch, rows, 1));
    // transformer scratch [VOC_MAX_NS * VOC_FCAP][.]
    const size_t M = (size_t)VOC_MAX_NS * VOC_FCAP;
    VTRY(valloc(e, v, &v->x, M * d)); VTRY(valloc(e, v, &v->xn, M * d)); VTRY(valloc(e, v, &v->xnb, M * d)); VTRY(valloc(e, v, &v->qkv, M * 3 * HH));
    VTRY(valloc(e, v, &v->att, M * HH)); VTRY(valloc(e, v, &v->g, M * c.d_ffn));
    VTRY(valloc(e, v, &v->kring, (size_t)c.n_layer * v->B * v->RW * HH)); VTRY(valloc(e, v, &v->vring, (size_t)c.n_layer * v->B * v->RW * HH));
    VTRY(valloc(e, v, &v->t1, (size_t)VOC_MAX_NS * scratch)); VTRY(valloc(e, v, &v->t2, (size_t)VOC_MAX_NS * scratch));
    if (v->up_bg) { size_t rows_up = (size_t)VOC_MAX_NS * VOC_FCAP; for (auto& u : v->U) rows_up *= u.r; VTRY(valloc(e, v, &v->upb, rows_up * d)); }
    v->pcm_stride = (size_t)(e->cfg.max_steps_cap + VOC_FCAP) * v->spf;  // + padding frames behind a finished utterance
    VTRY(valloc(e, v, &v->pcm, (size_t)v->B * v->pcm_stride));
    v->frames_done.assign(v->B, 0); v->last_flag.assign(v->B, 0);
    VTRY(valloc(e, v, &v->call_dev, 1));
    Q3_HIP(e, hipHostMalloc((void**)&v->call_host, 16 * sizeof(VCall)));
    Q3_HIP(e, hipStreamSynchronize(e->stream));
    return Q3TTS_OK;
}

void q3_voc_destroy(q3tts_engine* e) {
    Q3Voc* v = e->voc;
    if (!v) return;
    for (auto& kv : v->call_graphs) hipGraphExecDestroy(kv.second);
    for (auto& ev : v->call_ev) if (ev) hipEventDestroy(ev);
    if (v->call_host) hipHostFree(v->call_host);
    for (void* p : v->allocs) hipFree(p);
    delete v;
    e->voc = nullptr;
}

// zeroes every history block of a slot: grid (entries, 4)
__global__ void k_voc_zero(const Q3Voc::ZeroEnt* tab, int slot) {
    const Q3Voc::ZeroEnt z = tab[blockIdx.x];
    uint32_t* p = (uint32_t*)(z.base + (size_t)slot * z.bytes);
    const size_t n = z.bytes / 4;  // (bf16 / f32 blocks of H x C elements, C even: whole dwords)
    for (size_t i = (size_t)blockIdx.y * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.y * blockDim.x) p[i] = 0u;
}
int q3_voc_reset(q3tts_engine* e, int slot) {
    Q3Voc* v = e->voc;
    if (!v) return Q3TTS_OK;
    if (!v->zero_tab) {  // built at the first reset: the work buffers exist by then
        std::vector<Q3Voc::ZeroEnt> tab;
        auto add = [&](const VBuf& b) { if (b.H > 0) tab.push_back({(char*)b.hist, (unsigned long long)b.H * b.C * (b.bf16 ? 2 : 4)}); };
        add(v->pre_in);
        for (auto& u : v->U) add(u.dw_in);
        add(v->dec_in_in);
        for (auto& b : v->Bk) { add(b.ct_in); for (auto& r : b.res) add(r.c1_in); }
        add(v->out_in);
        v->n_zero = (int)tab.size();
        VTRY(valloc(e, v, &v->zero_tab, tab.size()));
        Q3_HIP(e, hipMemcpy(v->zero_tab, tab.data(), tab.size() * sizeof(Q3Voc::ZeroEnt), hipMemcpyHostToDevice));
    }
    if (v->n_zero) hipLaunchKernelGGL(k_voc_zero, dim3(v->n_zero, 4), dim3(256), 0, e->stream, v->zero_tab, slot);
    v->frames_done[slot] = 0; v->last_flag[slot] = 0;
    return Q3TTS_OK;
}

struct VSnake { float* y2 = nullptr; size_t stride = 0; int off = 0; const float* ea = nullptr; const float* ib = nullptr; int n = 1; int bf16 = 0; };
static VSnake snake_into(const VBuf& dst, const float* ea, const float* ib, int C) {
    VSnake k; k.y2 = dst.p; k.stride = dst.stride(); k.off = dst.H * dst.C; k.ea = ea; k.ib = ib; k.n = C; k.bf16 = dst.bf16; return k;
}
// the wide 7-tap convolutions run their K steps chunk by chunk (vstep), whichever kernel serves them
static bool vconv_chunked(const VConv& c) { return c.ntap == 7 && c.cin % 64 == 0 && c.cin >= 256; }
// Q3TTS_VOC_NOTAP=1: those convolutions on the ring / register-staged GEMMs instead of k_vconv_tap (same bits; A/B runs and tests)
static bool voc_tap() { const char* ev = getenv("Q3TTS_VOC_NOTAP"); return !(ev && atoi(ev)); }
// fewer workgroups than this (a draining batch, a single stream) leave most CUs idle under 256-row tiles: the finer 128 x 128 tiles serve them.
// Q3TTS_VOC_TAP_MIN overrides it (tests force k_vconv_tap onto one-slot calls with 1)
static long voc_tap_min() { const char* ev = getenv("Q3TTS_VOC_TAP_MIN"); return ev ? atol(ev) : 128; }
// Q3TTS_VOC_NORING=1: the register-staged kernel for bf16 A as well (A/B runs and the tests that compare the two)
static bool voc_ring() { const char* ev = getenv("Q3TTS_VOC_NORING"); return !(ev && atoi(ev)); }
// (its epilogue moves 4 columns at a time: nout and every row start are multiples of 4 elements for all convolutions of the vocoder)
static bool voc_ring_ok(const VGemm& g) { return voc_ring() && g.c.nout % 4 == 0 && g.y_off % 4 == 0 && g.y_stride % 4 == 0 && g.y2_off % 4 == 0 && g.y2_stride % 4 == 0; }
static void vgemm(hipStream_t s, const VConv& c, const float* x, size_t x_stride, int x_off, int ns, int T, float* y, size_t y_stride, int y_off,
                  int epi = 0, const float* scale = nullptr, int scale_n = 1, const VSnake* sk = nullptr, int store = 1, int a_bf16 = 0, int y_bf16 = 0) {
    VGemm g; g.x = x; g.x_stride = x_stride; g.x_off = x_off; g.T = T; g.M = ns * T; g.c = c; g.y = y; g.y_stride = y_stride; g.y_off = y_off;
    g.scale = scale; g.scale_n = scale_n; g.epi = epi; g.store = store; g.a_bf16 = a_bf16; g.y_bf16 = y_bf16;
    g.y2 = nullptr; g.y2_stride = 0; g.y2_off = 0; g.ea = g.ib = nullptr; g.snake_n = 1; g.y2_bf16 = 0;
    if (sk) { g.y2 = sk->y2; g.y2_stride = sk->stride; g.y2_off = sk->off; g.ea = sk->ea; g.ib = sk->ib; g.snake_n = sk->n; g.y2_bf16 = sk->bf16; }
    g.chunked = vconv_chunked(c) ? 1 : 0;
    // every kernel accumulates the same 32-wide K steps in the same order: the choice never changes a result
    const long tap_wgs = (long)(c.nout / 128) * ((ns * ((T + 127) / 128) + 1) / 2);  // (counted in 256-row workgroups for both forms)
    if (g.chunked && a_bf16 && voc_tap() && voc_ring_ok(g) && c.nout % 128 == 0 && 6 * c.dil <= 64 && epi != 4 &&
        ((T + 127) / 128) * 128 * 3 <= T * 4 &&  /* at most a quarter of the 128-row sub-tiles' rows beyond T */
        tap_wgs >= voc_tap_min()) {
        if (voc_polite()) launch_vconv_tap<1>(s, g); else launch_vconv_tap<2>(s, g);
    } else if (g.M <= 512 || epi == 4) {
        // the kernel is bound by what one CU's load path delivers: a narrow N runs 64 x 16 tiles to put a workgroup on
        // every CU instead of on half of them
        const long wg32 = (long)((c.nout + 31) / 32) * ((g.M + 63) / 64);
        if (epi != 4 && wg32 < 256) {
            dim3 grid((c.nout + 15) / 16, (g.M + 63) / 64);
            if (a_bf16) hipLaunchKernelGGL((k_vgemm_small<1, true>), grid, dim3(256), 0, s, g);
            else hipLaunchKernelGGL((k_vgemm_small<1, false>), grid, dim3(256), 0, s, g);
        } else {
            dim3 grid((c.nout + 31) / 32, (g.M + 63) / 64);
            if (a_bf16) hipLaunchKernelGGL((k_vgemm_small<2, true>), grid, dim3(256), 0, s, g);
            else hipLaunchKernelGGL((k_vgemm_small<2, false>), grid, dim3(256), 0, s, g);
        }
    } else if (c.nout % 128 != 0 && c.nout % 96 == 0) {
        dim3 grid(c.nout / 96, (g.M + 127) / 128);
        if (a_bf16 && voc_ring_ok(g)) launch_vgemm_ring<3>(s, g, grid);
        else if (a_bf16) hipLaunchKernelGGL((k_vgemm_lds<3, true>), grid, dim3(256), 0, s, g);
        else hipLaunchKernelGGL((k_vgemm_lds<3, false>), grid, dim3(256), 0, s, g);
    } else if (c.nout % 64 == 0 && (long)((c.nout + 127) / 128) * ((g.M + 127) / 128) < 192) {
        // too few 128 x 128 tiles for 256 CUs (the decoder's input convolution: 1024 rows x 1536 columns): 128 x 64 tiles, twice the workgroups
        dim3 grid(c.nout / 64, (g.M + 127) / 128);
        if (a_bf16 && voc_ring_ok(g)) launch_vgemm_ring<2>(s, g, grid);
        else if (a_bf16) hipLaunchKernelGGL((k_vgemm_lds<2, true>), grid, dim3(256), 0, s, g);
        else hipLaunchKernelGGL((k_vgemm_lds<2, false>), grid, dim3(256), 0, s, g);
    } else {
        dim3 grid((c.nout + 127) / 128, (g.M + 127) / 128);
        if (a_bf16 && voc_ring_ok(g)) launch_vgemm_ring<4>(s, g, grid);
        else if (a_bf16) hipLaunchKernelGGL((k_vgemm_lds<4, true>), grid, dim3(256), 0, s, g);
        else hipLaunchKernelGGL((k_vgemm_lds<4, false>), grid, dim3(256), 0, s, g);
    }
}
static bool resunit_ok(int C) {
    const char* ev = getenv("Q3TTS_VOC_NOFUSE");  // (read per call: the tests compare both paths in one process)
    const int off = ev ? atoi(ev) : 0;
    return !off && (C == 32 || C == 64 || C == 96 || C == 128 || C == 192);
}
template <int NT, int MT, int NWV = 4>
static void launch_resunit_t(hipStream_t s, const VResUnit& g, int ns) {
    constexpr int C = NT * 16, R = 64 * MT, LDA = C + 16;
    const size_t lds = voc_lds_floor(std::max(((size_t)(R + 6 * g.dil) * LDA + (size_t)2 * C * 32) * 2, (size_t)R * (C + 4) * 4));  // input tile + weight ring, later the f32 output tile
    static Q3PerDevice pd;
    pd.ensure(1, []() { hipFuncSetAttribute((const void*)k_voc_resunit<NT, MT, NWV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 1024); });
    hipLaunchKernelGGL((k_voc_resunit<NT, MT, NWV>), dim3((g.T + R - 1) / R, ns), dim3(64 * NWV), lds, s, g);
}
static void launch_resunit(hipStream_t s, const VRes& r, int ns, int T, int C, float* o, int store_o, const VSnake& sk) {
    VResUnit g;
    g.xin = r.c1_in.p; g.xin_stride = r.c1_in.stride(); g.T = T; g.dil = r.c1.dil;
    g.w1 = r.c1.w; g.w2 = r.c2.w; g.b1 = r.c1.b; g.b2 = r.c2.b; g.ea2 = r.ea2; g.ib2 = r.ib2;
    g.o = o; g.o_stride = (size_t)T * C; g.store_o = store_o;
    g.y2 = sk.y2; g.y2_stride = sk.stride; g.y2_off = sk.off; g.ea3 = sk.ea; g.ib3 = sk.ib;
    switch (C) {
        case 32: launch_resunit_t<2, 4>(s, g, ns); break;
        case 64: launch_resunit_t<4, 3>(s, g, ns); break;
        case 96: launch_resunit_t<6, 2>(s, g, ns); break;  // (beside the decoder, one workgroup per CU: 192-row tiles 233 us against 200, 232 VGPRs)
        case 128: launch_resunit_t<8, 1>(s, g, ns); break;
        default: if (voc_polite()) launch_resunit_t<12, 1>(s, g, ns); else launch_resunit_t<12, 2, 8>(s, g, ns); break;  // (64-row tiles, 4 waves, two workgroups per CU: 182 us; 128 rows with 4 waves: 215; 128 rows with 8 waves: 168)
    }
}
static void hist(hipStream_t s, const VCall& cl, const VCall* cld, VBuf& b, int T, int save) {
    if (b.H == 0) return;
    const size_t n = (size_t)b.H * b.C;
    const dim3 grid((unsigned)std::min<size_t>((n + 255) / 256, 64), cl.ns);
    if (b.bf16) hipLaunchKernelGGL((k_voc_hist<uint16_t>), grid, dim3(256), 0, s, cld, (uint16_t*)b.p, b.stride(), (uint16_t*)b.hist, b.H, b.C, T, save);
    else hipLaunchKernelGGL((k_voc_hist<float>), grid, dim3(256), 0, s, cld, b.p, b.stride(), b.hist, b.H, b.C, T, save);
}
// every history-bearing buffer with the rows it receives in a call of nf frames, in pipeline order; the ConvNeXt buffers (dw_in) are
// loaded with the others but saved by their own launch (their rows are modified in place right after the depthwise convolution)
static void hist_all(hipStream_t s, const VCall& cl, const VCall* cld, Q3Voc* v, int nf, int save) {
    VHistTab tab; tab.n = 0;
    auto add = [&](const VBuf& b, int T, bool with_save) {
        if (b.H == 0 || (save && !with_save) || tab.n >= 24) return;
        const size_t es = b.bf16 ? 2 : 4;
        tab.e[tab.n++] = {(char*)b.p, (char*)b.hist, (unsigned long long)(b.stride() * es), (unsigned long long)((size_t)b.H * b.C * es), (unsigned long long)((size_t)T * b.C * es)};
    };
    int T = nf;
    add(v->pre_in, T, true);
    for (auto& p : v->U) { T *= p.r; add(p.dw_in, T, false); }
    add(v->dec_in_in, T, true);
    for (auto& k : v->Bk) { add(k.ct_in, T, true); T *= k.r; for (auto& r : k.res) add(r.c1_in, T, true); }
    add(v->out_in, T, true);
    if (tab.n) hipLaunchKernelGGL(k_voc_hist_all, dim3(16, cl.ns, tab.n), dim3(256), 0, s, cld, tab, save);
}
// one batched streaming call: ns slots x nf new frames each (uniform nf <= VOC_FCAP)
// sliding-window attention of a call: the LDS-staged kernel for heads of <= 64 dims (Q3TTS_VOC_ATTN_OLD=1: the row-walking kernel; same bits)
static void voc_launch_attn(hipStream_t s, const VCall* cld, Q3Voc* v, float* kr, float* vr, int ns, int nf, int tiled) {
    const q3tts_vocoder_config& c = v->c;
    const char* ev = getenv("Q3TTS_VOC_ATTN_OLD");   // (read per launch: a test compares the two kernels in one process)
    const bool old = ev && atoi(ev);
    const size_t lds = (size_t)(c.sliding_window + nf - 1) * (2 * c.head_dim + 1) * sizeof(float);
    if (!old && c.head_dim <= 64 && c.head_dim % 4 == 0 && lds <= 48 * 1024) {
        hipLaunchKernelGGL(k_voc_attn_lds, dim3(c.n_head, ns), dim3(64 * nf), lds, s, cld, v->qkv, kr, vr, v->rope, c.n_head, c.head_dim, v->RW, c.sliding_window, v->att, tiled);
        return;
    }
    hipLaunchKernelGGL(k_voc_attn, dim3(c.n_head, ns), dim3(64 * nf), 0, s, cld, v->qkv, kr, vr, v->rope, c.n_head, c.head_dim, v->RW, c.sliding_window, v->att, tiled);
}
static int voc_call_body(q3tts_engine* e, const VCall& cl, hipStream_t s) {
    Q3Voc* v = e->voc;
    const VCall* cld = v->call_dev;
    const q3tts_vocoder_config& c = v->c;
    const int ns = cl.ns, nf = cl.nf, d = c.latent_dim, HH = c.n_head * c.head_dim, M = ns * nf;
    hist_all(s, cl, cld, v, nf, 0);
    // V1 + V2
    hipLaunchKernelGGL(k_voc_embed, dim3(nf, ns), dim3(128), 0, s, cld, e->codes, e->cfg.max_steps_cap, e->cfg.model.n_codebooks, v->cb_dev,
                       c.n_codebooks, c.codebook_size, c.codebook_dim, v->pre_in.p, v->pre_in.stride(), v->pre_in.H * v->pre_in.C);
    vgemm(s, v->pre, v->pre_in.p, v->pre_in.stride(), v->pre_in.H * v->pre_in.C, ns, nf, v->x, (size_t)nf * d, 0);
    // V3 transformer (rows m = s*nf + t)
    for (int l = 0; l < c.n_layer; ++l) {
        VLayer& L = v->L[l];
        // GEMM-only activations (normed input, attention output, SwiGLU output) are stored as the bf16 they would be rounded to
        float* kr = v->kring + (size_t)l * v->B * v->RW * HH; float* vr = v->vring + (size_t)l * v->B * v->RW * HH;
        if (v->tfm_bg) {  // the decoder's GEMM: A-tiled bf16 rows in, 8 K-slices per output (its tile choice never changes a result)
            Q3BGemm g{}; g.B = M;
            hipLaunchKernelGGL(k_voc_rmsnorm, dim3(M), dim3(64), 0, s, v->x, L.in_norm, c.rms_eps, d, v->xnb, 2);
            g.a = (const uint16_t*)v->xnb; g.w = L.qkv_t; g.K = d; g.N = 3 * HH; g.epi = Q3_EPI_STORE; g.y = v->qkv; g.ldy = 3 * HH;
            if (q3_launch_bgemm(g, s)) return q3_set_err(e, Q3TTS_ERR_INVALID, "vocoder: qkv GEMM shape");
            voc_launch_attn(s, cld, v, kr, vr, ns, nf, 1);
            g.a = (const uint16_t*)v->att; g.w = L.o_t; g.K = HH; g.N = d; g.epi = Q3_EPI_RESID; g.y = v->x; g.ldy = d; g.col_scale = L.ls_attn;
            if (q3_launch_bgemm(g, s)) return q3_set_err(e, Q3TTS_ERR_INVALID, "vocoder: o GEMM shape");
            hipLaunchKernelGGL(k_voc_rmsnorm, dim3(M), dim3(64), 0, s, v->x, L.post_norm, c.rms_eps, d, v->xnb, 2);
            g.a = (const uint16_t*)v->xnb; g.w = L.gu_t; g.K = d; g.N = 2 * c.d_ffn; g.epi = Q3_EPI_SWIGLU; g.y = nullptr; g.yb = (uint16_t*)v->g; g.col_scale = nullptr;
            if (q3_launch_bgemm(g, s)) return q3_set_err(e, Q3TTS_ERR_INVALID, "vocoder: gate/up GEMM shape");
            g.a = (const uint16_t*)v->g; g.w = L.down_t; g.K = c.d_ffn; g.N = d; g.epi = Q3_EPI_RESID; g.y = v->x; g.ldy = d; g.yb = nullptr; g.col_scale = L.ls_mlp;
            if (q3_launch_bgemm(g, s)) return q3_set_err(e, Q3TTS_ERR_INVALID, "vocoder: down GEMM shape");
            continue;
        }
        hipLaunchKernelGGL(k_voc_rmsnorm, dim3(M), dim3(64), 0, s, v->x, L.in_norm, c.rms_eps, d, v->xnb, 1);
        vgemm(s, L.qkv, v->xnb, 0, 0, 1, M, v->qkv, 0, 0, 0, nullptr, 1, nullptr, 1, 1);
        voc_launch_attn(s, cld, v, kr, vr, ns, nf, 0);
        vgemm(s, L.o, v->att, 0, 0, 1, M, v->x, 0, 0, 1, L.ls_attn, d, nullptr, 1, 1);
        hipLaunchKernelGGL(k_voc_rmsnorm, dim3(M), dim3(64), 0, s, v->x, L.post_norm, c.rms_eps, d, v->xnb, 1);
        vgemm(s, L.gu, v->xnb, 0, 0, 1, M, v->g, 0, 0, 4, nullptr, 1, nullptr, 1, 1, 1);  // gate | up in one launch, SwiGLU in the epilogue
        vgemm(s, L.down, v->g, 0, 0, 1, M, v->x, 0, 0, 1, L.ls_mlp, d, nullptr, 1, 1);
    }
    // V5a upsample stages; cur = [ns][T][d] contiguous per slot (stride T*d)
    const float* cur = v->xn; int T = nf; size_t cur_stride = (size_t)nf * d; int cur_off = 0;
    if (v->up_bg) {
        // the stages' single-tap GEMMs on the decoder's k_bgemm: A-tiled bf16 rows (row m = s * T + t) in, f32 results into the per-slot
        // work buffers (segmented rows), bias / GELU / LayerScale-residual in the epilogue
        hipLaunchKernelGGL(k_voc_rmsnorm, dim3(M), dim3(64), 0, s, v->x, v->final_norm, c.rms_eps, d, v->xnb, 2);
        const uint16_t* a_in = (const uint16_t*)v->xnb;
        for (size_t ui = 0; ui < v->U.size(); ++ui) {
            VUp& p = v->U[ui];
            Q3BGemm g{}; g.a = a_in; g.B = ns * T; g.w = p.ct_t; g.K = d; g.N = p.r * d; g.epi = Q3_EPI_STORE;  // [T][r*d] == [T*r][d]
            g.bias = p.ct.b; g.bias_n = p.ct.bias_n; g.y = p.dw_in.p + (size_t)p.dw_in.H * d; g.ldy = p.r * d; g.seg_rows = T; g.seg_stride = p.dw_in.stride();
            if (q3_launch_bgemm(g, s)) return q3_set_err(e, Q3TTS_ERR_INVALID, "vocoder: upsample ConvTranspose GEMM shape");
            T *= p.r;
            hipLaunchKernelGGL(k_voc_dw_ln, dim3(T, ns), dim3(64), (size_t)d * 4, s, p.dw_in.p, p.dw_in.stride(), p.dw_in.H, T, d, p.dw_w, p.dw_b, p.ln_w, p.ln_b, v->t1, 1);
            hist(s, cl, cld, p.dw_in, T, 1);  // history = the raw ConvTranspose output, saved before the in-place residual below
            Q3BGemm h{}; h.a = (const uint16_t*)v->t1; h.B = ns * T; h.w = p.pw1_t; h.K = d; h.N = 4 * d; h.epi = Q3_EPI_GELU;
            h.bias = p.pw1.b; h.bias_n = p.pw1.bias_n; h.yb = (uint16_t*)v->t2;
            if (q3_launch_bgemm(h, s)) return q3_set_err(e, Q3TTS_ERR_INVALID, "vocoder: pointwise-1 GEMM shape");
            Q3BGemm r{}; r.a = (const uint16_t*)v->t2; r.B = ns * T; r.w = p.pw2_t; r.K = 4 * d; r.N = d; r.epi = Q3_EPI_RESID;  // residual in place
            r.bias = p.pw2.b; r.bias_n = p.pw2.bias_n; r.col_scale = p.gamma; r.y = p.dw_in.p + (size_t)p.dw_in.H * d; r.ldy = d; r.seg_rows = T; r.seg_stride = p.dw_in.stride();
            if (ui + 1 < v->U.size()) r.yb = v->upb;  // the next stage's GEMM input
            if (q3_launch_bgemm(r, s)) return q3_set_err(e, Q3TTS_ERR_INVALID, "vocoder: pointwise-2 GEMM shape");
            a_in = v->upb;
            cur = p.dw_in.p; cur_stride = p.dw_in.stride(); cur_off = p.dw_in.H * d;
        }
    } else {
        hipLaunchKernelGGL(k_voc_rmsnorm, dim3(M), dim3(64), 0, s, v->x, v->final_norm, c.rms_eps, d, v->xn, 0);
        for (auto& p : v->U) {
            vgemm(s, p.ct, cur, cur_stride, cur_off, ns, T, p.dw_in.p, p.dw_in.stride(), p.dw_in.H * d);  // [T][r*d] == [T*r][d]
            T *= p.r;
            hipLaunchKernelGGL(k_voc_dw_ln, dim3(T, ns), dim3(64), (size_t)d * 4, s, p.dw_in.p, p.dw_in.stride(), p.dw_in.H, T, d, p.dw_w, p.dw_b, p.ln_w, p.ln_b, v->t1, 0);
            hist(s, cl, cld, p.dw_in, T, 1);  // history = the raw ConvTranspose output, saved before the in-place residual below
            vgemm(s, p.pw1, v->t1, (size_t)T * d, 0, ns, T, v->t2, (size_t)T * 4 * d, 0, 3);
            vgemm(s, p.pw2, v->t2, (size_t)T * 4 * d, 0, ns, T, p.dw_in.p, p.dw_in.stride(), p.dw_in.H * d, 1, p.gamma, d);  // residual in place
            cur = p.dw_in.p; cur_stride = p.dw_in.stride(); cur_off = p.dw_in.H * d;
        }
    }
    {   // timing experiment only (results are then garbage): Q3TTS_EXP_VOC_SKIP=2 ends the call here — the launch-bound front end (embedding,
        // transformer, up-sampling) alone beside the decoder, to split the interference between the two halves of a call (profiles/README.md)
        static const int exp_skip = getenv("Q3TTS_EXP_VOC_SKIP") ? atoi(getenv("Q3TTS_EXP_VOC_SKIP")) : 0;
        if (exp_skip == 2) { Q3_HIP(e, hipGetLastError()); return Q3TTS_OK; }
    }
    // V5b decoder
    hipLaunchKernelGGL(k_voc_rows_bf16, dim3((unsigned)(((size_t)T * d / 4 + 255) / 256), ns), dim3(256), 0, s, cur + cur_off, cur_stride,
                       (uint16_t*)v->dec_in_in.p + (size_t)v->dec_in_in.H * d, v->dec_in_in.stride(), T * d);
    int ch = c.decoder_dim;
    // Every SnakeBeta runs in the epilogue of the convolution that produces its input and lands directly in the
    // work buffer of the convolution that consumes it: dec_in -> blk0.ct_in; ct -> res0.c1_in; c1 -> (snake2) -> c2's
    // input; c2 -> next unit's c1_in / next block's ct_in / the final conv's window.
    {
        const VSnake sk = snake_into(v->Bk[0].ct_in, v->Bk[0].ea, v->Bk[0].ib, ch);
        vgemm(s, v->dec_in, v->dec_in_in.p, v->dec_in_in.stride(), v->dec_in_in.H * d, ns, T, v->t1, (size_t)T * ch, 0, 0, nullptr, 1, &sk, 0, 1);
    }
    float* z = v->t1; float* o = v->t2;
    for (size_t bi = 0; bi < v->Bk.size(); ++bi) {
        VBlk& k = v->Bk[bi];
        {
            const VSnake sk = snake_into(k.res[0].c1_in, k.res[0].ea, k.res[0].ib, k.cout);
            vgemm(s, k.ct, k.ct_in.p, k.ct_in.stride(), k.ct_in.H * k.cin, ns, T, o, (size_t)T * k.r * k.cout, 0, 0, nullptr, 1, &sk, 1, 1);
        }
        T *= k.r; ch = k.cout;
        for (int u = 0; u < 3; ++u) {
            VRes& r = k.res[u];
            if (resunit_ok(ch)) {  // narrow blocks: the whole residual unit in one pass over HBM
                VSnake sk;
                if (u < 2) { sk = snake_into(k.res[u + 1].c1_in, k.res[u + 1].ea, k.res[u + 1].ib, ch); }
                else if (bi + 1 < v->Bk.size()) { VBlk& nx = v->Bk[bi + 1]; sk = snake_into(nx.ct_in, nx.ea, nx.ib, ch); }
                else { sk = snake_into(v->out_in, v->oea, v->oib, ch); }
                launch_resunit(s, r, ns, T, ch, o, u < 2 ? 1 : 0, sk);
                continue;
            }
            {
                VSnake sk; sk.y2 = z; sk.stride = (size_t)T * ch; sk.off = 0; sk.ea = r.ea2; sk.ib = r.ib2; sk.n = ch; sk.bf16 = 1;  // snake2 -> z (bf16)
                vgemm(s, r.c1, r.c1_in.p, r.c1_in.stride(), r.c1_in.H * ch, ns, T, z, (size_t)T * ch, 0, 0, nullptr, 1, &sk, 0, 1);
            }
            VSnake sk;
            if (u < 2) { sk = snake_into(k.res[u + 1].c1_in, k.res[u + 1].ea, k.res[u + 1].ib, ch); }
            else if (bi + 1 < v->Bk.size()) { VBlk& nx = v->Bk[bi + 1]; sk = snake_into(nx.ct_in, nx.ea, nx.ib, ch); }
            else { sk = snake_into(v->out_in, v->oea, v->oib, ch); }
            vgemm(s, r.c2, z, (size_t)T * ch, 0, ns, T, o, (size_t)T * ch, 0, 2, nullptr, 1, &sk, u < 2 ? 1 : 0, 1);  // o += conv k1
        }
    }
    // V6
    if (ch % 8 == 0 && ch / 8 <= 32 && !getenv("Q3TTS_VOC_OUT_OLD")) {
        const int NS = ch / 8, G = std::min(16, 256 / NS), R = G * 8;
        static Q3PerDevice pd8;
        pd8.ensure(1, []() { hipFuncSetAttribute((const void*)k_voc_out8, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024); });
        hipLaunchKernelGGL(k_voc_out8, dim3((T + R - 1) / R, ns), dim3(256), (size_t)(R + 6) * (ch * 2 + 16) + (size_t)NS * R * 4, s,  /* (workgroups of a few us: no need to make room) */ cld, (const uint16_t*)v->out_in.p,
                           v->out_in.stride(), v->out_in.H, T, ch, v->out_w, v->out_b, v->pcm, v->pcm_stride, v->spf, G);
    } else
        hipLaunchKernelGGL(k_voc_out, dim3((T + 63) / 64, ns), dim3(256), (size_t)(70 * (ch + 1) + 7 * ch) * 4, s, cld, v->out_in.p, v->out_in.stride(), v->out_in.H, T, ch, v->out_w, v->out_b,
                           v->pcm, v->pcm_stride, v->spf);
    hist_all(s, cl, cld, v, nf, 1);
    Q3_HIP(e, hipGetLastError());
    return Q3TTS_OK;
}

// One vocoder call = ~100 launches whose shapes depend only on (slots, frames) and the launch mode: the slot / position table goes to the
// device in stream order and the launches are replayed as ONE hipGraph per (slots, frames, mode, switches) — issuing them one by one took the
// host 0.5 ms per 4-frame chunk, time in which the engine's loop launches no frame step. The first call of a shape runs eagerly (it also sets
// the kernels' attributes), the second is captured. Q3TTS_VOC_NO_GRAPH=1: always eager.
static unsigned long long voc_call_key(const VCall& cl) {
    unsigned long long k = (unsigned long long)cl.ns | ((unsigned long long)cl.nf << 8) | ((unsigned long long)(voc_polite() ? 1 : 0) << 12);
    unsigned long long h = 1469598103934665603ull;  // the launch-time switches (tests flip them inside one process)
    for (const char* name : {"Q3TTS_VOC_NORING", "Q3TTS_VOC_NOFUSE", "Q3TTS_VOC_NOTAP", "Q3TTS_VOC_TAP_MIN", "Q3TTS_VOC_OUT_OLD", "Q3TTS_VOC_ATTN_OLD"}) {
        const char* ev = getenv(name);
        for (const char* c = ev ? ev : "-"; *c; ++c) h = (h ^ (unsigned char)*c) * 1099511628211ull;
        h = (h ^ 0xFFu) * 1099511628211ull;
    }
    return k | (h << 16);
}
static int voc_call(q3tts_engine* e, const VCall& cl, hipStream_t s) {
    Q3Voc* v = e->voc;
    {   // upload through a pinned ring: entry i is reused only after its copy has been seen done
        const unsigned i = v->call_i++ & 15;
        if (v->call_ev[i]) Q3_HIP(e, hipEventSynchronize(v->call_ev[i]));
        else Q3_HIP(e, hipEventCreateWithFlags(&v->call_ev[i], hipEventDisableTiming));
        v->call_host[i] = cl;
        Q3_HIP(e, hipMemcpyAsync(v->call_dev, &v->call_host[i], sizeof(VCall), hipMemcpyHostToDevice, s));
        Q3_HIP(e, hipEventRecord(v->call_ev[i], s));
    }
    static const bool no_graph = [] { const char* ev = getenv("Q3TTS_VOC_NO_GRAPH"); return ev && atoi(ev); }();
    if (no_graph) return voc_call_body(e, cl, s);
    const unsigned long long key = voc_call_key(cl);
    auto it = v->call_graphs.find(key);
    if (it == v->call_graphs.end()) {
        if (!v->call_seen.count(key)) { v->call_seen.insert(key); return voc_call_body(e, cl, s); }
        hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr;
        Q3_HIP(e, hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        const int rc = voc_call_body(e, cl, s);
        const hipError_t er = hipStreamEndCapture(s, &graph);
        if (rc != Q3TTS_OK) { if (graph) hipGraphDestroy(graph); return rc; }
        if (er != hipSuccess || !graph) return q3_set_err(e, Q3TTS_ERR_DEVICE, std::string("vocoder call capture: ") + hipGetErrorString(er));
        const hipError_t ei = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        hipGraphDestroy(graph);
        if (ei != hipSuccess) return q3_set_err(e, Q3TTS_ERR_DEVICE, std::string("vocoder call graph: ") + hipGetErrorString(ei));
        it = v->call_graphs.emplace(key, exec).first;
    }
    Q3_HIP(e, hipGraphLaunch(it->second, s));
    return Q3TTS_OK;
}

// decode frames [f0, f0+nf) of ONE slot (codes already on the device); nf may exceed VOC_FCAP (split)
int q3_voc_decode(q3tts_engine* e, int slot, int f0, int nf, int is_last, hipStream_t s) {
    Q3Voc* v = e->voc;
    if (!v) return q3_set_err(e, Q3TTS_ERR_STATE, "engine has no vocoder");
    if (f0 != v->frames_done[slot]) return q3_set_err(e, Q3TTS_ERR_STATE, "vocoder frames must be consumed in order");
    while (nf > 0) {
        const int n = std::min(nf, VOC_FCAP);
        VCall cl; memset(&cl, 0, sizeof(cl));
        cl.ns = 1; cl.nf = n; cl.slot[0] = slot; cl.pos[0] = v->frames_done[slot];
        VTRY(voc_call(e, cl, s));
        v->frames_done[slot] += n; nf -= n;
    }
    if (is_last) v->last_flag[slot] = 1;
    return Q3TTS_OK;
}
// batched variant: the same nf (<= VOC_FCAP) new frames for every listed slot
int q3_voc_decode_batch(q3tts_engine* e, const int* slots, const int* real, int ns, int nf, hipStream_t s, int beside_decoder) {
    Q3Voc* v = e->voc;
    if (!v) return q3_set_err(e, Q3TTS_ERR_STATE, "engine has no vocoder");
    if (ns <= 0 || ns > VOC_MAX_NS || nf <= 0 || nf > VOC_FCAP) return q3_set_err(e, Q3TTS_ERR_INVALID, "vocoder batch shape");
    VCall cl; memset(&cl, 0, sizeof(cl));
    cl.ns = ns; cl.nf = nf;
    for (int i = 0; i < ns; ++i) { cl.slot[i] = slots[i]; cl.pos[i] = v->frames_done[slots[i]]; }
    g_voc_polite_now = beside_decoder && ns >= 16;  // (fewer slots: few and short-lived workgroups — greedy launches, the least latency)
    const int rc = voc_call(e, cl, s);
    g_voc_polite_now = false;
    VTRY(rc);
    for (int i = 0; i < ns; ++i) v->frames_done[slots[i]] += real ? real[i] : nf;
    return Q3TTS_OK;
}
void q3_voc_mark_last(q3tts_engine* e, int slot) { if (e->voc) e->voc->last_flag[slot] = 1; }

float* q3_voc_pcm(q3tts_engine* e, int slot) { return e->voc->pcm + (size_t)slot * e->voc->pcm_stride; }
// V4: frames are withheld by lookahead_frames until more input arrives or the slot is flushed with is_last
int q3_voc_samples(q3tts_engine* e, int slot) {
    Q3Voc* v = e->voc;
    int fr = v->frames_done[slot];
    if (!v->last_flag[slot]) fr = std::max(0, fr - v->c.lookahead_frames);
    return fr * v->spf;
}

extern "C" int q3tts_k_vocoder(q3tts_engine* e, const int32_t* codes, int32_t n_frames, int32_t chunk_frames, float* pcm_out, int32_t* n_samples_out) {
    if (!e || !codes || !pcm_out || !n_samples_out || n_frames <= 0) return q3_set_err(e, Q3TTS_ERR_INVALID, "null argument");
    if (!e->voc) return q3_set_err(e, Q3TTS_ERR_STATE, "engine created with with_vocoder = 0");
    if (n_frames > e->cfg.max_steps_cap) return q3_set_err(e, Q3TTS_ERR_INVALID, "n_frames exceeds max_steps_cap");
    Q3_HIP(e, hipSetDevice(e->cfg.device));
    const int ncb = e->cfg.model.n_codebooks;
    hipStream_t s = e->stream;
    Q3_HIP(e, hipMemcpyAsync(e->codes, codes, sizeof(int32_t) * (size_t)n_frames * ncb, hipMemcpyHostToDevice, s));  // slot 0
    VTRY(q3_voc_reset(e, 0));
    const int step = chunk_frames > 0 ? chunk_frames : n_frames;
    for (int f = 0; f < n_frames; f += step) {
        const int n = std::min(step, n_frames - f);
        VTRY(q3_voc_decode(e, 0, f, n, f + n >= n_frames, s));
    }
    const int ns = q3_voc_samples(e, 0);
    Q3_HIP(e, hipMemcpyAsync(pcm_out, q3_voc_pcm(e, 0), sizeof(float) * (size_t)ns, hipMemcpyDeviceToHost, s));
    Q3_HIP(e, hipStreamSynchronize(s));
    *n_samples_out = ns;
    return Q3TTS_OK;
}

// Measurement hook (bench.py's roofline_vocoder): the batched vocoder alone — n_slots slots x 4-frame chunks, `chunks` calls on the
// engine's stream with nothing else on the GPU, HIP events around them. Codes are seeded pseudo-random. *ms_per_chunk = the mean
// duration of one batched 4-frame call (n_slots x 4 frames of PCM).
#if defined(Q3_STAMPS) || defined(Q3_VOC_STAMPS)
static void voc_print_stamps() {
    unsigned long long st[2][4][16];
    if (hipMemcpyFromSymbol(st, HIP_SYMBOL(g_voc_stamps), sizeof(st)) != hipSuccess) return;
    for (int k = 0; k < 2; ++k)
        for (int w = 0; w < 3; ++w)
            fprintf(stderr, "resunit stamps (%s) wg %d (100 MHz ticks from entry): first weights requested %llu | input staged + barrier %llu | conv1 done %llu | snake tile done %llu | conv2 done %llu | stores issued %llu\n",
                    k == 0 ? "192 channels" : "96 channels", w, st[k][w][1] - st[k][w][0], st[k][w][2] - st[k][w][0], st[k][w][3] - st[k][w][0], st[k][w][4] - st[k][w][0], st[k][w][5] - st[k][w][0], st[k][w][6] - st[k][w][0]);
}
#endif
extern "C" int q3tts_k_vocoder_bench(q3tts_engine* e, int32_t n_slots, int32_t chunks, float* ms_per_chunk) {
    if (!e || !ms_per_chunk || n_slots <= 0 || chunks <= 0) return q3_set_err(e, Q3TTS_ERR_INVALID, "vocoder bench: bad argument");
    if (!e->voc) return q3_set_err(e, Q3TTS_ERR_STATE, "engine created with with_vocoder = 0");
    if (n_slots > e->B || n_slots > VOC_MAX_NS || (chunks + 2) * 4 > e->cfg.max_steps_cap) return q3_set_err(e, Q3TTS_ERR_INVALID, "vocoder bench: shape exceeds the engine's");
    Q3_HIP(e, hipSetDevice(e->cfg.device));
    const int ncb = e->cfg.model.n_codebooks, cap = e->cfg.max_steps_cap, cbs = e->cfg.vocoder.codebook_size;
    hipStream_t s = e->stream;
    std::vector<int32_t> codes((size_t)n_slots * cap * ncb);
    uint64_t st = 0x9E3779B97F4A7C15ULL;
    for (auto& c : codes) { st = q3_mix64(st + 0x1234567); c = (int32_t)(st % (uint64_t)cbs); }
    Q3_HIP(e, hipMemcpyAsync(e->codes, codes.data(), codes.size() * 4, hipMemcpyHostToDevice, s));
    Q3_HIP(e, hipStreamSynchronize(s));
    std::vector<int> slots(n_slots);
    for (int i = 0; i < n_slots; ++i) { slots[i] = i; VTRY(q3_voc_reset(e, i)); }
    for (int w = 0; w < 2; ++w) VTRY(q3_voc_decode_batch(e, slots.data(), nullptr, n_slots, 4, s, 0));  // warm-up (and fills the sliding window)
    Q3_HIP(e, hipEventRecord(e->ev0, s));
    for (int c = 0; c < chunks; ++c) VTRY(q3_voc_decode_batch(e, slots.data(), nullptr, n_slots, 4, s, 0));
    Q3_HIP(e, hipEventRecord(e->ev2, s));
    Q3_HIP(e, hipStreamSynchronize(s));
    float ms = 0.0f;
    Q3_HIP(e, hipEventElapsedTime(&ms, e->ev0, e->ev2));
    *ms_per_chunk = ms / (float)chunks;
#if defined(Q3_STAMPS) || defined(Q3_VOC_STAMPS)
    voc_print_stamps();
#endif
#ifdef Q3_VOC_STAMPS
    {
        unsigned long long st[4][8];
        if (hipMemcpyFromSymbol(st, HIP_SYMBOL(g_ring_stamps), sizeof(st)) == hipSuccess)
            for (int w = 0; w < 3; ++w)
                fprintf(stderr, "ring<3> stamps wg %d (10 ns ticks from entry): prologue issued %llu | first stage in LDS %llu | main loop done %llu | tile in LDS %llu | stores issued %llu | stores done %llu\n",
                        w, st[w][1] - st[w][0], st[w][2] - st[w][0], st[w][3] - st[w][0], st[w][4] - st[w][0], st[w][5] - st[w][0], st[w][6] - st[w][0]);
    }
#endif
    for (int i = 0; i < n_slots; ++i) VTRY(q3_voc_reset(e, i));
    return Q3TTS_OK;
}
