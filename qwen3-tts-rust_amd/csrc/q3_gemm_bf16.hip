// q3_gemm_bf16.hip — the canonical bf16-MFMA GEMM: GEMMs on v_mfma_f32_16x16x32_bf16 that stay bit-exact
// against the CPU oracle because the instruction's accumulation arithmetic is restated there
// (oracle/q3_oracle.c::q3o_mfma_bf16_dot32, pinned by tests and by hardware golden vectors).
//   k_gemm_bf16_norm_swiglu<..., EPI>   IN THE ENGINE: the Predictor's norm-fused GEMMs — gate/up (SwiGLU epilogue) and QKV (store)
//   k_gemm_bf16<...>                    prototype behind q3tts_k_gemm_bf16: plain GEMM on bf16 activation rows (the form O / down
//                                       take once their producers emit bf16)
//
// Canonical order: y[b][n] = ((((s_0 + s_1) + s_2) + ...) + s_7), s_w = the chain of MFMA steps over K-slice w (K/8
// contiguous columns, 32 per instruction, ascending), each instruction consuming its 32 products lane group by lane group:
// group g holds k = 4g..4g+3 and 16+4g..16+4g+3 of the 32-block — exactly what the tiled weight layout of DESIGN.md §2.1
// puts into one lane, so weights are shared with the exact path. Activations arrive already rounded to bf16.
// Workgroup = 8 waves = 8 K-slices of one 32 x 48 output tile (160 B of operands per k); a wave issues every
// load of its slice before its first MFMA, so all of a CU's operand bytes are in flight at once.
#include "q3_kernels.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int RT, int NT, int PER, bool RESID>
__global__ __launch_bounds__(512) void k_gemm_bf16(const uint16_t* __restrict__ x, int ldx, int B, const uint4* __restrict__ w, int K, int N,
                                                   float* __restrict__ y, int ldy) {
    extern __shared__ float part[];  // [8 waves][RT*NT*4 regs][64 lanes]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, kq = lane >> 4, r = lane & 15;
    const int nb0 = blockIdx.x * NT, row0 = blockIdx.y * RT * 16, kblocks = K >> 5, kb0 = wave * PER;
    uint2 alo[PER][RT], ahi[PER][RT]; uint4 bq[PER][NT];
#pragma unroll
    for (int s = 0; s < PER; ++s) {
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            const int row = min(row0 + 16 * i + r, B - 1);
            const uint16_t* p = x + (size_t)row * ldx + (size_t)(kb0 + s) * 32 + 4 * kq;
            alo[s][i] = *(const uint2*)p; ahi[s][i] = *(const uint2*)(p + 16);
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) bq[s][j] = w[((size_t)(nb0 + j) * kblocks + kb0 + s) * 64 + lane];
    }
    __builtin_amdgcn_sched_barrier(0);  // every load of the slice is issued before the first MFMA (the scheduler would otherwise sink them)
    f32x4 acc[RT][NT];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < PER; ++s) {
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            union { uint4 u; bf16x8 v; } a; a.u = make_uint4(alo[s][i].x, alo[s][i].y, ahi[s][i].x, ahi[s][i].y);
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                union { uint4 u; bf16x8 v; } b; b.u = bq[s][j];
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, acc[i][j], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) part[((size_t)wave * (RT * NT * 4) + (i * NT + j) * 4 + e) * 64 + lane] = acc[i][j][e];
    __syncthreads();
    // slice partials combined in slice order; element (tile t, reg e, lane l) -> row 16 i + 4 (l >> 4) + e, col 16 j + (l & 15)
    for (int o = threadIdx.x; o < RT * NT * 4 * 64; o += 512) {
        float v = part[o];
#pragma unroll
        for (int wv = 1; wv < 8; ++wv) v = v + part[(size_t)wv * (RT * NT * 4 * 64) + o];
        const int l = o & 63, e = (o >> 6) & 3, t = o >> 8, i = t / NT, j = t - i * NT;
        const int row = row0 + 16 * i + 4 * (l >> 4) + e;
        if (row < B) {
            float* yp = y + (size_t)row * ldy + (size_t)(nb0 + j) * 16 + (l & 15);
            *yp = RESID ? *yp + v : v;
        }
    }
}

// x: bf16 bits [B][ldx]; w: tiled bf16; K % 256 == 0 and K / 256 in {4, 8} (K = 1024, 2048), N % 48 == 0
int q3_launch_gemm_bf16(const uint16_t* x, int ldx, int B, const uint4* w, int K, int N, float* y, int ldy, hipStream_t s) {
    if (B < 1 || N % 48 || (K != 1024 && K != 2048)) return -1;
    const dim3 grid(N / 48, (B + 31) / 32);
    const size_t lds = (size_t)8 * 2 * 3 * 4 * 64 * 4;
    if (K == 1024) hipLaunchKernelGGL((k_gemm_bf16<2, 3, 4, false>), grid, dim3(512), lds, s, x, ldx, B, w, K, N, y, ldy);
    else hipLaunchKernelGGL((k_gemm_bf16<2, 3, 8, false>), grid, dim3(512), lds, s, x, ldx, B, w, K, N, y, ldy);
    return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// The same order with the fused RMSNorm prologue and the SwiGLU epilogue of the Predictor's gate/up GEMM (K = 1024):
//   xn[k] = bf16(x[k] * nw[k]);  raw[n] = canonical bf16 GEMM of xn;  y[n] = s_r * raw[n];  out = swiglu(y_gate, y_up)
//   ss_r: per (slice w, lane group kq) an fmaf chain over the lane's own k (ascending), S_w = (c0 + c1) + (c2 + c3),
//   ss = S_0 + ... + S_7, s_r = 1 / sqrtf(ss / K + eps)                                  
// The A operand is still the f32 residual stream (the first integration step of §16): converted in the kernel.
// ---------------------------------------------------------------------------------------------------------------
template <int RT, int NT, int PER, int EPI>  // EPI: Q3_EPI_SWIGLU, or Q3_EPI_STORE (y = s_r * raw: the Predictor's QKV)
__global__ __launch_bounds__(512) void k_gemm_bf16_norm_swiglu(const float* __restrict__ x, int ldx, int B, const uint4* __restrict__ w, int K, int N,
                                                               const float* __restrict__ nw, float eps, float* __restrict__ y, int ldy, int y_bf16) {
    extern __shared__ float part[];  // [8][RT*NT*4][64] partial tiles, then [8][4][RT*16] ss partials, then [RT*16] row scales
    float* ssp = part + (size_t)8 * RT * NT * 4 * 64;
    float* srow = ssp + 8 * 4 * RT * 16;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, kq = lane >> 4, r = lane & 15;
    const int nb0 = blockIdx.x * NT, row0 = blockIdx.y * RT * 16, kblocks = K >> 5, kb0 = wave * PER;
    f32x4 a0[PER][RT], a1[PER][RT], n0[PER], n1[PER]; u32x4 bq[PER][NT];
#pragma unroll
    for (int s = 0; s < PER; ++s) {
        const size_t kk = (size_t)(kb0 + s) * 32 + 4 * kq;
        n0[s] = *(const f32x4*)(nw + kk); n1[s] = *(const f32x4*)(nw + kk + 16);
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            const int row = min(row0 + 16 * i + r, B - 1);
            const float* p = x + (size_t)row * ldx + kk;
            a0[s][i] = *(const f32x4*)p; a1[s][i] = *(const f32x4*)(p + 16);
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) bq[s][j] = ((const u32x4*)w)[((size_t)(nb0 + j) * kblocks + kb0 + s) * 64 + lane];
    }
    // every operand of the slice is requested before anything is consumed: the empty asm statements pin the loaded values
    // here (without them the compiler sinks half of the loads between the MFMAs and the slice pays several round trips)
#pragma unroll
    for (int s = 0; s < PER; ++s) {
        asm volatile("" : "+v"(n0[s]), "+v"(n1[s]));
#pragma unroll
        for (int i = 0; i < RT; ++i) asm volatile("" : "+v"(a0[s][i]), "+v"(a1[s][i]));
#pragma unroll
        for (int j = 0; j < NT; ++j) asm volatile("" : "+v"(bq[s][j]));
    }
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc[RT][NT];
    float ss[RT];
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        ss[i] = 0.0f;
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int s = 0; s < PER; ++s) {
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            const f32x4 u = a0[s][i], v = a1[s][i];
            ss[i] = fmaf(u.x, u.x, ss[i]); ss[i] = fmaf(u.y, u.y, ss[i]); ss[i] = fmaf(u.z, u.z, ss[i]); ss[i] = fmaf(u.w, u.w, ss[i]);
            ss[i] = fmaf(v.x, v.x, ss[i]); ss[i] = fmaf(v.y, v.y, ss[i]); ss[i] = fmaf(v.z, v.z, ss[i]); ss[i] = fmaf(v.w, v.w, ss[i]);
            bf16x8 a;
            a[0] = (__bf16)(u.x * n0[s].x); a[1] = (__bf16)(u.y * n0[s].y); a[2] = (__bf16)(u.z * n0[s].z); a[3] = (__bf16)(u.w * n0[s].w);
            a[4] = (__bf16)(v.x * n1[s].x); a[5] = (__bf16)(v.y * n1[s].y); a[6] = (__bf16)(v.z * n1[s].z); a[7] = (__bf16)(v.w * n1[s].w);
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                union { u32x4 u4; bf16x8 v8; } b; b.u4 = bq[s][j];
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b.v8, acc[i][j], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        ssp[(wave * 4 + kq) * (RT * 16) + 16 * i + r] = ss[i];
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) part[((size_t)wave * (RT * NT * 4) + (i * NT + j) * 4 + e) * 64 + lane] = acc[i][j][e];
    }
    __syncthreads();
    if (threadIdx.x < RT * 16) {
        float tot = 0.0f;
#pragma unroll
        for (int wv = 0; wv < 8; ++wv) {
            const float* c = ssp + (size_t)wv * 4 * (RT * 16) + threadIdx.x;
            const float S = (c[0] + c[RT * 16]) + (c[2 * RT * 16] + c[3 * RT * 16]);
            tot = wv == 0 ? S : tot + S;
        }
        srow[threadIdx.x] = 1.0f / sqrtf(tot / (float)K + eps);
    }
    __syncthreads();
    for (int o = threadIdx.x; o < RT * NT * 4 * 64; o += 512) {
        const int l = o & 63, e = (o >> 6) & 3, t = o >> 8, i = t / NT, j = t - i * NT;
        const int rl = 16 * i + 4 * (l >> 4) + e, row = row0 + rl;
        const float sc = srow[rl];
        if (EPI == Q3_EPI_SWIGLU) {
            // gate lanes (column < 8 of a tile) finish one output each: swiglu(s * gate, s * up), up = the same row 8 columns further
            if ((l & 15) >= 8) continue;
            float g = part[o], u = part[o + 8];
#pragma unroll
            for (int wv = 1; wv < 8; ++wv) { g = g + part[(size_t)wv * (RT * NT * 4 * 64) + o]; u = u + part[(size_t)wv * (RT * NT * 4 * 64) + o + 8]; }
            if (row < B) {
                const size_t oi = (size_t)row * ldy + (size_t)(nb0 + j) * 8 + (l & 15);
                const float val = q3_swiglu(sc * g, sc * u);
                if (y_bf16) ((uint16_t*)y)[oi] = q3_bf16(val); else y[oi] = val;  // bf16 when the consumer is the bf16 down projection
            }
        } else {
            float v = part[o];
#pragma unroll
            for (int wv = 1; wv < 8; ++wv) v = v + part[(size_t)wv * (RT * NT * 4 * 64) + o];
            if (row < B) y[(size_t)row * ldy + (size_t)(nb0 + j) * 16 + (l & 15)] = sc * v;
        }
    }
}

// shapes this path takes (a function of the model configuration only, never of the row count: the oracle applies the same rule)
bool q3_gemm_bf16_norm_ok(int K, int N) { return (K == 512 || K == 1024) && N % 32 == 0; }
bool q3_gemm_bf16_norm_swiglu_ok(int K, int N) { return q3_gemm_bf16_norm_ok(K, N); }

template <int NT, int PER, int EPI>
static void launch_bf16_n(const float* x, int ldx, int B, const uint4* w, int K, int N, const float* nw, float eps, float* y, int ldy, hipStream_t s, int yb) {
    const dim3 grid(N / (16 * NT), (B + 31) / 32);
    const size_t lds = ((size_t)8 * 2 * NT * 4 * 64 + 8 * 4 * 32 + 32) * 4;
    hipLaunchKernelGGL((k_gemm_bf16_norm_swiglu<2, NT, PER, EPI>), grid, dim3(512), lds, s, x, ldx, B, w, K, N, nw, eps, y, ldy, yb);
}
template <int EPI>
static int launch_bf16_norm(const float* x, int ldx, int B, const uint4* w, int K, int N, const float* nw, float eps, float* y, int ldy, hipStream_t s, int yb) {
    if (B < 1 || !q3_gemm_bf16_norm_ok(K, N)) return -1;
    const int tiles = N / 16, NT = tiles % 3 == 0 ? 3 : (tiles % 2 == 0 ? 2 : 1);
#define L(NT_) do { if (K == 1024) launch_bf16_n<NT_, 4, EPI>(x, ldx, B, w, K, N, nw, eps, y, ldy, s, yb); else launch_bf16_n<NT_, 2, EPI>(x, ldx, B, w, K, N, nw, eps, y, ldy, s, yb); } while (0)
    if (NT == 3) L(3); else if (NT == 2) L(2); else L(1);
#undef L
    return 0;
}
int q3_launch_gemm_bf16_norm_swiglu(const float* x, int ldx, int B, const uint4* w, int K, int N, const float* nw, float eps, float* y, int ldy,
                                    hipStream_t s, int y_bf16) {
    return launch_bf16_norm<Q3_EPI_SWIGLU>(x, ldx, B, w, K, N, nw, eps, y, ldy, s, y_bf16);
}
// y[B][N] = s_r * canonical bf16 GEMM of bf16(x * nw) (fused RMSNorm prologue, plain store): the Predictor's QKV
int q3_launch_gemm_bf16_norm_store(const float* x, int ldx, int B, const uint4* w, int K, int N, const float* nw, float eps, float* y, int ldy,
                                   hipStream_t s) {
    return launch_bf16_norm<Q3_EPI_STORE>(x, ldx, B, w, K, N, nw, eps, y, ldy, s, 0);
}

// residual form for the Predictor's O and down projections: 16 x 16 tiles (N = d_model is small: 64 column tiles x B/16 row chunks
// fill the chip), bf16 rows from the attention / SwiGLU producers, K / 256 steps per wave all in flight
bool q3_gemm_bf16_plain_ok(int K) { return K == 512 || K == 1024 || K == 2048 || K == 3072; }
int q3_launch_gemm_bf16_resid(const uint16_t* x, int ldx, int B, const uint4* w, int K, int N, float* y, int ldy, hipStream_t s) {
    if (B < 1 || N % 16 || !q3_gemm_bf16_plain_ok(K)) return -1;
    const dim3 grid(N / 16, (B + 15) / 16);
    const size_t lds = (size_t)8 * 4 * 64 * 4;
#define L(PER_) hipLaunchKernelGGL((k_gemm_bf16<1, 1, PER_, true>), grid, dim3(512), lds, s, x, ldx, B, w, K, N, y, ldy)
    switch (K >> 8) { case 2: L(2); break; case 4: L(4); break; case 8: L(8); break; default: L(12); break; }
#undef L
    return 0;
}
