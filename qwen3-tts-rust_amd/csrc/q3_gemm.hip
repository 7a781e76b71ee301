// q3_gemm.hip — the exact skinny GEMM of the codec-token decoder (K1/K2/K6/K8/K9, P1-P4, H6 of SURVEY.md §8a).
//
//   y[B][N] = x[B][K] . W[N][K]^T, bf16 weights, f32 accumulate on v_mfma_f32_16x16x4_f32, whose result is
//   bit-for-bit a k-ordered fmaf chain (tools/probe_mfma.hip: 0/256 mismatches at K = 2048).
//
// Canonical order (restated by oracle/q3_oracle.c gemm_t):
//   16 contiguous K-slices; inside a slice, for 32-wide block kb, for t in 0..7, for kq in 0..3:
//   k = kb*32 + (t/4)*16 + kq*4 + (t%4); partials p0..p15 combine as q_w = p_2w + p_2w+1, y = q0+q1+...+q7 in order.
//
// HBM layout of W: tile (nb = n/16, kb = k/32) is 1 KiB; lane l = (kq = l>>4, n = l&15) owns the
// 8 weights of its 8 MFMA steps, element e <-> k = kb*32 + (e/4)*16 + kq*4 + (e%4). A wave-load is 1 KiB contiguous
// and every weight byte is read once per row chunk. With this k map an activation fragment load touches 64
// contiguous bytes per row (lane group kq reads 16 B at kq*16), i.e. whole half-lines.
//
// vmcnt is ONE in-order queue: waiting on a young L2 load also waits for every older HBM load. Hence:
//   k_gemm_small (rows*K <= 12288 floats; the latency-critical B = 1..6 decode case): 16 waves = 16 slices, the
//        workgroup stages x and the norm weights in LDS once (their loads are issued BEFORE the weight stream), so
//        the main loop's vmcnt only ever counts weights.
//   k_gemm_ring (larger batches, MFMA-bound): 8 waves x 2 adjacent slices; x fragments ride a register ring XPF
//        blocks deep issued BEFORE the weight refill of the same block; each wave reuses its x fragments for NT
//        column tiles (x traffic from L2 is the co-bottleneck at B = 64).
//
// Fused RMSNorm: the row scale commutes out of the K-sum, so a NORM GEMM computes
//   y[r][n] = s_r * SUM_canonical((x[r][k] * nw[k]) * W[n][k]),   s_r = 1 / sqrtf(ss_r / K + eps),
// and ss_r is accumulated from the very fragments the MFMAs consume, in the GEMM's own order: per (slice, kq) an
// fmaf chain over ascending k, S_slice = (c0 + c1) + (c2 + c3), Q_w = S_2w + S_2w+1, ss = Q_0 + Q_1 + ... + Q_7.
// No pre-kernel, no second pass over x, no normalised copy in HBM.
#include <cstdlib>

#include "q3_kernels.h"

#ifndef Q3_XCD_MAP
#define Q3_XCD_MAP 1  // -DQ3_XCD_MAP=0: plain blockIdx -> tile mapping (A/B measurements)
#endif

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// once-read weight stream: non-temporal 16-byte load (MI355X_MICROARCH.md, row nt-weights)
__device__ __forceinline__ uint4 ntload16(const uint4* p) {
    const u32x4 v = __builtin_nontemporal_load((const u32x4*)p);
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = v + __shfl_xor(v, m);
    return v;
}
__device__ __forceinline__ void unpack8(const uint4 wv, float* b) {
    b[0] = q3_u2f(wv.x << 16); b[1] = q3_u2f(wv.x & 0xffff0000u);
    b[2] = q3_u2f(wv.y << 16); b[3] = q3_u2f(wv.y & 0xffff0000u);
    b[4] = q3_u2f(wv.z << 16); b[5] = q3_u2f(wv.z & 0xffff0000u);
    b[6] = q3_u2f(wv.w << 16); b[7] = q3_u2f(wv.w & 0xffff0000u);
}
__device__ __forceinline__ float sq4(const float4 v, float acc) {
    acc = fmaf(v.x, v.x, acc); acc = fmaf(v.y, v.y, acc); acc = fmaf(v.z, v.z, acc); acc = fmaf(v.w, v.w, acc);
    return acc;
}

#define XLDS_MAX_FLOATS 12288
#ifdef Q3_STAMPS
#define STAMP(i) do { if (g.dbg && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) g.dbg[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP(i) do { } while (0)
#endif

// shared epilogue: thread (row, col) already holds the canonical sum s of its output element
template <int ROWS, int COLS>
__device__ __forceinline__ void epilogue(const Q3Gemm& g, float* sums, int tid, int nthreads, int row0, int nrows, int col_base) {
    const int epi = g.epi;
    if (epi == Q3_EPI_SWIGLU) {  // each 16-column tile = 8 gate columns then the 8 matching up columns
        __syncthreads();
        for (int o = tid; o < ROWS * (COLS / 2); o += nthreads) {
            const int row = o / (COLS / 2), hc = o - row * (COLS / 2);
            if (row >= nrows) continue;
            const int tile = hc >> 3, c = hc & 7;
            g.y[(size_t)(row0 + row) * g.ldy + (col_base >> 1) + hc] =
                q3_swiglu(sums[row * COLS + tile * 16 + c], sums[row * COLS + tile * 16 + 8 + c]);
        }
    } else if (epi == Q3_EPI_ARGMAX) {
        __syncthreads();
        if (tid < nrows) {
            unsigned long long best = 0;
            for (int c = 0; c < COLS; ++c) {
                const unsigned long long kk = q3_argmax_key(sums[tid * COLS + c], (uint32_t)(col_base + c));
                best = kk > best ? kk : best;
            }
            atomicMax(g.keys + (size_t)(row0 + tid) * g.key_stride, best);
        }
    }
}
__device__ __forceinline__ void store_elem(const Q3Gemm& g, float* sums, float s, size_t grow, int col, int lrow, int lcol, int COLS) {
    if (g.epi == Q3_EPI_STORE) {
        if (g.bias) s = s + g.bias[col];
        g.y[grow * g.ldy + col] = s;
    } else if (g.epi == Q3_EPI_RESID) {
        float* yp = g.y + grow * g.ldy + col;
        *yp = *yp + s;
    } else {
        sums[lrow * COLS + lcol] = s;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// small-batch kernel: 1024 threads, wave w = slice w, one 16-column tile per workgroup. BPS = K/512 (0 = runtime).
// ---------------------------------------------------------------------------------------------------------------
template <bool NORM, int BPS>
__global__ __launch_bounds__(1024) void k_gemm_small(Q3Gemm g) {
    extern __shared__ __attribute__((aligned(16))) float dsm[];  // x[nrows][K] then norm_w[K]
    __shared__ float red[16 * 16 * 17];
    __shared__ float sums[16 * 16];
    __shared__ float ssred[16 * 16];
    constexpr int WPF = 8;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int nb = blockIdx.x, nrows = g.B;
    STAMP(0);
    const int K = g.K, K4 = K >> 2;
    const int bps = BPS > 0 ? BPS : (K >> 9);
    const int kq = lane >> 4, li = lane & 15;
    const uint4* wp = g.w + ((size_t)nb * (K >> 5) + (size_t)wave * bps) * 64 + lane;
    const int koff = wave * (K >> 4) + kq * 4;
    uint4 wq[WPF];
    f32x4 acc = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
    // stage raw activations (contiguous rows: ldx == K) and norm weights; these loads are the OLDEST in the queue
    const int n4 = nrows * K4;  // <= 3072 float4
    float* nwl = dsm + (size_t)nrows * K;
    const float4* xsrc = (const float4*)g.x;
    // (named scalars, not arrays: hipcc keeps a load-now/store-later float4 array in scratch)
    const float4 xv0 = xsrc[min(tid, n4 - 1)], xv1 = xsrc[min(tid + 1024, n4 - 1)], xv2 = xsrc[min(tid + 2048, n4 - 1)];
    const float4* nsrc = (const float4*)(NORM ? g.norm_w : g.x);
    const float4 nv0 = nsrc[min(tid, K4 - 1)], nv1 = nsrc[min(tid + 1024, K4 - 1)];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < WPF; ++j)
        if (j < bps) wq[j] = ntload16(wp + (size_t)j * 64);
    __builtin_amdgcn_sched_barrier(0);
    if (tid < n4) ((float4*)dsm)[tid] = xv0;
    if (tid + 1024 < n4) ((float4*)dsm)[tid + 1024] = xv1;
    if (tid + 2048 < n4) ((float4*)dsm)[tid + 2048] = xv2;
    if (NORM) {
        if (tid < K4) ((float4*)nwl)[tid] = nv0;
        if (tid + 1024 < K4) ((float4*)nwl)[tid + 1024] = nv1;
    }
    STAMP(1);
    __syncthreads();
    STAMP(2);
    const int lr = li < nrows ? li : nrows - 1;  // padding rows replicate the last row; their results are dropped
    const float* xa = dsm + (size_t)lr * K + koff;
    const float* nwa = nwl + koff;
    float ss = 0.0f;  // NORM: this lane's (slice, kq) chain of the row's sum of squares
    for (int kb0 = 0; kb0 < bps; kb0 += WPF) {
#pragma unroll
        for (int j = 0; j < WPF; ++j) {
            const int kb = kb0 + j;
            if (kb < bps) {
                float b[8];
                unpack8(wq[j], b);
                __builtin_amdgcn_sched_barrier(0);
                if (kb + WPF < bps) wq[j] = ntload16(wp + (size_t)(kb + WPF) * 64);
                __builtin_amdgcn_sched_barrier(0);
                const float4 x0 = *(const float4*)(xa + kb * 32), x1 = *(const float4*)(xa + kb * 32 + 16);
                float a[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
                if (NORM) {
                    const float4 n0 = *(const float4*)(nwa + kb * 32), n1 = *(const float4*)(nwa + kb * 32 + 16);
                    const float nw[8] = {n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, n1.z, n1.w};
                    ss = sq4(x0, ss); ss = sq4(x1, ss);
#pragma unroll
                    for (int t = 0; t < 8; ++t) a[t] = a[t] * nw[t];
                }
#pragma unroll
                for (int t = 0; t < 8; ++t) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b[t], acc, 0, 0, 0);
            }
        }
    }
    STAMP(3);
    // D layout: lane holds rows 4*(lane>>4)+j, column lane&15
#pragma unroll
    for (int j = 0; j < 4; ++j) red[(wave * 16 + 4 * kq + j) * 17 + li] = acc[j];
    if (NORM) {
        ss = ss + __shfl_xor(ss, 16);
        ss = ss + __shfl_xor(ss, 32);  // (c0 + c1) + (c2 + c3): this slice's share of row li
        if (kq == 0) ssred[wave * 16 + li] = ss;
    }
    __syncthreads();
    STAMP(4);
    if (tid < 256) {
        const int row = tid >> 4, col = tid & 15;
        if (row < nrows) {
            float s = red[row * 17 + col] + red[(16 + row) * 17 + col];
#pragma unroll
            for (int w = 1; w < 8; ++w) s = s + (red[(2 * w * 16 + row) * 17 + col] + red[((2 * w + 1) * 16 + row) * 17 + col]);
            if (NORM) {
                float tot = ssred[row] + ssred[16 + row];
#pragma unroll
                for (int w = 1; w < 8; ++w) tot = tot + (ssred[2 * w * 16 + row] + ssred[(2 * w + 1) * 16 + row]);
                s = s * (1.0f / sqrtf(tot / (float)K + g.eps));
            }
            store_elem(g, sums, s, (size_t)row, nb * 16 + col, row, col, 16);
        }
    }
    STAMP(5);
    epilogue<16, 16>(g, sums, tid, 1024, 0, nrows, nb * 16);
    STAMP(6);
}

// ---------------------------------------------------------------------------------------------------------------
// ring kernel: 512 threads, wave w = slices 2w and 2w+1 (adjacent in k), RT row tiles x NT column tiles per wave.
// BPS = K/512 blocks per slice (0 = runtime).
// ---------------------------------------------------------------------------------------------------------------
template <int RT, int NT, int BPS, bool NORM>
__global__ __launch_bounds__(512, (RT <= 2 && NT == 1 && BPS > 0 && BPS <= 2) ? 4 : 2) void k_gemm_ring(Q3Gemm g) {
    extern __shared__ __attribute__((aligned(16))) float nwl[];  // NORM: norm_w[K]
    __shared__ float red[8 * RT * 16 * (NT * 16 + 1)];
    __shared__ float sums[RT * 16 * NT * 16];
    __shared__ float ssred[NORM ? 8 * RT * 16 : 1];
    __shared__ float srow[NORM ? RT * 16 : 1];
    constexpr int WPF = NT == 1 ? 8 : 4;
    constexpr int XPF0 = (RT == 4 || (NORM && RT * NT >= 6)) ? 2 : (RT == 2 ? 4 : 8);  // NORM keeps a[] apart from the ring: shallower ring for the widest tile
    constexpr int XPF = XPF0 < WPF ? XPF0 : WPF;  // the ring slot of block kb is kb % XPF == j % XPF only if XPF divides WPF
    constexpr int CP = NT * 16 + 1;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // XCD-aware tile mapping: workgroups are dealt round-robin to the 8 XCDs (each with its own L2) in linear-id order. With two
    // row chunks the two workgroups that read the same weight columns are made neighbours in time on the SAME XCD (linear ids 8
    // apart), so the second one finds the weight tiles in that L2 instead of fetching them again a whole dispatch round later.
    int nbt_ = blockIdx.x, rc_ = blockIdx.y;
    if (gridDim.y == 2 && (gridDim.x & 7) == 0 && Q3_XCD_MAP) {
        const int id = blockIdx.x + gridDim.x * blockIdx.y;
        rc_ = (id >> 3) & 1; nbt_ = (id & 7) + ((id >> 4) << 3);
    }
    const int nbt = nbt_, row0 = rc_ * (RT * 16);
    STAMP(0);
    const int nrows = min(RT * 16, g.B - row0);
    const int K = g.K;
    const int bps = BPS > 0 ? BPS : (K >> 9);
    const int nblk = 2 * bps;  // this wave's blocks (two adjacent slices)
    const int kq = lane >> 4, li = lane & 15;
    const size_t tile_stride = (size_t)(K >> 5) * 64;
    const uint4* wp = g.w + (size_t)nbt * NT * tile_stride + (size_t)wave * nblk * 64 + lane;
    const int koff = wave * (K >> 3) + kq * 4;
    uint4 wq[WPF][NT];
    float4 xq[XPF][RT][2];
    f32x4 acc[RT][NT], accA[RT][NT];
    float ss[RT], ssA[RT];  // NORM: (slice, kq) chains of the rows' sums of squares
    const float* xr[RT];
    // NORM: every wave parks the norm weights of ITS OWN k range (K/8 floats) in a wave-private LDS strip: no workgroup
    // barrier in the prologue (writer and reader are the same wave); these loads are the OLDEST in the queue
    const int KW4 = K >> 5;  // float4 per wave
    const float4* nsrc = (const float4*)(NORM ? g.norm_w + wave * (K >> 3) : g.x);
    float4 nv0, nv1, nv2, nv3;
    if (NORM) {
        nv0 = nsrc[min(lane, KW4 - 1)]; nv1 = nsrc[min(lane + 64, KW4 - 1)];
        nv2 = nsrc[min(lane + 128, KW4 - 1)]; nv3 = nsrc[min(lane + 192, KW4 - 1)];
    }
    // RESID: the residual operand is fetched up front instead of at the very end
    constexpr int NOUT = (RT * 16 * NT * 16 + 511) / 512;
    float yres[NOUT];
    if (g.epi == Q3_EPI_RESID) {
#pragma unroll
        for (int i = 0; i < NOUT; ++i) {
            const int o = tid + i * 512;
            const int row = o / (NT * 16), col = o - row * (NT * 16);
            yres[i] = (o < RT * 16 * NT * 16 && row < nrows) ? g.y[(size_t)(row0 + row) * g.ldy + nbt * NT * 16 + col] : 0.0f;
        }
    }
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        int lrw = r * 16 + li;
        if (lrw >= nrows) lrw = nrows - 1;  // padding rows replicate the last row; their results are dropped
        xr[r] = g.x + (size_t)(row0 + lrw) * g.ldx + koff;
#pragma unroll
        for (int c = 0; c < NT; ++c) { acc[r][c] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f}; accA[r][c] = acc[r][c]; }
        ss[r] = 0.0f; ssA[r] = 0.0f;
    }
#pragma unroll
    for (int j = 0; j < XPF; ++j)
        if (j < nblk) {
#pragma unroll
            for (int r = 0; r < RT; ++r) { xq[j][r][0] = *(const float4*)(xr[r] + j * 32); xq[j][r][1] = *(const float4*)(xr[r] + j * 32 + 16); }
        }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < WPF; ++j)
        if (j < nblk) {
#pragma unroll
            for (int c = 0; c < NT; ++c) wq[j][c] = ntload16(wp + c * tile_stride + (size_t)j * 64);
        }
    __builtin_amdgcn_sched_barrier(0);
    float* nww = nwl + wave * (K >> 3);
    if (NORM) {
        if (lane < KW4) ((float4*)nww)[lane] = nv0;
        if (lane + 64 < KW4) ((float4*)nww)[lane + 64] = nv1;
        if (lane + 128 < KW4) ((float4*)nww)[lane + 128] = nv2;
        if (lane + 192 < KW4) ((float4*)nww)[lane + 192] = nv3;  // K <= 8192
    }
    const float* nwa = nww + kq * 4;
    STAMP(1);
    for (int kb0 = 0; kb0 < nblk; kb0 += WPF) {
#pragma unroll
        for (int j = 0; j < WPF; ++j) {
            const int kb = kb0 + j;
            if (kb < nblk) {
                if (kb == bps) {  // slice 2w done: park its partial, start slice 2w+1 from +0
#pragma unroll
                    for (int r = 0; r < RT; ++r) {
#pragma unroll
                        for (int c = 0; c < NT; ++c) { accA[r][c] = acc[r][c]; acc[r][c] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f}; }
                        ssA[r] = ss[r]; ss[r] = 0.0f;
                    }
                }
                float b[NT][8];
#pragma unroll
                for (int c = 0; c < NT; ++c) unpack8(wq[j][c], b[c]);
                float a[RT][8];
#pragma unroll
                for (int r = 0; r < RT; ++r) {
                    const float4 x0 = xq[j % XPF][r][0], x1 = xq[j % XPF][r][1];
                    a[r][0] = x0.x; a[r][1] = x0.y; a[r][2] = x0.z; a[r][3] = x0.w; a[r][4] = x1.x; a[r][5] = x1.y; a[r][6] = x1.z; a[r][7] = x1.w;
                    if (NORM) { ss[r] = sq4(x0, ss[r]); ss[r] = sq4(x1, ss[r]); }
                }
                if (NORM) {
                    const float4 n0 = *(const float4*)(nwa + kb * 32), n1 = *(const float4*)(nwa + kb * 32 + 16);
                    const float nw[8] = {n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, n1.z, n1.w};
#pragma unroll
                    for (int r = 0; r < RT; ++r)
#pragma unroll
                        for (int t = 0; t < 8; ++t) a[r][t] = a[r][t] * nw[t];
                }
                __builtin_amdgcn_sched_barrier(0);
                if (kb + XPF < nblk) {  // activations first: they must stay OLDER than the weight refill below
#pragma unroll
                    for (int r = 0; r < RT; ++r) {
                        xq[j % XPF][r][0] = *(const float4*)(xr[r] + (kb + XPF) * 32);
                        xq[j % XPF][r][1] = *(const float4*)(xr[r] + (kb + XPF) * 32 + 16);
                    }
                }
                if (kb + WPF < nblk) {
#pragma unroll
                    for (int c = 0; c < NT; ++c) wq[j][c] = ntload16(wp + c * tile_stride + (size_t)(kb + WPF) * 64);
                }
                __builtin_amdgcn_sched_barrier(0);
#ifdef Q3_STAMPS
                if (kb == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); STAMP(2); }
#endif
                // independent (row tile, column tile) chains interleave; each chain still sees t = 0..7 in order
#pragma unroll
                for (int t = 0; t < 8; ++t)
#pragma unroll
                    for (int r = 0; r < RT; ++r)
#pragma unroll
                        for (int c = 0; c < NT; ++c) acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r][t], b[c][t], acc[r][c], 0, 0, 0);
            }
        }
    }
    STAMP(3);
    // q_w = p_2w + p_2w+1, then the eight q meet in LDS and are summed in order
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int c = 0; c < NT; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) red[((wave * RT + r) * 16 + 4 * kq + j) * CP + c * 16 + li] = accA[r][c][j] + acc[r][c][j];
    if (NORM) {
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            float sa = ssA[r] + __shfl_xor(ssA[r], 16), sb = ss[r] + __shfl_xor(ss[r], 16);
            sa = sa + __shfl_xor(sa, 32); sb = sb + __shfl_xor(sb, 32);  // S_2w, S_2w+1 = (c0 + c1) + (c2 + c3)
            if (kq == 0) ssred[(wave * RT + r) * 16 + li] = sa + sb;
        }
    }
    __syncthreads();
    STAMP(4);
    if (NORM) {
        if (tid < RT * 16) {
            float tot = ssred[tid];
#pragma unroll
            for (int w = 1; w < 8; ++w) tot = tot + ssred[w * RT * 16 + tid];
            srow[tid] = 1.0f / sqrtf(tot / (float)K + g.eps);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < NOUT; ++i) {
        const int o = tid + i * 512;
        const int row = o / (NT * 16), col = o - row * (NT * 16);
        if (o >= RT * 16 * NT * 16 || row >= nrows) continue;
        float s = red[row * CP + col];
#pragma unroll
        for (int w = 1; w < 8; ++w) s = s + red[(w * RT * 16 + row) * CP + col];
        if (NORM) s = s * srow[row];
        if (g.epi == Q3_EPI_RESID) g.y[(size_t)(row0 + row) * g.ldy + nbt * NT * 16 + col] = yres[i] + s;
        else store_elem(g, sums, s, (size_t)(row0 + row), nbt * NT * 16 + col, row, col, NT * 16);
    }
    STAMP(5);
    epilogue<RT * 16, NT * 16>(g, sums, tid, 512, row0, nrows, nbt * NT * 16);
#ifdef Q3_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    STAMP(6);
}

// ---------------------------------------------------------------------------------------------------------------
// fat-wave kernel: 256 threads = 4 waves, ONE wave per SIMD with up to 512 VGPRs, wave w = the four adjacent K-slices
// 4w .. 4w+3. The ring kernel's two waves per SIMD take turns rather than interleave and each is limited by what the
// CU's load path delivers per MFMA (profiles/README.md); a wave that owns a 4-6x larger output tile (RT x NT up to 2 x 6 /
// 4 x 4) does 3-4x the MFMA work per loaded byte and per ring slot, so the same rings now run ahead of the MFMA pipe.
// Same canonical order: per slice one fmaf chain, q_2w = p_4w + p_4w+1, q_2w+1 = p_4w+2 + p_4w+3, y = q0 + ... + q7.
// ---------------------------------------------------------------------------------------------------------------
template <int RT, int NT, bool NORM>
__global__ __launch_bounds__(256, 1) void k_gemm_fat(Q3Gemm g) {
    extern __shared__ __attribute__((aligned(16))) float dsm[];  // red [8][RT*16][NT*16+1] | sums [RT*16][NT*16] | nwl [K]
    __shared__ float ssred[NORM ? 8 * RT * 16 : 1];
    __shared__ float srow[NORM ? RT * 16 : 1];
    constexpr int WPF = 2;  // (register budget: the tile's accumulators and their parked copies come first)
    constexpr int XPF = 2;
    constexpr int CP = NT * 16 + 1;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int nbt = blockIdx.x, row0 = blockIdx.y * (RT * 16);
    STAMP(0);
    const int nrows = min(RT * 16, g.B - row0);
    const int K = g.K, bps = K >> 9, nblk = 4 * bps;
    const int kq = lane >> 4, li = lane & 15;
    const size_t tile_stride = (size_t)(K >> 5) * 64;
    const uint4* wp = g.w + (size_t)nbt * NT * tile_stride + (size_t)wave * nblk * 64 + lane;
    const int koff = wave * (K >> 2) + kq * 4;
    float* red = dsm;
    float* sums = red + 8 * RT * 16 * CP;
    float* nwl = sums + RT * 16 * NT * 16;
    uint4 wq[WPF][NT];
    float4 xq[XPF][RT][2];
    f32x4 acc[RT][NT], accA[RT][NT];
    float ss[RT], ssA[RT], qs0[RT];
    const float* xr[RT];
    const int KW4 = K >> 4;  // float4 of norm weights per wave (K/4 floats)
    const float4* nsrc = (const float4*)(NORM ? g.norm_w + wave * (K >> 2) : g.x);
    float* nww = nwl + wave * (K >> 2);
    if (NORM) {  // wave-private strip of the norm weights (K <= 8192: up to 8 float4 per lane)
        for (int i = lane; i < KW4; i += 64) ((float4*)nww)[i] = nsrc[i];
    }
    constexpr int NOUT = (RT * 16 * NT * 16 + 255) / 256;
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        int lrw = r * 16 + li;
        if (lrw >= nrows) lrw = nrows - 1;
        xr[r] = g.x + (size_t)(row0 + lrw) * g.ldx + koff;
#pragma unroll
        for (int c = 0; c < NT; ++c) { acc[r][c] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f}; accA[r][c] = acc[r][c]; }
        ss[r] = 0.0f; ssA[r] = 0.0f; qs0[r] = 0.0f;
    }
#pragma unroll
    for (int j = 0; j < XPF; ++j) {
#pragma unroll
        for (int r = 0; r < RT; ++r) { xq[j][r][0] = *(const float4*)(xr[r] + j * 32); xq[j][r][1] = *(const float4*)(xr[r] + j * 32 + 16); }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < WPF; ++j) {
#pragma unroll
        for (int c = 0; c < NT; ++c) wq[j][c] = ntload16(wp + c * tile_stride + (size_t)j * 64);
    }
    __builtin_amdgcn_sched_barrier(0);
    const float* nwa = nww + kq * 4;
    STAMP(1);
    for (int kb0 = 0; kb0 < nblk; kb0 += WPF) {  // nblk = 4*bps is a multiple of WPF
#pragma unroll
        for (int j = 0; j < WPF; ++j) {
            const int kb = kb0 + j;
            if (kb == bps || kb == 3 * bps) {  // slice 4w (4w+2) done: park its partial
#pragma unroll
                for (int r = 0; r < RT; ++r) {
#pragma unroll
                    for (int c = 0; c < NT; ++c) { accA[r][c] = acc[r][c]; acc[r][c] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f}; }
                    ssA[r] = ss[r]; ss[r] = 0.0f;
                }
            } else if (kb == 2 * bps) {  // first pair done: q_2w = p_4w + p_4w+1
#pragma unroll
                for (int r = 0; r < RT; ++r) {
#pragma unroll
                    for (int c = 0; c < NT; ++c) {  // straight to the reduction buffer: no register copy of q_2w is kept
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) red[(((2 * wave) * RT + r) * 16 + 4 * kq + jj) * CP + c * 16 + li] = accA[r][c][jj] + acc[r][c][jj];
                        acc[r][c] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
                    }
                    if (NORM) {
                        float sa = ssA[r] + __shfl_xor(ssA[r], 16), sb = ss[r] + __shfl_xor(ss[r], 16);
                        sa = sa + __shfl_xor(sa, 32); sb = sb + __shfl_xor(sb, 32);
                        qs0[r] = sa + sb;
                    }
                    ss[r] = 0.0f;
                }
            }
            float b[NT][8];
#pragma unroll
            for (int c = 0; c < NT; ++c) unpack8(wq[j][c], b[c]);
            float a[RT][8];
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                const float4 x0 = xq[j % XPF][r][0], x1 = xq[j % XPF][r][1];
                a[r][0] = x0.x; a[r][1] = x0.y; a[r][2] = x0.z; a[r][3] = x0.w; a[r][4] = x1.x; a[r][5] = x1.y; a[r][6] = x1.z; a[r][7] = x1.w;
                if (NORM) { ss[r] = sq4(x0, ss[r]); ss[r] = sq4(x1, ss[r]); }
            }
            if (NORM) {
                const float4 n0 = *(const float4*)(nwa + kb * 32), n1 = *(const float4*)(nwa + kb * 32 + 16);
                const float nw[8] = {n0.x, n0.y, n0.z, n0.w, n1.x, n1.y, n1.z, n1.w};
#pragma unroll
                for (int r = 0; r < RT; ++r)
#pragma unroll
                    for (int t = 0; t < 8; ++t) a[r][t] = a[r][t] * nw[t];
            }
            __builtin_amdgcn_sched_barrier(0);
            {   // refills are unconditional (clamped past the end): activations first, they must stay OLDER than the weights
                const int kx = min(kb + XPF, nblk - 1), kw = min(kb + WPF, nblk - 1);
#pragma unroll
                for (int r = 0; r < RT; ++r) {
                    xq[j % XPF][r][0] = *(const float4*)(xr[r] + kx * 32);
                    xq[j % XPF][r][1] = *(const float4*)(xr[r] + kx * 32 + 16);
                }
#pragma unroll
                for (int c = 0; c < NT; ++c) wq[j][c] = ntload16(wp + c * tile_stride + (size_t)kw * 64);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 8; ++t)
#pragma unroll
                for (int r = 0; r < RT; ++r)
#pragma unroll
                    for (int c = 0; c < NT; ++c) acc[r][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r][t], b[c][t], acc[r][c], 0, 0, 0);
        }
    }
    STAMP(3);
    // q_2w (already in LDS) and q_2w+1 = p_4w+2 + p_4w+3 meet the other waves' and are summed in order
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int c = 0; c < NT; ++c)
#pragma unroll
            for (int j = 0; j < 4; ++j) red[(((2 * wave + 1) * RT + r) * 16 + 4 * kq + j) * CP + c * 16 + li] = accA[r][c][j] + acc[r][c][j];
    if (NORM) {
#pragma unroll
        for (int r = 0; r < RT; ++r) {
            float sa = ssA[r] + __shfl_xor(ssA[r], 16), sb = ss[r] + __shfl_xor(ss[r], 16);
            sa = sa + __shfl_xor(sa, 32); sb = sb + __shfl_xor(sb, 32);
            if (kq == 0) { ssred[((2 * wave) * RT + r) * 16 + li] = qs0[r]; ssred[((2 * wave + 1) * RT + r) * 16 + li] = sa + sb; }
        }
    }
    __syncthreads();
    STAMP(4);
    if (NORM) {
        if (tid < RT * 16) {
            float tot = ssred[tid];
#pragma unroll
            for (int w = 1; w < 8; ++w) tot = tot + ssred[w * RT * 16 + tid];
            srow[tid] = 1.0f / sqrtf(tot / (float)K + g.eps);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < NOUT; ++i) {
        const int o = tid + i * 256;
        const int row = o / (NT * 16), col = o - row * (NT * 16);
        if (o >= RT * 16 * NT * 16 || row >= nrows) continue;
        float s = red[row * CP + col];
#pragma unroll
        for (int w = 1; w < 8; ++w) s = s + red[(w * RT * 16 + row) * CP + col];
        if (NORM) s = s * srow[row];
        store_elem(g, sums, s, (size_t)(row0 + row), nbt * NT * 16 + col, row, col, NT * 16);
    }
    STAMP(5);
    epilogue<RT * 16, NT * 16>(g, sums, tid, 256, row0, nrows, nbt * NT * 16);
    STAMP(6);
}

template <int RT, int NT>
static void launch_fat(const Q3Gemm& g, dim3 grid, hipStream_t s) {
    const bool norm = g.norm_w != nullptr;
    const size_t lds = ((size_t)8 * RT * 16 * (NT * 16 + 1) + (size_t)RT * 16 * NT * 16 + (norm ? (size_t)g.K : 0)) * 4;
    static Q3PerDevice pd;  // dynamic LDS above 64 KiB has to be allowed per kernel and per device (K <= 8192 bounds it)
    pd.ensure(lds, [&]() {
        hipFuncSetAttribute((const void*)k_gemm_fat<RT, NT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipFuncSetAttribute((const void*)k_gemm_fat<RT, NT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    });
    if (norm) hipLaunchKernelGGL((k_gemm_fat<RT, NT, true>), grid, dim3(256), lds, s, g);
    else hipLaunchKernelGGL((k_gemm_fat<RT, NT, false>), grid, dim3(256), lds, s, g);
}

template <bool NORM>
static void launch_small(const Q3Gemm& g, dim3 grid, size_t lds, hipStream_t s) {
#define L(BPS_) hipLaunchKernelGGL((k_gemm_small<NORM, BPS_>), grid, dim3(1024), lds, s, g)
    switch (g.K >> 9) {
        case 1: L(1); break;
        case 2: L(2); break;
        case 4: L(4); break;
        case 6: L(6); break;
        case 12: L(12); break;
        default: L(0); break;
    }
#undef L
}
template <int RT, int NT>
static void launch_ring(const Q3Gemm& g, dim3 grid, hipStream_t s) {
    const bool norm = g.norm_w != nullptr;
    const size_t lds = norm ? (size_t)g.K * 4 : 0;
#define L(BPS_)                                                                                              \
    do {                                                                                                     \
        if (norm) hipLaunchKernelGGL((k_gemm_ring<RT, NT, BPS_, true>), grid, dim3(512), lds, s, g);         \
        else hipLaunchKernelGGL((k_gemm_ring<RT, NT, BPS_, false>), grid, dim3(512), 0, s, g);               \
    } while (0)
#define LN(BPS_) hipLaunchKernelGGL((k_gemm_ring<RT, NT, BPS_, false>), grid, dim3(512), 0, s, g)
    switch (g.K >> 9) {  // norm GEMMs have K = d_model: 1, 2, 4 (anything else: the runtime-K instance)
        case 1: L(1); break;
        case 2: L(2); break;
        case 4: if (norm && RT == 4) L(0); else L(4); break;  // the unrolled K = 2048 NORM instance of RT = 4 spills
        case 6: if (norm) L(0); else LN(6); break;
        case 12: if (norm) L(0); else LN(12); break;
        default: L(0); break;
    }
#undef L
#undef LN
}

void q3_launch_gemm(const Q3Gemm& g, hipStream_t s) {
    const bool norm = g.norm_w != nullptr;
    if (g.B <= 16 && g.ldx == g.K && (size_t)g.B * g.K <= XLDS_MAX_FLOATS && g.K <= 8192) {
        const size_t lds = ((size_t)g.B * g.K + (norm ? g.K : 0)) * 4;
        dim3 grid(g.N / 16);
        if (norm) launch_small<true>(g, grid, lds, s); else launch_small<false>(g, grid, lds, s);
        return;
    }
    const int tiles = g.N / 16;
    // bit 1 (64 x 64 tiles for many rows, i.e. prefill: 78 -> 72.5 ms for 1 984 prompt rows) is on by default; bit 0
    // (32 x 96 tiles for the 12 288-column gate/up GEMM at decode batch) measured slower than the ring kernel
    // (7.78 vs 7.52 ms per frame step) and stays off
    static const int fat = getenv("Q3_GEMM_FAT") ? atoi(getenv("Q3_GEMM_FAT")) : 2;
    if (fat && g.B > 32 && g.K % 512 == 0 && g.K <= 8192) {
        // very wide N at decode batch: 32 x 96 tiles still give every CU a workgroup
        if ((fat & 1) && g.B <= 64 && tiles % 6 == 0 && (long)(tiles / 6) * ((g.B + 31) / 32) >= 256) { launch_fat<2, 6>(g, dim3(tiles / 6, (g.B + 31) / 32), s); return; }
        // many rows (prefill): 64 x 64 tiles
        if ((fat & 2) && g.B > 64 && tiles % 4 == 0 && (long)(tiles / 4) * ((g.B + 63) / 64) >= 256) { launch_fat<4, 4>(g, dim3(tiles / 4, (g.B + 63) / 64), s); return; }
    }
    int NT = 1;
    if (g.B > 16) {
        // x-fragment traffic from L2 scales with 1/NT and is the co-bottleneck at B = 64: widest NT that still
        // gives every CU a workgroup (2 row chunks of 32 rows each)
        if (tiles >= 384 && tiles % 3 == 0) NT = 3;
        else if (tiles >= 256 && tiles % 2 == 0) NT = 2;
    }
    // Per-CU operand bytes per k are 64*RT (x, f32) + 32*NT (weights, bf16): with ~128 column tiles and 64 rows a 16 x 32
    // workgroup tile (RT 1, NT 2: 128 B/k) still fills 256 CUs and moves 20 % less than 32 x 16 (160 B/k) — the launch is
    // bound by what a CU's load path delivers, not by MFMA issue (profiles/README.md). Q3_RT1NT2=0 restores 32 x 16.
    static const int rt1nt2 = getenv("Q3_RT1NT2") ? atoi(getenv("Q3_RT1NT2")) : 1;
    if (rt1nt2 && NT == 1 && g.B > 32 && tiles % 2 == 0 && (long)(tiles / 2) * ((g.B + 15) / 16) >= 256) {
        dim3 grid(tiles / 2, (g.B + 15) / 16);
        launch_ring<1, 2>(g, grid, s);
        return;
    }
    if (NT > 1) {
        dim3 grid(tiles / NT, (g.B + 31) / 32);
        if (NT == 3) launch_ring<2, 3>(g, grid, s); else launch_ring<2, 2>(g, grid, s);
        return;
    }
    int RT = 1;
    if (g.B > 16) {
        RT = g.B > 32 ? 4 : 2;
        static const int rtmax = getenv("Q3_RTMAX") ? atoi(getenv("Q3_RTMAX")) : 4;  // tuning knob
        if (RT > rtmax) RT = rtmax;
        static const int minwg = getenv("Q3_MINWG") ? atoi(getenv("Q3_MINWG")) : 256;  // tuning knob
        while (RT > 1 && (long)tiles * ((g.B + RT * 16 - 1) / (RT * 16)) < minwg) RT >>= 1;
    }
    dim3 grid(tiles, (g.B + RT * 16 - 1) / (RT * 16));
    if (RT == 4) launch_ring<4, 1>(g, grid, s);
    else if (RT == 2) launch_ring<2, 1>(g, grid, s);
    else launch_ring<1, 1>(g, grid, s);
}
