// q3_bgemm8.hip — the decoder's GEMM in ggml's Q8_0 x Q8_0 arithmetic (W8A8) on v_mfma_i32_16x16x32_i8: what llama.cpp computes for the
// reference's default model directory gguf_q8_0 (/root/reference/src/tts/engine.rs:91-95, README.md:29-32: Q8_0 weights; llama.cpp's CPU
// and GPU back ends quantise the activations of a mul_mat to Q8_0 blocks as well and multiply block by block: vec_dot_q8_0_q8_0).
// q3tts_engine_config.talker_q8_0 = 2; oracle q3o_bgemm_q8a8_raw (oracle/q3_oracle_bf16.c).
//
// Canonical order: a block of 32 k contributes  p_b = f32(sumi_b) * (f32(d_w) * f32(d_x)),  sumi_b = the EXACT int32 sum of its 32 int8
// products (one v_mfma_i32_16x16x32_i8 from a zero accumulator: integer arithmetic has no order), the scales multiplied first — ggml's
// `sumf += sumi * (dx * dy)` without contraction;  t_w = the blocks of K slice w added in ascending order from +0 (plain f32 adds);
// RAW = ((t_0 + t_1) + ...) + t_7. No weight or activation is ever widened: 1.06 bytes per operand element on both sides.
//
// Operands: the weights in the tiled Q8 layout of q3_kernels.h (tile PAIR (n/16, k/64) = 1 KiB, f16 scales [N][K/32]); the activations
// in the same form with rows in place of columns (q3_q8_off) and their block scales as [K/64][row tiles][4 row quads][2 blocks][4 rows] f16
// (q3_q8_scale_idx): the 8 scales a lane needs for one (row tile, block pair) are one 16-byte load. Producers write both (this kernel's
// RESID / SWIGLU epilogues, the attention kernels, k_pred_next, k_norm_inputs): an activation is quantised where it is produced, from
// its f32 value, by ggml's rule (d = amax / 127, id = d ? 1 / d : 0, q = roundf(v * id), d kept as f16).
//
// Workgroup = 8 waves = the 8 K slices of one (16 RT) x (16 NT) tile, DP block pairs of operands in flight per wave; slice partials meet
// in LDS and are added in slice order, as in q3_bgemm.hip. Epilogues: STORE (y = s_r RAW), RESID (x += RAW; the consumer's operand
// v = x * nw_next quantised per 32 columns + the tile sums of squares: NT even), SWIGLU (h = swiglu(s_r gate, s_r up) quantised per 32
// columns: a 16-column weight tile is 8 gate + 8 up columns, so NT = 4).
#include "q3_kernels.h"

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float b8_f16(uint32_t bits) { return (float)__builtin_bit_cast(_Float16, (unsigned short)bits); }
__device__ __forceinline__ long b8_pack(uint32_t lo, uint32_t hi) { return (long)(((unsigned long)hi << 32) | (unsigned long)lo); }

#define B8_PH(RT_, NT_) ((8 * (RT_) * (NT_) > 64) ? 2 : 1)  // slice-reduction phases: <= 64 KiB of LDS
template <int RT, int NT, int DP, bool ALIGNED>
__global__ __launch_bounds__(512) void k_bgemm8(Q3BGemm g) {
    extern __shared__ float part[];  // [8 waves][TRP regs][64 lanes]
    __shared__ float srow[64];
    constexpr int TR = RT * NT * 4;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, kq = lane >> 4, r = lane & 15, tid = threadIdx.x;
    int cg = blockIdx.x, rc = blockIdx.y;
    if (gridDim.y > 1 && (gridDim.x & 7) == 0) {  // row chunks of one column group: linear ids 8 apart (same XCD), as in k_bgemm
        const int id = blockIdx.x + gridDim.x * blockIdx.y, G = gridDim.y;
        rc = (id >> 3) % G; cg = (id & 7) + (id / (8 * G)) * 8;
    }
    const int nb0 = cg * NT, row0 = rc * RT * 16, kblocks = g.K >> 5, kpairs = g.K >> 6, npair = g.K >> 9, kp0 = wave * npair;
    const int B = g.B;
    // operand pointers: 16 bytes per lane per (tile, block pair)
    const u32x4* ap[RT]; const u32x4* asp[RT];
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        const int R = g.a_row0 + min(row0 + 16 * i + r, B - 1);
        ap[i] = (const u32x4*)g.a + ((size_t)(R >> 4) * kpairs + kp0) * 64 + kq * 16 + (R & 15);
        const int Rb = min(g.a_row0 + row0 + 16 * i, g.a_row0 + B - 1);   // (aligned: a_row0 is a multiple of 16; a row tile past the last row's re-reads that tile: its results are dropped)
        asp[i] = (const u32x4*)g.ascale + ((size_t)kp0 * g.a_rt16 + (Rb >> 4)) * 4 + kq;
    }
    const u32x4* wq[NT]; const uint32_t* sq[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        wq[j] = (const u32x4*)g.w + ((size_t)(nb0 + j) * kpairs + kp0) * 64 + lane;
        sq[j] = (const uint32_t*)(g.wscale + (size_t)((nb0 + j) * 16 + r) * kblocks + 2 * kp0);
    }
    constexpr int NR = 2 * RT;
    float sp0[NR], sp1[NR];
    if (g.ssp) {
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const float* sp = g.ssp + (size_t)min(row0 + wave + 8 * i, B - 1) * g.ld_ssp;
            sp0[i] = lane < g.ntiles ? sp[lane] : 0.0f;
            sp1[i] = lane + 64 < g.ntiles ? sp[lane + 64] : 0.0f;
        }
    }
    u32x4 aq[DP][RT], as[DP][RT], bq[DP][NT]; uint32_t bs[DP][NT];
    auto load_pair = [&](int slot, int p) {
#pragma unroll
        for (int j = 0; j < NT; ++j) { bq[slot][j] = wq[j][(size_t)p * 64]; bs[slot][j] = sq[j][p]; }
#pragma unroll
        for (int i = 0; i < RT; ++i) {
            aq[slot][i] = ap[i][(size_t)p * 64];
            if constexpr (ALIGNED) as[slot][i] = asp[i][(size_t)p * g.a_rt16 * 4];
            else {  // rows that do not start a tile (a single row picked out of a prefill batch): the lane's 4 rows x 2 blocks one by one
                uint32_t sv[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int Rr = g.a_row0 + min(row0 + 16 * i + 4 * kq + e, B - 1);
                    const uint16_t* s8 = g.ascale + q3_q8_scale_idx(Rr, 2 * (kp0 + p), g.a_rt16);
                    sv[e] = s8[0]; sv[4 + e] = s8[4];
                }
                as[slot][i] = (u32x4){sv[0] | (sv[1] << 16), sv[2] | (sv[3] << 16), sv[4] | (sv[5] << 16), sv[6] | (sv[7] << 16)};
            }
        }
    };
#pragma unroll
    for (int p = 0; p < DP; ++p) load_pair(p, min(p, npair - 1));
    __builtin_amdgcn_sched_barrier(0);
    if (g.ssp) {
        float av[NR];
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            float a = sp0[i];
            if (lane + 64 < g.ntiles) a = a + sp1[i];
            if (g.ntiles > 128) {
                const float* sp = g.ssp + (size_t)min(row0 + wave + 8 * i, B - 1) * g.ld_ssp;
                for (int t = lane + 128; t < g.ntiles; t += 64) a = a + sp[t];
            }
            av[i] = a;
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1)
#pragma unroll
            for (int i = 0; i < NR; ++i) av[i] = av[i] + __shfl_xor(av[i], m);
        if (lane == 0) {
#pragma unroll
            for (int i = 0; i < NR; ++i) srow[wave + 8 * i] = 1.0f / sqrtf(av[i] / (float)g.d_norm + g.eps);
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    f32x2 acc[RT][NT][2];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) { acc[i][j][0] = (f32x2){0.f, 0.f}; acc[i][j][1] = (f32x2){0.f, 0.f}; }
    const i32x4 zero4 = (i32x4){0, 0, 0, 0};
    for (int p0 = 0; p0 < npair; p0 += DP) {
#pragma unroll
        for (int dp = 0; dp < DP; ++dp) {
            const int p = p0 + dp;
            if (p < npair) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    float dw[NT]; long bb[NT];
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        dw[j] = b8_f16(h ? bs[dp][j] >> 16 : bs[dp][j] & 0xffffu);
                        bb[j] = h ? b8_pack(bq[dp][j].z, bq[dp][j].w) : b8_pack(bq[dp][j].x, bq[dp][j].y);
                    }
#pragma unroll
                    for (int i = 0; i < RT; ++i) {
                        const long aa = h ? b8_pack(aq[dp][i].z, aq[dp][i].w) : b8_pack(aq[dp][i].x, aq[dp][i].y);
                        const uint32_t s01 = h ? as[dp][i].z : as[dp][i].x, s23 = h ? as[dp][i].w : as[dp][i].y;  // f16 scales of rows 4 kq + 0..3
                        const f32x2 dx01 = (f32x2){b8_f16(s01 & 0xffffu), b8_f16(s01 >> 16)}, dx23 = (f32x2){b8_f16(s23 & 0xffffu), b8_f16(s23 >> 16)};
#pragma unroll
                        for (int j = 0; j < NT; ++j) {
                            const i32x4 P = __builtin_amdgcn_mfma_i32_16x16x32_i8(aa, bb[j], zero4, 0, 0, 0);
                            const f32x2 dwv = (f32x2){dw[j], dw[j]};
                            const f32x2 sc01 = dwv * dx01, sc23 = dwv * dx23;                       // f32(d_w) * f32(d_x)
                            const f32x2 pr01 = (f32x2){(float)P[0], (float)P[1]} * sc01, pr23 = (f32x2){(float)P[2], (float)P[3]} * sc23;
                            acc[i][j][0] = acc[i][j][0] + pr01; acc[i][j][1] = acc[i][j][1] + pr23;  // plain adds, blocks ascending
                        }
                        __builtin_amdgcn_sched_barrier(0);   // (one row tile's products at a time: keeps the integer results of 16 MFMAs from being live at once)
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (p + DP < npair) load_pair(dp, p + DP);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    constexpr int PH = B8_PH(RT, NT), RTP = RT / PH, TRP = TR / PH;
    static_assert(RT % PH == 0, "two phases split the row tiles");
    const int epi = g.epi;
#pragma unroll
    for (int ph = 0; ph < PH; ++ph) {
        if (ph) __syncthreads();
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                if (i / RTP != ph) continue;
                const int t = (i - ph * RTP) * NT + j;
                part[((size_t)wave * TRP + t * 4 + 0) * 64 + lane] = acc[i][j][0][0]; part[((size_t)wave * TRP + t * 4 + 1) * 64 + lane] = acc[i][j][0][1];
                part[((size_t)wave * TRP + t * 4 + 2) * 64 + lane] = acc[i][j][1][0]; part[((size_t)wave * TRP + t * 4 + 3) * 64 + lane] = acc[i][j][1][1];
            }
        __syncthreads();
        // one thread per (row tile of the phase, register e, lane l): it owns row 16 i + 4 (l >> 4) + e and column l & 15 of all NT tiles
        for (int it = tid; it < RTP * 256; it += 512) {   // (whole waves: 256 items per row tile)
            const int l = it & 63, e = (it >> 6) & 3, il = it >> 8, i = ph * RTP + il;
            const int rl = 16 * i + 4 * (l >> 4) + e, row = row0 + rl, c = l & 15;
            const bool live = row < B;
            const float sc = g.ssp ? srow[rl] : 1.0f;
            float v[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int o = ((il * NT + j) * 4 + e) * 64 + l;
                float s = part[o];
#pragma unroll
                for (int wv = 1; wv < 8; ++wv) s = s + part[(size_t)wv * (TRP * 64) + o];
                v[j] = s;
            }
            if (epi == Q3_EPI_STORE) {
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    if (live) g.y[(size_t)row * g.ldy + (nb0 + j) * 16 + c] = g.ssp ? sc * v[j] : v[j];
            } else if (epi == Q3_EPI_RESID) {
                float u[NT];
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const int col = (nb0 + j) * 16 + c;
                    const float xv = (live ? g.y[(size_t)row * g.ldy + col] : 0.0f) + v[j];
                    if (live) g.y[(size_t)row * g.ldy + col] = xv;
                    u[j] = xv * g.nw_next[col];
                    float sqv = xv * xv;
                    sqv = sqv + __shfl_xor(sqv, 1); sqv = sqv + __shfl_xor(sqv, 2); sqv = sqv + __shfl_xor(sqv, 4); sqv = sqv + __shfl_xor(sqv, 8);
                    if (live && c == 0) g.ssp_out[(size_t)row * g.ld_ssp_out + (nb0 + j)] = sqv;
                }
#pragma unroll
                for (int j = 0; j < NT; j += 2) {  // a block = the 32 columns of tiles j, j + 1 (nb0 is even: NT is)
                    float amax = fmaxf(fabsf(u[j]), fabsf(u[j + 1]));
#pragma unroll
                    for (int m = 1; m <= 8; m <<= 1) amax = fmaxf(amax, __shfl_xor(amax, m));
                    const float d = amax / 127.0f, id = d != 0.0f ? 1.0f / d : 0.0f;
                    const int col = (nb0 + j) * 16 + c;
                    if (live) {
                        ((int8_t*)g.yb)[q3_q8_off(row, col, g.N >> 6)] = (int8_t)(int)roundf(u[j] * id);
                        ((int8_t*)g.yb)[q3_q8_off(row, col + 16, g.N >> 6)] = (int8_t)(int)roundf(u[j + 1] * id);
                        if (c == 0) g.yscale[q3_q8_scale_idx(row, col >> 5, g.y_rt16)] = __builtin_bit_cast(unsigned short, (_Float16)d);
                    }
                }
            } else {  // Q3_EPI_SWIGLU: gate = columns 0-7 of a tile, up = the same row 8 columns further
                float hv[NT];
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const float up = __shfl(v[j], (l & 48) | ((c + 8) & 15));
                    hv[j] = q3_swiglu(sc * v[j], sc * up);   // (meaningful on lanes c < 8)
                }
#pragma unroll
                for (int j = 0; j < NT; j += 4) {  // a block = 32 h columns = the gate halves of tiles j .. j + 3
                    float amax = fmaxf(fmaxf(fabsf(hv[j]), fabsf(hv[j + 1])), fmaxf(fabsf(hv[j + 2]), fabsf(hv[j + 3])));
#pragma unroll
                    for (int m = 1; m <= 4; m <<= 1) amax = fmaxf(amax, __shfl_xor(amax, m));   // (stays inside the 8 gate lanes)
                    const float d = amax / 127.0f, id = d != 0.0f ? 1.0f / d : 0.0f;
                    if (live && c < 8) {
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) ((int8_t*)g.yb)[q3_q8_off(row, (nb0 + j + jj) * 8 + c, g.N >> 7)] = (int8_t)(int)roundf(hv[j + jj] * id);
                        if (c == 0) g.yscale[q3_q8_scale_idx(row, ((nb0 + j) * 8) >> 5, g.y_rt16)] = __builtin_bit_cast(unsigned short, (_Float16)d);
                    }
                }
            }
        }
    }
}

template <int RT, int NT>
struct B8Inst {
    static constexpr int DP = (RT + NT) <= 4 ? 3 : 2;  // block pairs in flight per wave (a pair = two 32-wide steps)
    static constexpr size_t lds = (size_t)8 * RT * NT * 4 * 64 * 4 / B8_PH(RT, NT);
    static void prepare() { if (lds + 1024 > 65536) hipFuncSetAttribute((const void*)k_bgemm8<RT, NT, DP, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); }
    static void launch(const Q3BGemm& g, dim3 grid, hipStream_t s) { hipLaunchKernelGGL((k_bgemm8<RT, NT, DP, true>), grid, dim3(512), lds, s, g); }
};
#define B8_EACH(X) X(1, 2) X(2, 2) X(4, 2) X(1, 4) X(2, 4)   // ((4, 4) needs more than 256 registers per wave: gate/up runs 32-row chunks)
static void b8_prepare() {
    static Q3PerDevice pd;
    pd.ensure(1, []() {
#define P(RT_, NT_) B8Inst<RT_, NT_>::prepare();
        B8_EACH(P)
#undef P
    });
}
// tile choice: as q3_launch_bgemm (fewest operand bytes per workgroup round: 16 RT + 17 NT bytes per k); RESID needs NT even, SWIGLU NT = 4
void q3_bgemm8_pick(const Q3BGemm& g, int* rt, int* nt) {
    const int tiles = g.N / 16;
    int bestRT = 1, bestNT = 2; long bestCost = -1, bestWgs = 0;
    for (int RT = 1; RT <= 4; RT *= 2)
        for (int NT = 2; NT <= 4; NT *= 2) {
            if (tiles % NT) continue;
            if (g.epi == Q3_EPI_SWIGLU && NT != 4) continue;
            if (RT == 4 && NT == 4) continue;
            if (g.B > 64 && RT != 4 && !(RT == 2 && NT == 4)) continue;   // many rows (prefill): the largest row chunk of the column width
            const long chunks = (g.B + 16 * RT - 1) / (16 * RT);
            if (g.B <= 64 && RT > 1 && 16 * (RT / 2) * chunks >= g.B) continue;  // a smaller RT covers the rows with the same chunk count
            const long wgs = (long)(tiles / NT) * chunks, rounds = (wgs + 255) / 256;
            const long cost = rounds * ((16L * RT + 17L * NT) * g.K + 24000L);
            if (bestCost < 0 || cost < bestCost || (cost == bestCost && wgs > bestWgs)) { bestCost = cost; bestRT = RT; bestNT = NT; bestWgs = wgs; }
        }
    *rt = bestRT; *nt = bestNT;
}
int q3_launch_bgemm8(const Q3BGemm& g, hipStream_t s) {
    if (g.B < 1 || g.N % 32 || g.K % 512 || g.K < 512 || !g.a || !g.w || !g.wscale || !g.ascale || g.a_row0 < 0 || g.a_rt16 < 1) return -1;
    if (g.epi != Q3_EPI_STORE && g.epi != Q3_EPI_RESID && g.epi != Q3_EPI_SWIGLU) return -1;
    if ((g.a_row0 & 15) && g.B != 1) return -1;   // unaligned first row: the single-row pick of the prefill head only
    if (g.epi == Q3_EPI_RESID && (!g.yb || !g.yscale || !g.nw_next || !g.ssp_out || g.N % 64 || g.y_rt16 < 1)) return -1;
    if (g.epi == Q3_EPI_SWIGLU && (!g.yb || !g.yscale || g.N % 128 || g.y_rt16 < 1)) return -1;
    if (g.ssp && g.ntiles < 1) return -1;
    b8_prepare();
    if (g.a_row0 & 15) {  // one row that does not start a tile (the last prompt row's head at prefill): the scales are fetched row by row
        if ((g.N / 16) % 2) return -1;
        const size_t lds12 = B8Inst<1, 2>::lds;
        hipLaunchKernelGGL((k_bgemm8<1, 2, 2, false>), dim3(g.N / 32, 1), dim3(512), lds12, s, g);
        return 0;
    }
    int rt, nt; q3_bgemm8_pick(g, &rt, &nt);
    if ((g.N / 16) % nt) return -1;
    const dim3 grid(g.N / 16 / nt, (g.B + 16 * rt - 1) / (16 * rt));
#define L(RT_, NT_) if (rt == RT_ && nt == NT_) { B8Inst<RT_, NT_>::launch(g, grid, s); return 0; }
    B8_EACH(L)
#undef L
    return -1;
}
