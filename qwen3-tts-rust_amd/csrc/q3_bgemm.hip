// q3_bgemm.hip — the decoder's GEMM: bf16 rows x tiled bf16 weights on v_mfma_f32_16x16x32_bf16, bit-exact against the CPU
// oracle because the instruction's accumulation arithmetic is restated there (oracle/q3_oracle_bf16.c, pinned by hardware
// golden vectors). Serves K1/K2/K6-K9 of the Talker and P1-P4 of the Predictor (llama_decode behind
// /root/reference/src/models/llama/mod.rs:442-451), decode and prefill alike.
//
// Canonical order: RAW[r][n] = ((((s_0 + s_1) + s_2) + ...) + s_7), s_w = the chain of MFMA steps over K-slice w
// (K/8 contiguous columns, 32 per instruction, ascending); lane group g of an instruction holds k = 4g..4g+3 and 16+4g..16+4g+3 of
// the 32-block — what the tiled weight layout of DESIGN.md §2.1 puts into one lane. A row's result never depends on the other
// rows, the tile shape or the grid: every (RT, NT) instance and every row count give the same bits.
//
// RMSNorm is split between producer and consumer: a residual epilogue (EPI_RESID) writes, next to the f32
// residual stream x, the consumer's A operand xb = bf16(x * nw_next) and one partial sum of squares per 16-column tile; the
// norm GEMM that follows reduces the partials to the row scale s_r while its first operands are in flight and applies
// y = s_r * RAW in its epilogue. No f32 activation is read by a GEMM, no pre-kernel, no second pass over x.
//
// Workgroup = 8 waves = the 8 K-slices of one (RT*16) x (NT*16) output tile; a wave keeps D steps of operands in flight
// (register ring, refilled right after the MFMAs that consumed a slot); slice partials meet in LDS and are added in order.
#include "q3_kernels.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float bg_wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = v + __shfl_xor(v, m);
    return v;
}

__device__ __forceinline__ size_t bg_yoff(const Q3BGemm& g, int row) {
    return g.seg_rows ? (size_t)(row / g.seg_rows) * g.seg_stride + (size_t)(row % g.seg_rows) * g.ldy : (size_t)row * g.ldy;
}
__device__ __forceinline__ float bg_gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }

__device__ __forceinline__ u32x4 bg_ldw(const u32x4* p, bool nt) { return nt ? __builtin_nontemporal_load(p) : *p; }

// 8 int8 quants (lo = elements 0..3, hi = 4..7 of the lane's fragment) as a bf16 MFMA operand: f32(int8) has an exact bf16 in its upper half
__device__ __forceinline__ bf16x8 bg_q8_frag(uint32_t lo, uint32_t hi) {
    union { u32x4 u; bf16x8 v; } r;
#define Q8F(w_, b_) __float_as_uint((float)(int)(signed char)((w_) >> (8 * (b_))))
    r.u.x = (Q8F(lo, 0) >> 16) | (Q8F(lo, 1) & 0xffff0000u); r.u.y = (Q8F(lo, 2) >> 16) | (Q8F(lo, 3) & 0xffff0000u);
    r.u.z = (Q8F(hi, 0) >> 16) | (Q8F(hi, 1) & 0xffff0000u); r.u.w = (Q8F(hi, 2) >> 16) | (Q8F(hi, 3) & 0xffff0000u);
#undef Q8F
    return r.v;
}
__device__ __forceinline__ float bg_f16(uint32_t bits) { return (float)__builtin_bit_cast(_Float16, (unsigned short)bits); }

// Q8: the weights are ggml Q8_0 blocks in the tiled Q8 layout (q3_kernels.h): 1.06 bytes per weight instead of 2 through
// the same ring; every 32-wide K step is one block: P = MFMA from a zero accumulator, acc = fmaf(f32(d), P, acc).
#define BG_PH(RT_, NT_) ((8 * (RT_) * (NT_) > 64 && ((RT_) * (NT_)) % 2 == 0) ? 2 : 1)  // slice-reduction phases (k_bgemm, BgInst::lds)
template <int RT, int NT, int D, bool NTW, bool Q8>
__global__ __launch_bounds__(512) void k_bgemm(Q3BGemm g) {
    extern __shared__ float part[];  // [8 waves][RT*NT*4 regs][64 lanes]
    __shared__ float srow[64];
    Q3_STAMP(g, 0);
    constexpr int TR = RT * NT * 4;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, kq = lane >> 4, r = lane & 15;
    // workgroups are dealt round-robin to the 8 XCDs in linear-id order: the row chunks of one column group get ids 8 apart
    // (same XCD, neighbours in time), so the group's weight tiles are fetched into that L2 once
    int cg = blockIdx.x, rc = blockIdx.y;
    if (gridDim.y > 1 && (gridDim.x & 7) == 0) {
        const int id = blockIdx.x + gridDim.x * blockIdx.y, G = gridDim.y;
        rc = (id >> 3) % G; cg = (id & 7) + (id / (8 * G)) * 8;
    }
    const int nb0 = cg * NT, row0 = rc * RT * 16, kblocks = g.K >> 5, per = g.K >> 8, kb0 = wave * per;
    const int B = g.B;
    const u32x4* ap[RT];  // A-tiled rows: one contiguous KiB per (row tile, k block) when the row tile is aligned, 16 B per lane always
#pragma unroll
    for (int i = 0; i < RT; ++i) {
        const int R = g.a_row0 + min(row0 + 16 * i + r, B - 1);
        ap[i] = (const u32x4*)g.a + ((size_t)(R >> 4) * kblocks + kb0) * 64 + kq * 16 + (R & 15);
    }
    const u32x4* wp[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) wp[j] = (const u32x4*)g.w + ((size_t)(nb0 + j) * kblocks + kb0) * 64 + lane;
    // Load order = the order the results are needed in (vmcnt retires in order): first the norm GEMMs' tile partials (a few hundred
    // bytes per wave from L2), whose reduction to the row scales then runs WHILE the operands are in flight (in-kernel timestamps,
    // tools/chain_stamps.hip: done after the main loop it cost ~1 us of every norm GEMM); then the weights (HBM / Infinity Cache: the
    // long latency) and the rows (L2), D steps of both; last what only the epilogue reads (residual, next norm weight, bias, scale).
    // Every load below is unconditional (out-of-range steps re-read step per - 1): straight-line code, so the counted waits are exact.
    // norm GEMMs: wave w owns the row scales of rows w, w + 8, ...: lane j adds its tiles j, j + 64, ... in ascending order, then the
    // 64-lane butterfly
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
    constexpr int DQ = (D + 1) & ~1, DP = DQ / 2;  // Q8: an even number of steps in flight = DP block pairs (one 16-byte weight load covers two steps)
    u32x4 aq[Q8 ? DQ : D][RT], bq[Q8 ? DP : D][NT];
    uint32_t sc8[Q8 ? DP : 1][NT];                 // Q8: the two f16 block scales of a pair
    const u32x4* wq[NT]; const uint32_t* sq[NT];
    const int npair = per >> 1;
    if constexpr (Q8) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            wq[j] = (const u32x4*)g.w + ((size_t)(nb0 + j) * (kblocks >> 1) + (kb0 >> 1)) * 64 + lane;
            sq[j] = (const uint32_t*)(g.wscale + (size_t)((nb0 + j) * 16 + r) * kblocks + kb0);
        }
#pragma unroll
        for (int p = 0; p < DP; ++p) {
            const int pc = min(p, npair - 1);
#pragma unroll
            for (int j = 0; j < NT; ++j) { bq[p][j] = bg_ldw(wq[j] + (size_t)pc * 64, NTW); sc8[p][j] = sq[j][pc]; }
        }
#pragma unroll
        for (int s = 0; s < DQ; ++s) {
            const size_t so = (size_t)min(s, per - 1) * 64;
#pragma unroll
            for (int i = 0; i < RT; ++i) aq[s][i] = ap[i][so];
        }
    } else {
#pragma unroll
        for (int s = 0; s < D; ++s) {
            const size_t so = (size_t)min(s, per - 1) * 64;
#pragma unroll
            for (int j = 0; j < NT; ++j) bq[s][j] = bg_ldw(wp[j] + so, NTW);
        }
#pragma unroll
        for (int s = 0; s < D; ++s) {
            const size_t so = (size_t)min(s, per - 1) * 64;
#pragma unroll
            for (int i = 0; i < RT; ++i) aq[s][i] = ap[i][so];
        }
    }
    // epilogue operands, fetched up front instead of at the very end (each was a dependent round trip behind the slice reduction)
    constexpr int NOUT = (TR * 64 + 511) / 512;
    float yres[NOUT], nwv[NOUT], bsv[NOUT], csv[NOUT];
#pragma unroll
    for (int it = 0; it < NOUT; ++it) { yres[it] = 0.0f; nwv[it] = 0.0f; bsv[it] = 0.0f; csv[it] = 1.0f; }
    if (g.epi == Q3_EPI_RESID) {
#pragma unroll
        for (int it = 0; it < NOUT; ++it) {
            const int o = threadIdx.x + it * 512;
            const int l = o & 63, e = (o >> 6) & 3, t = o >> 8, i = t / NT, j = t - i * NT;
            const int row = row0 + 16 * i + 4 * (l >> 4) + e, col = (nb0 + min(j, NT - 1)) * 16 + (l & 15);
            const bool in = o < TR * 64;
            yres[it] = (in && row < B) ? g.y[bg_yoff(g, row) + col] : 0.0f;
            if (g.nw_next) nwv[it] = in ? g.nw_next[col] : 0.0f;
            if (g.col_scale) csv[it] = in ? g.col_scale[col] : 1.0f;
        }
    }
    if (g.bias && g.epi != Q3_EPI_SWIGLU) {
#pragma unroll
        for (int it = 0; it < NOUT; ++it) {
            const int o = threadIdx.x + it * 512;
            const int l = o & 63, t = o >> 8, i = t / NT, j = t - i * NT;
            const int col = (nb0 + min(j, NT - 1)) * 16 + (l & 15);
            bsv[it] = o < TR * 64 ? g.bias[col % g.bias_n] : 0.0f;
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    Q3_STAMP(g, 1);
    if (g.ssp) {  // the NR butterflies are independent chains: interleaved; they wait for the partials only (the first loads issued)
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
    f32x4 acc[RT][NT];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if constexpr (Q8) {
        const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int p0 = 0; p0 < npair; p0 += DP) {
#pragma unroll
            for (int dp = 0; dp < DP; ++dp) {
                const int p = p0 + dp;
                if (p < npair) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        float dsc[NT]; bf16x8 bf[NT];
#pragma unroll
                        for (int j = 0; j < NT; ++j) {
                            dsc[j] = bg_f16(h ? sc8[dp][j] >> 16 : sc8[dp][j] & 0xffffu);
                            bf[j] = h ? bg_q8_frag(bq[dp][j].z, bq[dp][j].w) : bg_q8_frag(bq[dp][j].x, bq[dp][j].y);
                        }
#pragma unroll
                        for (int i = 0; i < RT; ++i) {
                            union { u32x4 u; bf16x8 v; } a; a.u = aq[2 * dp + h][i];
#pragma unroll
                            for (int j = 0; j < NT; ++j) {
                                const f32x4 P = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, bf[j], zero4, 0, 0, 0);
                                acc[i][j][0] = fmaf(dsc[j], P[0], acc[i][j][0]); acc[i][j][1] = fmaf(dsc[j], P[1], acc[i][j][1]);
                                acc[i][j][2] = fmaf(dsc[j], P[2], acc[i][j][2]); acc[i][j][3] = fmaf(dsc[j], P[3], acc[i][j][3]);
                            }
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (p + DP < npair) {
#pragma unroll
                        for (int j = 0; j < NT; ++j) { bq[dp][j] = bg_ldw(wq[j] + (size_t)(p + DP) * 64, NTW); sc8[dp][j] = sq[j][p + DP]; }
#pragma unroll
                        for (int h = 0; h < 2; ++h)
#pragma unroll
                            for (int i = 0; i < RT; ++i) aq[2 * dp + h][i] = ap[i][(size_t)(2 * (p + DP) + h) * 64];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    } else
    for (int s0 = 0; s0 < per; s0 += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int s = s0 + d;
            if (s < per) {
#pragma unroll
                for (int i = 0; i < RT; ++i) {
                    union { u32x4 u; bf16x8 v; } a; a.u = aq[d][i];
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        union { u32x4 u; bf16x8 v; } b; b.u = bq[d][j];
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.v, b.v, acc[i][j], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#ifdef Q3_STAMPS
                if (s == 0) { asm volatile("" :: "v"(acc[0][0][0])); Q3_STAMP(g, 2); }
#endif
                if (s + D < per) {
#pragma unroll
                    for (int j = 0; j < NT; ++j) bq[d][j] = bg_ldw(wp[j] + (size_t)(s + D) * 64, NTW);
#pragma unroll
                    for (int i = 0; i < RT; ++i) aq[d][i] = ap[i][(size_t)(s + D) * 64];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
#ifdef Q3_STAMPS
    asm volatile("" :: "v"(acc[0][0][0])); Q3_STAMP(g, 3);
#endif
    // The slice buffer holds TRP = TR / PH registers per wave: instances whose 8 x TR x 256 bytes exceed 64 KiB ((4,3): 96) reduce in PH = 2
    // phases of half their tiles, so that no decode workgroup needs more than 64 KiB of LDS — what is left beside one resident vocoder
    // workgroup. Same sums in the same slice order per element; a thread's elements keep their places (o = tid + 512 it).
    constexpr int PH = BG_PH(RT, NT), TP = RT * NT / PH, TRP = TR / PH, NOUTP = NOUT / PH;
    static_assert(PH == 1 || ((RT * NT) % 2 == 0 && (TRP * 64) % 512 == 0), "two phases need an even tile count and whole items per phase");
    const int epi = g.epi;
#pragma unroll
    for (int ph = 0; ph < PH; ++ph) {
        if (ph) __syncthreads();  // every wave is done reading the previous phase's partials
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                if ((i * NT + j) / TP != ph) continue;
#pragma unroll
                for (int e = 0; e < 4; ++e) part[((size_t)wave * TRP + ((i * NT + j) - ph * TP) * 4 + e) * 64 + lane] = acc[i][j][e];
            }
        __syncthreads();
        if (ph == 0) Q3_STAMP(g, 4);
        // slice partials combined in slice order; element (tile t, reg e, lane l) -> row 16 i + 4 (l >> 4) + e, col 16 j + (l & 15)
        if (epi == Q3_EPI_SWIGLU) {
            // one thread per (gate, up) pair: pair p of the 32 in a (tile, register) block of 64 lanes sits in lanes l = 16 (p / 8) + p % 8 (gate,
            // column c = p % 8 < 8 of the tile) and l + 8 (up, the same row 8 columns further); every thread has work (a loop over all
            // TR * 64 elements left the up lanes idle: half of the epilogue's issue slots)
            constexpr int NP = (TRP * 32 + 511) / 512;
#pragma unroll
            for (int it = 0; it < NP; ++it) {
                const int p = threadIdx.x + it * 512;
                if (p >= TRP * 32) break;  // (TRP * 32 is a multiple of 128: whole waves leave together)
                const int q = p & 31, l = ((q >> 3) << 4) + (q & 7), o = ((p >> 5) << 6) + l;
                const int e = (o >> 6) & 3, t = ph * TP + (o >> 8), i = t / NT, j = t - i * NT;
                const int rl = 16 * i + 4 * (l >> 4) + e, row = row0 + rl, c = l & 15;
                const float sc = g.ssp ? srow[rl] : 1.0f;
                float gt = part[o], up = part[o + 8];
#pragma unroll
                for (int wv = 1; wv < 8; ++wv) { gt = gt + part[(size_t)wv * (TRP * 64) + o]; up = up + part[(size_t)wv * (TRP * 64) + o + 8]; }
                if (row < B) g.yb[q3_atile_off(row, (nb0 + j) * 8 + c, g.N >> 6)] = q3_bf16(q3_swiglu(sc * gt, sc * up));
            }
            continue;
        }
#pragma unroll
        for (int itp = 0; itp < NOUTP; ++itp) {
            const int it = ph * NOUTP + itp, o = threadIdx.x + itp * 512;  // o: place in this phase's buffer; tid + 512 it in the whole tile set
            if (o >= TRP * 64) break;  // whole waves leave together (TRP * 64 is a multiple of 256)
            const int l = o & 63, e = (o >> 6) & 3, t = ph * TP + (o >> 8), i = t / NT, j = t - i * NT;
            const int rl = 16 * i + 4 * (l >> 4) + e, row = row0 + rl, c = l & 15;
            const int col = (nb0 + j) * 16 + c;
            const bool live = row < B;
            const float sc = g.ssp ? srow[rl] : 1.0f;
            float v = part[o];
#pragma unroll
            for (int wv = 1; wv < 8; ++wv) v = v + part[(size_t)wv * (TRP * 64) + o];
            if (g.bias) v = v + bsv[it];
            if (epi == Q3_EPI_STORE) {
                if (live) g.y[bg_yoff(g, row) + col] = g.ssp ? sc * v : v;
            } else if (epi == Q3_EPI_GELU) {
                if (live) g.yb[q3_atile_off(row, col, g.N >> 5)] = q3_bf16(bg_gelu_erf(v));
            } else if (epi == Q3_EPI_RESID) {
                const float xv = g.col_scale ? yres[it] + csv[it] * v : yres[it] + v;
                if (live) g.y[bg_yoff(g, row) + col] = xv;
                if (g.yb && !g.nw_next && live) g.yb[q3_atile_off(row, col, g.N >> 5)] = q3_bf16(xv);
                if (g.nw_next) {  // the consumer's norm inputs: bf16(x * nw) and the tile's sum of squares (16-lane butterfly)
                    if (live) g.yb[q3_atile_off(row, col, g.N >> 5)] = q3_bf16(xv * nwv[it]);
                    float sq = xv * xv;
                    sq = sq + __shfl_xor(sq, 1); sq = sq + __shfl_xor(sq, 2); sq = sq + __shfl_xor(sq, 4); sq = sq + __shfl_xor(sq, 8);
                    if (live && c == 0) g.ssp_out[(size_t)row * g.ld_ssp_out + (nb0 + j)] = sq;
                }
            } else {  // Q3_EPI_ARGMAX: the largest key of the tile's 16 columns goes to keys[row][column tile]; the consumer takes the maximum
                      // over the row's N/16 entries (an atomicMax per (row, tile) on one word per row cost ~8 us of the launch)
                unsigned long long key = q3_argmax_key(g.ssp ? sc * v : v, (uint32_t)col);
#pragma unroll
                for (int m = 1; m <= 8; m <<= 1) { const unsigned long long ok = __shfl_xor(key, m); key = ok > key ? ok : key; }
                if (live && c == 0) g.keys[(size_t)row * g.key_stride + (nb0 + j)] = key;
            }
        }
    }
#ifdef Q3_STAMPS
    Q3_STAMP(g, 5);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    Q3_STAMP(g, 6);
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// The same GEMM for MANY rows (prefill: thousands of prompt rows in one pass): k_bgemm re-reads its weights for every 64-row chunk and
// runs ~12 MFMAs per wave between memory waits, which is right for decode (weights are the traffic) and leaves a 3 000-row prefill at
// ~240 TFLOP/s. Here a workgroup (4 waves, 2 x 2) owns a 128 x 128 tile, each wave a 64 x 64 part over the WHOLE K, and the operand
// tiles arrive by LDS-DMA in a 4-stage ring (one 32-wide K step per stage: 8 A tiles + 8 B tiles of 1 KiB, both already stored in
// fragment order, so a DMA instruction moves one fragment and a fragment read is lane-linear: no swizzle, no bank conflicts).
// The canonical order is kept by construction: a wave runs the MFMA chain of K-slice w from zero accumulators, then
// total = (w == 0) ? chain : total + chain (plain f32 adds, slices ascending) — the bits k_bgemm produces from its 8 waves.
// Tile t = (row tile fastest) goes to XCD t / (tiles / 8): an XCD owns a band of column tiles with all their row tiles, so a weight
// tile is fetched into one L2 while the (smaller) row operand streams through.
// Epilogues STORE / RESID / SWIGLU straight from the D layout (lane (kq, c): rows 4 kq + e, column c of a tile).
// ---------------------------------------------------------------------------------------------------------------------
#define BB_NS 4
__global__ __launch_bounds__(256) void k_bgemm_big(Q3BGemm g) {
    extern __shared__ __attribute__((aligned(16))) char bb_ring[];  // [BB_NS][A: 8 row tiles x 1 KiB | B: 8 column tiles x 1 KiB]
    __shared__ float srow[128];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, wm = wave >> 1, wn = wave & 1, kq = lane >> 4, c = lane & 15;
    int bx = blockIdx.x, by = blockIdx.y;
    {
        const int total = gridDim.x * gridDim.y;
        if ((total & 7) == 0) {
            const int id = blockIdx.x + gridDim.x * blockIdx.y, t = (id & 7) * (total >> 3) + (id >> 3);
            by = t % gridDim.y; bx = t / gridDim.y;
        }
    }
    const int B = g.B, kblocks = g.K >> 5, per = g.K >> 8;  // per = steps per K-slice
    const int rt0 = (g.a_row0 >> 4) + by * 8, rt_last = (g.a_row0 + B - 1) >> 4, nb0 = bx * 8;
    // loader: wave w brings A tiles 2w, 2w+1 and B tiles 2w, 2w+1 of every stage
    const u32x4* asrc[2]; const u32x4* bsrc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        asrc[i] = (const u32x4*)g.a + (size_t)min(rt0 + wave * 2 + i, rt_last) * kblocks * 64 + lane;
        bsrc[i] = (const u32x4*)g.w + (size_t)(nb0 + wave * 2 + i) * kblocks * 64 + lane;
    }
    auto issue = [&](int step) {
        char* st = bb_ring + (size_t)(step % BB_NS) * 16384;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(asrc[i] + (size_t)step * 64),
                                             (__attribute__((address_space(3))) void*)(st + (wave * 2 + i) * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bsrc[i] + (size_t)step * 64),
                                             (__attribute__((address_space(3))) void*)(st + 8192 + (wave * 2 + i) * 1024), 16, 0, 0);
        }
    };
#pragma unroll
    for (int p = 0; p < BB_NS - 1; ++p) issue(p);  // (K >= 256: at least 8 steps)
    // row scales of the split RMSNorm: wave w owns rows w, w + 4, ... of the tile (lane j adds its tiles j, j + 64, ... ascending, then
    // the 64-lane butterfly: DESIGN.md §4.2)
    if (g.ssp) {  // 16 rows per trip: their partials are requested together (one round trip, not one per row)
        for (int i0 = 0; i0 < 32; i0 += 16) {
            float a0[16], a1[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float* sp = g.ssp + (size_t)min(by * 128 + wave + 4 * (i0 + i), B - 1) * g.ld_ssp;
                a0[i] = lane < g.ntiles ? sp[lane] : 0.0f;
                a1[i] = lane + 64 < g.ntiles ? sp[lane + 64] : 0.0f;
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                float a = a0[i];
                if (lane + 64 < g.ntiles) a = a + a1[i];
                if (g.ntiles > 128) {
                    const float* sp = g.ssp + (size_t)min(by * 128 + wave + 4 * (i0 + i), B - 1) * g.ld_ssp;
                    for (int t = lane + 128; t < g.ntiles; t += 64) a = a + sp[t];
                }
                a0[i] = a;
            }
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1)
#pragma unroll
                for (int i = 0; i < 16; ++i) a0[i] = a0[i] + __shfl_xor(a0[i], m);
            if (lane == 0) {
#pragma unroll
                for (int i = 0; i < 16; ++i) srow[wave + 4 * (i0 + i)] = 1.0f / sqrtf(a0[i] / (float)g.d_norm + g.eps);
            }
        }
    }
    f32x4 acc[4][4], tot[4][4];
    const int nsteps = kblocks;
    int in_slice = 0, slice = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int step = 0; step < nsteps; ++step) {
        if (step + BB_NS - 2 < nsteps) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((BB_NS - 2) * 4) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        Q3_LDS_BARRIER();  // (not __syncthreads(): that would wait for every stage in flight, q3_kernels.h)
        if (step + BB_NS - 1 < nsteps) issue(step + BB_NS - 1);
        const char* st = bb_ring + (size_t)(step % BB_NS) * 16384;
        bf16x8 a_[4], b_[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) a_[i] = *(const bf16x8*)(st + (wm * 4 + i) * 1024 + lane * 16);
#pragma unroll
        for (int j = 0; j < 4; ++j) b_[j] = *(const bf16x8*)(st + 8192 + (wn * 4 + j) * 1024 + lane * 16);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_[i], b_[j], acc[i][j], 0, 0, 0);
        if (++in_slice == per) {  // the slice's chain is complete: RAW = ((s_0 + s_1) + ...) + s_7
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (slice == 0) tot[i][j] = acc[i][j];
                    else { tot[i][j][0] = tot[i][j][0] + acc[i][j][0]; tot[i][j][1] = tot[i][j][1] + acc[i][j][1]; tot[i][j][2] = tot[i][j][2] + acc[i][j][2]; tot[i][j][3] = tot[i][j][3] + acc[i][j][3]; }
                    acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
                }
            in_slice = 0; ++slice;
        }
    }
    __syncthreads();  // srow
    const int epi = g.epi;
    if (epi == Q3_EPI_RESID) {  // the residual operands of the whole 64 x 64 part in one batch of loads (the chain accumulators are free now)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int row = min(by * 128 + wm * 64 + i * 16 + 4 * kq + e, B - 1);
                    acc[i][j][e] = g.y[(size_t)row * g.ldy + (nb0 + wn * 4 + j) * 16 + c];
                }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int rl = wm * 64 + i * 16 + 4 * kq + e, row = by * 128 + rl;
            const bool live = row < B;
            const float sc = g.ssp ? srow[rl] : 1.0f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int tile = nb0 + wn * 4 + j, col = tile * 16 + c;
                const float v = tot[i][j][e];
                if (epi == Q3_EPI_SWIGLU) {  // gate = columns 0-7 of a tile, up = the same row 8 columns further (lane + 8)
                    const float up = __shfl(v, (lane + 8) & 63);
                    if (live && c < 8) g.yb[q3_atile_off(row, tile * 8 + c, g.N >> 6)] = q3_bf16(q3_swiglu(sc * v, sc * up));
                } else if (epi == Q3_EPI_STORE) {
                    if (live) g.y[(size_t)row * g.ldy + col] = g.ssp ? sc * v : v;
                } else {  // Q3_EPI_RESID
                    const float xv = acc[i][j][e] + v;
                    if (live) g.y[(size_t)row * g.ldy + col] = xv;
                    if (g.nw_next) {
                        if (live) g.yb[q3_atile_off(row, col, g.N >> 5)] = q3_bf16(xv * g.nw_next[col]);
                        float sq = xv * xv;
                        sq = sq + __shfl_xor(sq, 1); sq = sq + __shfl_xor(sq, 2); sq = sq + __shfl_xor(sq, 4); sq = sq + __shfl_xor(sq, 8);
                        if (live && c == 0) g.ssp_out[(size_t)row * g.ld_ssp_out + tile] = sq;
                    }
                }
            }
        }
}
static bool bg_big_ok(const Q3BGemm& g) {
    return g.B >= 256 && !g.wscale && (g.epi == Q3_EPI_STORE || g.epi == Q3_EPI_RESID || g.epi == Q3_EPI_SWIGLU) && g.N % 128 == 0 && g.a_row0 % 16 == 0 &&
           !g.bias && !g.col_scale && !g.seg_rows && !(g.epi == Q3_EPI_RESID && g.yb && !g.nw_next);
}
static void bg_launch_big(const Q3BGemm& g, hipStream_t s) {
    const dim3 grid(g.N / 128, (g.B + 127) / 128);
    hipLaunchKernelGGL(k_bgemm_big, grid, dim3(256), BB_NS * 16384, s, g);
}

// Operand steps in flight per wave (a step = 4 (RT + NT) registers; a K slice of 256 is 8 steps): 8, 5, 4, 3, 2 for RT + NT = <= 3, 4, 5, 6, 7,
// which keeps every instance at <= ~146 registers per wave. Round 2 first ran 5-8 steps everywhere (160-230 registers): the shallow
// depths are 3 % faster on the Talker's gate/up GEMM and 20 % on launches with more than one workgroup per CU (tools/bgemm_tune.hip), and
// the whole job gains ~2.5 % with or without the vocoder alongside (bench.py A/B on one box: 551 -> 567 and 645 -> 661 audio-sec/s).
// For RT + NT = 7 depths 1, 2 and 3 time the same (16.9 / 16.7 / 16.7 us on the Talker's gate/up): eight waves per workgroup hide the
// latency, not the depth.
template <int RT, int NT>
struct BgInst {
    static constexpr int S = RT + NT;
    static constexpr int D = S <= 3 ? 8 : (S == 4 ? 5 : (S == 5 ? 4 : (S == 6 ? 3 : 2)));
    static constexpr size_t lds = (size_t)8 * RT * NT * 4 * 64 * 4 / BG_PH(RT, NT);
    static void prepare() {  // dynamic LDS above 64 KiB has to be allowed per kernel
        if (lds > 65536) {
            hipFuncSetAttribute((const void*)k_bgemm<RT, NT, D, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipFuncSetAttribute((const void*)k_bgemm<RT, NT, D, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipFuncSetAttribute((const void*)k_bgemm<RT, NT, D, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipFuncSetAttribute((const void*)k_bgemm<RT, NT, D, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        }
    }
    static void launch(const Q3BGemm& g, dim3 grid, hipStream_t s) {
        // once-read weight streams (the Talker at decode: every tile goes to exactly one workgroup) take non-temporal loads
        const bool nt = g.w_once && grid.y == 1;
        if (g.wscale) {  // Q8_0 weights
            if (nt) hipLaunchKernelGGL((k_bgemm<RT, NT, D, true, true>), grid, dim3(512), lds, s, g);
            else hipLaunchKernelGGL((k_bgemm<RT, NT, D, false, true>), grid, dim3(512), lds, s, g);
        } else if (nt) hipLaunchKernelGGL((k_bgemm<RT, NT, D, true, false>), grid, dim3(512), lds, s, g);
        else hipLaunchKernelGGL((k_bgemm<RT, NT, D, false, false>), grid, dim3(512), lds, s, g);
    }
};
#define BG_EACH(X) X(1, 1) X(1, 2) X(1, 3) X(2, 1) X(2, 2) X(2, 3) X(3, 1) X(3, 2) X(3, 3) X(4, 1) X(4, 2) X(4, 3)
// once per process, outside any stream capture (the engine calls it before it records its graphs)
static int g_lds_cap_kb = 0;  // Q3TTS_BG_LDS_CAP (KiB), read once: tile instances whose slice buffer is larger are not chosen (experiment: DESIGN.md §16)
static int g_big_policy = 0;  // Q3TTS_BG_BIG, read once: 1 = the many-row kernel whenever it is eligible, -1 = never, 0 = when it fills the chip
void q3_bgemm_prepare() {  // once per DEVICE (function attributes are per device: q3tts_node_* drives several from one process)
    static Q3PerDevice pd;
    static std::once_flag env_once;
    std::call_once(env_once, []() {
        const char* ev = getenv("Q3TTS_BG_BIG"); g_big_policy = ev ? atoi(ev) : 0;
        ev = getenv("Q3TTS_BG_LDS_CAP"); g_lds_cap_kb = ev ? atoi(ev) : 0;
    });
    pd.ensure(1, []() {
        hipFuncSetAttribute((const void*)k_bgemm_big, hipFuncAttributeMaxDynamicSharedMemorySize, BB_NS * 16384);
#define P(RT_, NT_) BgInst<RT_, NT_>::prepare();
        BG_EACH(P)
#undef P
    });
}

// Tile choice: a launch is bound by the operand bytes a CU's load path takes in (x from L2, weights from HBM / L2), so pick the
// (RT, NT) with the fewest bytes per workgroup-round — 32 (RT + NT) bytes per k — counted over ceil(workgroups / 256) rounds.
// The choice never changes a result (see the header).
// tuning hook (tools/bgemm_tune.hip): force one tile instance for every following launch; (0, 0) restores the cost model
static int g_force_rt = 0, g_force_nt = 0;
void q3_bgemm_force(int rt, int nt) { g_force_rt = rt; g_force_nt = nt; }
// test hook (q3tts_k_bgemm_policy): the many-row kernel always (1) / never (-1) / by the fill rule (0), overriding Q3TTS_BG_BIG
void q3_bgemm_big_policy(int policy) { q3_bgemm_prepare(); g_big_policy = policy; }

// The instance the launcher takes for a shape (also q3tts_k_bgemm_pick: bench.py names the kernel symbol of a probed launch from it
// instead of hard-coding what the cost model is assumed to choose). big = 1: k_bgemm_big.
static void bg_pick(const Q3BGemm& g, int* rt, int* nt, int* big) {
    *big = 0;
    if (g.B >= 256 && g_force_rt == 0 && bg_big_ok(g)) {
        // Q3TTS_BG_BIG: 1 = the many-row kernel whenever it is eligible, -1 = never (A/B runs and the tests: the results are the same bits);
        // default: when it fills the chip — at least one 128 x 128 tile per CU (prefill; the vocoder's 256-row GEMMs stay on k_bgemm)
        const int policy = g_big_policy;
        const long wgs = (long)(g.N / 128) * ((g.B + 127) / 128);
        if (policy > 0 || (policy == 0 && wgs >= 256)) { *big = 1; *rt = 8; *nt = 8; return; }
    }
    const int tiles = g.N / 16;
    int bestRT = 1, bestNT = 1; long bestCost = -1, bestWgs = 0;
    for (int RT = 1; RT <= 4; ++RT)
        for (int NT = 1; NT <= 3; ++NT) {
            if (tiles % NT) continue;
            if (g_lds_cap_kb > 0 && 8 * RT * NT / BG_PH(RT, NT) > g_lds_cap_kb && !(RT == 1 && NT == 1)) continue;
            if (g.B > 64 && RT != 4 && !(RT == 2 && g.B <= 128)) continue;  // many rows (prefill): 64-row chunks
            const long chunks = (g.B + 16 * RT - 1) / (16 * RT);
            if (g.B <= 64 && RT > 1 && 16 * (RT - 1) * chunks >= g.B) continue;  // a smaller RT covers the rows with the same chunk count
            const long wgs = (long)(tiles / NT) * chunks, rounds = (wgs + 255) / 256;
            const long cost = rounds * ((g.wscale ? 32L * RT + 17L * NT : 32L * (RT + NT)) * g.K + 24000L);  // + a fixed cost per round (ramp, reduction); Q8_0 weights: 17 bytes per k per column tile
            if (bestCost < 0 || cost < bestCost || (cost == bestCost && (wgs > bestWgs || (wgs == bestWgs && NT > bestNT)))) { bestCost = cost; bestRT = RT; bestNT = NT; bestWgs = wgs; }
        }
    if (g_force_rt > 0 && g_force_nt > 0 && tiles % g_force_nt == 0) { bestRT = g_force_rt; bestNT = g_force_nt; }
    *rt = bestRT; *nt = bestNT;
}
// rt/nt/d: the tile instance and its ring depth; ntw: non-temporal weight loads (once-read weights, one row chunk); big: k_bgemm_big
void q3_bgemm_pick(const Q3BGemm& g, int* rt, int* nt, int* d, int* ntw, int* big) {
    q3_bgemm_prepare();
    bg_pick(g, rt, nt, big);
    *d = 0; *ntw = 0;
    if (*big) return;
#define X(RT_, NT_) if (*rt == RT_ && *nt == NT_) *d = BgInst<RT_, NT_>::D;
    BG_EACH(X)
#undef X
    *ntw = (g.w_once && (g.B + 16 * *rt - 1) / (16 * *rt) == 1) ? 1 : 0;
}

static int bg_check(const Q3BGemm& g) {
    if (g.B < 1 || g.N % 16 || g.K % 256 || g.K < 256 || !g.a || !g.w || g.a_row0 < 0) return -1;
    if (g.wscale && g.K % 512) return -1;  // Q8_0: two blocks per 16-byte weight load, an even number of blocks per K slice
    if (g.yb && (((g.epi == Q3_EPI_RESID || g.epi == Q3_EPI_GELU) && g.N % 32) || (g.epi == Q3_EPI_SWIGLU && g.N % 64))) return -1;  // the A-tiled output has N (N/2) columns in 32-blocks
    if ((g.epi == Q3_EPI_SWIGLU || g.epi == Q3_EPI_GELU) && (!g.yb)) return -1;
    if (g.bias && g.bias_n < 1) return -1;
    if (g.ssp && g.ntiles < 1) return -1;
    return 0;
}

int q3_launch_bgemm(const Q3BGemm& g, hipStream_t s) {
    if (bg_check(g)) return -1;
    q3_bgemm_prepare();
    int bestRT, bestNT, big;
    bg_pick(g, &bestRT, &bestNT, &big);
    if (big) { bg_launch_big(g, s); return 0; }
    const dim3 grid(g.N / 16 / bestNT, (g.B + 16 * bestRT - 1) / (16 * bestRT));
#define L(RT_, NT_) if (bestRT == RT_ && bestNT == NT_) { BgInst<RT_, NT_>::launch(g, grid, s); return 0; }
    BG_EACH(L)
#undef L
    return -1;
}

// ---------------------------------------------------------------------------------------------------------------------
// norm inputs of f32 rows (prompt rows before prefill; test hook): xb = bf16(x * nw), ssp[t] = sum of squares of tile t.
// 256 threads per row; 16 consecutive lanes own one tile.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_norm_inputs(const float* x, int ldx, int d, const float* nw, uint16_t* xb, int xb_row0, float* ssp, int ld_ssp) {
    const int row = blockIdx.x;
    for (int i = threadIdx.x; i < d; i += 256)
        q3_norm_out(x[(size_t)row * ldx + i], nw[i], xb + q3_atile_off(xb_row0 + row, i, d >> 5), ssp + (size_t)row * ld_ssp + (i >> 4), (i & 15) == 0);
}
// the W8A8 consumer's form: v = x * nw quantised per 32 columns by ggml's rule (q3_q8_out32: a half wave = one block), ssp as above
__global__ __launch_bounds__(256) void k_norm_inputs_q8(const float* x, int ldx, int d, const float* nw, int8_t* xq, uint16_t* xscale, int rt16, float* ssp, int ld_ssp) {
    const int row = blockIdx.x;
    for (int i = threadIdx.x; i < d; i += 256) {
        const float v = x[(size_t)row * ldx + i];
        q3_q8_out32(v * nw[i], row, i, d >> 6, rt16, xq, xscale);
        float sq = v * v;
        sq = sq + __shfl_xor(sq, 1); sq = sq + __shfl_xor(sq, 2); sq = sq + __shfl_xor(sq, 4); sq = sq + __shfl_xor(sq, 8);
        if ((i & 15) == 0) ssp[(size_t)row * ld_ssp + (i >> 4)] = sq;
    }
}
void q3_launch_norm_inputs_q8(const float* x, int ldx, int rows, int d, const float* nw, int8_t* xq, uint16_t* xscale, int rt16, float* ssp, int ld_ssp, hipStream_t s) {
    hipLaunchKernelGGL(k_norm_inputs_q8, dim3(rows), dim3(256), 0, s, x, ldx, d, nw, xq, xscale, rt16, ssp, ld_ssp);
}
// rows [0, rows) of x -> A-tiled rows [xb_row0, xb_row0 + rows) of xb, ssp rows [0, rows)
void q3_launch_norm_inputs(const float* x, int ldx, int rows, int d, const float* nw, uint16_t* xb, int xb_row0, float* ssp, int ld_ssp, hipStream_t s) {
    hipLaunchKernelGGL(k_norm_inputs, dim3(rows), dim3(256), 0, s, x, ldx, d, nw, xb, xb_row0, ssp, ld_ssp);
}
