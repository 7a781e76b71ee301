// Two numbers that decide whether the Predictor pass (27 launches of ~6.6 us, weights resident in the Infinity Cache) should become ONE
// persistent kernel whose workgroups are grouped per XCD (/root/reference/src/tts/engine.rs:575-610):
//   (i)  the cost of a seam INSIDE an XCD: 32 workgroups that share one L2 hand a 16 KiB activation block to each other (every workgroup
//        writes its 512 B, arrives on a per-XCD counter, waits for its group, reads the whole block past its L1) — no cross-XCD coherence,
//        so no L2 write-back / bypass; and
//   (ii) per-CU ingest when all eight XCDs stream the SAME 157 MB (every group needs every weight): 8 x the bytes out of the Infinity Cache;
//   (iii) both together in the Predictor's schedule: 27 phases per pass whose per-workgroup weight shares add up to 1/32 of 157 MB + a head,
//        the ring of the next phase requested BEFORE the seam (weights never depend on activations), against 178 us per pass today.
// Groups are formed at run time: a workgroup reads its XCC id (s_getreg HW_REG_XCC_ID), takes a ticket on its XCD's counter and learns the
// group size after one chip-wide rendezvous. Every wait is bounded (~20 ms), a timeout raises a flag that ends all loops.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/xcd_bench tools/xcd_bench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct Ctl {
    unsigned* tickets;   // [8][64]: per-XCD ticket counter (one 256-byte line each)
    unsigned* arrive;    // [8][64]: per-XCD phase counter
    unsigned* all;       // chip-wide rendezvous counter (once per launch)
    unsigned* failed;
    unsigned* gsize;     // [8] group sizes seen (host print)
};

__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15u; }  // HW_REG_XCC_ID[3:0]

// L2-level primitives (one XCD): atomics execute in the L2, loads with sc0 miss the (non-coherent, write-through) vector L1
__device__ __forceinline__ void l2_add(unsigned* p, unsigned v) { asm volatile("global_atomic_add %0, %1, off" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ unsigned l2_add_ret(unsigned* p, unsigned v) {
    unsigned r; asm volatile("global_atomic_add %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p), "v"(v) : "memory"); return r;
}
// the counter as the L2 holds it: an atomic that adds nothing (atomics always execute in the L2; a load with sc0 may still hit a stale L1 line:
// four of eight XCDs hung on such a poll)
__device__ __forceinline__ unsigned l2_load(const unsigned* p) { return l2_add_ret(const_cast<unsigned*>(p), 0u); }
// after a seam: drop this CU's vector-L1 lines (workgroup-scope invalidate), so that plain loads of the block are served by the L2
// inv 0: buffer_inv sc0 (measured: does NOT drop the L1 lines here — stale reads unless the weight stream happens to evict them); 1: buffer_inv sc1;
// 2: no invalidate, the block is read with sc1 loads; 3: with sc0 sc1 loads
__device__ __forceinline__ void l1_inv(int inv) {
    if (inv == 0) asm volatile("buffer_inv sc0" ::: "memory");
    else if (inv == 1) asm volatile("buffer_inv sc1" ::: "memory");
}
__device__ __forceinline__ u32x4 blk_load(const u32x4* p, int inv) {
    u32x4 r;
    if (inv == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory");
    else if (inv == 3) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(r) : "v"(p) : "memory");
    else r = *p;
    return r;
}

// wait until *ctr >= target (thread 0 of the workgroup polls, the others wait at the barrier); false on timeout / failure elsewhere
__device__ __forceinline__ bool wait_ge(const unsigned* ctr, unsigned target, unsigned* failed, bool agent_scope) {
    __shared__ int ok;
    if (threadIdx.x == 0) {
        int good = 1; unsigned spins = 0;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        for (;;) {
            const unsigned v = agent_scope ? __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : l2_load(ctr);
            if ((int)(v - target) >= 0) break;
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 255) == 0) {
                if (__builtin_amdgcn_s_memrealtime() - t0 > 2000000ull || __hip_atomic_load(failed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    __hip_atomic_store(failed, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); good = 0; break;
                }
            }
        }
        ok = good;
    }
    __syncthreads();
    return ok != 0;
}

// mode 0: seams only; 1: ingest only (one "pass": the workgroup's 1/n share of wbytes, chunked over `phases`); 2: both, ring of the next
// phase requested before the seam. share[phase] = KiB of weights this workgroup streams in that phase (sums to its share of the matrix set).
template <int NWAVE, int INFL>
__global__ __launch_bounds__(NWAVE * 64) void k_pass(Ctl c, const u32x4* __restrict__ w, size_t wkib_total, const int* __restrict__ share, int phases, int passes,
                                                     u32x4* act, int mode, int inv, uint32_t* sink) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, tid = threadIdx.x;
    const unsigned x = xcc_id();
    __shared__ unsigned s_t, s_n;
    if (tid == 0) {
        s_t = l2_add_ret(c.tickets + x * 64, 1u);   // ticket within the XCD (counters are zeroed by the host before every launch)
        __hip_atomic_fetch_add(c.all, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!wait_ge(c.all, gridDim.x, c.failed, true)) { if (tid == 0) __hip_atomic_fetch_or(c.failed, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }   // everybody has a ticket: the group sizes are final
    if (tid == 0) { s_n = l2_load(c.tickets + x * 64); if (s_t == 0) c.gsize[x] = s_n; }
    __syncthreads();
    const unsigned t = s_t, n = s_n;
    // this workgroup's slice of the weight set: KiB [t * per, (t + 1) * per) of wkib_total, walked phase by phase
    const size_t per = wkib_total / n;
    const u32x4* wp = w + (size_t)t * per * 64 + lane;
    u32x4 acc = (u32x4){0, 0, 0, 0};
    u32x4* myact = act + (size_t)x * 2 * 1024;   // per XCD: two 16 KiB blocks (phase parity)
    unsigned arrived = 0, wrong = 0;
    u32x4 pre[INFL];      // mode 2: the first INFL KiB of the NEXT phase, requested before the seam (weights never depend on activations)
    bool have_pre = false;
    for (int pass = 0; pass < passes; ++pass) {
        size_t k = 0;   // KiB consumed of this pass
        for (int p = 0; p < phases; ++p) {
            const int kib = mode == 0 ? 0 : share[p];
            // ---- stream this phase's weights: wave wv takes KiB k + wv * INFL .. + INFL, then NWAVE * INFL further; INFL loads per wave in flight
            int k0 = wave * INFL;
            if (have_pre) {
#pragma unroll
                for (int j = 0; j < INFL; ++j) acc ^= pre[j];
                k0 += NWAVE * INFL; have_pre = false;
            }
            for (; k0 < kib; k0 += NWAVE * INFL) {
                u32x4 v[INFL];
#pragma unroll
                for (int j = 0; j < INFL; ++j) {
                    const size_t kk = k + (size_t)min(k0 + j, kib - 1);
                    v[j] = wp[(kk % per) * 64];
                }
#pragma unroll
                for (int j = 0; j < INFL; ++j) acc ^= v[j];
            }
            k += (size_t)kib;
            if (mode == 1) continue;
            // ---- seam: write my 16 KiB / n share, arrive, wait for the group, read the whole block past L1
            u32x4* blk = myact + (size_t)(p & 1) * 1024;
            const int mine = 1024 / (int)n;   // u32x4 per workgroup
            if (tid < mine) blk[t * mine + tid] = (u32x4){acc.x + (unsigned)p, t, (unsigned)pass * 64u + (unsigned)p, (unsigned)tid};
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the store has reached the L2
            __syncthreads();
            arrived += n;
            if (tid == 0) l2_add(c.arrive + x * 64, 1u);
            if (mode == 2 && wave != 0) {  // the next phase's first ring goes out now, behind the store (wave 0 polls: its loads return in order, it carries none)
                const int pn = p + 1 < phases ? p + 1 : 0;
                const int kn = share[pn];
                const size_t kb = p + 1 < phases ? k : 0;
                if (kn > 0 && wave * INFL < kn) {
#pragma unroll
                    for (int j = 0; j < INFL; ++j) pre[j] = wp[((kb + (size_t)min(wave * INFL + j, kn - 1)) % per) * 64];
                    have_pre = true;
                }
            }
            if (!wait_ge(c.arrive + x * 64, arrived, c.failed, false)) return;
            l1_inv(inv);
            for (int i = tid; i < 1024; i += NWAVE * 64) {
                const u32x4 r = blk_load(blk + i, inv);
                wrong += (r.y != (unsigned)(i / mine)) + (r.z != (unsigned)pass * 64u + (unsigned)p) + (r.w != (unsigned)(i % mine));   // every share is this phase's
                acc.y ^= r.x;
            }
        }
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
    if (wrong) atomicAdd(sink + 1, wrong);
}

template <int NWAVE, int INFL>
static int run(hipStream_t s, Ctl c, const u32x4* w, size_t wkib, const int* share_dev, int phases, int passes, u32x4* act, int mode, int inv, uint32_t* sink, const char* what, double kib_per_wg_pass) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e30f;
    unsigned gs[8] = {0};
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipMemsetAsync(c.tickets, 0, 8 * 256, s)); CK(hipMemsetAsync(c.arrive, 0, 8 * 256, s)); CK(hipMemsetAsync(c.all, 0, 256, s)); CK(hipMemsetAsync(c.gsize, 0, 64, s));
        CK(hipEventRecord(a, s));
        hipLaunchKernelGGL((k_pass<NWAVE, INFL>), dim3(256), dim3(NWAVE * 64), 0, s, c, w, wkib, share_dev, phases, passes, act, mode, inv, sink);
        CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b));
        float ms = 0; hipEventElapsedTime(&ms, a, b);
        unsigned f = 0; CK(hipMemcpy(&f, c.failed, 4, hipMemcpyDeviceToHost));
        if (f) {
            unsigned tk[8 * 64], ar[8 * 64], al = 0;
            CK(hipMemcpy(tk, c.tickets, sizeof(tk), hipMemcpyDeviceToHost)); CK(hipMemcpy(ar, c.arrive, sizeof(ar), hipMemcpyDeviceToHost)); CK(hipMemcpy(&al, c.all, 4, hipMemcpyDeviceToHost));
            printf("%s: TIMED OUT (flag %u); rendezvous %u of 256; tickets", what, f, al);
            for (int x = 0; x < 8; ++x) printf(" %u", tk[x * 64]);
            printf("; arrivals");
            for (int x = 0; x < 8; ++x) printf(" %u", ar[x * 64]);
            printf("\n");
            return 1;
        }
        if (rep > 0 && ms < best) best = ms;
        CK(hipMemcpy(gs, c.gsize, 32, hipMemcpyDeviceToHost));
    }
    const double us_pass = best * 1e3 / passes;
    printf("%-58s inv %d, %d waves, %2d loads in flight: %8.2f us per pass (%5.2f us per phase)", what, inv, NWAVE, INFL, us_pass, us_pass / phases);
    if (kib_per_wg_pass > 0) printf("  %6.1f GB/s per CU, %5.2f TB/s chip", kib_per_wg_pass * 1024.0 / (us_pass * 1e-6) / 1e9, 256.0 * kib_per_wg_pass * 1024.0 / (us_pass * 1e-6) / 1e12);
    uint32_t hs[2] = {0, 0}; CK(hipMemcpy(hs, sink, 8, hipMemcpyDeviceToHost));
    printf("   groups %u %u %u %u %u %u %u %u  stale/wrong words read: %u\n", gs[0], gs[1], gs[2], gs[3], gs[4], gs[5], gs[6], gs[7], hs[1]);
    CK(hipMemset(sink, 0, 8));
    return 0;
}

int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    Ctl c;
    CK(hipMalloc(&c.tickets, 8 * 256)); CK(hipMalloc(&c.arrive, 8 * 256)); CK(hipMalloc(&c.all, 256)); CK(hipMalloc(&c.failed, 256)); CK(hipMalloc(&c.gsize, 64));
    CK(hipMemset(c.failed, 0, 256));
    // the Predictor's matrices: 5 x (QKV 1024 -> 4096, O 2048 -> 1024, gate/up 1024 -> 6144, down 3072 -> 1024) + one head 1024 -> 2048, bf16
    const size_t layer_kib = (size_t)(4096 * 1024 + 1024 * 2048 + 6144 * 1024 + 1024 * 3072) * 2 / 1024, head_kib = (size_t)2048 * 1024 * 2 / 1024;
    const size_t wkib = 5 * layer_kib + head_kib;   // 157 MB + 4 MB
    u32x4* w; u32x4* act; uint32_t* sink; int* share_dev;
    CK(hipMalloc(&w, wkib * 1024)); CK(hipMemset(w, 1, wkib * 1024)); CK(hipMalloc(&act, 8 * 2 * 16384)); CK(hipMalloc(&sink, 64)); CK(hipMemset(sink, 0, 64));
    // 27 phases per pass: per block QKV | attention (no weights) | O | gate/up | down, then head | next-row glue (no weights); shares for groups of 32
    std::vector<int> share;
    for (int l = 0; l < 5; ++l) { share.push_back(4096 * 1024 * 2 / 1024 / 32); share.push_back(0); share.push_back(1024 * 2048 * 2 / 1024 / 32); share.push_back(6144 * 1024 * 2 / 1024 / 32); share.push_back(1024 * 3072 * 2 / 1024 / 32); }
    share.push_back((int)head_kib / 32); share.push_back(0);
    double kib_pass = 0; for (int v : share) kib_pass += v;
    CK(hipMalloc(&share_dev, share.size() * 4)); CK(hipMemcpy(share_dev, share.data(), share.size() * 4, hipMemcpyHostToDevice));
    const int phases = (int)share.size(), passes = 15;
    printf("weight set %.1f MB (every XCD group streams all of it: %.2f GB per pass chip-wide), %d phases per pass, %d passes per launch, %.0f KiB per workgroup per pass\n",
           wkib / 1024.0, 8.0 * wkib / 1048576.0, phases, passes, kib_pass);
    int rc = 0;
    for (int inv = 0; inv < 4 && !rc; ++inv) rc |= run<8, 8>(s, c, w, wkib, share_dev, phases, passes, act, 0, inv, sink, "(i) seams only: 512 B write + arrive + wait + 16 KiB read", 0);
    rc |= run<4, 8>(s, c, w, wkib, share_dev, phases, passes, act, 0, 1, sink, "(i) seams only: 512 B write + arrive + wait + 16 KiB read", 0);
    if (rc) return 1;
    rc |= run<8, 8>(s, c, w, wkib, share_dev, phases, passes, act, 1, 1, sink, "(ii) ingest only, all XCDs read the same 161 MB", kib_pass);
    rc |= run<8, 16>(s, c, w, wkib, share_dev, phases, passes, act, 1, 1, sink, "(ii) ingest only, all XCDs read the same 161 MB", kib_pass);
    rc |= run<4, 16>(s, c, w, wkib, share_dev, phases, passes, act, 1, 1, sink, "(ii) ingest only, all XCDs read the same 161 MB", kib_pass);
    if (rc) return 1;
    rc |= run<8, 8>(s, c, w, wkib, share_dev, phases, passes, act, 2, 1, sink, "(iii) Predictor schedule: ingest + a seam per phase", kib_pass);
    rc |= run<8, 16>(s, c, w, wkib, share_dev, phases, passes, act, 2, 1, sink, "(iii) Predictor schedule: ingest + a seam per phase", kib_pass);
    rc |= run<4, 16>(s, c, w, wkib, share_dev, phases, passes, act, 2, 1, sink, "(iii) Predictor schedule: ingest + a seam per phase", kib_pass);
    return rc;
}
