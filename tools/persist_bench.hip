// Would a PERSISTENT Talker step — 256 workgroups that stay resident through all 28 blocks and meet at a chip-wide barrier where a launch
// boundary is today — beat the launch chain (53.7 us per block for 100.7 MB of weights; HBM alone: ~14 us)? Weights never depend on
// activations, so a workgroup can request the first ring of the NEXT phase's weights before it waits at the barrier: the 4.5 us of a
// cross-XCD hand-off (tools/barrier_bench.hip) would run under the weight stream instead of in front of it.
// Model of one block = 5 phases with the Talker's bytes at 64 rows (QKV 16.8 MB, attention: 18 MB of K / V, O 8.4 MB, gate/up 50.3 MB, down
// 25.2 MB; cold: the weight set rotates through 6 GiB). Per phase a workgroup (8 waves) streams its 1 / 256 share with INFL KiB per wave in
// flight, the first ring issued BEFORE the barrier of the previous phase; hands over 1 KiB with write-through stores (sc0 sc1); after the
// barrier reads the phase's whole operand (the rows of its tile: 32 ... 256 KiB) with PLAIN loads from a buffer no earlier phase has used
// (so the XCD's L2 serves all but the first reader: what a kernel boundary gives today). Every wait is bounded (~20 ms).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/persist_bench tools/persist_bench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ bool grid_barrier(unsigned* count, unsigned target, unsigned* failed) {
    __shared__ int ok;
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int good = 1; unsigned spins = 0;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while ((int)(__hip_atomic_load(count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 255) == 0 && (__builtin_amdgcn_s_memrealtime() - t0 > 2000000ull || __hip_atomic_load(failed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                __hip_atomic_store(failed, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); good = 0; break;
            }
        }
        ok = good;
    }
    __syncthreads();
    return ok != 0;
}

struct Phase { int w_kib; int a_kib; };   // per workgroup: KiB of weights to stream; KiB of activations (shared operand) to read after the barrier
__constant__ Phase c_ph[8];

template <int INFL>
__global__ __launch_bounds__(512) void k_persist(const u32x4* __restrict__ w, size_t w_u4_total, const u32x4* __restrict__ act, size_t act_u4_total, u32x4* outb,
                                                 int nph, int layers, unsigned* count, unsigned* failed, int with_barrier, int prefetch, uint32_t* sink) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, tid = threadIdx.x;
    u32x4 acc = (u32x4){0, 0, 0, 0};
    size_t wpos = (size_t)blockIdx.x * 4096;   // this workgroup's cursor into the weight set (u32x4 units), advanced phase by phase, wrapped
    size_t apos = 0;
    unsigned nbar = 0;
    u32x4 pre[INFL]; bool have_pre = false;
    for (int l = 0; l < layers; ++l)
        for (int p = 0; p < nph; ++p) {
            const int kib = c_ph[p].w_kib;
            const u32x4* wp = w + (wpos % (w_u4_total - (size_t)1 << 20)) + lane;
            int k0 = wave * INFL;
            if (have_pre) {
#pragma unroll
                for (int j = 0; j < INFL; ++j) acc ^= pre[j];
                k0 += 8 * INFL; have_pre = false;
            }
            for (; k0 < kib; k0 += 8 * INFL) {
                u32x4 v[INFL];
#pragma unroll
                for (int j = 0; j < INFL; ++j) v[j] = __builtin_nontemporal_load(wp + (size_t)min(k0 + j, kib - 1) * 64);
#pragma unroll
                for (int j = 0; j < INFL; ++j) acc ^= v[j];
            }
            wpos += (size_t)kib * 64 * 256;   // (the other workgroups' shares lie in between: every byte of the set is read once per pass)
            // hand-over: 1 KiB per workgroup, written through to memory
            if (tid < 64) {
                u32x4* o = outb + ((size_t)(nbar & 63) * gridDim.x + blockIdx.x) * 64 + tid;
                asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(o), "v"(acc) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            // the next phase's first ring goes out before the wait (weights never depend on activations)
            const int pn = p + 1 < nph ? p + 1 : 0, kn = c_ph[pn].w_kib;
            if (prefetch && wave != 0 && wave * INFL < kn) {
                const u32x4* wn = w + (wpos % (w_u4_total - (size_t)1 << 20)) + lane;
#pragma unroll
                for (int j = 0; j < INFL; ++j) pre[j] = __builtin_nontemporal_load(wn + (size_t)min(wave * INFL + j, kn - 1) * 64);
                have_pre = true;
            }
            ++nbar;
            if (with_barrier && !grid_barrier(count, nbar * gridDim.x, failed)) return;
            // the phase's operand: rows written by everybody, read by everybody (plain loads: the XCD's L2 shares them), a fresh buffer per phase
            const int akib = c_ph[pn].a_kib;
            const u32x4* ap = act + (apos % (act_u4_total - (size_t)1 << 16)) + lane;
            for (int k = wave; k < akib; k += 8) { const u32x4 v = ap[(size_t)k * 64]; acc.y ^= v.x; }
            apos += (size_t)akib * 64;
        }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const size_t WB = (size_t)6 << 30, AB = (size_t)256 << 20;
    u32x4 *w, *act, *outb; unsigned *count, *failed; uint32_t* sink;
    CK(hipMalloc(&w, WB)); CK(hipMemset(w, 1, WB)); CK(hipMalloc(&act, AB)); CK(hipMemset(act, 2, AB)); CK(hipMalloc(&outb, (size_t)64 * 256 * 1024));
    CK(hipMalloc(&count, 256)); CK(hipMalloc(&failed, 256)); CK(hipMalloc(&sink, 64)); CK(hipMemset(failed, 0, 256)); CK(hipMemset(sink, 0, 64));
    // per-workgroup shares at 64 rows: QKV 2048 -> 4096 | attention (K / V of ~150 keys x 64 slots) | O | gate/up | down ; operands read after the barrier
    const Phase ph[5] = {{16384 / 256, 128}, {18432 / 256, 32}, {8192 / 256, 64}, {49152 / 256, 256}, {24576 / 256, 192}};   // operand KiB as the launcher's tiles read them: (2,2) 32 rows | a slot's q / k / v | (1,2) 16 rows | (4,3) 64 rows | (1,2) 16 rows x 6144
    CK(hipMemcpyToSymbol(HIP_SYMBOL(c_ph), ph, sizeof(ph)));
    double mb_layer = 0; for (auto& p : ph) mb_layer += p.w_kib * 256.0 / 1024.0;
    const int layers = 28;
    printf("one block = 5 phases, %.1f MB of cold weights (+ K / V), %d blocks per launch, 256 workgroups x 512 threads\n", mb_layer, layers);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int variant = 0; variant < 4; ++variant) {
        const int with_barrier = variant != 0, prefetch = variant >= 2, infl16 = variant == 3;
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipMemsetAsync(count, 0, 256, s));
            CK(hipEventRecord(a, s));
            if (infl16) hipLaunchKernelGGL(k_persist<16>, dim3(256), dim3(512), 0, s, w, WB / 16, act, AB / 16, outb, 5, layers, count, failed, with_barrier, prefetch, sink);
            else hipLaunchKernelGGL(k_persist<8>, dim3(256), dim3(512), 0, s, w, WB / 16, act, AB / 16, outb, 5, layers, count, failed, with_barrier, prefetch, sink);
            CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b));
            float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
            unsigned f = 0; CK(hipMemcpy(&f, failed, 4, hipMemcpyDeviceToHost));
            if (f) { printf("variant %d: TIMED OUT\n", variant); return 1; }
            if (rep > 0 && ms < best) best = ms;
        }
        printf("%-86s %7.2f us per block (%5.2f us per phase), %5.2f TB/s of weights\n",
               variant == 0 ? "no barriers (the stream alone: what HBM allows)" : variant == 1 ? "chip-wide barrier per phase, next ring issued AFTER the barrier" :
               variant == 2 ? "chip-wide barrier per phase, next ring (8 KiB per wave) issued BEFORE the barrier" : "chip-wide barrier per phase, next ring (16 KiB per wave) issued BEFORE the barrier",
               best * 1e3 / layers, best * 1e3 / layers / 5, mb_layer * 1e6 / (best * 1e-3 / layers) / 1e12);
    }
    printf("today: 53.7 us per block as five launches (profiles/r03/frame_step_timeline.txt)\n");
    return 0;
}
