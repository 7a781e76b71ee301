// Per-CU ingest probe: how many bytes per second ONE workgroup per CU (512 threads = 8 waves, the k_bgemm geometry) takes in when it
// streams a private slab of once-read "weights" (HBM / Infinity Cache) plus a shared slab of "rows" (L2), through
//   (a) global_load_dwordx4 into registers, 16 loads per lane in flight (what k_bgemm did in round 2 so far), and
//   (b) global_load_lds_dwordx4 (LDS-DMA) into a wave-private LDS ring, then ds_read_b128 (what MI355X_MICROARCH.md's ring numbers use).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/ingest_bench tools/ingest_bench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// per workgroup: wkb KiB of private weights (contiguous), xkb KiB of shared rows; wave w takes every 8th KiB
template <int INFLIGHT>
__global__ __launch_bounds__(512) void k_vgpr(const u32x4* __restrict__ w, const u32x4* __restrict__ x, int wkb, int xkb, uint32_t* sink) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const u32x4* wp = w + (size_t)blockIdx.x * wkb * 64 + lane;
    const u32x4* xp = x + lane;
    u32x4 acc = (u32x4){0, 0, 0, 0};
    const int total = wkb + xkb;  // KiB units, interleaved: weights first then rows
    for (int k0 = wave * INFLIGHT; k0 < total; k0 += 8 * INFLIGHT) {
        u32x4 v[INFLIGHT];
#pragma unroll
        for (int j = 0; j < INFLIGHT; ++j) {
            const int k = min(k0 + j, total - 1);
            v[j] = k < wkb ? wp[(size_t)k * 64] : xp[(size_t)(k - wkb) * 64];
        }
#pragma unroll
        for (int j = 0; j < INFLIGHT; ++j) acc ^= v[j];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

template <int INFLIGHT>
__global__ __launch_bounds__(512) void k_glds(const u32x4* __restrict__ w, const u32x4* __restrict__ x, int wkb, int xkb, uint32_t* sink) {
    extern __shared__ __attribute__((aligned(16))) u32x4 ring[];  // [8 waves][2 halves][INFLIGHT][64 lanes]
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const u32x4* wp = w + (size_t)blockIdx.x * wkb * 64 + lane;
    const u32x4* xp = x + lane;
    u32x4 acc = (u32x4){0, 0, 0, 0};
    const int total = wkb + xkb;
    u32x4* mine = ring + (size_t)wave * 2 * INFLIGHT * 64;
    auto issue = [&](int k0, int half) {
#pragma unroll
        for (int j = 0; j < INFLIGHT; ++j) {
            const int k = min(k0 + j, total - 1);
            const u32x4* src = k < wkb ? wp + (size_t)k * 64 : xp + (size_t)(k - wkb) * 64;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(mine + (size_t)(half * INFLIGHT + j) * 64), 16, 0, 0);
        }
    };
    int half = 0;
    issue(wave * INFLIGHT, 0);
    for (int k0 = wave * INFLIGHT; k0 < total; k0 += 8 * INFLIGHT) {
        const int kn = k0 + 8 * INFLIGHT;
        if (kn < total) issue(kn, half ^ 1);
        if (kn < total) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(INFLIGHT) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < INFLIGHT; ++j) acc ^= mine[(size_t)(half * INFLIGHT + j) * 64 + lane];
        half ^= 1;
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

template <class F> static float timeit(hipStream_t s, int iters, F launch) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    launch(0); hipStreamSynchronize(s);
    hipEventRecord(a, s);
    for (int i = 0; i < iters; ++i) launch(i + 1);
    hipEventRecord(b, s); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f / iters;
}

int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const size_t WBYTES = (size_t)6 << 30;  // weights region: rotate through 6 GiB so every launch is cold
    u32x4 *w, *x; uint32_t* sink;
    CK(hipMalloc(&w, WBYTES)); CK(hipMalloc(&x, 1 << 20)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(w, 1, WBYTES)); CK(hipMemset(x, 2, 1 << 20)); CK(hipMemset(sink, 0, 64));
    const int NWG = 256, iters = 40;
    hipFuncSetAttribute((const void*)k_glds<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 2 * 8 * 1024);
    hipFuncSetAttribute((const void*)k_glds<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 2 * 4 * 1024);
    printf("256 workgroups x 512 threads; us per launch (minus ~1.6 us launch) -> GB/s per CU over weights + rows\n");
    for (int cfg = 0; cfg < 5; ++cfg) {
        const int wkb = cfg == 0 ? 64 : cfg == 1 ? 196 : cfg == 2 ? 196 : cfg == 3 ? 1024 : 0;
        const int xkb = cfg == 0 ? 64 : cfg == 1 ? 0 : cfg == 2 ? 256 : cfg == 3 ? 0 : 256;
        const size_t per_launch = (size_t)NWG * wkb * 1024;
        const size_t nslots = per_launch ? WBYTES / per_launch : 1;
        auto off = [&](int i) { return w + (per_launch ? ((size_t)i % nslots) * (per_launch / 16) : 0); };
        const float a16 = timeit(s, iters, [&](int i) { hipLaunchKernelGGL(k_vgpr<16>, dim3(NWG), dim3(512), 0, s, off(i), x, wkb, xkb, sink); });
        const float a8 = timeit(s, iters, [&](int i) { hipLaunchKernelGGL(k_vgpr<8>, dim3(NWG), dim3(512), 0, s, off(i), x, wkb, xkb, sink); });
        const float b8 = timeit(s, iters, [&](int i) { hipLaunchKernelGGL(k_glds<8>, dim3(NWG), dim3(512), 8 * 2 * 8 * 1024, s, off(i), x, wkb, xkb, sink); });
        const float b4 = timeit(s, iters, [&](int i) { hipLaunchKernelGGL(k_glds<4>, dim3(NWG), dim3(512), 8 * 2 * 4 * 1024, s, off(i), x, wkb, xkb, sink); });
        const double kb = wkb + xkb;
        auto gbs = [&](float us) { return kb * 1024.0 / ((us - 1.6) * 1e-6) / 1e9; };
        printf("weights %4d KiB (cold) + rows %3d KiB (shared, L2) per WG: vgpr16 %6.2f us (%5.1f GB/s/CU)  vgpr8 %6.2f (%5.1f)  glds8 %6.2f (%5.1f)  glds4 %6.2f (%5.1f)\n",
               wkb, xkb, a16, gbs(a16), a8, gbs(a8), b8, gbs(b8), b4, gbs(b4));
    }
    return 0;
}
