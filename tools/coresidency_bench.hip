// Do short dependent launches get through while a bulk kernel owns the CUs? (the decoder's frame step beside the vocoder)
//   hipcc --offload-arch=gfx950 -O3 -o tools/coresidency_bench tools/coresidency_bench.hip && tools/coresidency_bench
// Stream 1 runs a "hog" shaped like the vocoder's ring GEMM: thousands of workgroups of 256 threads, 176 VGPRs, 66 KiB of LDS (two per CU:
// 352 of 512 VGPRs per SIMD and 132 of 160 KiB of LDS taken), each alive for ~30 us. Stream 2 replays a chain of 200 dependent launches of
// 256 workgroups in three shapes:
//   big    8 waves x 128 VGPRs, 48 KiB LDS   (the decoder's k_bgemm<2,3,4>: does not fit beside two hog workgroups)
//   mid    8 waves x  80 VGPRs, 16 KiB LDS   (fits: 2 x 80 = 160 VGPRs per SIMD, 28 KiB of LDS are free)
//   small  4 waves x 160 VGPRs, 16 KiB LDS   (fits: 1 x 160)
// and prints the chain's time per launch alone and with the hog running. If "mid" / "small" keep their solo pace beside the hog while "big"
// slows to the hog's workgroup lifetime, a decode kernel built to the leftover resources would take the vocoder off the frame step's clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ void spin_us(float us) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz
    const unsigned long long dt = (unsigned long long)(us * 100.0f);
    while (__builtin_amdgcn_s_memrealtime() - t0 < dt) __builtin_amdgcn_s_sleep(8);
}

__global__ __launch_bounds__(256) void k_hog(float* out, float us) {
    extern __shared__ float sm[];
    asm volatile("v_mov_b32 v175, 0" ::: "v175");  // 176 VGPRs allocated
    sm[threadIdx.x] = (float)threadIdx.x;
    __syncthreads();
    spin_us(us);
    if (threadIdx.x == 0 && sm[1] < 0.0f) out[blockIdx.x] = sm[0];
}
template <int NV>
__device__ __forceinline__ void touch_vgpr();
template <> __device__ __forceinline__ void touch_vgpr<128>() { asm volatile("v_mov_b32 v127, 0" ::: "v127"); }
template <> __device__ __forceinline__ void touch_vgpr<80>() { asm volatile("v_mov_b32 v79, 0" ::: "v79"); }
template <> __device__ __forceinline__ void touch_vgpr<160>() { asm volatile("v_mov_b32 v159, 0" ::: "v159"); }
template <> __device__ __forceinline__ void touch_vgpr<152>() { asm volatile("v_mov_b32 v151, 0" ::: "v151"); }
template <int NTHR, int NV>
__global__ __launch_bounds__(NTHR) void k_link(const float* in, float* out, float us) {
    extern __shared__ float sm[];
    touch_vgpr<NV>();
    sm[threadIdx.x] = in[(blockIdx.x * NTHR + threadIdx.x) & 65535];
    __syncthreads();
    spin_us(us);
    out[(blockIdx.x * NTHR + threadIdx.x) & 65535] = sm[threadIdx.x ^ 1] + 1.0f;
}

template <int NTHR, int NV>
static double chain(hipStream_t s, float* a, float* b, int lds, int n) {
    CK(hipFuncSetAttribute((const void*)k_link<NTHR, NV>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k_link<NTHR, NV>), dim3(256), dim3(NTHR), lds, s, (i & 1) ? b : a, (i & 1) ? a : b, 2.0f);
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL((k_link<NTHR, NV>), dim3(256), dim3(NTHR), lds, s, (i & 1) ? b : a, (i & 1) ? a : b, 2.0f);
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3 / n;
}

int main() {
    float *a, *b, *h;
    CK(hipMalloc(&a, 65536 * 4)); CK(hipMalloc(&b, 65536 * 4)); CK(hipMalloc(&h, 1 << 20));
    CK(hipMemset(a, 0, 65536 * 4)); CK(hipMemset(b, 0, 65536 * 4));
    hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    const int hog_lds = 66 * 1024;
    CK(hipFuncSetAttribute((const void*)k_hog, hipFuncAttributeMaxDynamicSharedMemorySize, hog_lds));
    const int n = 200;
    for (int with_hog = 0; with_hog < 2; ++with_hog) {
        for (int shape = 0; shape < 3; ++shape) {
            for (float hog_us : {30.0f, 80.0f}) {
                if (!with_hog && hog_us > 30.0f) continue;
                if (with_hog)  // ~25 ms of hog: 256 CUs x 2 workgroups per round
                    hipLaunchKernelGGL(k_hog, dim3((unsigned)(512 * (25000.0f / hog_us))), dim3(256), hog_lds, s1, h, hog_us);
                double us = shape == 0 ? chain<512, 128>(s2, a, b, 48 * 1024, n) : shape == 1 ? chain<512, 80>(s2, a, b, 16 * 1024, n) : chain<256, 160>(s2, a, b, 16 * 1024, n);
                CK(hipStreamSynchronize(s1));
                printf("%-5s chain (256 workgroups, 2 us of work per launch) %s: %7.2f us per launch\n", shape == 0 ? "big" : shape == 1 ? "mid" : "small",
                       with_hog ? (hog_us > 30.0f ? "beside a hog of 80-us workgroups" : "beside a hog of 30-us workgroups") : "alone", us);
            }
        }
    }
    // the other way round: the hog leaves room — ONE workgroup per CU (81 KiB of LDS: a second does not fit), 4 waves x 176 VGPRs; the chain has
    // the decoder's largest footprint (8 waves x 152 VGPRs, 48 KiB)
    CK(hipFuncSetAttribute((const void*)k_hog, hipFuncAttributeMaxDynamicSharedMemorySize, 81 * 1024));
    for (float hog_us : {30.0f, 80.0f}) {
        hipLaunchKernelGGL(k_hog, dim3((unsigned)(256 * (25000.0f / hog_us))), dim3(256), 81 * 1024, s1, h, hog_us);
        const double us = chain<512, 152>(s2, a, b, 48 * 1024, n);
        CK(hipStreamSynchronize(s1));
        printf("decoder-sized chain (8 waves x 152 VGPRs, 48 KiB) beside ONE %d-us hog workgroup per CU (81 KiB, 4 x 176 VGPRs): %7.2f us per launch\n", (int)hog_us, us);
    }
    return 0;
}
