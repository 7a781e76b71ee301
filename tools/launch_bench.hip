// Launch-floor probe: dependent kernel chains on one stream (eager and hipGraph) at several grid shapes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_empty(float* p) { if (p == nullptr && threadIdx.x == 12345) p[0] = 0; }
__global__ __launch_bounds__(512) void k_touch(float* p, int n) {  // every WG reads 1 float4 per thread (produced by the previous kernel) and writes it back
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    float4 v = ((float4*)p)[i % (n / 4)];
    v.x += 1.0f;
    ((float4*)p)[i % (n / 4)] = v;
}
__global__ __launch_bounds__(512) void k_lds(float* p, int n) {  // same plus a 64 KB LDS allocation and two barriers
    extern __shared__ float sm[];
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    sm[threadIdx.x] = p[i % n];
    __syncthreads();
    float v = sm[(threadIdx.x + 64) & 511];
    __syncthreads();
    p[i % n] = v + 1.0f;
}

template <class F> static int run(const char* name, hipStream_t s, int iters, F launch) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 20; ++i) launch();
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(a, s));
    for (int i = 0; i < iters; ++i) launch();
    CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b));
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    printf("%-44s eager %7.2f us/kernel", name, ms * 1e3 / iters);
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < iters; ++i) launch();
    CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    CK(hipEventRecord(a, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b));
    hipEventElapsedTime(&ms, a, b);
    printf("   graph %7.2f us/kernel\n", ms * 1e3 / iters);
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
    return 0;
}

int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    float* p; const int n = 1 << 22; CK(hipMalloc(&p, n * 4)); CK(hipMemset(p, 0, n * 4));
    const int iters = 500;
    run("empty <<<1,64>>>", s, iters, [&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, p); });
    run("empty <<<256,512>>>", s, iters, [&] { hipLaunchKernelGGL(k_empty, dim3(256), dim3(512), 0, s, p); });
    run("empty <<<512,512>>>", s, iters, [&] { hipLaunchKernelGGL(k_empty, dim3(512), dim3(512), 0, s, p); });
    run("touch <<<8,512>>>", s, iters, [&] { hipLaunchKernelGGL(k_touch, dim3(8), dim3(512), 0, s, p, n); });
    run("touch <<<256,512>>>", s, iters, [&] { hipLaunchKernelGGL(k_touch, dim3(256), dim3(512), 0, s, p, n); });
    run("touch <<<512,512>>>", s, iters, [&] { hipLaunchKernelGGL(k_touch, dim3(512), dim3(512), 0, s, p, n); });
    run("touch <<<2048,512>>>", s, iters, [&] { hipLaunchKernelGGL(k_touch, dim3(2048), dim3(512), 0, s, p, n); });
    run("lds64K <<<256,512>>>", s, iters, [&] { hipLaunchKernelGGL(k_lds, dim3(256), dim3(512), 65536, s, p, n); });
    run("lds64K <<<512,512>>>", s, iters, [&] { hipLaunchKernelGGL(k_lds, dim3(512), dim3(512), 65536, s, p, n); });
    run("touch <<<256,1024>>>", s, iters, [&] { hipLaunchKernelGGL(k_touch, dim3(256), dim3(1024), 0, s, p, n); });
    return 0;
}
