// Does kernel-argument PRELOAD (gfx940+: the dispatcher puts the first kernel arguments into SGPRs at wave launch;
// -mllvm -amdgpu-kernarg-preload-count=16) shorten a dependent launch chain? Every launch of the frame step starts with s_load of its
// arguments (a struct passed by value) and can compute no address before they arrive. A replayed hipGraph of 400 dependent launches of
// 256 x 512 threads, each reading one value per thread through pointers that come (a) out of a by-value struct, (b) as leading scalar
// arguments. Build twice:  hipcc --offload-arch=gfx950 -O3 [-mllvm -amdgpu-kernarg-preload-count=16] -o tools/kernarg_bench[_pre] tools/kernarg_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
struct Args { const float* in; float* out; int n; int pad[40]; };
__global__ __launch_bounds__(512) void k_struct(Args a) { const int i = blockIdx.x * 512 + threadIdx.x; if (i < a.n) a.out[i] = a.in[i] + 1.0f; }
__global__ __launch_bounds__(512) void k_scalar(const float* in, float* out, int n, Args rest) { const int i = blockIdx.x * 512 + threadIdx.x; if (i < n) out[i] = in[i] + 1.0f + (float)rest.pad[39] * 0.0f; }
int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    float *a, *b; const int n = 256 * 512;
    CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMemset(a, 0, n * 4)); CK(hipMemset(b, 0, n * 4));
    const int NL = 400;
    for (int mode = 0; mode < 2; ++mode) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < NL; ++i) {
            Args ar{}; ar.in = (i & 1) ? b : a; ar.out = (i & 1) ? a : b; ar.n = n;
            if (mode == 0) hipLaunchKernelGGL(k_struct, dim3(256), dim3(512), 0, s, ar);
            else hipLaunchKernelGGL(k_scalar, dim3(256), dim3(512), 0, s, ar.in, ar.out, n, ar);
        }
        CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
        }
        printf("%s arguments: %.3f us per dependent launch (graph replay, 256 x 512 threads, one load + one store per thread)\n", mode == 0 ? "by-value struct" : "leading scalar  ", best * 1e3f / NL);
    }
    return 0;
}
