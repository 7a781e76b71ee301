// Tile sweep of the decoder's GEMM (csrc/q3_bgemm.hip) at the engine's shapes: every (RT, NT) instance the launcher could pick, with
// the weights cold (rotating through copies larger than L2 + Infinity Cache: the Talker) or hot (one copy: the Predictor, whose
// weights are re-read 15 times per frame). Prints us per launch inside a replayed hipGraph of 50 dependent launches.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I qwen3-tts-rust_amd/csrc -o tools/bgemm_tune tools/bgemm_tune.hip -L qwen3-tts-rust_amd/csrc -lq3tts -Wl,-rpath,'$ORIGIN/../qwen3-tts-rust_amd/csrc'
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "q3_kernels.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// touches a weight matrix once (16 bytes per lane, results folded into a value that is never stored): after it the matrix sits in
// the memory-side Infinity Cache (and partly in the L2s) — the experiment at the end asks what a GEMM gains from that
__global__ __launch_bounds__(256) void k_touch(const uint4* w, size_t n16, unsigned* sink) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) { const uint4 v = w[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x9e3779b9u) sink[0] = acc;
}

struct Shape { const char* name; int M, K, N, epi, scaled, cold, q8 = 0; };  // q8: ggml Q8_0 blocks kept in block form (1.0625 bytes per weight)

int main(int argc, char** argv) {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const Shape shapes[] = {
        {"T qkv", 64, 2048, 4096, Q3_EPI_STORE, 1, 1}, {"T o", 64, 2048, 2048, Q3_EPI_RESID, 0, 1}, {"T gate/up", 64, 2048, 12288, Q3_EPI_SWIGLU, 1, 1},
        {"T down", 64, 6144, 2048, Q3_EPI_RESID, 0, 1}, {"T head", 64, 2048, 3072, Q3_EPI_STORE, 1, 1},
        {"T qkv", 48, 2048, 4096, Q3_EPI_STORE, 1, 1}, {"T o", 48, 2048, 2048, Q3_EPI_RESID, 0, 1}, {"T gate/up", 48, 2048, 12288, Q3_EPI_SWIGLU, 1, 1},
        {"T down", 48, 6144, 2048, Q3_EPI_RESID, 0, 1},
        {"T qkv", 32, 2048, 4096, Q3_EPI_STORE, 1, 1}, {"T o", 32, 2048, 2048, Q3_EPI_RESID, 0, 1}, {"T gate/up", 32, 2048, 12288, Q3_EPI_SWIGLU, 1, 1},
        {"T down", 32, 6144, 2048, Q3_EPI_RESID, 0, 1},
        {"T gate/up", 1, 2048, 12288, Q3_EPI_SWIGLU, 1, 1}, {"T down", 1, 6144, 2048, Q3_EPI_RESID, 0, 1},
        {"P qkv", 64, 1024, 4096, Q3_EPI_STORE, 1, 0}, {"P o", 64, 2048, 1024, Q3_EPI_RESID, 0, 0}, {"P gate/up", 64, 1024, 6144, Q3_EPI_SWIGLU, 1, 0},
        {"P down", 64, 3072, 1024, Q3_EPI_RESID, 0, 0}, {"P head", 64, 1024, 2048, Q3_EPI_ARGMAX, 1, 0},
        {"P qkv", 48, 1024, 4096, Q3_EPI_STORE, 1, 0}, {"P o", 48, 2048, 1024, Q3_EPI_RESID, 0, 0}, {"P gate/up", 48, 1024, 6144, Q3_EPI_SWIGLU, 1, 0},
        {"P down", 48, 3072, 1024, Q3_EPI_RESID, 0, 0},
        {"P qkv", 128, 1024, 4096, Q3_EPI_STORE, 1, 0}, {"P gate/up", 128, 1024, 6144, Q3_EPI_SWIGLU, 1, 0}, {"P down", 128, 3072, 1024, Q3_EPI_RESID, 0, 0},
        {"T8 qkv", 64, 2048, 4096, Q3_EPI_STORE, 1, 1, 1}, {"T8 o", 64, 2048, 2048, Q3_EPI_RESID, 0, 1, 1}, {"T8 gate/up", 64, 2048, 12288, Q3_EPI_SWIGLU, 1, 1, 1},
        {"T8 down", 64, 6144, 2048, Q3_EPI_RESID, 0, 1, 1}, {"T8 head", 64, 2048, 3072, Q3_EPI_STORE, 1, 1, 1}, {"T8 gate/up", 1, 2048, 12288, Q3_EPI_SWIGLU, 1, 1, 1}, {"T8 down", 1, 6144, 2048, Q3_EPI_RESID, 0, 1, 1},
    };
    uint16_t* wsc; CK(hipMalloc(&wsc, (size_t)16384 * 256 * 2)); CK(hipMemset(wsc, 0x2c, (size_t)16384 * 256 * 2));
    const size_t WBYTES = (size_t)2 << 30;
    uint4* w; CK(hipMalloc(&w, WBYTES)); CK(hipMemset(w, 0x3c, WBYTES));
    uint16_t *a, *yb; float *y, *ssp, *sso, *nw; unsigned long long* keys;
    CK(hipMalloc(&a, 256 * 8192 * 2)); CK(hipMemset(a, 0x3c, 256 * 8192 * 2));
    CK(hipMalloc(&yb, 256 * 16384 * 2)); CK(hipMalloc(&y, 256 * 16384 * 4)); CK(hipMemset(y, 0, 256 * 16384 * 4));
    CK(hipMalloc(&ssp, 256 * 512 * 4)); CK(hipMemset(ssp, 0x3c, 256 * 512 * 4)); CK(hipMalloc(&sso, 256 * 1024 * 4));
    CK(hipMalloc(&nw, 16384 * 4)); CK(hipMemset(nw, 0x3c, 16384 * 4)); CK(hipMalloc(&keys, 256 * 1024 * 8));
    q3_bgemm_prepare();
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (const Shape& sh : shapes) {
        const size_t wb = (size_t)sh.N * sh.K * (sh.q8 ? 1 : 2);
        const int copies = sh.cold ? (int)(WBYTES / wb) : 1;
        printf("%-10s M=%3d K=%4d N=%5d %s:", sh.name, sh.M, sh.K, sh.N, sh.cold ? "cold" : "hot ");
        float best = 1e9f; int brt = 0, bnt = 0; float chosen = 0;
        for (int pass = 0; pass < 2; ++pass)
            for (int rt = 1; rt <= 4; ++rt)
                for (int nt = 1; nt <= 3; ++nt) {
                    if (pass == 0 && (rt != 1 || nt != 1)) continue;  // pass 0: the launcher's own choice
                    if (pass == 1 && (sh.N / 16) % nt) continue;
                    q3_bgemm_force(pass == 0 ? 0 : rt, pass == 0 ? 0 : nt);
                    hipGraph_t g; hipGraphExec_t ge;
                    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
                    const int iters = 50;
                    for (int i = 0; i < iters; ++i) {
                        Q3BGemm q{}; q.a = a; q.B = sh.M; q.w = (const uint4*)((const char*)w + (size_t)(i % copies) * wb); q.K = sh.K; q.N = sh.N; q.w_once = sh.cold;
                        if (sh.q8) q.wscale = wsc;
                        if (sh.scaled) { q.ssp = ssp; q.ld_ssp = sh.K / 16; q.ntiles = sh.K / 16; q.d_norm = sh.K; q.eps = 1e-6f; }
                        q.epi = sh.epi; q.y = y; q.ldy = sh.N; q.yb = yb; q.keys = keys; q.key_stride = sh.N / 16;
                        if (sh.epi == Q3_EPI_RESID) { q.nw_next = nw; q.ssp_out = sso; q.ld_ssp_out = sh.N / 16; }
                        if (q3_launch_bgemm(q, s)) { printf(" launch refused\n"); return 1; }
                    }
                    CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
                    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
                    float tot = 0;
                    for (int rep = 0; rep < 3; ++rep) {
                        CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
                        float ms = 0; hipEventElapsedTime(&ms, e0, e1); tot += ms;
                    }
                    const float us = tot * 1e3f / (3 * iters);
                    if (pass == 0) { chosen = us; printf(" launcher %5.2f |", us); }
                    else { printf(" (%d,%d) %5.2f", rt, nt, us); if (us < best) { best = us; brt = rt; bnt = nt; } }
                    hipGraphExecDestroy(ge); hipGraphDestroy(g);
                }
        printf("  -> best (%d,%d) %.2f us (launcher %+.0f%%)\n", brt, bnt, best, 100.0f * (chosen - best) / best);
    }
    // Does a GEMM run faster when its (otherwise cold) weights were read once just before — i.e. are in the Infinity Cache?
    {
        q3_bgemm_force(0, 0);
        unsigned* sink; CK(hipMalloc(&sink, 64));
        const Shape tsh[2] = {{"T gate/up", 64, 2048, 12288, Q3_EPI_SWIGLU, 1, 1}, {"T down", 64, 6144, 2048, Q3_EPI_RESID, 0, 1}};
        for (const Shape& sh : tsh) {
            const size_t wb = (size_t)sh.N * sh.K * 2;
            const int copies = (int)(WBYTES / wb);
            for (int pre = 0; pre < 2; ++pre) {
                hipGraph_t g; hipGraphExec_t ge;
                CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
                const int iters = 30;
                for (int i = 0; i < iters; ++i) {
                    const uint4* wi = (const uint4*)((const char*)w + (size_t)(i % copies) * wb);
                    if (pre) hipLaunchKernelGGL(k_touch, dim3(1024), dim3(256), 0, s, wi, wb / 16, sink);
                    Q3BGemm q{}; q.a = a; q.B = sh.M; q.w = wi; q.K = sh.K; q.N = sh.N; q.w_once = 1;
                    if (sh.scaled) { q.ssp = ssp; q.ld_ssp = sh.K / 16; q.ntiles = sh.K / 16; q.d_norm = sh.K; q.eps = 1e-6f; }
                    q.epi = sh.epi; q.y = y; q.ldy = sh.N; q.yb = yb;
                    if (sh.epi == Q3_EPI_RESID) { q.nw_next = nw; q.ssp_out = sso; q.ld_ssp_out = sh.N / 16; }
                    if (q3_launch_bgemm(q, s)) return 1;
                }
                CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
                CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
                float tot = 0;
                for (int rep = 0; rep < 3; ++rep) {
                    CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
                    float ms = 0; hipEventElapsedTime(&ms, e0, e1); tot += ms;
                }
                printf("%-10s M=64, cold weights%s: %.2f us per %s\n", sh.name, pre ? ", each touched by a streaming kernel right before" : "",
                       tot * 1e3f / (3 * iters), pre ? "(touch + GEMM) pair" : "GEMM");
                hipGraphExecDestroy(ge); hipGraphDestroy(g);
            }
        }
    }
    // Two half-batches side by side: does a chain of Predictor-layer GEMMs at 32 rows, run twice on two streams at once, finish sooner
    // than one chain at 64 rows? (The chains are latency-bound; the question is whether two can share the CUs without slowing down.)
    {
        q3_bgemm_force(0, 0);
        hipStream_t s2; CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
        hipEvent_t f0, f1; CK(hipEventCreate(&f0)); CK(hipEventCreate(&f1));
        const Shape layer[4] = {{"P qkv", 0, 1024, 4096, Q3_EPI_STORE, 1, 0}, {"P o", 0, 2048, 1024, Q3_EPI_RESID, 0, 0},
                                {"P gate/up", 0, 1024, 6144, Q3_EPI_SWIGLU, 1, 0}, {"P down", 0, 3072, 1024, Q3_EPI_RESID, 0, 0}};
        auto build = [&](hipStream_t st, int M, int half, hipGraphExec_t* ge) -> int {
            hipGraph_t g;
            CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
            for (int l = 0; l < 10; ++l)
                for (const Shape& sh : layer) {
                    Q3BGemm q{}; q.a = a + (size_t)half * 64 * 8192; q.B = M; q.w = (const uint4*)((const char*)w + (size_t)(l * 4 + (&sh - layer)) * (16 << 20)); q.K = sh.K; q.N = sh.N;
                    if (sh.scaled) { q.ssp = ssp + half * 64 * 512; q.ld_ssp = sh.K / 16; q.ntiles = sh.K / 16; q.d_norm = sh.K; q.eps = 1e-6f; }
                    q.epi = sh.epi; q.y = y + (size_t)half * 64 * 16384; q.ldy = sh.N; q.yb = yb + (size_t)half * 64 * 16384;
                    if (sh.epi == Q3_EPI_RESID) { q.nw_next = nw; q.ssp_out = sso + half * 64 * 1024; q.ld_ssp_out = sh.N / 16; }
                    if (q3_launch_bgemm(q, st)) return 1;
                }
            CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(ge, g, nullptr, nullptr, 0));
            return 0;
        };
        for (int M : {64, 32, 16}) {
            hipGraphExec_t ga, gb;
            if (build(s, M, 0, &ga) || build(s2, M, 1, &gb)) { printf("dual: build failed\n"); return 1; }
            CK(hipGraphLaunch(ga, s)); CK(hipGraphLaunch(gb, s2)); CK(hipDeviceSynchronize());
            float one = 0, two = 0;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ga, s)); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
                float ms = 0; hipEventElapsedTime(&ms, e0, e1); one += ms;
                CK(hipEventRecord(e0, s)); CK(hipStreamWaitEvent(s2, e0, 0));
                CK(hipGraphLaunch(ga, s)); CK(hipGraphLaunch(gb, s2));
                CK(hipEventRecord(f1, s2)); CK(hipStreamWaitEvent(s, f1, 0)); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
                hipEventElapsedTime(&ms, e0, e1); two += ms;
            }
            printf("Predictor-layer GEMM chain (10 layers x 4 GEMMs, hot weights), %2d rows: one chain %.1f us; two chains at once on two streams %.1f us (%.2fx one)\n",
                   M, one * 1e3f / 3, two * 1e3f / 3, two / one);
        }
    }
    return 0;
}
