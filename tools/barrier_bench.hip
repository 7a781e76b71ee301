// Grid-barrier probe: what one in-kernel hand-off between two dependent phases costs on MI355X when 256 workgroups x 512 threads
// stay resident (the alternative to one launch per phase: 1.6-2.0 us inside a replayed hipGraph, tools/launch_bench.hip).
// Each phase: every workgroup writes a 4 KiB record (so there is real data to make visible across the 8 non-coherent XCD L2s),
// arrives at the barrier (agent-scope release), waits for all (agent-scope acquire), then reads the record of a workgroup that
// sits on ANOTHER XCD (block id + 1) and checks it. Every wait is bounded: a barrier that does not complete in ~50 ms sets a
// flag and the kernel runs to its end without further waiting.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/barrier_bench tools/barrier_bench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Bar { unsigned* count; unsigned* failed; };

// mode 0: every wave fences at agent scope (L2 write-back + invalidate per wave)
// mode 1: only thread 0's arrive / wait carry agent-scope release / acquire; the other waves order through the workgroup barrier
// mode 2, 3: no agent-scope fence at all: the handed-off DATA is written and read past the non-coherent L2 (sc1 accesses)
__device__ __forceinline__ void grid_barrier(const Bar& b, unsigned target, int mode) {
    const int sleep = 1;
    if (mode == 0) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");  // this wave's stores written back beyond its XCD's L2
    else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (threadIdx.x == 0) {
        if (mode >= 2) __hip_atomic_fetch_add(b.count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else __hip_atomic_fetch_add(b.count, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while ((mode >= 2 ? __hip_atomic_load(b.count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                          : __hip_atomic_load(b.count, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)) < target) {
            if (sleep) __builtin_amdgcn_s_sleep(2);
            if (++spins > (1u << 22) || __hip_atomic_load(b.failed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                __hip_atomic_store(b.failed, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
    }
    __syncthreads();
    if (mode == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // stale lines of other XCDs' data dropped from L1 / L2
    else __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

__device__ __forceinline__ void st_sc1(uint32_t* p, uint32_t v, int mode) {
    if (mode == 2) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (mode == 3) asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
    else *p = v;
}
__device__ __forceinline__ uint32_t ld_sc1(const uint32_t* p, int mode) {
    if (mode == 2) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (mode == 3) { uint32_t v; asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); return v; }
    return *p;
}

__global__ __launch_bounds__(512) void k_phases(Bar b, uint32_t* rec, int phases, int sleep, unsigned base, unsigned* bad) {
    const int nb = gridDim.x, me = blockIdx.x, other = (me + 1) % nb;
    unsigned wrong = 0;
    for (int p = 0; p < phases; ++p) {
        // 4 KiB record per workgroup per phase parity (two buffers so a fast workgroup cannot overwrite what a slow one still reads)
        uint32_t* mine = rec + ((size_t)(p & 1) * nb + me) * 1024;
        st_sc1(mine + threadIdx.x, base + (unsigned)p * 1000003u + me * 513u + threadIdx.x, sleep);
        st_sc1(mine + 512 + threadIdx.x, ~(base + (unsigned)p), sleep);
        if (sleep == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        grid_barrier(b, base + (unsigned)(p + 1) * nb, sleep);
        const uint32_t* theirs = rec + ((size_t)(p & 1) * nb + other) * 1024;
        const uint32_t got = ld_sc1(theirs + threadIdx.x, sleep), got2 = ld_sc1(theirs + 512 + threadIdx.x, sleep);
        wrong += got != base + (unsigned)p * 1000003u + other * 513u + threadIdx.x;
        wrong += got2 != ~(base + (unsigned)p);
    }
    if (wrong) atomicAdd(bad, wrong);
}

// the same data movement as one kernel per phase (what the frame step does today), for the comparison inside a graph
__global__ __launch_bounds__(512) void k_one_phase(uint32_t* rec, int p, unsigned base, unsigned* bad) {
    const int nb = gridDim.x, me = blockIdx.x, other = (me + 1) % nb;
    if (p > 0) {
        const uint32_t* theirs = rec + ((size_t)((p - 1) & 1) * nb + other) * 1024;
        const uint32_t got = theirs[threadIdx.x];
        if (got != base + (unsigned)(p - 1) * 1000003u + other * 513u + threadIdx.x) atomicAdd(bad, 1u);
    }
    uint32_t* mine = rec + ((size_t)(p & 1) * nb + me) * 1024;
    mine[threadIdx.x] = base + (unsigned)p * 1000003u + me * 513u + threadIdx.x;
    mine[512 + threadIdx.x] = ~(base + (unsigned)p);
}

int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    unsigned *count, *failed, *bad; uint32_t* rec;
    const int NB = 256, PH = 200;
    CK(hipMalloc(&count, 256)); CK(hipMalloc(&failed, 256)); CK(hipMalloc(&bad, 256)); CK(hipMalloc(&rec, (size_t)2 * NB * 4096));
    CK(hipMemset(count, 0, 256)); CK(hipMemset(failed, 0, 256)); CK(hipMemset(bad, 0, 256));
    hipEvent_t a, c; hipEventCreate(&a); hipEventCreate(&c);
    unsigned base = 0;
    for (int sleep = 0; sleep < 4; ++sleep)
        for (int rep = 0; rep < 3; ++rep) {
            Bar b{count, failed};
            CK(hipEventRecord(a, s));
            hipLaunchKernelGGL(k_phases, dim3(NB), dim3(512), 0, s, b, rec, PH, sleep, base, bad);
            CK(hipEventRecord(c, s)); CK(hipEventSynchronize(c));
            float ms = 0; hipEventElapsedTime(&ms, a, c);
            unsigned hf = 0, hb = 0; CK(hipMemcpy(&hf, failed, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
            printf("persistent: %d workgroups x 512, %d phases, mode=%d: %.2f us per phase (write 4 KiB + barrier + read 4 KiB from another XCD); timed-out=%u wrong=%u\n",
                   NB, PH, sleep, ms * 1e3f / PH, hf, hb);
            base += (unsigned)PH * NB;  // the counter keeps counting: no reset between launches
            if (hf) return 1;
            CK(hipMemset(bad, 0, 4));
        }
    {  // one launch per phase, replayed from a graph
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int p = 0; p < PH; ++p) hipLaunchKernelGGL(k_one_phase, dim3(NB), dim3(512), 0, s, rec, p, 7u, bad);
        CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(a, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(c, s)); CK(hipEventSynchronize(c));
            float ms = 0; hipEventElapsedTime(&ms, a, c);
            unsigned hb = 0; CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
            printf("graph     : one launch per phase, %d phases: %.2f us per phase; wrong=%u\n", PH, ms * 1e3f / PH, hb);
        }
    }
    return 0;
}
