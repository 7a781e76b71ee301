// Probe of v_mfma_f32_16x16x32_bf16's accumulation arithmetic (what a CPU oracle would have to restate to make bf16-MFMA
// GEMMs bit-exact): n independent cases, one wave each; case i: A [16][32] bf16, B [32][16] bf16, C [16][16] f32 -> D.
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/probe_bf16_mfma.hip -o tools/libprobe_bf16_mfma.so
#include <hip/hip_runtime.h>
#include <cstdint>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__global__ void k_case(const uint16_t* A, const uint16_t* B, const float* C, float* D, int chain) {
    const int l = threadIdx.x, cs = blockIdx.x;
    const uint16_t* a = A + (size_t)cs * chain * 16 * 32; const uint16_t* b = B + (size_t)cs * chain * 32 * 16;
    f32x4 acc;
    for (int j = 0; j < 4; ++j) acc[j] = C[(size_t)cs * 256 + (4 * (l >> 4) + j) * 16 + (l & 15)];
    for (int s = 0; s < chain; ++s) {  // `chain` MFMAs accumulate into the same registers (K = 32 * chain)
        union { bf16x8 v; uint16_t u[8]; } af, bf;
        for (int j = 0; j < 8; ++j) {
            af.u[j] = a[(size_t)s * 512 + (l & 15) * 32 + 8 * (l >> 4) + j];
            bf.u[j] = b[(size_t)s * 512 + (8 * (l >> 4) + j) * 16 + (l & 15)];
        }
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af.v, bf.v, acc, 0, 0, 0);
    }
    for (int j = 0; j < 4; ++j) D[(size_t)cs * 256 + (4 * (l >> 4) + j) * 16 + (l & 15)] = acc[j];
}

extern "C" int probe_bf16_mfma(const uint16_t* A, const uint16_t* B, const float* C, float* D, int n, int chain) {
    uint16_t *dA, *dB; float *dC, *dD;
    const size_t na = (size_t)n * chain * 512 * 2, nc = (size_t)n * 256 * 4;
    if (hipMalloc((void**)&dA, na) || hipMalloc((void**)&dB, na) || hipMalloc((void**)&dC, nc) || hipMalloc((void**)&dD, nc)) return -1;
    hipMemcpy(dA, A, na, hipMemcpyHostToDevice); hipMemcpy(dB, B, na, hipMemcpyHostToDevice); hipMemcpy(dC, C, nc, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_case, dim3(n), dim3(64), 0, 0, dA, dB, dC, dD, chain);
    const int rc = hipDeviceSynchronize() == hipSuccess ? 0 : -2;
    hipMemcpy(D, dD, nc, hipMemcpyDeviceToHost);
    hipFree(dA); hipFree(dB); hipFree(dC); hipFree(dD);
    return rc;
}
