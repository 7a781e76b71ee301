// Probe: does v_mfma_f32_16x16x4_f32 reproduce a k-ordered fmaf chain bit-for-bit?
// Also: are f32 divide / sqrtf / rintf correctly rounded (CPU == GPU bitwise)?
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/probe_mfma.hip -o tools/probe_mfma
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

static inline uint64_t mix64(uint64_t z) {
    z ^= z >> 30; z *= 0xbf58476d1ce4e5b9ULL; z ^= z >> 27; z *= 0x94d049bb133111ebULL; z ^= z >> 31; return z;
}
static float rnd(uint64_t i, float scale) {
    uint64_t h = mix64(i * 0x9E3779B97F4A7C15ULL + 12345);
    int s = (int)(h & 0xffff) + (int)((h >> 16) & 0xffff) + (int)((h >> 32) & 0xffff) + (int)(h >> 48) - 131070;
    return (float)s * scale;
}

// one wave: A [16][K] f32 row-major, B [K][16] f32; canonical: for kb (step 4): MFMA with lane (i=l&15,k=l>>4)
__global__ void k_mfma(const float* A, const float* Bm, float* D, int K) {
    int l = threadIdx.x;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < K; k0 += 4) {
        float a = A[(l & 15) * K + k0 + (l >> 4)];
        float b = Bm[(k0 + (l >> 4)) * 16 + (l & 15)];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    }
    for (int j = 0; j < 4; ++j) D[(4 * (l >> 4) + j) * 16 + (l & 15)] = acc[j];
}

__device__ __host__ inline float spec_expf(float x) {
    if (x < -87.0f) return 0.0f;
    float n = rintf(x * 1.44269504088896341f);
    float r = fmaf(n, -0.693145751953125f, x);
    r = fmaf(n, -1.42860682030941723212e-6f, r);
    float p = 1.0f / 720.0f;
    p = fmaf(p, r, 1.0f / 120.0f);
    p = fmaf(p, r, 1.0f / 24.0f);
    p = fmaf(p, r, 1.0f / 6.0f);
    p = fmaf(p, r, 0.5f);
    p = fmaf(p, r, 1.0f);
    p = fmaf(p, r, 1.0f);
    int e = (int)n + 127;
    union { uint32_t u; float f; } s; s.u = (uint32_t)e << 23;
    return p * s.f;
}

__global__ void k_scalar(const float* x, const float* y, float* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = x[i] / y[i];
    out[n + i] = sqrtf(fabsf(x[i]));
    out[2 * n + i] = spec_expf(x[i]);
    out[3 * n + i] = 1.0f / sqrtf(fabsf(y[i]) + 1e-6f);
    out[4 * n + i] = x[i] / (1.0f + spec_expf(-x[i]));
}

int main() {
    const int K = 2048;
    std::vector<float> A(16 * K), B(K * 16), D(256), R(256);
    for (int i = 0; i < 16 * K; ++i) A[i] = rnd(i, 3e-5f);
    for (int i = 0; i < K * 16; ++i) {
        float v = rnd(1000000 + i, 6e-7f);
        uint32_t u; memcpy(&u, &v, 4); u &= 0xffff0000u; memcpy(&v, &u, 4);  // bf16-representable
        B[i] = v;
    }
    float *dA, *dB, *dD;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dD, 256 * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    k_mfma<<<1, 64>>>(dA, dB, dD, K);
    hipMemcpy(D.data(), dD, 256 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            float acc = 0.f;
            for (int k = 0; k < K; ++k) acc = fmaf(A[i * K + k], B[k * 16 + j], acc);
            R[i * 16 + j] = acc;
            if (memcmp(&acc, &D[i * 16 + j], 4)) { if (bad < 5) printf("mismatch [%d][%d] cpu=%.9g gpu=%.9g\n", i, j, acc, D[i * 16 + j]); ++bad; }
        }
    printf("MFMA_F32_CHAIN mismatches=%d/256 (K=%d)\n", bad, K);

    const int n = 1 << 20;
    std::vector<float> x(n), y(n), out(5 * n);
    for (int i = 0; i < n; ++i) { x[i] = rnd(5000000 + i, 2.5e-4f); y[i] = rnd(9000000 + i, 1e-4f); if (y[i] == 0) y[i] = 1.f; }
    float *dx, *dy, *dout;
    hipMalloc(&dx, n * 4); hipMalloc(&dy, n * 4); hipMalloc(&dout, 5 * n * 4);
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(dy, y.data(), n * 4, hipMemcpyHostToDevice);
    k_scalar<<<n / 256, 256>>>(dx, dy, dout, n);
    hipMemcpy(out.data(), dout, 5 * n * 4, hipMemcpyDeviceToHost);
    int b[5] = {0, 0, 0, 0, 0};
    for (int i = 0; i < n; ++i) {
        float r0 = x[i] / y[i], r1 = sqrtf(fabsf(x[i])), r2 = spec_expf(x[i]), r3 = 1.0f / sqrtf(fabsf(y[i]) + 1e-6f);
        float r4 = x[i] / (1.0f + spec_expf(-x[i]));
        b[0] += memcmp(&r0, &out[i], 4) != 0;
        b[1] += memcmp(&r1, &out[n + i], 4) != 0;
        b[2] += memcmp(&r2, &out[2 * n + i], 4) != 0;
        b[3] += memcmp(&r3, &out[3 * n + i], 4) != 0;
        b[4] += memcmp(&r4, &out[4 * n + i], 4) != 0;
    }
    printf("SCALAR mismatches div=%d sqrt=%d exp=%d rsqrt=%d silu=%d of %d\n", b[0], b[1], b[2], b[3], b[4], n);
    double maxrel = 0;
    for (int i = 0; i < n; ++i) { double e = exp((double)x[i]); double rel = fabs(spec_expf(x[i]) - e) / e; if (rel > maxrel) maxrel = rel; }
    printf("spec_expf max rel err vs exp(double) = %.3g\n", maxrel);
    return (bad || b[0] || b[1] || b[2] || b[3] || b[4]) ? 1 : 0;
}
