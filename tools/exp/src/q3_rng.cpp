// q3_rng.cpp — host-side `rand 0.8` StdRng stream (ChaCha12, rand_chacha 0.3 block layout) used to draw the
// Talker sampler's uniforms. Reference call sites: src/models/llama/mod.rs:648 (StdRng::seed_from_u64) and
// :757 (`let r: f32 = rng.gen()`). The crate is a Cargo dependency that is not vendored in /root/reference
// (Cargo.toml:18, Cargo.lock git-ignored), so this restates its published algorithm:
//   seed_from_u64: rand_core 0.6 PCG32 expansion of the u64 into the 32-byte key;
//   ChaCha12, 64-bit block counter in words 12-13, stream id 0, four blocks buffered per refill;
//   gen::<f32>() = (next_u32() >> 8) * 2^-24.
#include <stdint.h>
#include <string.h>

static inline uint32_t rotl(uint32_t v, int n) { return (v << n) | (v >> (32 - n)); }
static inline void qr(uint32_t& a, uint32_t& b, uint32_t& c, uint32_t& d) {
    a += b; d ^= a; d = rotl(d, 16);
    c += d; b ^= c; b = rotl(b, 12);
    a += b; d ^= a; d = rotl(d, 8);
    c += d; b ^= c; b = rotl(b, 7);
}
static void chacha_block(const uint32_t in[16], int rounds, uint32_t out[16]) {
    uint32_t x[16];
    memcpy(x, in, 64);
    for (int i = 0; i < rounds; i += 2) {
        qr(x[0], x[4], x[8], x[12]); qr(x[1], x[5], x[9], x[13]); qr(x[2], x[6], x[10], x[14]); qr(x[3], x[7], x[11], x[15]);
        qr(x[0], x[5], x[10], x[15]); qr(x[1], x[6], x[11], x[12]); qr(x[2], x[7], x[8], x[13]); qr(x[3], x[4], x[9], x[14]);
    }
    for (int i = 0; i < 16; ++i) out[i] = x[i] + in[i];
}

void q3_stdrng_f32(uint64_t state, int n, float* out) {
    uint32_t st[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u};
    const uint64_t MUL = 6364136223846793005ULL, INC = 11634580027462260723ULL;
    for (int i = 0; i < 8; ++i) {
        state = state * MUL + INC;
        const uint32_t xs = (uint32_t)(((state >> 18) ^ state) >> 27);
        const uint32_t rot = (uint32_t)(state >> 59);
        st[4 + i] = (xs >> rot) | (xs << ((32 - rot) & 31));
    }
    st[12] = st[13] = st[14] = st[15] = 0;
    uint32_t buf[64];
    int idx = 64;
    for (int i = 0; i < n; ++i) {
        if (idx >= 64) {
            for (int b = 0; b < 4; ++b) {
                chacha_block(st, 12, buf + 16 * b);
                if (++st[12] == 0) ++st[13];
            }
            idx = 0;
        }
        out[i] = (float)(buf[idx++] >> 8) * (1.0f / 16777216.0f);
    }
}
