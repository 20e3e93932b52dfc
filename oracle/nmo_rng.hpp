// nmo_rng.hpp — random stream of the CPU oracle.  TEST INFRASTRUCTURE ONLY (see nmo_math.hpp).
//
// The reference draws all randomness from third-party crates that are NOT under /root/reference and are
// not pinned by a lockfile (Cargo.toml:21-22: rand 0.10 with the `chacha` feature, rand_distr 0.6).
// Their published algorithms are restated here from the specifications:
//   * ChaCha (D. J. Bernstein, "ChaCha, a variant of Salsa20"), 8 rounds, 64-bit block counter in
//     words 12-13 and 64-bit stream id in words 14-15 (the rand_chacha layout); the generator output is
//     the keystream read as consecutive little-endian u32 words, next_u64 = lo word then hi word.
//   * SeedableRng::seed_from_u64: PCG32 (XSH-RR) expansion of the u64 into the 32-byte key.
//   * StandardUniform: bool = sign bit of one u32; f64 = 53 high bits of one u64 times 2^-53.
//   * Bernoulli (random_bool): p_int = (p * 2^64) as u64, true iff next_u64 < p_int; p == 1 consumes nothing.
//   * Uniform<f64>::new(lo, hi).sample: 52-bit fraction in [1,2) minus 1, times scale plus lo.
//   * StandardNormal: 256-layer ziggurat (Marsaglia & Tsang 2000) with R = 3.654152885361008796,
//     V = 4.92867323399e-3, tables x_0 = V/f(R), x_1 = R, x_{i+1} = f^-1(V/x_i + f(x_i)), x_256 = 0.
// PARITY UNPINNED: the reference's tests hold no fixed-seed golden values and the crates cannot be built
// here, so agreement of this stream with the real crates is best effort (ChaCha itself is pinned by the
// cipher's published keystream vectors, tests/golden/chacha_kat.json).
// Call sites in the reference: src/sampler.rs:761, :1105-1106; src/nuts.rs:200, :334;
// src/dynamics/hamiltonian.rs:113; src/stepsize/adapt.rs:259-261; src/math/cpu_math.rs:185, :567-575.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include "nmo_math.hpp"

namespace nmo {

static inline uint32_t rotl32(uint32_t x, int r) { return (x << r) | (x >> (32 - r)); }

static inline void chacha_block(const uint32_t key[8], uint64_t counter, uint64_t stream, int rounds,
                                uint32_t out[16]) {
    uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u,
                      key[0], key[1], key[2], key[3], key[4], key[5], key[6], key[7],
                      (uint32_t)counter, (uint32_t)(counter >> 32), (uint32_t)stream, (uint32_t)(stream >> 32)};
    uint32_t x[16];
    for (int i = 0; i < 16; ++i) x[i] = s[i];
#define NMO_QR(a, b, c, d)                                  \
    x[a] += x[b]; x[d] ^= x[a]; x[d] = rotl32(x[d], 16);    \
    x[c] += x[d]; x[b] ^= x[c]; x[b] = rotl32(x[b], 12);    \
    x[a] += x[b]; x[d] ^= x[a]; x[d] = rotl32(x[d], 8);     \
    x[c] += x[d]; x[b] ^= x[c]; x[b] = rotl32(x[b], 7);
    for (int r = 0; r < rounds; r += 2) {
        NMO_QR(0, 4, 8, 12) NMO_QR(1, 5, 9, 13) NMO_QR(2, 6, 10, 14) NMO_QR(3, 7, 11, 15)
        NMO_QR(0, 5, 10, 15) NMO_QR(1, 6, 11, 12) NMO_QR(2, 7, 8, 13) NMO_QR(3, 4, 9, 14)
    }
#undef NMO_QR
    for (int i = 0; i < 16; ++i) out[i] = x[i] + s[i];
}

// The word stream w_0, w_1, ... of ChaCha8(key, stream): block b holds words 16b..16b+15.
struct ChaCha8Rng {
    uint32_t key[8];
    uint64_t stream = 0;
    uint64_t pos = 0;           // next u32 word
    uint32_t buf[16];
    uint64_t buf_block = ~0ull;

    static ChaCha8Rng from_seed(const uint8_t seed[32]) {
        ChaCha8Rng r;
        for (int i = 0; i < 8; ++i)
            r.key[i] = (uint32_t)seed[4 * i] | ((uint32_t)seed[4 * i + 1] << 8) |
                       ((uint32_t)seed[4 * i + 2] << 16) | ((uint32_t)seed[4 * i + 3] << 24);
        return r;
    }
    // rand_core::SeedableRng::seed_from_u64 (PCG32 expansion)
    static ChaCha8Rng seed_from_u64(uint64_t state) {
        const uint64_t MUL = 6364136223846793005ull, INC = 11634580027462260723ull;
        uint8_t seed[32];
        for (int c = 0; c < 8; ++c) {
            state = state * MUL + INC;
            uint32_t xorshifted = (uint32_t)(((state >> 18) ^ state) >> 27);
            uint32_t rot = (uint32_t)(state >> 59);
            uint32_t x = (xorshifted >> rot) | (xorshifted << ((32 - rot) & 31));
            seed[4 * c] = (uint8_t)x; seed[4 * c + 1] = (uint8_t)(x >> 8);
            seed[4 * c + 2] = (uint8_t)(x >> 16); seed[4 * c + 3] = (uint8_t)(x >> 24);
        }
        return from_seed(seed);
    }
    void set_stream(uint64_t s) { stream = s; buf_block = ~0ull; }
    uint32_t next_u32() {
        uint64_t b = pos >> 4;
        if (b != buf_block) { chacha_block(key, b, stream, 8, buf); buf_block = b; }
        return buf[pos++ & 15];
    }
    uint64_t next_u64() { uint64_t lo = next_u32(); uint64_t hi = next_u32(); return (hi << 32) | lo; }
    // ChaCha8Rng::try_from_rng(self): a new generator keyed by the next 32 bytes
    ChaCha8Rng fork() {
        uint8_t seed[32];
        for (int i = 0; i < 8; ++i) {
            uint32_t w = next_u32();
            seed[4 * i] = (uint8_t)w; seed[4 * i + 1] = (uint8_t)(w >> 8);
            seed[4 * i + 2] = (uint8_t)(w >> 16); seed[4 * i + 3] = (uint8_t)(w >> 24);
        }
        return from_seed(seed);
    }
    bool random_bool_std() { return (int32_t)next_u32() < 0; }
    double random_f64() { return (double)(next_u64() >> 11) * (1.0 / 9007199254740992.0); }
};

// rng.random_bool(p): returns -1 when p is outside [0,1] (the reference panics there).
static inline int random_bool(ChaCha8Rng& rng, double p) {
    if (!(p >= 0.0 && p < 1.0)) {
        if (p == 1.0) return 1;
        return -1;
    }
    uint64_t p_int = (uint64_t)(p * 18446744073709551616.0);
    uint64_t v = rng.next_u64();
    return v < p_int ? 1 : 0;
}

// Uniform::new(low, high) then one sample (reference src/stepsize/adapt.rs:259-261)
struct UniformF64 {
    double low, scale;
    static UniformF64 make(double low, double high) {
        const double max_rand = 1.0 - 2.220446049250313e-16;
        double scale = high - low;
        while (scale * max_rand + low >= high) scale = u2f(f2u(scale) - 1);
        return {low, scale};
    }
    double sample(ChaCha8Rng& rng) const {
        double v12 = u2f((rng.next_u64() >> 12) | 0x3ff0000000000000ull);
        return (v12 - 1.0) * scale + low;
    }
};

// rand_distr ships FIXED tables (ZIG_NORM_X / ZIG_NORM_F: its generator's doubles after a '%.18f' round trip); they are
// restated by tools/gen_ziggurat_tables.py into nmo_zig_tables.hpp as hex floats — no run-time libm, same bits everywhere.
#include "nmo_zig_tables.hpp"
struct ZigguratTables {
    double x[257], f[257];
    double r;
    ZigguratTables() {
        static const double X[257] = NM_ZIG_NORM_X, F[257] = NM_ZIG_NORM_F;
        r = NM_ZIG_NORM_R;
        for (int i = 0; i < 257; ++i) { x[i] = X[i]; f[i] = F[i]; }
    }
};
static inline const ZigguratTables& zig_tables() { static ZigguratTables t; return t; }

// rand_distr::StandardNormal::sample::<f64>
static inline double standard_normal(ChaCha8Rng& rng, const Ctx& m) {
    const ZigguratTables& T = zig_tables();
    for (;;) {
        uint64_t bits = rng.next_u64();
        size_t i = (size_t)(bits & 0xff);
        double u = u2f((bits >> 12) | 0x4000000000000000ull) - 3.0;   // [2,4) - 3 -> [-1,1)
        double x = u * T.x[i];
        if (std::fabs(x) < T.x[i + 1]) return x;
        if (i == 0) {
            double xx = 1.0, yy = 0.0;
            while (-2.0 * yy < xx * xx) {
                // Open01: 52-bit fraction in (0,1)
                double a = u2f((rng.next_u64() >> 12) | 0x3ff0000000000000ull) - (1.0 - 2.220446049250313e-16 / 2.0);
                double b = u2f((rng.next_u64() >> 12) | 0x3ff0000000000000ull) - (1.0 - 2.220446049250313e-16 / 2.0);
                xx = m.ln(a) / T.r;
                yy = m.ln(b);
            }
            return u < 0.0 ? xx - T.r : T.r - xx;
        }
        if (T.f[i + 1] + (T.f[i] - T.f[i + 1]) * rng.random_f64() < m.exp(-x * x / 2.0)) return x;
    }
}

}  // namespace nmo
