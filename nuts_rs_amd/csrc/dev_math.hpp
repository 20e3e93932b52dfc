// dev_math.hpp — device-side scalar math, wavefront reductions and the ChaCha8 word stream (gfx950).
//
// One wavefront (64 lanes) owns one chain.  A chain vector of padded length DP = 64*DPL lives in HBM as
// [DP] doubles; lane l holds, for m = 0..DPL/2-1, the PAIR of elements d = 2*(m*64 + l) + {0,1} in registers,
// so every global access is a 16-byte-per-lane, 1-KiB-per-wave coalesced double2 transaction.
//
// Arithmetic contract: compiled with -ffp-contract=off; FMAs only where the reference
// writes mul_add (src/math/util.rs:161-168, :273-280, :376-379, :426-429, :477-480); reductions over dim are
// a per-lane serial sum in (m, j) order followed by an xor butterfly with offsets 1,2,4,8,16,32;
// exp / ln are the fdlibm algorithms written out in binary64 operations (the platform libm the reference
// calls through Rust's f64::exp is not bit-stable across machines; these are, and the CPU oracle has the
// same sequences so GPU and oracle agree bit-for-bit).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "detmath_tables.hpp"

namespace nm {

#define NM_DEV __device__ __forceinline__

#define NM_HD __host__ __device__ __forceinline__
#ifndef NM_PACKED_LF2
#define NM_PACKED_LF2 1       // the leapfrog's two sums (logp term, kinetic energy) too
#endif
#ifndef NM_PACKED_TESTS_N2
#define NM_PACKED_TESTS_N2 1     // the level-1 test's two sums as well
#endif
#ifndef NM_PACKED_SUMS
#define NM_PACKED_SUMS 1      // several sums of a wavefront through one transposed butterfly (wave_sum_packed); 0: one butterfly per value (rounds 1-4)
#endif
NM_HD uint64_t d2u(double x) { return __builtin_bit_cast(uint64_t, x); }
NM_HD double u2d(uint64_t u) { return __builtin_bit_cast(double, u); }
// Tile mode (nuts_tile.hpp): a block holds 16 independent chains, one wavefront each, so "the chain's block" is the
// wavefront: tid() is the lane, chain-wide synchronisation is wave-local; real block barriers exist only in the tile
// kernel's own rendezvous code.  A translation unit is compiled entirely in one mode.
#ifndef NM_TILE_MODE
#define NM_TILE_MODE 0
#endif
#define NM_ONE_WAVE_BLOCK (NM_TILE_MODE || blockDim.x == 64)
NM_DEV int lane_id() { return (int)(threadIdx.x & 63); }
// Block-wide synchronisation point.  With one wavefront per block the LDS pipeline is already in order (a lane sees
// what another lane of its wave wrote earlier), so nothing has to be waited for: __syncthreads() would still stall
// on every outstanding global store (s_waitcnt vmcnt(0)) — a compiler-level barrier is all that is needed.
NM_DEV void block_sync(bool single_wave) {
    if (single_wave || NM_TILE_MODE) asm volatile("" ::: "memory");
    else __syncthreads();
}
NM_DEV void chain_sync() { block_sync(false); }         // all threads of the CHAIN (the block, or in tile mode the wavefront)
NM_DEV int tid() { return NM_TILE_MODE ? (int)(threadIdx.x & 63) : (int)threadIdx.x; }   // thread within the chain's block (64*W threads)
// wave index inside the block; IS wave-uniform, but anything derived from threadIdx is divergent to the compiler
// unless it goes through readfirstlane (guide T20) — and values loaded through a "divergent" index poison everything
NM_DEV int wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

// ---- wavefront reductions ---------------------------------------------------------------------
// Sum over the 64 lanes, result identical in every lane.  Pairing = xor butterfly with offsets 1,2,4,8,16,32
// (the documented reduction order; oracle/nmo_math.hpp gpu_reduce).  Implemented without LDS traffic:
// offsets 1,2 are DPP quad permutes; for 4 and 8 every lane of a quad / of an 8-group already holds the same
// partial sum, so the DPP half-mirror / mirror (lane i <- 7-i / 15-i) pairs exactly the groups that xor 4 /
// xor 8 would, with identical bits; the four row sums are then combined as (r0+r1)+(r2+r3) from readlanes,
// which is what the xor 16 / xor 32 steps compute in every lane.
template <int CTRL>
NM_DEV double dpp_mov(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
NM_DEV double readlane_f64(double x, int lane) {
    int lo = __builtin_amdgcn_readlane(__double2loint(x), lane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(x), lane);
    return __hiloint2double(hi, lo);
}
NM_DEV double swap_add16(double x) {            // rows (r0, r1, r2, r3) -> (r0 + r1, r0 + r1, r2 + r3, r2 + r3)
    const unsigned lo = (unsigned)__double2loint(x), hi = (unsigned)__double2hiint(x);
    const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
}
NM_DEV double swap_add32(double x) {            // halves (lo, hi) -> lo + hi in every lane
    const unsigned lo = (unsigned)__double2loint(x), hi = (unsigned)__double2hiint(x);
    const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
}
NM_DEV double wave_sum(double x) {
    x = x + dpp_mov<0xB1>(x);    // quad_perm [1,0,3,2]  : xor 1
    x = x + dpp_mov<0x4E>(x);    // quad_perm [2,3,0,1]  : xor 2
    x = x + dpp_mov<0x141>(x);   // row_half_mirror      : pairs the two quads of each 8 (= xor 4)
    x = x + dpp_mov<0x140>(x);   // row_mirror           : pairs the two halves of each row of 16 (= xor 8)
    const double r0 = readlane_f64(x, 0), r1 = readlane_f64(x, 16), r2 = readlane_f64(x, 32), r3 = readlane_f64(x, 48);
    return (r0 + r1) + (r2 + r3);
}
NM_DEV double wave_rows(double x) {              // wave_sum's first four steps: the sum of the lane's row of 16, in every lane of the row
    x = x + dpp_mov<0xB1>(x);
    x = x + dpp_mov<0x4E>(x);
    x = x + dpp_mov<0x141>(x);
    return x + dpp_mov<0x140>(x);
}
// two sums at once (independent chains interleave in the issue stream)
NM_DEV void wave_sum2(double& a, double& b) {
    a = a + dpp_mov<0xB1>(a);  b = b + dpp_mov<0xB1>(b);
    a = a + dpp_mov<0x4E>(a);  b = b + dpp_mov<0x4E>(b);
    a = a + dpp_mov<0x141>(a); b = b + dpp_mov<0x141>(b);
    a = a + dpp_mov<0x140>(a); b = b + dpp_mov<0x140>(b);
    const double a0 = readlane_f64(a, 0), a1 = readlane_f64(a, 16), a2 = readlane_f64(a, 32), a3 = readlane_f64(a, 48);
    const double b0 = readlane_f64(b, 0), b1 = readlane_f64(b, 16), b2 = readlane_f64(b, 32), b3 = readlane_f64(b, 48);
    a = (a0 + a1) + (a2 + a3);
    b = (b0 + b1) + (b2 + b3);
}
// ---- several sums at once: the transposed butterfly (round 5) ---------------------------------------------------------------------
// A lone wavefront pays for the NUMBER of instructions it issues (tools/probes/ubench_issue.hip), and wave_sum above is 27 of them per
// value: six for a U-turn test group are 162.  The butterfly of N values can share its exchanges: at the xor-1 step a lane KEEPS one value
// of a pair (A, B) and GIVES the other to its partner, so one add serves two values; after the xor-2 step a quad's four lanes hold four
// values, after the xor-4 step an eight-group's lanes hold eight — the remaining steps work on ONE register whatever N is.  Each value
// still goes through exactly wave_sum's tree — (i, i^1), (i, i^2), the two quads of an eight-group, the two halves of a row, then
// (r0 + r1) + (r2 + r3) over the rows, IEEE addition being commutative — so every total has wave_sum's bits.
// wave_sum pairs quads and half rows with the DPP mirrors (lane i <-> 7 - i, i <-> 15 - i), which is only the xor pairing when the paired
// lanes hold the same thing; here they hold DIFFERENT values, so the value a lane keeps is chosen mirror-symmetrically: with b0..b3 the
// low bits of the lane id, value id = (b0 ^ b2) + 2 (b1 ^ b2) + 4 (b2 ^ b3) — every partner under i^1, i^2, 7 - i, 15 - i that must hold
// the same value does, every partner that must hold the other one does.  Total of value id v: in lane packed_lane(v) of every row.
// The last two steps use gfx950's v_permlane16_swap / v_permlane32_swap (rows / halves exchanged between two registers: with the same
// value in both, their sum is x + xor16(x), x + xor32(x)) instead of eight v_readlane and scalar-operand adds.
__host__ __device__ constexpr int packed_lane(int v) { return (((v >> 2) & 1) << 2) | ((((v >> 1) ^ (v >> 2)) & 1) << 1) | ((v ^ (v >> 2)) & 1); }
// one transposed step: a lane with `second` keeps b and gives a, the others keep a and give b; result = kept + partner's gift
template <int CTRL>
NM_DEV double fold_pair(double a, double b, bool second) {
    const double keep = second ? b : a, give = second ? a : b;
    return keep + dpp_mov<CTRL>(give);
}
// N = 2 .. 8 values -> one register: lane packed_lane(v) of every row holds the wave total of value v.  (Where a pair has only one member
// the plain step x + dpp(x) is taken: the lanes that would have kept the missing value hold a second copy of its sibling's partial — the
// sibling's tree is the same, and every lane of the result holds the total of SOME value < N: a sign test over all lanes sees exactly the N totals.)
template <int N>
NM_DEV double wave_sum_packed(const double (&v)[N]) {
    static_assert(N >= 2 && N <= 8, "2 .. 8 values");
    const int l = lane_id();
    const bool c1 = ((l ^ (l >> 2)) & 1) != 0, c2 = (((l >> 1) ^ (l >> 2)) & 1) != 0, c3 = (((l >> 2) ^ (l >> 3)) & 1) != 0;
    constexpr int NQ = (N + 1) / 2, NH = (NQ + 1) / 2;
    double q[NQ];                                  // after the xor-1 step: q[j] = values (2 j, 2 j + 1)
#pragma unroll
    for (int j = 0; j < NQ; ++j)
        q[j] = 2 * j + 1 < N ? fold_pair<0xB1>(v[2 * j], v[2 * j + 1 < N ? 2 * j + 1 : 0], c1) : v[2 * j] + dpp_mov<0xB1>(v[2 * j]);
    double h[NH];                                  // after the xor-2 step: h[k] = pairs (2 k, 2 k + 1)
#pragma unroll
    for (int k = 0; k < NH; ++k)
        h[k] = 2 * k + 1 < NQ ? fold_pair<0x4E>(q[2 * k], q[2 * k + 1 < NQ ? 2 * k + 1 : 0], c2) : q[2 * k] + dpp_mov<0x4E>(q[2 * k]);
    double x = NH == 2 ? fold_pair<0x141>(h[0], h[NH - 1], c3) : h[0] + dpp_mov<0x141>(h[0]);      // row_half_mirror: the two quads of an eight-group
    x = x + dpp_mov<0x140>(x);                                                                     // row_mirror: the two halves of a row
    x = swap_add16(x);
    return swap_add32(x);
}

// Block-wide sums for a chain that spans W waves.  Each wave reduces with DPP, lane 0 of every wave publishes its
// total in LDS, one barrier, then every thread adds the W totals in wave order (w = 0 first) — the documented
// cross-wave order (oracle gpu_reduce).  Two LDS buffers alternate so one barrier per reduction is enough.
constexpr int RED_MAX_VALUES = 6;

// ---- chains wider than one block (dim > 4096): NM_CLUSTER_MODE translation units (kern_cluster.hip) -------------------------
// `k` co-resident blocks ("members") own consecutive 4096-element slices of ONE chain and run the same instruction stream:
// every scalar decision is a function of block sums, random words and per-chain scalars, so it is enough that every sum is
// the same in all members.  A block's Reducer result is therefore followed by an exchange: each member publishes its
// partial sums in the chain's mailbox, all meet at a counter, and every member adds the k partials in member order
// (oracle gpu_reduce with gpu_slice: slice totals added in slice order).  Agent-scope release / acquire atomics carry the
// data between compute units (and XCDs); the grid never exceeds the resident capacity, so the spin cannot deadlock.
#ifndef NM_CLUSTER_MODE
#define NM_CLUSTER_MODE 0
#endif
constexpr int CL_MAX_MEMBERS = 32;          // blocks per chain at most (dim <= 32 x 4096): the members of a chain share an XCD, which has 32 CUs
constexpr int CL_BOX_WORDS = 6;              // u64 words of a chain's mailbox per (member, value): [2][k][V] counted + [2][k][2 V] tagged
struct ClusterLink {
    unsigned long long* box;     // [2][k][RED_MAX_VALUES] bit patterns of the members' partial sums (two epochs alternate), then
                                 // [2][k][2 RED_MAX_VALUES] tagged half-words of the same-XCD protocol
    unsigned long long* cnt;     // arrivals since the launch began
    unsigned long long epoch;    // exchanges this member has completed since the launch began
    int k, member;
    int same_xcd;                // every member reported the same XCC_ID at kernel start: the XCD's L2 is their coherence point
    int dead;                    // an exchange timed out (a member never arrived: the device did not hold the whole grid at once?):
                                 // no further waiting, the chain ends with an error status instead of hanging the device
};
// the XCD this wave runs on (HW_REG_XCC_ID, bits 3:0)
NM_DEV int xcc_id() { return (int)(__builtin_amdgcn_s_getreg(((4 - 1) << 11) | 20) & 0xfu); }

template <int W>
struct Reducer {
    double* buf;   // LDS [2][RED_MAX_VALUES][W] (+ RED_MAX_VALUES + 1 + 2 RED_MAX_VALUES CL_MAX_MEMBERS in cluster mode)
    int par;
    bool packed_tests = false;   // the six sums of a level-k / top-level U-turn test only (any_sign<6>): the 16-doubles tiling, where packing everything spills
    bool packed = false;   // several sums through ONE transposed butterfly (wave_sum_packed: same bits).  Set by the kernel per tiling: measured
                           // (2 doubles per lane) K3 +6.6 %, dim 100 +7 %; (4) dim 256 +4 %; (8) dim 512 -12 %; (16) K2 -24 % (96 more bytes of scratch): profiles/r05k_*, r05l_*
#if NM_CLUSTER_MODE
    ClusterLink cl;      // by value: a pointer to a link inside the chain's context would pin the whole context in scratch memory
    NM_DEV void init(double* lds) { buf = lds; par = 0; }       // (cl is set by the kernel; k <= 1: no exchange)
    template <int N>
    NM_DEV void cluster_combine(double (&v)[N]) {
        ClusterLink& L = cl;
        if (L.dead) return;
        double* lds_out = buf + 2 * RED_MAX_VALUES * W;
        constexpr unsigned long long SPIN_LIMIT = 1ull << 27;        // x s_sleep 1 (~64 cycles): several seconds
#ifndef NM_CLUSTER_COUNTED
        if (L.same_xcd) {
            // All members sit on one XCD (verified when the kernel started): its L2 is their point of coherence and performs the
            // agent-scope atomics, so no cache needs writing back or invalidating (the release / acquire pair of the general
            // protocol below costs about as much again as the whole leapfrog).  Lock-free on top of that: every partial sum travels
            // as two 8-byte words {tag = exchange number, 32 bits of the value} — single-copy atomic each, in any order — and a
            // reader knows a value is this exchange's when both tags match: one store and one round of polling loads instead of
            // store / wait / count / poll / load.  A slot is reused two exchanges later, which no member reaches before every
            // member has read this one (it must complete the exchange in between first).
            const unsigned k = (unsigned)L.k, total = 2u * N * k;
            unsigned long long* tb = L.box + 2ull * k * RED_MAX_VALUES + (L.epoch & 1ull) * (2ull * k * RED_MAX_VALUES);
            unsigned long long* stage = reinterpret_cast<unsigned long long*>(lds_out + RED_MAX_VALUES + 1);
            const unsigned tag = (unsigned)(L.epoch + 1ull) | 0x80000000u;     // never the 0 of a mailbox word nobody has written yet
            if (threadIdx.x < 64u) {
                const unsigned l = threadIdx.x;
                if (l < 2u * N) {
                    double mine = v[0];
#pragma unroll
                    for (int i = 1; i < N; ++i) mine = (l >> 1) == (unsigned)i ? v[i] : mine;
                    const unsigned long long bits = d2u(mine);
                    const unsigned half = (l & 1u) ? (unsigned)(bits >> 32) : (unsigned)bits;
                    __hip_atomic_store(&tb[(unsigned)L.member * (2u * RED_MAX_VALUES) + l], ((unsigned long long)tag << 32) | half,
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                unsigned long long spins = 0;
                bool timed_out = false;
                for (;;) {
                    bool ok = true;
                    for (unsigned w = l; w < total; w += 64u) {
                        const unsigned m = w / (2u * N), j = w % (2u * N);
                        const unsigned long long got = __hip_atomic_load(&tb[m * (2u * RED_MAX_VALUES) + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ok = ok && (unsigned)(got >> 32) == tag;
                        stage[w] = got;
                    }
                    if (__ballot(!ok) == 0ull) break;
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > SPIN_LIMIT) { timed_out = true; break; }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");        // the staged words, for the lanes that add them up
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                if (l == 0) lds_out[RED_MAX_VALUES] = timed_out ? 1.0 : 0.0;
                if (l < (unsigned)N) {                                       // the k partials in member order
                    double t = 0.0;
                    for (unsigned m = 0; m < k; ++m) {
                        const unsigned long long lo = stage[m * 2u * N + 2u * l], hi = stage[m * 2u * N + 2u * l + 1u];
                        const double x = u2d((hi << 32) | (lo & 0xffffffffull));
                        t = m == 0 ? x : t + x;
                    }
                    lds_out[l] = t;
                }
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < N; ++i) v[i] = lds_out[i];
            if (lds_out[RED_MAX_VALUES] != 0.0) L.dead = 1;
            __syncthreads();
            L.epoch += 1ull;
            return;
        }
#endif
        if (threadIdx.x == 0) {
            unsigned long long spins = 0;
            bool timed_out = false;
            unsigned long long* mine = L.box + ((L.epoch & 1ull) * (unsigned long long)L.k + (unsigned long long)L.member) * RED_MAX_VALUES;
#pragma unroll
            for (int i = 0; i < N; ++i) __hip_atomic_store(&mine[i], (unsigned long long)d2u(v[i]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long want = (L.epoch + 1ull) * (unsigned long long)L.k;
            if (L.same_xcd) {
                // all members sit on one XCD (verified when the kernel started): its L2 is their point of coherence, the
                // agent-scope atomics are performed there, and no cache needs writing back or invalidating — the release /
                // acquire pair below costs about as much again as the whole leapfrog (measured: 39 -> 22 us at dim 8192)
                __builtin_amdgcn_s_waitcnt(0);             // the partial sums are in the L2 before the arrival is counted
                (void)__hip_atomic_fetch_add(L.cnt, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                while (__hip_atomic_load(L.cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) { __builtin_amdgcn_s_sleep(1); if (++spins > SPIN_LIMIT) { timed_out = true; break; } }
            } else {
                (void)__hip_atomic_fetch_add(L.cnt, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                while (__hip_atomic_load(L.cnt, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) { __builtin_amdgcn_s_sleep(1); if (++spins > SPIN_LIMIT) { timed_out = true; break; } }
            }
            lds_out[RED_MAX_VALUES] = timed_out ? 1.0 : 0.0;
            const unsigned long long* all = L.box + (L.epoch & 1ull) * (unsigned long long)L.k * RED_MAX_VALUES;
#pragma unroll
            for (int i = 0; i < N; ++i) {
                double t = u2d(__hip_atomic_load(&all[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                for (int m = 1; m < L.k; ++m) t = t + u2d(__hip_atomic_load(&all[m * RED_MAX_VALUES + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                lds_out[i] = t;
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] = lds_out[i];
        if (lds_out[RED_MAX_VALUES] != 0.0) L.dead = 1;
        __syncthreads();
        L.epoch += 1ull;
    }
#else
    NM_DEV void init(double* lds) { buf = lds; par = 0; }
#endif
    template <int N>
    NM_DEV void sum_n(double (&v)[N]) {
        static_assert(N <= RED_MAX_VALUES, "too many values");
        if (NM_PACKED_SUMS && (packed || (packed_tests && N == 2 && NM_PACKED_LF2))) {
            if constexpr (N == 1) v[0] = swap_add32(swap_add16(wave_rows(v[0])));
            else {
                const double pk = wave_sum_packed<N>(v);       // one shared butterfly (same tree, same bits per value), the totals read out as uniform values
#pragma unroll
                for (int i = 0; i < N; ++i) v[i] = readlane_f64(pk, packed_lane(i));
            }
        } else {
            if (N == 1) v[0] = wave_sum(v[0]);
            else {
#pragma unroll
                for (int i = 0; i + 1 < N; i += 2) wave_sum2(v[i], v[i + 1]);
                if (N & 1) v[N - 1] = wave_sum(v[N - 1]);
            }
        }
        if (W != 1) {
            double* b = buf + par * (RED_MAX_VALUES * W);
            if (lane_id() == 0) {
#pragma unroll
                for (int i = 0; i < N; ++i) b[i * W + wave_id()] = v[i];
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < N; ++i) {
                double t = b[i * W];
#pragma unroll
                for (int w = 1; w < W; ++w) t = t + b[i * W + w];
                v[i] = t;
            }
            par ^= 1;
        }
#if NM_CLUSTER_MODE
        if (cl.k > 1) cluster_combine<N>(v);
#endif
    }
    // is any of the N block sums negative (neg) / positive (!neg)?  (the U-turn tests: is_turning's "< 0" with the direction folded into the
    // sign, nuts_kernels.hpp turning_regs).  One wavefront per chain with packed sums: every lane of the packed register holds one of the N totals,
    // so the answer is a compare and a ballot — no total is read out.  Same decision as sum_n + N scalar compares.
    template <int N>
    NM_DEV bool any_sign(double (&v)[N], bool neg) {
        if (NM_PACKED_SUMS && (packed || (packed_tests && (N == 6 || (N == 2 && NM_PACKED_TESTS_N2)))) && W == 1 && !NM_CLUSTER_MODE) {
            const double pk = wave_sum_packed<N>(v);
            return __ballot(neg ? pk < 0. : pk > 0.) != 0ull;
        }
        sum_n(v);
        bool any = false;
#pragma unroll
        for (int i = 0; i < N; ++i) any = any | (neg ? v[i] < 0. : v[i] > 0.);
        return any;
    }
    NM_DEV double sum(double x) { double v[1] = {x}; sum_n(v); return v[0]; }
    NM_DEV void sum2(double& a, double& b) { double v[2] = {a, b}; sum_n(v); a = v[0]; b = v[1]; }
    // true iff `ok` holds in every thread of the block
    NM_DEV bool all(bool ok) {
        const bool wave_ok = __ballot(!ok) == 0ull;
        if (W == 1 && !NM_CLUSTER_MODE) return wave_ok;
        return sum(wave_ok ? 0.0 : 1.0) == 0.0;
    }
};

// value of lane `src` (wave-uniform lane index) as a wave-uniform value
NM_DEV uint64_t wave_bcast_u64(uint64_t x, int src) {
    uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)x, src);
    uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(x >> 32), src);
    return ((uint64_t)hi << 32) | lo;
}
// Tell the compiler a value is wave-uniform (it is: every lane computed it from uniform inputs).  Results of
// real calls (dexp/dlog are not inlined) count as divergent otherwise, which would turn all the scalar tree
// logic into exec-masked control flow held in VGPRs.
NM_DEV double uniform_f64(double x) {
    int lo = __builtin_amdgcn_readfirstlane(__double2loint(x));
    int hi = __builtin_amdgcn_readfirstlane(__double2hiint(x));
    return __hiloint2double(hi, lo);
}

// ---- deterministic exp / ln / ln_1p ---------------------------------------------------------------
// Table-driven (Tang-style) reductions with tables from tools/gen_detmath_tables.py; every step is one IEEE-754
// binary64 operation (+, -, *, fma, round-to-nearest-even, power-of-two scaling), no division, ~12 dependent
// operations: these run on the latency-critical scalar path of every tree merge.  oracle/nmo_math.hpp executes
// the same sequence on the host (bit-identical results); tests bound both against libm (< 1 ulp).
//   exp:  x = n (ln2/64) + r, n = 64 k + j:  exp x = 2^k T[j] (1 + p(r))
//   ln :  x = 2^k m, m in [sqrt 1/2, sqrt 2), j = rint(64 m), z = m R[j] - 1:  ln x = k ln2 - ln R[j] + log1p(z)
// On the device the tables live in LDS (2.1 KiB per block, copied in by dm_init_lds() at the start of every kernel):
// a lookup costs an LDS read instead of a trip to L2 (two dependent ones per logaddexp, ~1 us in the tree's merges).
static constexpr double DM_T_HI[64] = DM_EXP_T_HI, DM_T_LO[64] = DM_EXP_T_LO;
static constexpr double DM_R[47] = DM_LOG_R, DM_F_HI[47] = DM_LOG_F_HI, DM_F_LO[47] = DM_LOG_F_LO;
constexpr int DM_OFF_T_HI = 0, DM_OFF_T_LO = 64, DM_OFF_R = 128, DM_OFF_F_HI = 175, DM_OFF_F_LO = 222, DM_LDS_DOUBLES = 269;
#if defined(__HIP_DEVICE_COMPILE__)
// (16-byte aligned and a multiple of 16 bytes long: the kernels' own LDS structures follow it, and their 128-bit accesses — sigma / mu tiles, L[1] —
// were 8 bytes off a 16-byte boundary for as long as this array was 269 doubles: every ds_read_b128 / ds_write_b128 of the hot loops misaligned)
alignas(16) static __shared__ double dm_lds[(DM_LDS_DOUBLES + 1) / 2 * 2 + 2];
#define DM_TAB(name, off, idx) dm_lds[(off) + (idx)]
#else
#define DM_TAB(name, off, idx) name[idx]
#endif
// every kernel that may evaluate exp / ln calls this first (all threads)
static __device__ __forceinline__ void dm_init_lds() {
#if defined(__HIP_DEVICE_COMPILE__)
    for (int i = (int)threadIdx.x; i < DM_LDS_DOUBLES; i += (int)blockDim.x) {
        double v;
        if (i < DM_OFF_T_LO) v = DM_T_HI[i];
        else if (i < DM_OFF_R) v = DM_T_LO[i - DM_OFF_T_LO];
        else if (i < DM_OFF_F_HI) v = DM_R[i - DM_OFF_R];
        else if (i < DM_OFF_F_LO) v = DM_F_HI[i - DM_OFF_F_HI];
        else v = DM_F_LO[i - DM_OFF_F_LO];
        dm_lds[i] = v;
    }
    __syncthreads();
#endif
}

// U = true: the argument is wave-uniform; the table index is moved to a scalar register so that the lookups are
// scalar loads (their own counter) instead of vector loads, which would queue behind every outstanding store.
template <bool U>
static __host__ __device__ __forceinline__ int dm_index(int j) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (U) return __builtin_amdgcn_readfirstlane(j);
#endif
    return j;
}

template <bool U>
static __host__ __device__ __forceinline__ double dexp_branchy(double x) {
    if (x != x) return x;
    if (x > 7.09782712893383973096e+02) return __builtin_inf();
    if (x < -7.45133219101941108420e+02) return 0.0;
    const double nf = __builtin_rint(x * DM_EXP_INV_L);
    const double r1 = __builtin_fma(-nf, DM_EXP_L_HI, x);
    const double r = __builtin_fma(-nf, DM_EXP_L_LO, r1);
    const int n = dm_index<U>((int)nf);
    const int j = n & 63, k = n >> 6;
    const double r2 = r * r;
    const double a = __builtin_fma(r, DM_EXP_E3, DM_EXP_E2);
    const double b = __builtin_fma(r, DM_EXP_E5, DM_EXP_E4);
    const double q = __builtin_fma(r2, __builtin_fma(r2, DM_EXP_E6, b), a);
    const double p = __builtin_fma(r2, q, r);                 // expm1(r)
    const double th = DM_TAB(DM_T_HI, DM_OFF_T_HI, j);
    const double sum = __builtin_fma(th, p, DM_TAB(DM_T_LO, DM_OFF_T_LO, j));
    return __builtin_ldexp(th + sum, k);
}
// (round 5) dexp_impl / dlog_impl / dlog1p_impl ARE the branch-free forms below (exp_sl, log_sl, log1p_sl: same operation sequences,
// same results on every double — tests/cpp/merge_math_check.hip compares them with the *_branchy originals above); the template
// parameter (a wave-uniform argument: scalar table loads) is accepted and ignored
template <bool U> static __host__ __device__ __forceinline__ double dexp_impl(double x);
template <bool U> static __host__ __device__ __forceinline__ double dlog_impl(double x);
template <bool U> static __host__ __device__ __forceinline__ double dlog1p_impl(double x);
#ifndef NM_DETMATH_INLINE
#define NM_DETMATH_INLINE 0      // 1: exp / ln / ln_1p / merge_math / sin-cos / expm1 inlined at every call site: NO out-of-line device call in the unit.
                                 // The build sets it for the kernels of the small tilings (<= 4 doubles per lane: nuts_launch.hpp NM_TU_PART 1; DESIGN §22,
                                 // fourth incident: every instantiation that has gone wrong was one of them) and for user modules of those tilings; the
                                 // 8- and 16-doubles tilings keep the calls (inlined, K2 loses 11 %: profiles/r05ac_inline_all_ab.txt)
#endif
#if NM_DETMATH_INLINE
#define NM_DM_CALL __forceinline__
#else
#define NM_DM_CALL __noinline__
#endif
static __host__ __device__ NM_DM_CALL double dexp(double x);

// ln(x) + c / x for finite x > 0 (c = 0: plain ln; ln_1p passes the rounding error of 1 + t)
template <bool U>
static __host__ __device__ __forceinline__ double dlog_core(double x, double c) {
    uint64_t u = d2u(x);
    int k = 0;
    if (u < 0x0010000000000000ull) {                          // subnormal
        x *= 1.80143985094819840000e+16;                      // 2^54
        u = d2u(x);
        k = -54;
    }
    const uint64_t mant = u & 0x000fffffffffffffull;
    const int up = mant >= 0x6a09e667f3bcdull;                // mantissa >= sqrt(2): halve it
    k += (int)(u >> 52) - 1023 + up;
    const double m = u2d(mant | ((uint64_t)(1023 - up) << 52));    // [sqrt 1/2, sqrt 2)
    const int j = dm_index<U>((int)__builtin_rint(m * 64.0) - DM_LOG_J0);
    const double rj = DM_TAB(DM_R, DM_OFF_R, j);
    const double z = __builtin_fma(m, rj, -1.0);
    const double dk = (double)k;
    const double z2 = z * z, z4 = z2 * z2;
    const double p01 = __builtin_fma(z, DM_LOG_C3, DM_LOG_C2), p23 = __builtin_fma(z, DM_LOG_C5, DM_LOG_C4);
    const double p45 = __builtin_fma(z, DM_LOG_C7, DM_LOG_C6), p67 = __builtin_fma(z, DM_LOG_C9, DM_LOG_C8);
    const double q0 = __builtin_fma(z2, p23, p01), q1 = __builtin_fma(z2, p67, p45);
    const double Q = __builtin_fma(z4, __builtin_fma(z4, DM_LOG_C10, q1), q0);
    const int kc = k < -1000 ? -1000 : (k > 1000 ? 1000 : k);
    const double corr = (c * rj) * u2d((uint64_t)(1023 - kc) << 52);
    const double lo = __builtin_fma(dk, DM_LN2_LO, DM_TAB(DM_F_LO, DM_OFF_F_LO, j)) + corr;
    const double t = __builtin_fma(z2, Q, lo);
    const double hk = dk * DM_LN2_HI;                         // exact
    const double fh = DM_TAB(DM_F_HI, DM_OFF_F_HI, j);
    const double s1 = hk + fh, e1 = (hk - s1) + fh;
    const double s2 = s1 + z, e2 = (s1 - s2) + z;
    return s2 + ((e1 + e2) + t);
}

template <bool U>
static __host__ __device__ __forceinline__ double dlog_branchy(double x) {
    if (x != x) return x;
    if (x < 0.0) return __builtin_nan("");
    if (x == 0.0) return -__builtin_inf();
    if (__builtin_isinf(x)) return x;
    return dlog_core<U>(x, 0.0);
}
// ln(1 + x): ln of the rounded sum plus the first-order term of its rounding error
template <bool U>
static __host__ __device__ __forceinline__ double dlog1p_branchy(double x) {
    const double u = 1.0 + x;
    if (u == 1.0) return x;
    if (!(u == u) || __builtin_isinf(u) || !(u > 0.0)) return dlog_branchy<U>(u);
    return dlog_core<U>(u, x - (u - 1.0));
}
static __host__ __device__ NM_DM_CALL double dlog(double x);
static __host__ __device__ NM_DM_CALL double dlog1p(double x);
// per-lane logaddexp (reference src/math/util.rs:6-19)
NM_DEV double logaddexp_lane(double a, double b) {
    if (a == b) return a + dlog(2.0);
    double diff = a - b;
    if (diff > 0.) return a + dlog1p(dexp(-diff));
    if (diff < 0.) return b + dlog1p(dexp(diff));
    return diff;
}
// wave-uniform variants: same arithmetic, result marked uniform
// (calls, not inlined copies with scalar table loads: measured, the inlined form is 3 % slower on K2 — the waits it
// removes from the calls reappear at the next memory operation, and the extra code costs registers)
NM_DEV double uexp(double x) { return uniform_f64(dexp(x)); }
NM_DEV double ulog(double x) { return uniform_f64(dlog(x)); }
NM_DEV double ulog1p(double x) { return uniform_f64(dlog1p(x)); }
// reference src/math/util.rs:6-19
NM_DEV double logaddexp(double a, double b) {
    if (a == b) return a + ulog(2.0);
    double diff = a - b;
    if (diff > 0.) return a + ulog1p(uexp(-diff));
    if (diff < 0.) return b + ulog1p(uexp(diff));
    return diff;
}

// ---- the merge's scalar arithmetic as ONE straight-line routine (round 5) --------------------------------------------------
// A lone wavefront issues one instruction per ~4.3 cycles whatever its kind and whether or not it depends on the one before
// (tools/probes/ubench_issue.hip), a taken branch costs ~25 cycles, a VALU compare feeding a scalar branch ~38, a call ~78: what a
// merge costs is the NUMBER of instructions on its path, not the latency of its dependent chain.  merge_into's arithmetic
// (src/nuts.rs:172-207: logaddexp = exp + ln_1p, exp, the Bernoulli compare) through the general-purpose dexp / dlog1p is ~400
// instructions (exec-masked special cases, 64-bit constants built from two s_mov each, three calls); here it is ONE routine
// without a branch: the main path of every function is evaluated unconditionally on operands that cannot fault (table indices are
// masked, conversions saturate), the special cases are selects at the end, and the operation sequence of the main path is EXACTLY
// dexp_impl / dlog1p_impl / dlog_core's (same bits; tests/test_merge_math.py runs both on the host over special and random operands).
//   exp_sl(x)       == dexp_impl(x) for every x
//   log1p_unit(x)   == dlog1p_impl(x) for x in [0, 1] or NaN (what exp(-|d|) can be)
NM_HD int dm_cvt_i32(double x) {           // v_cvt_i32_f64: saturating, NaN -> 0 (a C cast of an out-of-range value is undefined)
#if defined(__HIP_DEVICE_COMPILE__)
    // the compiler's own conversion of an operand clamped into range first (v_max_f64 / v_min_f64 return the other operand for a NaN):
    // defined for every double; operands that were in range — the only ones whose result is used — convert as before
    return (int)__builtin_fmin(__builtin_fmax(x, -2.0e9), 2.0e9);
#else
    return x != x ? 0 : (x >= 2147483647.0 ? 2147483647 : (x <= -2147483648.0 ? (int)(-2147483647 - 1) : (int)x));
#endif
}
NM_HD double exp_sl(double x) {
    const double nf = __builtin_rint(x * DM_EXP_INV_L);
    const double r1 = __builtin_fma(-nf, DM_EXP_L_HI, x);
    const double r = __builtin_fma(-nf, DM_EXP_L_LO, r1);
    const int n = dm_cvt_i32(nf);
    const int j = n & 63, k = n >> 6;
    const double r2 = r * r;
    const double a = __builtin_fma(r, DM_EXP_E3, DM_EXP_E2);
    const double b = __builtin_fma(r, DM_EXP_E5, DM_EXP_E4);
    const double q = __builtin_fma(r2, __builtin_fma(r2, DM_EXP_E6, b), a);
    const double p = __builtin_fma(r2, q, r);
    const double th = DM_TAB(DM_T_HI, DM_OFF_T_HI, j);
    const double sum = __builtin_fma(th, p, DM_TAB(DM_T_LO, DM_OFF_T_LO, j));
    double res = __builtin_ldexp(th + sum, k);
    res = x > 7.09782712893383973096e+02 ? __builtin_inf() : res;
    res = x < -7.45133219101941108420e+02 ? 0.0 : res;
    return x != x ? x : res;
}
NM_HD double log1p_unit(double x) {
    const double u = 1.0 + x;                                  // [1, 2] or NaN
    const double c = x - (u - 1.0);
    const uint64_t ub = d2u(u);
    const uint64_t mant = ub & 0x000fffffffffffffull;
    const int up = mant >= 0x6a09e667f3bcdull;
    const int k = (int)(ub >> 52) - 1023 + up;                 // 0, 1 (or anything for NaN: discarded)
    const double m = u2d(mant | ((uint64_t)(1023 - up) << 52));
    const int j = (int)__builtin_rint(m * 64.0) - DM_LOG_J0;   // m in [sqrt 1/2, sqrt 2) whatever u's bits: 0..46
    const double rj = DM_TAB(DM_R, DM_OFF_R, j);
    const double z = __builtin_fma(m, rj, -1.0);
    const double dk = (double)k;
    const double z2 = z * z, z4 = z2 * z2;
    const double p01 = __builtin_fma(z, DM_LOG_C3, DM_LOG_C2), p23 = __builtin_fma(z, DM_LOG_C5, DM_LOG_C4);
    const double p45 = __builtin_fma(z, DM_LOG_C7, DM_LOG_C6), p67 = __builtin_fma(z, DM_LOG_C9, DM_LOG_C8);
    const double q0 = __builtin_fma(z2, p23, p01), q1 = __builtin_fma(z2, p67, p45);
    const double Q = __builtin_fma(z4, __builtin_fma(z4, DM_LOG_C10, q1), q0);
    const int kc = k < -1000 ? -1000 : (k > 1000 ? 1000 : k);
    const double corr = (c * rj) * u2d((uint64_t)(1023 - kc) << 52);
    const double lo = __builtin_fma(dk, DM_LN2_LO, DM_TAB(DM_F_LO, DM_OFF_F_LO, j)) + corr;
    const double t = __builtin_fma(z2, Q, lo);
    const double hk = dk * DM_LN2_HI;
    const double fh = DM_TAB(DM_F_HI, DM_OFF_F_HI, j);
    const double s1 = hk + fh, e1 = (hk - s1) + fh;
    const double s2 = s1 + z, e2 = (s1 - s2) + z;
    double res = s2 + ((e1 + e2) + t);
    res = u == 1.0 ? x : res;
    return u != u ? u : res;
}
// ln(x) + c / x on every double, ln(1 + x) on every double: dlog_impl / dlog1p_impl without a branch (the sub-normal rescaling and
// the special cases are selects).  Same operation sequence on the main path, same results everywhere (tests/cpp/merge_math_check.hip).
NM_HD double log_core_sl(double x, double c) {
    uint64_t u = d2u(x);
    const bool sub = u < 0x0010000000000000ull;
    const double xs = x * 1.80143985094819840000e+16;           // 2^54
    u = sub ? d2u(xs) : u;
    int k = sub ? -54 : 0;
    const uint64_t mant = u & 0x000fffffffffffffull;
    const int up = mant >= 0x6a09e667f3bcdull;
    k += (int)(u >> 52) - 1023 + up;
    const double m = u2d(mant | ((uint64_t)(1023 - up) << 52));
    const int j = (int)__builtin_rint(m * 64.0) - DM_LOG_J0;
    const double rj = DM_TAB(DM_R, DM_OFF_R, j);
    const double z = __builtin_fma(m, rj, -1.0);
    const double dk = (double)k;
    const double z2 = z * z, z4 = z2 * z2;
    const double p01 = __builtin_fma(z, DM_LOG_C3, DM_LOG_C2), p23 = __builtin_fma(z, DM_LOG_C5, DM_LOG_C4);
    const double p45 = __builtin_fma(z, DM_LOG_C7, DM_LOG_C6), p67 = __builtin_fma(z, DM_LOG_C9, DM_LOG_C8);
    const double q0 = __builtin_fma(z2, p23, p01), q1 = __builtin_fma(z2, p67, p45);
    const double Q = __builtin_fma(z4, __builtin_fma(z4, DM_LOG_C10, q1), q0);
    const int kc = k < -1000 ? -1000 : (k > 1000 ? 1000 : k);
    const double corr = (c * rj) * u2d((uint64_t)(1023 - kc) << 52);
    const double lo = __builtin_fma(dk, DM_LN2_LO, DM_TAB(DM_F_LO, DM_OFF_F_LO, j)) + corr;
    const double t = __builtin_fma(z2, Q, lo);
    const double hk = dk * DM_LN2_HI;
    const double fh = DM_TAB(DM_F_HI, DM_OFF_F_HI, j);
    const double s1 = hk + fh, e1 = (hk - s1) + fh;
    const double s2 = s1 + z, e2 = (s1 - s2) + z;
    return s2 + ((e1 + e2) + t);
}
NM_HD double log_sl(double x) {
    double res = log_core_sl(x, 0.0);
    res = __builtin_isinf(x) ? x : res;                         // (+inf; -inf is negative: NaN below)
    res = x == 0.0 ? -__builtin_inf() : res;
    res = x < 0.0 ? __builtin_nan("") : res;
    return x != x ? x : res;
}
NM_HD double log1p_sl(double x) {
    const double u = 1.0 + x;
    double res = log_core_sl(u, x - (u - 1.0));
    // dlog1p_impl: u == 1 -> x; u NaN, infinite or not > 0 -> dlog_impl(u)
    res = __builtin_isinf(u) ? u : res;
    res = u == 0.0 ? -__builtin_inf() : res;
    res = u < 0.0 ? __builtin_nan("") : res;
    res = u != u ? u : res;
    return u == 1.0 ? x : res;
}
#ifndef NM_BRANCH_FREE_MATH
#define NM_BRANCH_FREE_MATH 1     // 0: the general-purpose forms of rounds 1-4 behind the same names (bisecting builds)
#endif
#if NM_BRANCH_FREE_MATH
template <bool U> static __host__ __device__ __forceinline__ double dexp_impl(double x) { return exp_sl(x); }
template <bool U> static __host__ __device__ __forceinline__ double dlog_impl(double x) { return log_sl(x); }
template <bool U> static __host__ __device__ __forceinline__ double dlog1p_impl(double x) { return log1p_sl(x); }
static __host__ __device__ NM_DM_CALL double dexp(double x) { return exp_sl(x); }
static __host__ __device__ NM_DM_CALL double dlog(double x) { return log_sl(x); }
static __host__ __device__ NM_DM_CALL double dlog1p(double x) { return log1p_sl(x); }
#else
template <bool U> static __host__ __device__ __forceinline__ double dexp_impl(double x) { return dexp_branchy<U>(x); }
template <bool U> static __host__ __device__ __forceinline__ double dlog_impl(double x) { return dlog_branchy<U>(x); }
template <bool U> static __host__ __device__ __forceinline__ double dlog1p_impl(double x) { return dlog1p_branchy<U>(x); }
static __host__ __device__ NM_DM_CALL double dexp(double x) { return dexp_branchy<false>(x); }
static __host__ __device__ NM_DM_CALL double dlog(double x) { return dlog_branchy<false>(x); }
static __host__ __device__ NM_DM_CALL double dlog1p(double x) { return dlog1p_branchy<false>(x); }
#endif
// merge_into's scalars (reference src/nuts.rs:172-207).  (w_lo, w_hi): the NEXT u64 of the chain's stream, read but not consumed by
// the caller.  flags: bit 0 take other's draw, bit 1 the u64 was consumed (random_bool drew), bit 2 fatal (p outside [0, 1]: the reference panics)
struct MergeOut { double total; uint32_t flags; };
NM_HD MergeOut merge_math_impl(double a, double b, uint32_t is_main, uint32_t w_lo, uint32_t w_hi) {
    const double diff = a - b;
    const double e = exp_sl(diff > 0. ? -diff : diff);
    const double lp = log1p_unit(e);
    double total = (diff > 0. ? a : b) + lp;
    total = (diff > 0. || diff < 0.) ? total : diff;                        // neither: NaN (util.rs:18)
    total = a == b ? a + 0x1.62e42fefa39efp-1 : total;                      // = dlog(2.0), bit for bit (tests/test_merge_math.py)
    const double self_log_size = is_main ? a : total;
    const bool ge = b >= self_log_size;
    const double p_ = exp_sl(b - self_log_size);
    const bool in01 = p_ >= 0.0 && p_ < 1.0;                                // random_bool(p): p outside [0, 1) draws nothing
    // (u64)(p * 2^64) for p in [0, 1): the integer part of an exact product
    const double ph = __builtin_floor(p_ * 4294967296.0);                   // high 32 bits
    const double pl = __builtin_floor(__builtin_fma(p_, 18446744073709551616.0, -ph * 4294967296.0));
    const uint64_t p_int = in01 ? (((uint64_t)(uint32_t)ph << 32) | (uint64_t)(uint32_t)pl) : 0ull;
    const uint64_t w = ((uint64_t)w_hi << 32) | w_lo;
    const bool draws = !ge && in01;
    const bool take = ge || (in01 ? w < p_int : p_ == 1.0);
    const bool fatal = !ge && !in01 && !(p_ == 1.0);
    MergeOut o;
    o.total = total;
    o.flags = (take && !fatal ? 1u : 0u) | (draws ? 2u : 0u) | (fatal ? 4u : 0u);
    return o;
}
static __device__ NM_DM_CALL MergeOut merge_math(double a, double b, uint32_t is_main, uint32_t w_lo, uint32_t w_hi) { return merge_math_impl(a, b, is_main, w_lo, w_hi); }

// exp(x) - 1 for the isokinetic momentum refresh (reference f64::exp_m1, transformed_hamiltonian.rs:800-801): the same
// operation sequence as oracle/nmo_math.hpp det_expm1 (Taylor series to x^14 for |x| <= 0.35, else exp(x) - 1)
template <bool INL>
static __host__ __device__ __forceinline__ double dexpm1_impl(double x) {
    if (!(__builtin_fabs(x) <= 0.35)) return (INL ? dexp_impl<false>(x) : dexp(x)) - 1.0;
    double p = 1.1470745597729725e-11;                                 // 1/14!
    p = __builtin_fma(x, p, 1.6059043836821613e-10);    // 1/13!
    p = __builtin_fma(x, p, 2.08767569878681e-09);    // 1/12!
    p = __builtin_fma(x, p, 2.505210838544172e-08);    // 1/11!
    p = __builtin_fma(x, p, 2.755731922398589e-07);    // 1/10!
    p = __builtin_fma(x, p, 2.7557319223985893e-06);    // 1/9!
    p = __builtin_fma(x, p, 2.48015873015873e-05);    // 1/8!
    p = __builtin_fma(x, p, 0.0001984126984126984);    // 1/7!
    p = __builtin_fma(x, p, 0.001388888888888889);    // 1/6!
    p = __builtin_fma(x, p, 0.008333333333333333);    // 1/5!
    p = __builtin_fma(x, p, 0.041666666666666664);    // 1/4!
    p = __builtin_fma(x, p, 0.16666666666666666);    // 1/3!
    p = __builtin_fma(x, p, 0.5);
    return __builtin_fma(x * x, p, x);
}
// (out of line for the one-chain kernels; kernels with several chains per wavefront use dexpm1_impl<true>: nothing out of line under a
// branch that is not uniform over the wavefront)
static __host__ __device__ NM_DM_CALL double dexpm1(double x) { return dexpm1_impl<false>(x); }

// sin / cos of a step size (the ExactNormal trajectory kind; reference f64::sin / f64::cos, src/math/util.rs:580-581):
// Cody-Waite reduction by pi/2 in two fma steps and the classic minimax kernels on [-pi/4, pi/4]; the same operation
// sequence as oracle/nmo_math.hpp det_sincos (bit-identical), ~1 ulp from libm.
static __host__ __device__ __forceinline__ double2 dsincos_impl(double x) {     // (sin x, cos x)
    if (!(x == x) || __builtin_isinf(x)) return make_double2(__builtin_nan(""), __builtin_nan(""));
    const double ax = __builtin_fabs(x);
    const double nf = __builtin_rint(ax * 6.36619772367581382433e-01);
    double r = __builtin_fma(-nf, 1.57079632679489655800e+00, ax);
    r = __builtin_fma(-nf, 6.12323399573676603587e-17, r);
    const double z = r * r;
    double ps = __builtin_fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = __builtin_fma(z, ps, 2.75573137070700676789e-06);
    ps = __builtin_fma(z, ps, -1.98412698298579493134e-04);
    ps = __builtin_fma(z, ps, 8.33333333332248946124e-03);
    ps = __builtin_fma(z, ps, -1.66666666666666324348e-01);
    const double sr = __builtin_fma(r * z, ps, r);
    double pc = __builtin_fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = __builtin_fma(z, pc, -2.75573143513906633035e-07);
    pc = __builtin_fma(z, pc, 2.48015872894767294178e-05);
    pc = __builtin_fma(z, pc, -1.38888888888741095749e-03);
    pc = __builtin_fma(z, pc, 4.16666666666666019037e-02);
    const double cr = __builtin_fma(z * z, pc, __builtin_fma(-0.5, z, 1.0));
    const int q = (int)((long long)nf & 3);
    double s_, c_;
    if (q == 0) { s_ = sr; c_ = cr; }
    else if (q == 1) { s_ = cr; c_ = -sr; }
    else if (q == 2) { s_ = -sr; c_ = -cr; }
    else { s_ = -cr; c_ = sr; }
    return make_double2(x < 0.0 ? -s_ : s_, c_);
}
// out of line for the one-chain kernels (every lane of the wavefront calls it together); kernels with several chains per wavefront call
// dsincos_impl: no out-of-line call under a branch that is not uniform over the wavefront
static __host__ __device__ NM_DM_CALL double2 dsincos(double x) { return dsincos_impl(x); }

NM_DEV bool is_finite(double x) { return __builtin_fabs(x) < __builtin_inf(); }
NM_DEV double clampd(double v, double lo, double hi) {   // f64::clamp: NaN stays NaN
    if (v < lo) return lo;
    if (v > hi) return hi;
    return v;
}
// f64::min(self, other): if one is NaN returns the other
NM_DEV double fmin_rs(double a, double b) { return __builtin_fmin(a, b); }

// ---- ChaCha8 word stream (rand's ChaCha8Rng semantics; see oracle/nmo_rng.hpp for the restated spec) -----
NM_DEV void chacha8_block(const uint32_t* key, uint64_t counter, uint64_t stream, uint32_t (&out)[16]) {
    uint32_t s[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u,
                      key[0], key[1], key[2], key[3], key[4], key[5], key[6], key[7],
                      (uint32_t)counter, (uint32_t)(counter >> 32), (uint32_t)stream, (uint32_t)(stream >> 32)};
    uint32_t x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = s[i];
#define NM_QR(a, b, c, d)                                                   \
    x[a] += x[b]; x[d] ^= x[a]; x[d] = __builtin_rotateleft32(x[d], 16);    \
    x[c] += x[d]; x[b] ^= x[c]; x[b] = __builtin_rotateleft32(x[b], 12);    \
    x[a] += x[b]; x[d] ^= x[a]; x[d] = __builtin_rotateleft32(x[d], 8);     \
    x[c] += x[d]; x[b] ^= x[c]; x[b] = __builtin_rotateleft32(x[b], 7);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        NM_QR(0, 4, 8, 12) NM_QR(1, 5, 9, 13) NM_QR(2, 6, 10, 14) NM_QR(3, 7, 11, 15)
        NM_QR(0, 5, 10, 15) NM_QR(1, 6, 11, 12) NM_QR(2, 7, 8, 13) NM_QR(3, 4, 9, 14)
    }
#undef NM_QR
#pragma unroll
    for (int i = 0; i < 16; ++i) out[i] = x[i] + s[i];
}

// Wave-uniform generator.  The 64 lanes produce 64 consecutive blocks (1024 words) into an LDS cache; every
// lane then reads the same word (LDS broadcast), so all scalar control flow driven by the stream is uniform.
constexpr int RNG_CACHE_WORDS = 512;    // 32 ChaCha blocks, produced by lanes 0..31
struct DevRng {
    const uint32_t* key;   // 8 words (LDS)
    uint64_t pos;      // next u32 word of the stream
    uint64_t base;     // stream position of cache word 0 (multiple of 16)
    uint32_t* cache;   // LDS, at least RNG_CACHE_WORDS words
    uint32_t cap;      // valid words in the cache (RNG_CACHE_WORDS after a refill; more while a bulk fill lends its buffer)
    bool one_wave;     // the block is one wavefront (read once: `blockDim.x` is a load from the dispatch packet, and a refill that asks
                       // for it again waits for every outstanding memory operation of the wavefront)

    NM_DEV void init(const uint32_t* k, uint64_t p, uint32_t* lds) {
        one_wave = NM_ONE_WAVE_BLOCK;
        key = k;
        pos = p;
        base = p + 16;   // invalid: forces a refill on first use
        cache = lds;
        cap = RNG_CACHE_WORDS;
    }
    NM_DEV bool has(uint64_t nwords) const { return pos >= base && (pos - base) + nwords <= (uint64_t)cap; }
    NM_DEV void refill() {
        block_sync(one_wave);
        base = pos & ~15ull;
        cap = RNG_CACHE_WORDS;
        if (tid() < RNG_CACHE_WORDS / 16) {
            uint32_t out[16];
            chacha8_block(key, (base >> 4) + (uint64_t)tid(), 0ull, out);
            uint4* dst = reinterpret_cast<uint4*>(cache + tid() * 16);
            dst[0] = make_uint4(out[0], out[1], out[2], out[3]);
            dst[1] = make_uint4(out[4], out[5], out[6], out[7]);
            dst[2] = make_uint4(out[8], out[9], out[10], out[11]);
            dst[3] = make_uint4(out[12], out[13], out[14], out[15]);
        }
        block_sync(one_wave);
    }
    NM_DEV uint32_t next_u32() {
        if (!has(1)) refill();
        uint32_t w = cache[pos - base];
        pos += 1;
        return w;
    }
    NM_DEV uint64_t next_u64() {
        if (!has(2)) refill();
        uint64_t lo = cache[pos - base], hi = cache[pos - base + 1];
        pos += 2;
        return (hi << 32) | lo;
    }
    // StandardUniform: bool = sign bit of a u32; f64 = 53 high bits of a u64
    NM_DEV bool random_bool_std() { return (int32_t)next_u32() < 0; }
    NM_DEV double random_f64() { return (double)(next_u64() >> 11) * (1.0 / 9007199254740992.0); }
    // Bernoulli (rng.random_bool(p), reference src/nuts.rs:200): 1/0, or -1 for p outside [0,1] (reference panics)
    NM_DEV int random_bool(double p) {
        if (!(p >= 0.0 && p < 1.0)) return p == 1.0 ? 1 : -1;
        uint64_t p_int = (uint64_t)(p * 18446744073709551616.0);
        return next_u64() < p_int ? 1 : 0;
    }
};

struct ZigTables { const double* x; const double* f; };   // 257 entries each, HBM (L1/L2 resident)
constexpr double ZIG_R = 3.654152885361008796;   // rand_distr's ZIG_NORM_R; the tables are the fixed constants of zig_tables.hpp

// Tail of rand_distr's ziggurat loop for ONE sample whose first fast-path test failed; uniform over the wave.
NM_DEV double normal_slow_path(DevRng& rng, uint64_t bits, ZigTables T) {
    for (;;) {
        int i = (int)(bits & 0xff);
        double u = u2d((bits >> 12) | 0x4000000000000000ull) - 3.0;
        double x = u * T.x[i];
        if (__builtin_fabs(x) < T.x[i + 1]) return x;
        if (i == 0) {
            double xx = 1.0, yy = 0.0;
            while (-2.0 * yy < xx * xx) {
                double a = u2d((rng.next_u64() >> 12) | 0x3ff0000000000000ull) - (1.0 - 2.220446049250313e-16 / 2.0);
                double b = u2d((rng.next_u64() >> 12) | 0x3ff0000000000000ull) - (1.0 - 2.220446049250313e-16 / 2.0);
                xx = ulog(a) / ZIG_R;
                yy = ulog(b);
            }
            return u < 0.0 ? xx - ZIG_R : ZIG_R - xx;
        }
        if (T.f[i + 1] + (T.f[i] - T.f[i + 1]) * rng.random_f64() < uexp(-x * x / 2.0)) return x;
        bits = rng.next_u64();
    }
}

// `count` StandardNormal variates of the stream, in stream order, into stage[0..count) (LDS or global scratch).
// 64 samples are attempted per pass from 64 consecutive u64 of the stream (one per lane); the samples before
// the first lane whose fast-path test fails are exactly what the sequential algorithm would have produced;
// that lane's sample is finished on the slow path and the pass restarts behind it.
NM_DEV void fill_standard_normals(DevRng& rng, double* stage, int count, ZigTables T) {
    const int lane = lane_id();
    int i = 0;
    while (i < count) {
        if (!rng.has(128)) rng.refill();
        const int nvalid = (count - i) < 64 ? (count - i) : 64;
        const uint32_t off = (uint32_t)(rng.pos - rng.base) + 2u * (uint32_t)lane;
        uint64_t bits = ((uint64_t)rng.cache[off + 1] << 32) | rng.cache[off];
        int zi = (int)(bits & 0xff);
        double u = u2d((bits >> 12) | 0x4000000000000000ull) - 3.0;
        double x = u * T.x[zi];
        bool ok = __builtin_fabs(x) < T.x[zi + 1];
        uint64_t fail = __ballot(lane < nvalid && !ok);
        int nacc = fail ? (int)__builtin_ctzll(fail) : nvalid;
        if (lane < nacc) stage[i + lane] = x;
        rng.pos += 2ull * (uint64_t)nacc;
        i += nacc;
        if (fail) {
            uint64_t fbits = wave_bcast_u64(bits, nacc);
            rng.pos += 2;
            double xs = normal_slow_path(rng, fbits, T);
            if (lane == 0) stage[i] = xs;
            i += 1;
        }
    }
    chain_sync();
}

// ---------------------------------------------------------------------------------------------------------------
// Bulk variant for the momentum refresh.  Everything the stream consumes inside `array_gaussian` is a whole u64, so
// from the start position the stream is a fixed grid of u64 "cells"; each cell is either the fast-path candidate of
// one sample or is swallowed by the slow path of an earlier failing cell.  Instead of re-aligning 64 lanes after every
// rejection (a chain of ~28 dependent passes for 1024 samples), one chunk does
//   1. all ChaCha blocks of up to 64*P cells into LDS (all threads),
//   2. the fast-path test of every cell, P independent passes (table gathers all in flight), x kept in registers,
//      rejection masks by ballot,
//   3. a wave-uniform walk over the ~1.2 % rejected cells in stream order: the slow path of each (reading the stream
//      through the same LDS words) tells how many cells it swallowed,
//   4. a scatter of every surviving x to its sample index (cell index minus the cells swallowed before it).
// Sample values and the final stream position are those of the sequential algorithm, bit for bit.
// ---------------------------------------------------------------------------------------------------------------
constexpr int ZIG_FMAX = 64;      // slow-path events handled per chunk (a chunk that meets more simply ends early)
NM_DEV int uniform_i32(int x) { return __builtin_amdgcn_readfirstlane(x); }
// small wave-uniform tables kept one entry per lane (a VGPR read with a scalar lane index costs a few cycles; the
// same table in LDS would put a ~100-cycle round trip into every step of the scalar walk)
NM_DEV int lane_get(int v, int idx) { return __builtin_amdgcn_readlane(v, idx); }
NM_DEV double lane_get_f64(double v, int idx) {
    const uint64_t b = d2u(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, idx);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), idx);
    return u2d(((uint64_t)hi << 32) | lo);
}

// `count` StandardNormal variates in stream order into samp[0..count) (LDS).  wbuf: LDS, 128*P + 16 words; P <= 64.
#ifndef NM_PROF
#define NM_PROF 0
#endif
#if NM_PROF
#define NM_MARK_F(slot)                                                           \
    {                                                                             \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();             \
        if (blockIdx.x == 0 && threadIdx.x == 0) (void)__hip_atomic_fetch_add(&prof[slot], now_ - prof_t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     \
        prof_t = now_;                                                            \
    }
#else
#define NM_MARK_F(slot)
#endif
template <int P>
NM_DEV void fill_standard_normals_bulk(DevRng& rng, uint32_t* wbuf, double* samp, int count, ZigTables T, int nthreads,
                                       unsigned long long* prof, unsigned long long& prof_t) {
    static_assert(P <= 64, "rejection masks are kept one per lane");
    const int lane = lane_id();
    uint32_t* const small_cache = rng.cache;
    int produced = 0;
    block_sync(nthreads == 64);
    while (produced < count) {
        const int need = count - produced;
        // cells of this chunk: the samples wanted plus room for the cells the slow paths will swallow (~2 % of them)
        int nc = need + ((need >> 4) > 8 ? (need >> 4) : 8);
        if (nc > 64 * P) nc = 64 * P;
        const uint64_t pos0 = rng.pos;
        const uint64_t b0 = pos0 >> 4;
        const int nb = (int)(((pos0 + 2ull * (uint64_t)nc - 1ull) >> 4) - b0) + 1;
        uint64_t* const lmask = reinterpret_cast<uint64_t*>(small_cache + 256);    // [2][P] bit masks, see step 3
        if (lane < 2 * P) lmask[lane] = 0ull;
        for (int b = tid(); b < nb; b += nthreads) {          // 1. the words of all cells
            uint32_t out[16];
            chacha8_block(rng.key, b0 + (uint64_t)b, 0ull, out);
            uint4* dst = reinterpret_cast<uint4*>(wbuf + b * 16);
            dst[0] = make_uint4(out[0], out[1], out[2], out[3]);
            dst[1] = make_uint4(out[4], out[5], out[6], out[7]);
            dst[2] = make_uint4(out[8], out[9], out[10], out[11]);
            dst[3] = make_uint4(out[12], out[13], out[14], out[15]);
        }
        block_sync(nthreads == 64);
        NM_MARK_F(8)
        rng.cache = wbuf; rng.base = b0 << 4; rng.cap = (uint32_t)nb * 16u;     // the slow path reads the same words
        const int w0 = (int)(pos0 - rng.base);
        // 2. fast-path test of every cell.  No branch between the passes (cells past nc hold stale words: tested,
        // ignored), so all table gathers are in flight together.  Lane p keeps the rejection mask of pass p.
        double xr[P];
        int* const flist = reinterpret_cast<int*>(small_cache);   // rejected cells in stream order (+ 64 dummy slots)
        int nrej = 0;
        {
            uint64_t cb[P];
            double tx[P], tn[P];
#pragma unroll
            for (int p = 0; p < P; ++p) {                         // all LDS reads, then all table gathers, then the math:
                const int c = 64 * p + lane;                      // no branch in between, so the loads overlap
                cb[p] = ((uint64_t)wbuf[w0 + 2 * c + 1] << 32) | wbuf[w0 + 2 * c];
            }
#pragma unroll
            for (int p = 0; p < P; ++p) {
                const int zi = (int)(cb[p] & 0xff);
                tx[p] = T.x[zi];
                tn[p] = T.x[zi + 1];
            }
#pragma unroll
            for (int p = 0; p < P; ++p) {
                const int c = 64 * p + lane;
                const double u = u2d((cb[p] >> 12) | 0x4000000000000000ull) - 3.0;
                const double x = u * tx[p];
                const bool rej = !(__builtin_fabs(x) < tn[p]) && c < nc;
                xr[p] = x;
                const uint64_t fail = __ballot(rej);
                const int rank = nrej + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(fail >> 32),
                                                                       __builtin_amdgcn_mbcnt_lo((uint32_t)fail, 0u));
                flist[(rej && rank < ZIG_FMAX) ? rank : ZIG_FMAX + lane] = c;   // unconditional store
                nrej += (int)__builtin_popcountll(fail);
            }
        }
        // 2b. the slow path of every rejected cell at once, one lane each.  What a rejected cell f yields depends
        // only on the cells behind it: the wedge test uses cell f+1 (accept: f's own x, 1 cell swallowed); if that
        // rejects, cell f+2 is the next candidate (fast-path accept: its x, 2 cells swallowed).  Base-layer tails,
        // a second rejection in a row and cells at the end of the chunk are left to the scalar routine in step 3.
        const int nlist = nrej < ZIG_FMAX ? nrej : ZIG_FMAX;
        int fcell, sp_r;
        double sp_x;
        bool sp_scalar;
        fcell = 0; sp_r = 0; sp_x = 0.0; sp_scalar = false;
        if (nlist > 0) {
            fcell = lane < nlist ? flist[lane] : 0;
            const int wa = w0 + 2 * fcell;
            const uint64_t bits0 = ((uint64_t)wbuf[wa + 1] << 32) | wbuf[wa];
            const uint64_t bits1 = ((uint64_t)wbuf[wa + 3] << 32) | wbuf[wa + 2];
            const uint64_t bits2 = ((uint64_t)wbuf[wa + 5] << 32) | wbuf[wa + 4];
            const int i0 = (int)(bits0 & 0xff);
            const double u0 = u2d((bits0 >> 12) | 0x4000000000000000ull) - 3.0;
            const double x0 = u0 * T.x[i0];
            const double u01 = (double)(bits1 >> 11) * (1.0 / 9007199254740992.0);
            const bool wedge = T.f[i0 + 1] + (T.f[i0] - T.f[i0 + 1]) * u01 < dexp(-x0 * x0 / 2.0);
            const int i2 = (int)(bits2 & 0xff);
            const double u2 = u2d((bits2 >> 12) | 0x4000000000000000ull) - 3.0;
            const double x2 = u2 * T.x[i2];
            const bool ok2 = __builtin_fabs(x2) < T.x[i2 + 1];
            sp_r = wedge ? 1 : 2;
            sp_x = wedge ? x0 : x2;
            sp_scalar = i0 == 0 || fcell + 2 >= nc || (!wedge && !ok2);
        }
        NM_MARK_F(9)
        NM_MARK_F(12)
        // 3. walk over the rejected cells in stream order: next unconsumed cell `cur`, samples so far `j`.  Lane q keeps
        // two 64-bit masks over the cells of pass q: `sw` = swallowed by a slow path, `nk` = swallowed or rejected (no
        // fast-path sample).  Lane e keeps event e: its sample index and value.
        int cur = 0, j = 0, nf = 0;
        uint64_t sw = 0, nk = 0;
        int flj = 0;
        double flx = 0.0;
        bool open = true;
        const uint64_t scalar_mask = __ballot(sp_scalar);
        // Common case (97 %): every rejected cell is an event of its own — none needs the scalar routine, none lies in
        // the cells an earlier one swallows, the list is complete.  Then nothing is sequential: the cells swallowed
        // before event g are a prefix sum over the lanes (r is 1 or 2: two ballots), its sample index follows, and the
        // per-pass masks are OR-ed together in LDS by the event lanes.
        const int rr = lane < nlist ? sp_r : 0;
        const uint64_t m1 = __ballot(rr == 1), m2 = __ballot(rr == 2);
        const int s_before = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m1, 0u)) +
                             2 * (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m2 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m2, 0u));
        const int prev_end = __builtin_amdgcn_update_dpp(-1, fcell + rr, 0x138, 0xf, 0xf, false);     // lane g-1's last swallowed cell
        const bool chained = lane >= 1 && lane < nlist && fcell <= prev_end;
        const bool fast = scalar_mask == 0ull && nrej <= nlist && __ballot(chained) == 0ull;
        bool ev_active = false;
        int ev_j = 0;
        if (fast) {
            ev_j = fcell - s_before;                          // sample index of event g inside the chunk
            ev_active = lane < nlist && ev_j < need;          // events behind the last sample wanted are never reached
            const uint64_t am = __ballot(ev_active);
            const int s_act = (int)__builtin_popcountll(m1 & am) + 2 * (int)__builtin_popcountll(m2 & am);
            j = (nc - s_act) < need ? (nc - s_act) : need;
            cur = j + s_act;
            if (ev_active) {                                  // cells [f, f+1+r) no fast-path sample, [f+1, f+1+r) swallowed
                const int q = fcell >> 6, off = fcell & 63;
                const uint64_t nk0 = ((1ull << (rr + 1)) - 1ull) << off;
                const uint64_t sw0 = off == 63 ? 0ull : (((1ull << rr) - 1ull) << (off + 1));
                (void)__hip_atomic_fetch_or(&lmask[q], sw0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                (void)__hip_atomic_fetch_or(&lmask[P + q], nk0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const int spill = off + rr + 1 - 64;          // cells of the range that belong to the next pass
                if (spill > 0) {
                    const uint64_t nx = (1ull << spill) - 1ull;
                    (void)__hip_atomic_fetch_or(&lmask[q + 1], nx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    (void)__hip_atomic_fetch_or(&lmask[P + q + 1], nx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
            sw = lane < P ? lmask[lane] : 0ull;
            nk = lane < P ? lmask[P + lane] : 0ull;
            open = false;
        }
        for (int g = 0; g < nlist && !fast; ++g) {
            const int f = lane_get(fcell, g);
            if (f < cur) continue;                            // swallowed by an earlier slow path
            if (j + (f - cur) >= need) { cur += need - j; j = need; open = false; break; }
            j += f - cur;
            int r;
            double xs;
            if ((scalar_mask >> g) & 1ull) {
                const uint32_t lo = (uint32_t)uniform_i32((int)wbuf[w0 + 2 * f]);
                const uint32_t hi = (uint32_t)uniform_i32((int)wbuf[w0 + 2 * f + 1]);
                rng.pos = pos0 + 2ull * (uint64_t)(f + 1);
                xs = normal_slow_path(rng, ((uint64_t)hi << 32) | lo, T);
                r = (int)((rng.pos - pos0) >> 1) - (f + 1);
            } else {
                r = lane_get(sp_r, g);
                xs = lane_get_f64(sp_x, g);
            }
            // cells [f, f+1+r) give no fast-path sample, [f+1, f+1+r) are swallowed.  The bit ranges are scalar work;
            // only the update of the owning lane's masks touches vector registers.
            for (int a = f, end = (f + 1 + r < 64 * P ? f + 1 + r : 64 * P); a < end;) {
                const int q = a >> 6, off = a & 63;
                const int take = (end - a) < (64 - off) ? (end - a) : (64 - off);
                const uint64_t bits = (take >= 64 ? ~0ull : ((1ull << take) - 1ull)) << off;
                const uint64_t sbits = a == f ? (bits & ~(1ull << off)) : bits;
                nk |= lane == q ? bits : 0ull;
                sw |= lane == q ? sbits : 0ull;
                a += take;
            }
            flj = lane == nf ? j : flj;
            flx = lane == nf ? xs : flx;
            nf += 1;
            j += 1;
            cur = f + 1 + r;
            if (j >= need || cur >= nc) { open = false; break; }
        }
        if (nrej > nlist && !fast) open = false;              // more rejections than list slots: the rest next chunk
        if (open && cur < nc) {                               // tail after the last event
            const int take = (nc - cur) < (need - j) ? (nc - cur) : (need - j);
            j += take; cur += take;
        }
        rng.pos = pos0 + 2ull * (uint64_t)cur;
        NM_MARK_F(10)
        // 4. scatter, branch-free: a surviving cell's sample index is its index minus the swallowed cells before it;
        // lanes with nothing to store write to a dummy slot.  Then the slow-path samples, one lane per event.
        {
            double* const dummy = reinterpret_cast<double*>(small_cache) + 64 + lane;
            const int swlo = (int)(uint32_t)sw, swhi = (int)(uint32_t)(sw >> 32);
            const int nklo = (int)(uint32_t)nk, nkhi = (int)(uint32_t)(nk >> 32);
            int base = 0;
#pragma unroll
            for (int p = 0; p < P; ++p) {
                const uint32_t s_lo = (uint32_t)lane_get(swlo, p), s_hi = (uint32_t)lane_get(swhi, p);
                const uint64_t n_p = ((uint64_t)(uint32_t)lane_get(nkhi, p) << 32) | (uint32_t)lane_get(nklo, p);
                const int c = 64 * p + lane;
                const int below = (int)__builtin_amdgcn_mbcnt_hi(s_hi, __builtin_amdgcn_mbcnt_lo(s_lo, 0u));
                const bool keep = !((n_p >> lane) & 1ull) && c < cur;
                double* dst = keep ? samp + (produced + c - base - below) : dummy;
                *dst = xr[p];
                base += (int)__builtin_popcount(s_lo) + (int)__builtin_popcount(s_hi);
            }
            double* dst = fast ? (ev_active ? samp + (produced + ev_j) : dummy) : (lane < nf ? samp + (produced + flj) : dummy);
            *dst = fast ? sp_x : flx;
        }
        produced += j;
        block_sync(nthreads == 64);
        NM_MARK_F(11)
    }
    // Hand the already generated words behind the final position to the small cache: the tree's direction bits
    // and Bernoulli draws start there, and a copy is ~10x cheaper than regenerating 32 blocks.
    const uint64_t nbase = rng.pos & ~15ull;
    const uint64_t gen_end = rng.base + (uint64_t)rng.cap;
    if (nbase >= rng.base && gen_end >= nbase + 16) {
        const uint32_t avail = gen_end - nbase > (uint64_t)RNG_CACHE_WORDS ? (uint32_t)RNG_CACHE_WORDS : (uint32_t)(gen_end - nbase);
        const uint32_t* src = wbuf + (nbase - rng.base);
        constexpr int PER_LANE = RNG_CACHE_WORDS / 64;      // 8 words = two 16-byte moves
        static_assert(PER_LANE == 8, "copy below moves 8 words per lane");
        const uint32_t o = (uint32_t)lane * PER_LANE;
        const uint4 q0 = *reinterpret_cast<const uint4*>(src + o), q1 = *reinterpret_cast<const uint4*>(src + o + 4);
        block_sync(nthreads == 64);                                      // every wave has read its words before any wave stores
        *reinterpret_cast<uint4*>(small_cache + o) = q0;
        *reinterpret_cast<uint4*>(small_cache + o + 4) = q1;
        rng.cache = small_cache; rng.base = nbase; rng.cap = avail;
    } else {
        rng.cache = small_cache;
        rng.base = rng.pos + 16;  // nothing usable: refill on first use
        rng.cap = RNG_CACHE_WORDS;
    }
}

}  // namespace nm
