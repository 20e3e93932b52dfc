// sl_math_device_check.hip — ON THE DEVICE: the branch-free exp / ln / ln_1p of csrc/dev_math.hpp against the general-purpose forms of
// rounds 1-4, bit for bit, over random bit patterns, dense ranges around every special case, and merge_math_impl against the original
// merge sequence.  (tests/cpp/merge_math_check.hip is the same comparison on the host.)
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I nuts_rs_amd/csrc -I include tools/probes/sl_math_device_check.hip -o tools/probes/sl_math_device_check
#include "dev_math.hpp"
#include <cstdio>
using namespace nm;
__device__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }
__device__ bool same(double a, double b) { return d2u(a) == d2u(b) || (a != a && b != b); }
__global__ void check(unsigned long long* bad, unsigned long long* first, uint64_t seed, int kind) {
    dm_init_lds();
    const uint64_t id = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (int it = 0; it < 64; ++it) {
        const uint64_t r = mix(seed + id * 64 + it);
        double x;
        switch (kind) {
        case 0: x = u2d(r); break;                                                        // any bit pattern
        case 1: x = ((double)(r >> 11) * (1.0 / 9007199254740992.0) - 0.5) * 1600.0; break;   // exp's whole range
        case 2: x = (double)(r >> 11) * (1.0 / 9007199254740992.0); break;                 // [0, 1)
        case 3: x = u2d(r & 0x000fffffffffffffull); break;                                // sub-normals
        case 4: x = u2d((r & 0x800fffffffffffffull) | ((uint64_t)(1020 + (r >> 60) % 8) << 52)); break;   // around 1 / the sqrt 2 boundary, both signs
        default: x = ((double)(r >> 11) * (1.0 / 9007199254740992.0) - 0.5) * 80.0; break;
        }
        int w = 0;
        if (!same(exp_sl(x), dexp_branchy<false>(x))) w |= 1;
        if (!same(log_sl(x), dlog_branchy<false>(x))) w |= 2;
        if (!same(log1p_sl(x), dlog1p_branchy<false>(x))) w |= 4;
        if (x >= 0.0 && x <= 1.0 && !same(log1p_unit(x), dlog1p_branchy<false>(x))) w |= 8;
        // the merge: a = x, b = a second value near or far
        const uint64_t r2 = mix(r);
        const double y = (it & 1) ? x + ((double)(r2 >> 11) * (1.0 / 9007199254740992.0) - 0.5) * 1e-3 : ((double)(r2 >> 11) * (1.0 / 9007199254740992.0) - 0.5) * 80.0;
        if (kind >= 5 || kind == 1) {
            const bool is_main = (r2 >> 7) & 1;
            const MergeOut o = merge_math_impl(x, y, is_main ? 1u : 0u, (uint32_t)r2, (uint32_t)(r2 >> 32));
            // the original sequence
            double total;
            if (x == y) total = x + dlog_branchy<false>(2.0);
            else { const double d = x - y; total = d > 0. ? x + dlog1p_branchy<false>(dexp_branchy<false>(-d)) : (d < 0. ? y + dlog1p_branchy<false>(dexp_branchy<false>(d)) : d); }
            const double self = is_main ? x : total;
            uint32_t fl;
            if (y >= self) fl = 1;
            else { const double p = dexp_branchy<false>(y - self);
                   if (!(p >= 0.0 && p < 1.0)) fl = p == 1.0 ? 1u : 4u;
                   else fl = 2u | (r2 < (uint64_t)(p * 18446744073709551616.0) ? 1u : 0u); }
            if (!same(o.total, total) || o.flags != fl) w |= 16;
        }
        if (w) { atomicAdd(bad, 1ull); atomicCAS(first, 0ull, d2u(x) ^ ((unsigned long long)w << 60)); }
    }
}
int main() {
    unsigned long long *bad, *first, hb = 0, hf = 0, total = 0;
    (void)hipMalloc(&bad, 8); (void)hipMalloc(&first, 8);
    for (int kind = 0; kind < 7; ++kind) {
        (void)hipMemset(bad, 0, 8); (void)hipMemset(first, 0, 8);
        hipLaunchKernelGGL(check, dim3(8192), dim3(256), 0, 0, bad, first, 0x1234567ull * (kind + 1), kind);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(&hf, first, 8, hipMemcpyDeviceToHost);
        printf("kind %d: %llu operands, %llu mismatches (first: %016llx)\n", kind, 8192ull * 256 * 64, hb, hf);
        total += hb;
    }
    printf("TOTAL mismatches %llu\n", total);
    return total ? 1 : 0;
}
