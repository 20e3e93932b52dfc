// packed_sum_check.hip — wave_sum_packed<N> (csrc/dev_math.hpp, round 5) against wave_sum per value, bit for bit, N = 2 .. 8, random data
// with mixed magnitudes / signs / zeros; every lane of the packed result must hold the total of some value < N.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I nuts_rs_amd/csrc -I include tools/probes/packed_sum_check.hip -o tools/probes/packed_sum_check
#include "dev_math.hpp"
#include <cstdio>
using namespace nm;
__device__ uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x; }
template <int N>
__device__ int one(uint64_t seed) {
    double v[N], ref[N];
    for (int i = 0; i < N; ++i) {
        const uint64_t r = mix(seed * 977 + i * 64 + threadIdx.x);
        const int e = (int)((r >> 52) % 40) - 20;
        double x = ((double)(r >> 11) * (1.0 / 9007199254740992.0) - 0.5) * __builtin_ldexp(1.0, e);
        if ((r & 31) == 0) x = 0.0;
        v[i] = x; ref[i] = wave_sum(x);
    }
    const double pk = wave_sum_packed<N>(v);
    int bad = 0;
    for (int i = 0; i < N; ++i) if (d2u(readlane_f64(pk, packed_lane(i))) != d2u(ref[i])) bad |= 1 << i;
    bool some = false;
    for (int i = 0; i < N; ++i) some = some || d2u(pk) == d2u(ref[i]);
    if (__ballot(!some) != 0ull) bad |= 1 << 8;
    return bad;
}
__global__ void k(int* out, uint64_t seed) {
    int bad[7];
    bad[0] = one<2>(seed); bad[1] = one<3>(seed + 1); bad[2] = one<4>(seed + 2); bad[3] = one<5>(seed + 3); bad[4] = one<6>(seed + 4); bad[5] = one<7>(seed + 5); bad[6] = one<8>(seed + 6);
    if (threadIdx.x == 0) for (int i = 0; i < 7; ++i) if (bad[i]) atomicOr(&out[i], bad[i]);
}
int main() {
    int* d; int h[7];
    (void)hipMalloc(&d, 28); (void)hipMemset(d, 0, 28);
    for (int it = 0; it < 200; ++it) hipLaunchKernelGGL(k, dim3(256), dim3(64), 0, 0, d, 12345ull + 1000ull * it + 0);
    (void)hipDeviceSynchronize(); (void)hipMemcpy(h, d, 28, hipMemcpyDeviceToHost);
    int tot = 0;
    for (int i = 0; i < 7; ++i) { printf("N = %d: mismatch mask %x\n", i + 2, h[i]); tot |= h[i]; }
    printf("%s\n", tot ? "FAILED" : "all totals bit-identical to wave_sum");
    return tot ? 1 : 0;
}
