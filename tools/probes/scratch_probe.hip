// scratch_probe.hip — does private (scratch) memory of concurrently resident waves alias when a dispatch has more than
// 1024 waves?  Every wave fills a private array (forced into scratch by dynamic indexing) with a pattern unique to
// (block, lane, i), spins a while, and checks it.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/scratch_probe.hip -o /tmp/scratch_probe && /tmp/scratch_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int N = 96;     // doubles per lane: 768 B of scratch per lane
__global__ __launch_bounds__(64) void probe(unsigned long long* bad, int spin, const int* perm) {
    double a[N];
    const unsigned long long tag = ((unsigned long long)blockIdx.x << 20) | ((unsigned long long)threadIdx.x << 8);
    for (int i = 0; i < N; ++i) a[perm[i]] = __longlong_as_double((long long)(0x3ff0000000000000ull | tag | (unsigned)i));
    long long t0 = clock64();
    while (clock64() - t0 < spin) {}
    unsigned long long errs = 0;
    for (int i = 0; i < N; ++i)
        if (__double_as_longlong(a[perm[i]]) != (long long)(0x3ff0000000000000ull | tag | (unsigned)i)) errs++;
    if (errs) atomicAdd(&bad[blockIdx.x >= 1024 ? 1 : 0], errs);
}
int main() {
    unsigned long long* d_bad; int* d_perm;
    hipMalloc(&d_bad, 16); hipMalloc(&d_perm, N * sizeof(int));
    std::vector<int> perm(N);
    for (int i = 0; i < N; ++i) perm[i] = (i * 37) % N;
    hipMemcpy(d_perm, perm.data(), N * sizeof(int), hipMemcpyHostToDevice);
    for (int grid : {512, 1024, 1100, 2048, 4096, 8192}) {
        hipMemset(d_bad, 0, 16);
        hipLaunchKernelGGL(probe, dim3(grid), dim3(64), 0, 0, d_bad, 2000000, d_perm);
        hipDeviceSynchronize();
        unsigned long long h[2];
        hipMemcpy(h, d_bad, 16, hipMemcpyDeviceToHost);
        printf("grid %5d: corrupted private words in blocks < 1024: %llu, in blocks >= 1024: %llu\n", grid, h[0], h[1]);
    }
    return 0;
}
