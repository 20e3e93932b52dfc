// exchange_probe.hip — what one block-sum exchange between the members of a wide chain costs (dev_math.hpp
// Reducer::cluster_combine), in isolation: pairs / quads / ... of 256-thread blocks do nothing but exchanges.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I nuts_rs_amd/csrc [-DNM_CLUSTER_COUNTED] tools/probes/exchange_probe.hip -o /tmp/exchange_probe
#define NM_CLUSTER_MODE 1
#include "dev_math.hpp"
#include <cstdio>
#include <vector>
using namespace nm;
template <int N>
__global__ __launch_bounds__(256) void probe(unsigned long long* box, unsigned long long* cnt, int k, int iters, int same, double* out, long long* cycles) {
    __shared__ double red[2 * RED_MAX_VALUES * 4 + RED_MAX_VALUES + 1 + 2 * RED_MAX_VALUES * CL_MAX_MEMBERS];
    const unsigned cl_member = (blockIdx.x / 8u) % (unsigned)k;
    const unsigned long long cl_id = (unsigned long long)(blockIdx.x / (8u * k)) * 8u + blockIdx.x % 8u;
    ClusterLink L;
    L.box = box + cl_id * (unsigned long long)CL_BOX_WORDS * k * RED_MAX_VALUES; L.cnt = cnt + cl_id;
    L.k = k; L.member = (int)cl_member; L.epoch = 0; L.same_xcd = 0; L.dead = 0;
    Reducer<4> r;
    r.init(red);
    r.cl = L;
    double v0[2] = {(double)xcc_id(), 1.0};
    r.template cluster_combine<2>(v0);              // warm-up with the safe protocol
    r.cl.same_xcd = same;
    double acc = 0.0;
    const long long t0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
        double v[N];
#pragma unroll
        for (int j = 0; j < N; ++j) v[j] = (double)(cl_member + 1) * (double)(i + j);
        r.template cluster_combine<N>(v);
        acc += v[0];
    }
    const long long t1 = wall_clock64();
    if (threadIdx.x == 0) { out[blockIdx.x] = acc; cycles[blockIdx.x] = t1 - t0; if (r.cl.dead) out[blockIdx.x] = -1.0; }
}
int main() {
    const int iters = 20000;
    for (int k : {2, 4, 16}) for (int same : {1, 0}) for (int grid : {8 * k, 256 / (8 * k) * 8 * k}) {
        unsigned long long *box, *cnt; double* out; long long* cyc;
        const int n_clusters = grid / k;
        hipMalloc(&box, (size_t)n_clusters * CL_BOX_WORDS * k * RED_MAX_VALUES * 8); hipMalloc(&cnt, n_clusters * 8);
        hipMemset(box, 0, (size_t)n_clusters * CL_BOX_WORDS * k * RED_MAX_VALUES * 8); hipMemset(cnt, 0, n_clusters * 8);
        hipMalloc(&out, grid * 8); hipMalloc(&cyc, grid * 8);
        hipLaunchKernelGGL(probe<2>, dim3(grid), dim3(256), 0, 0, box, cnt, k, iters, same, out, cyc);
        hipDeviceSynchronize();
        std::vector<double> o(grid); std::vector<long long> c(grid);
        hipMemcpy(o.data(), out, grid * 8, hipMemcpyDeviceToHost); hipMemcpy(c.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
        double expect = 0.0;
        for (int i = 0; i < iters; ++i) { double t = 0.0; for (int m = 0; m < k; ++m) t = (m == 0 ? 0.0 : t) + (double)(m + 1) * (double)i; expect += t; }
        int bad = 0; long long mx = 0;
        for (int b = 0; b < grid; ++b) { bad += o[b] != expect; mx = c[b] > mx ? c[b] : mx; }
        printf("k=%2d same_xcd=%d grid=%3d: %.3f us per exchange (wall clock 100 MHz), wrong blocks %d\n", k, same, grid, (double)mx / 100.0 / iters, bad);
        hipFree(box); hipFree(cnt); hipFree(out); hipFree(cyc);
    }
    return 0;
}
