// mfma_f64_rate.hip — what does v_mfma_f64_16x16x4_f64 sustain on this GPU?  (round 4: the lockstep kernel's products run at
// ~112 shader cycles per instruction per SIMD with four dependent chains per SIMD, loads and LDS reads removed.)
// Each wavefront issues ITER x CHAINS instructions: CHAINS independent accumulators, each a dependent chain.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_f64_rate.hip -o tools/probes/mfma_f64_rate && tools/probes/mfma_f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));
template <int CHAINS>
__global__ void k(double* out, int iters, double a0, double b0) {
    v4d acc[CHAINS];
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) acc[c] = v4d{0.0, 0.0, 0.0, 0.0};
    double a = a0 + threadIdx.x * 1e-9, b = b0 - threadIdx.x * 1e-9;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0.0;
#pragma unroll
    for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) out[gridDim.x * blockDim.x] = (double)(t1 - t0);
}
template <int CHAINS>
void run(int waves_per_simd, int iters) {
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const int threads = 64 * 4 * waves_per_simd;     // one block per CU, waves_per_simd waves on each of its 4 SIMDs
    double* out;
    hipMalloc(&out, ((size_t)cus * threads + 1) * sizeof(double));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<CHAINS>, dim3(cus), dim3(threads), 0, 0, out, 10, 1.0, 1.0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<CHAINS>, dim3(cus), dim3(threads), 0, 0, out, iters, 1.0, 1.0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    double ticks = 0;
    hipMemcpy(&ticks, out + (size_t)cus * threads, sizeof(double), hipMemcpyDeviceToHost);
    const double n_per_simd = (double)iters * CHAINS * waves_per_simd;
    const double tflops = n_per_simd * 4.0 * cus * 2048.0 / (ms * 1e-3) / 1e12;
    printf("waves/SIMD %d chains/wave %d: %.3f ms, %.1f ns per instruction per SIMD (= %.1f cycles at 2.4 GHz), %.1f s_memtime ticks per instruction per SIMD, %.1f TFLOP/s\n",
           waves_per_simd, CHAINS, ms, ms * 1e6 / n_per_simd, ms * 1e6 / n_per_simd * 2.4, ticks / n_per_simd, tflops);
    hipFree(out);
}
int main() {
    const int iters = 20000;
    for (int w : {1, 2, 4, 8}) { run<1>(w, iters); run<2>(w, iters); run<4>(w, iters); }
    return 0;
}
