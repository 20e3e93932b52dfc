// mfma_f64_probe.hip — what v_mfma_f64_16x16x4_f64 computes, bit for bit (the guide documents layouts, not rounding):
// D = A(16x4) B(4x16) + C against candidate host formulas on operands with wildly different magnitudes.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/mfma_f64_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ void k(const double* A, const double* B, const double* C, double* D, int trials) {
    const int l = threadIdx.x;
    for (int t = 0; t < trials; ++t) {
        const double a = A[t * 64 + (l & 15) * 4 + (l >> 4)];            // A[i = l&15][k = l>>4], stored [16][4]
        const double b = B[t * 64 + (l >> 4) * 16 + (l & 15)];           // B[k = l>>4][j = l&15], stored [4][16]
        v4d c;
        for (int r = 0; r < 4; ++r) c[r] = C[t * 256 + ((l >> 4) + 4 * r) * 16 + (l & 15)];   // row = (l>>4) + 4 r, col = l&15
        v4d d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
        for (int r = 0; r < 4; ++r) D[t * 256 + ((l >> 4) + 4 * r) * 16 + (l & 15)] = d[r];
    }
}
int main() {
    const int T = 2000;
    std::mt19937_64 g(1);
    std::uniform_real_distribution<double> u(-1, 1);
    std::uniform_int_distribution<int> e(-30, 30);
    std::vector<double> A(T * 64), B(T * 64), C(T * 256), D(T * 256);
    for (auto& x : A) x = std::ldexp(u(g), e(g));
    for (auto& x : B) x = std::ldexp(u(g), e(g));
    for (auto& x : C) x = std::ldexp(u(g), e(g));
    double *dA, *dB, *dC, *dD;
    hipMalloc(&dA, A.size() * 8); hipMalloc(&dB, B.size() * 8); hipMalloc(&dC, C.size() * 8); hipMalloc(&dD, D.size() * 8);
    hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice);
    hipMemcpy(dC, C.data(), C.size() * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD, T);
    if (hipMemcpy(D.data(), dD, D.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) { printf("hip error\n"); return 1; }
    long n = 0, m_fwd = 0, m_rev = 0, m_exact = 0, m_pair = 0, m_unfused = 0;
    for (int t = 0; t < T; ++t)
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                const double* a = &A[t * 64 + i * 4];
                double b[4];
                for (int kk = 0; kk < 4; ++kk) b[kk] = B[t * 64 + kk * 16 + j];
                const double c = C[t * 256 + i * 16 + j], d = D[t * 256 + i * 16 + j];
                double f = c; for (int kk = 0; kk < 4; ++kk) f = std::fma(a[kk], b[kk], f);
                double r = c; for (int kk = 3; kk >= 0; --kk) r = std::fma(a[kk], b[kk], r);
                __float128 ex = c; for (int kk = 0; kk < 4; ++kk) ex += (__float128)a[kk] * b[kk];
                const double p = std::fma(a[0], b[0], a[1] * b[1]) + std::fma(a[2], b[2], a[3] * b[3]) + c;
                double uf = c; for (int kk = 0; kk < 4; ++kk) uf = uf + a[kk] * b[kk];
                ++n;
                m_fwd += std::memcmp(&f, &d, 8) == 0; m_rev += std::memcmp(&r, &d, 8) == 0;
                const double exd = (double)ex; m_exact += std::memcmp(&exd, &d, 8) == 0;
                m_pair += std::memcmp(&p, &d, 8) == 0; m_unfused += std::memcmp(&uf, &d, 8) == 0;
            }
    printf("{\"outputs\": %ld, \"fma_chain_k_ascending\": %ld, \"fma_chain_k_descending\": %ld, \"exact_sum_one_rounding\": %ld, \"pairwise\": %ld, \"unfused_sequential\": %ld}\n",
           n, m_fwd, m_rev, m_exact, m_pair, m_unfused);
    return 0;
}
