// ubench_latency.hip — single-wavefront instruction latencies on gfx950 (round 5): what does a DEPENDENT chain cost per
// instruction against the issue rate of independent ones?  One wavefront on one SIMD, s_memtime around N instructions.
// hipcc --offload-arch=gfx950 -O3 tools/probes/ubench_latency.hip -o /tmp/ubench && /tmp/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define N 256
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)
#define REP256(x) REP64(x) REP64(x) REP64(x) REP64(x)
__device__ __forceinline__ uint64_t now() { uint64_t t; asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; }

__global__ void k_fma_dep(double* out, uint64_t* cyc, double a, double b) {
    double x = a + threadIdx.x;
    uint64_t t0 = now();
    REP256(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));)
    uint64_t t1 = now();
    out[threadIdx.x] = x; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_fma_ind(double* out, uint64_t* cyc, double a, double b) {
    double x0 = a + threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    uint64_t t0 = now();
    REP64(asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b));)
    uint64_t t1 = now();
    out[threadIdx.x] = x0 + x1 + x2 + x3; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_fma_ind2(double* out, uint64_t* cyc, double a, double b) {
    double x0 = a + threadIdx.x, x1 = x0 + 1;
    uint64_t t0 = now();
    REP64(asm volatile("v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3" : "+v"(x0), "+v"(x1) : "v"(a), "v"(b));)
    uint64_t t1 = now();
    out[threadIdx.x] = x0 + x1; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_add_dep(double* out, uint64_t* cyc, double a, double b) {
    double x = a + threadIdx.x;
    uint64_t t0 = now();
    REP256(asm volatile("v_add_f64 %0, %0, %1" : "+v"(x) : "v"(a));)
    uint64_t t1 = now();
    out[threadIdx.x] = x; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_mul_dep(double* out, uint64_t* cyc, double a, double b) {
    double x = a + threadIdx.x;
    uint64_t t0 = now();
    REP256(asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(a));)
    uint64_t t1 = now();
    out[threadIdx.x] = x; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_i32_dep(double* out, uint64_t* cyc, double a, double b) {
    uint32_t x = threadIdx.x, y = (uint32_t)a;
    uint64_t t0 = now();
    REP256(asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(y));)
    uint64_t t1 = now();
    out[threadIdx.x] = x; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_i32_ind(double* out, uint64_t* cyc, double a, double b) {
    uint32_t x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, y = (uint32_t)a;
    uint64_t t0 = now();
    REP64(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(y));)
    uint64_t t1 = now();
    out[threadIdx.x] = x0 + x1 + x2 + x3; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_cndmask_dep(double* out, uint64_t* cyc, double a, double b) {   // v_cmp -> vcc -> v_cndmask chain
    double x = a + threadIdx.x;
    uint32_t lo = (uint32_t)threadIdx.x, hi = 7;
    uint64_t t0 = now();
    REP256(asm volatile("v_cmp_gt_f64 vcc, %1, %2\n v_cndmask_b32 %0, %0, %3, vcc" : "+v"(lo) : "v"(x), "v"(a), "v"(hi) : "vcc");)
    uint64_t t1 = now();
    out[threadIdx.x] = lo; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_rfl_dep(double* out, uint64_t* cyc, double a, double b) {        // VALU -> readfirstlane -> SALU -> VALU round trip
    uint32_t x = threadIdx.x; uint32_t s;
    uint64_t t0 = now();
    REP256(asm volatile("v_readfirstlane_b32 %1, %0\n s_add_u32 %1, %1, 1\n v_add_u32 %0, %0, %1" : "+v"(x), "=s"(s) :: "scc");)
    uint64_t t1 = now();
    out[threadIdx.x] = x; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_salu_dep(double* out, uint64_t* cyc, double a, double b) {
    uint32_t s = (uint32_t)(uint64_t)cyc;
    uint64_t t0 = now();
    REP256(asm volatile("s_add_u32 %0, %0, 3" : "+s"(s) :: "scc");)
    uint64_t t1 = now();
    out[threadIdx.x] = (double)(int)__builtin_amdgcn_readfirstlane((int)s); if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_dpp_dep(double* out, uint64_t* cyc, double a, double b) {        // one butterfly step: 2 dpp movs + add
    double x = a + threadIdx.x;
    uint64_t t0 = now();
    REP64(x += __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(x), 0xb1, 0xf, 0xf, true), __builtin_amdgcn_update_dpp(0, __double2loint(x), 0xb1, 0xf, 0xf, true));
          asm volatile("" : "+v"(x));)
    uint64_t t1 = now();
    out[threadIdx.x] = x; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_lds_dep(double* out, uint64_t* cyc, double a, double b) {        // dependent ds_read_b32 pointer chase
    __shared__ uint32_t sh[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) sh[i] = ((i + 64) & 1023) * 4;
    __syncthreads();
    uint32_t p = threadIdx.x * 4;
    uint64_t t0 = now();
    REP256(asm volatile("ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)" : "+v"(p) :: "memory");)
    uint64_t t1 = now();
    out[threadIdx.x] = p; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_lds128_dep(double* out, uint64_t* cyc, double a, double b) {     // ds_read_b128 (the sigma / mu / L[1] reads), address dependent
    __shared__ uint4 sh[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) sh[i] = make_uint4(((i + 64) & 1023) * 16, 0, 0, 0);
    __syncthreads();
    uint4 p = make_uint4(threadIdx.x * 16, 0, 0, 0);
    uint64_t t0 = now();
    REP256(asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(p) : "v"(p.x) : "memory");)
    uint64_t t1 = now();
    out[threadIdx.x] = p.x; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_gload_dep(double* out, uint64_t* cyc, const uint32_t* chase, int stride_unused) {   // dependent global loads (pointer chase through a buffer)
    uint32_t p = threadIdx.x;
    uint64_t t0 = now();
    for (int i = 0; i < 64; ++i) { p = chase[p]; }
    uint64_t t1 = now();
    out[threadIdx.x] = p; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_smem_dep(double* out, uint64_t* cyc, const uint32_t* chase, int) {
    uint32_t p = 0;
    uint64_t t0 = now();
    for (int i = 0; i < 64; ++i) { p = __builtin_amdgcn_readfirstlane(chase[__builtin_amdgcn_readfirstlane(p)]); }
    uint64_t t1 = now();
    out[threadIdx.x] = p; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_store_load(double* out, uint64_t* cyc, double* buf, int) {       // a tile store followed by a load of OTHER data: does the load wait for the store?
    double2 v = make_double2(threadIdx.x, 1.0);
    double2* b2 = reinterpret_cast<double2*>(buf);
    uint64_t acc = 0; double2 r = v;
    for (int i = 0; i < 16; ++i) {
        uint64_t t0 = now();
        for (int m = 0; m < 8; ++m) b2[(i * 16 + m) * 64 + threadIdx.x + 65536] = v;          // 8 KB store
        r = b2[(i * 16 + 8) * 64 + threadIdx.x];                                             // load elsewhere
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        uint64_t t1 = now();
        acc += t1 - t0; v.x += r.x;
    }
    out[threadIdx.x] = v.x; if (threadIdx.x == 0) cyc[0] = acc / 16;
}
__global__ void k_load_only(double* out, uint64_t* cyc, double* buf, int) {
    double2 v = make_double2(threadIdx.x, 1.0);
    double2* b2 = reinterpret_cast<double2*>(buf);
    uint64_t acc = 0; double2 r = v;
    for (int i = 0; i < 16; ++i) {
        uint64_t t0 = now();
        r = b2[(i * 16 + 8) * 64 + threadIdx.x + (i & 1) * 4096];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        uint64_t t1 = now();
        acc += t1 - t0; v.x += r.x;
    }
    out[threadIdx.x] = v.x; if (threadIdx.x == 0) cyc[0] = acc / 16;
}
__device__ __noinline__ double callee(double x) { return x * 1.0000001 + 0.5; }
__global__ void k_call(double* out, uint64_t* cyc, double a, double b) {
    double x = a + threadIdx.x;
    uint64_t t0 = now();
    for (int i = 0; i < 64; ++i) x = callee(x);
    uint64_t t1 = now();
    out[threadIdx.x] = x; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_branch(double* out, uint64_t* cyc, double a, int n) {            // uniform taken branches
    uint32_t s = (uint32_t)__builtin_amdgcn_readfirstlane(n);
    uint64_t t0 = now();
    REP64(asm volatile("s_cmp_eq_u32 %0, 0\n s_cbranch_scc1 1f\n s_add_u32 %0, %0, 1\n1:\n s_cmp_lg_u32 %0, 0\n s_cbranch_scc0 2f\n s_nop 0\n2:" : "+s"(s) :: "scc");)
    uint64_t t1 = now();
    out[threadIdx.x] = s; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

// a dependent chain interleaved with K independent instructions per chain step: is the chain's latency hidden at the issue rate?
__global__ void k_mix1(double* out, uint64_t* cyc, double a, double b) {
    double x = a + threadIdx.x, y0 = x + 1;
    uint64_t t0 = now();
    REP256(asm volatile("v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %2, %3, %2" : "+v"(x), "+v"(y0) : "v"(a), "v"(b));)
    uint64_t t1 = now();
    out[threadIdx.x] = x + y0; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_mix2(double* out, uint64_t* cyc, double a, double b) {
    double x = a + threadIdx.x, y0 = x + 1, y1 = x + 2;
    uint64_t t0 = now();
    REP256(asm volatile("v_fma_f64 %0, %0, %3, %4\n v_fma_f64 %1, %3, %4, %3\n v_fma_f64 %2, %3, %4, %4" : "+v"(x), "+v"(y0), "+v"(y1) : "v"(a), "v"(b));)
    uint64_t t1 = now();
    out[threadIdx.x] = x + y0 + y1; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_mix3(double* out, uint64_t* cyc, double a, double b) {
    double x = a + threadIdx.x, y0 = x + 1, y1 = x + 2, y2 = x + 3;
    uint64_t t0 = now();
    REP256(asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %4, %5, %4\n v_fma_f64 %2, %4, %5, %5\n v_fma_f64 %3, %5, %5, %4" : "+v"(x), "+v"(y0), "+v"(y1), "+v"(y2) : "v"(a), "v"(b));)
    uint64_t t1 = now();
    out[threadIdx.x] = x + y0 + y1 + y2; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
// two dependent chains interleaved
__global__ void k_two_chains(double* out, uint64_t* cyc, double a, double b) {
    double x = a + threadIdx.x, y = x + 1;
    uint64_t t0 = now();
    REP256(asm volatile("v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3" : "+v"(x), "+v"(y) : "v"(a), "v"(b));)
    uint64_t t1 = now();
    out[threadIdx.x] = x + y; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
// SALU between dependent VALU: do scalar instructions issue in the chain's shadow?
__global__ void k_valu_salu(double* out, uint64_t* cyc, double a, double b) {
    double x = a + threadIdx.x; uint32_t s = (uint32_t)(uint64_t)cyc;
    uint64_t t0 = now();
    REP256(asm volatile("v_fma_f64 %0, %0, %2, %3\n s_add_u32 %1, %1, 3" : "+v"(x), "+s"(s) : "v"(a), "v"(b) : "scc");)
    uint64_t t1 = now();
    out[threadIdx.x] = x + (double)(int)__builtin_amdgcn_readfirstlane((int)s); if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
// calibration: ticks of s_memtime per second (hipEvent wall clock around a long spin)
__global__ void k_spin(double* out, uint64_t* cyc, double a, int iters) {
    double x = a + threadIdx.x;
    uint64_t t0 = now();
    for (int i = 0; i < iters; ++i) { REP64(asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(x) : "v"(a));) }
    uint64_t t1 = now();
    out[threadIdx.x] = x; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    double* out; uint64_t* cyc; uint32_t* chase; double* buf;
    hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 64); hipMalloc(&buf, 64 << 20);
    std::vector<uint32_t> h(1 << 24);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (uint32_t)((i * 1048583ull + 12345) % h.size());     // 64 MiB table, scattered
    hipMalloc(&chase, h.size() * 4); hipMemcpy(chase, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    std::vector<uint32_t> h2(1 << 24); for (size_t i = 0; i < h2.size(); ++i) h2[i] = (uint32_t)((i + 64) & 4095);           // stays inside 16 KiB: cache hits
    uint32_t* chase_hit; hipMalloc(&chase_hit, h2.size() * 4); hipMemcpy(chase_hit, h2.data(), h2.size() * 4, hipMemcpyHostToDevice);
    hipMemset(buf, 0, 64 << 20);
    uint64_t c;
#define RUN(name, count, ...) do { for (int w = 0; w < 3; ++w) { hipLaunchKernelGGL(name, dim3(1), dim3(64), 0, 0, out, cyc, __VA_ARGS__); hipDeviceSynchronize(); } hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); \
    printf("%-16s %8llu ticks / %4d = %7.2f ticks each (s_memtime: 100 MHz ticks -> x%.0f cycles at 2.4 GHz = %.1f cycles)\n", #name, (unsigned long long)c, count, (double)c / count, 24.0, (double)c / count * 24.0); } while (0)
    RUN(k_fma_dep, 256, 1.0000001, 0.5);
    RUN(k_fma_ind, 256, 1.0000001, 0.5);
    RUN(k_fma_ind2, 256, 1.0000001, 0.5);
    RUN(k_add_dep, 256, 1.0000001, 0.5);
    RUN(k_mul_dep, 256, 1.0000001, 0.5);
    RUN(k_i32_dep, 256, 3.0, 0.5);
    RUN(k_i32_ind, 256, 3.0, 0.5);
    RUN(k_cndmask_dep, 256, 1.5, 0.5);
    RUN(k_rfl_dep, 256, 1.5, 0.5);
    RUN(k_salu_dep, 256, 1.5, 0.5);
    RUN(k_dpp_dep, 64, 1.5, 0.5);
    RUN(k_lds_dep, 256, 1.5, 0.5);
    RUN(k_lds128_dep, 256, 1.5, 0.5);
    RUN(k_gload_dep, 64, chase, 0);
    RUN(k_gload_dep, 64, chase_hit, 0);
    RUN(k_smem_dep, 64, chase, 0);
    RUN(k_smem_dep, 64, chase_hit, 0);
    RUN(k_store_load, 1, buf, 0);
    RUN(k_load_only, 1, buf, 0);
    RUN(k_call, 64, 1.5, 0.5);
    RUN(k_branch, 128, 1.5, 5);
    RUN(k_mix1, 256, 1.0000001, 0.5);
    RUN(k_mix2, 256, 1.0000001, 0.5);
    RUN(k_mix3, 256, 1.0000001, 0.5);
    RUN(k_two_chains, 256, 1.0000001, 0.5);
    RUN(k_valu_salu, 256, 1.0000001, 0.5);
    {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int w = 0; w < 2; ++w) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, 0, out, cyc, 1.0000001, 200000);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
        }
        float ms = 0; hipEventElapsedTime(&ms, e0, e1); hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("k_spin: %llu ticks in %.3f ms -> %.1f MHz tick rate; %.2f ns per dependent fma (12.8e6 of them)\n", (unsigned long long)c, ms, (double)c / ms / 1e3, ms * 1e6 / 12.8e6);
    }
    return 0;
}
