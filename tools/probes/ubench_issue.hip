// ubench_issue.hip — steady-state issue / dependency costs of ONE wavefront on gfx950 (round 5).  Every pattern runs 200 x 64 times in a
// loop (instruction cache warm, timing overhead amortised); s_memtime ticks at the 2.4 GHz shader clock (calibrated against hipEvents).
// hipcc --offload-arch=gfx950 -O3 tools/probes/ubench_issue.hip -o tools/probes/ubench_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)
#define ITERS 200
#define S4(x) x x x x
#define S16(x) S4(x) S4(x) S4(x) S4(x)
__device__ __forceinline__ uint64_t now() { uint64_t t; asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; }
#define KERNEL(name, decl, body, result)                                                        \
    __global__ void name(double* out, uint64_t* cyc, double a, double b) {                      \
        decl;                                                                                   \
        uint64_t t0 = now();                                                                    \
        for (int i_ = 0; i_ < ITERS; ++i_) { REP4(body) }                                      \
        uint64_t t1 = now();                                                                    \
        out[threadIdx.x] = (double)(result); if (threadIdx.x == 0) cyc[0] = t1 - t0;            \
    }
KERNEL(fma64_chain1, double x = a + threadIdx.x, asm volatile(S16("v_fma_f64 %0, %0, %1, %2" "\n") : "+v"(x) : "v"(a), "v"(b));, x)
KERNEL(fma64_chain2, double x = a + threadIdx.x; double y = x + 1, asm volatile(S16("v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3" "\n") : "+v"(x), "+v"(y) : "v"(a), "v"(b));, x + y)
KERNEL(fma64_chain3, double x = a + threadIdx.x; double y = x + 1; double z = x + 2, asm volatile(S16("v_fma_f64 %0, %0, %3, %4\n v_fma_f64 %1, %1, %3, %4\n v_fma_f64 %2, %2, %3, %4" "\n") : "+v"(x), "+v"(y), "+v"(z) : "v"(a), "v"(b));, x + y + z)
KERNEL(fma64_chain4, double x = a + threadIdx.x; double y = x + 1; double z = x + 2; double w = x + 3, asm volatile(S16("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5" "\n") : "+v"(x), "+v"(y), "+v"(z), "+v"(w) : "v"(a), "v"(b));, x + y + z + w)
KERNEL(fma64_dep_plus1ind, double x = a + threadIdx.x; double y = x + 1, asm volatile(S16("v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %2, %3, %2" "\n") : "+v"(x), "+v"(y) : "v"(a), "v"(b));, x + y)
KERNEL(fma64_dep_plus2ind, double x = a + threadIdx.x; double y = x + 1; double z = x + 2, asm volatile(S16("v_fma_f64 %0, %0, %3, %4\n v_fma_f64 %1, %3, %4, %3\n v_fma_f64 %2, %3, %4, %4" "\n") : "+v"(x), "+v"(y), "+v"(z) : "v"(a), "v"(b));, x + y + z)
KERNEL(add64_chain1, double x = a + threadIdx.x, asm volatile(S16("v_add_f64 %0, %0, %1" "\n") : "+v"(x) : "v"(a));, x)
KERNEL(fma32_chain1, float x = (float)a + threadIdx.x; float fa = (float)a; float fb = (float)b, asm volatile(S16("v_fma_f32 %0, %0, %1, %2" "\n") : "+v"(x) : "v"(fa), "v"(fb));, x)
KERNEL(fma32_chain4, float x = (float)a + threadIdx.x; float y = x + 1; float z = x + 2; float w = x + 3; float fa = (float)a; float fb = (float)b, asm volatile(S16("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5" "\n") : "+v"(x), "+v"(y), "+v"(z), "+v"(w) : "v"(fa), "v"(fb));, x + y + z + w)
KERNEL(i32_chain1, uint32_t x = threadIdx.x; uint32_t y = (uint32_t)a, asm volatile(S16("v_add_u32 %0, %0, %1" "\n") : "+v"(x) : "v"(y));, x)
KERNEL(i32_chain4, uint32_t x = threadIdx.x; uint32_t x1 = x + 1; uint32_t x2 = x + 2; uint32_t x3 = x + 3; uint32_t y = (uint32_t)a, asm volatile(S16("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" "\n") : "+v"(x), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(y));, x + x1 + x2 + x3)
KERNEL(salu_chain1, uint32_t s = (uint32_t)(uint64_t)cyc, asm volatile(S16("s_add_u32 %0, %0, 3" "\n") : "+s"(s) :: "scc");, (int)__builtin_amdgcn_readfirstlane((int)s))
KERNEL(salu_chain2, uint32_t s = (uint32_t)(uint64_t)cyc; uint32_t s2 = s + 1, asm volatile(S16("s_add_u32 %0, %0, 3\n s_add_u32 %1, %1, 5" "\n") : "+s"(s), "+s"(s2) :: "scc");, (int)__builtin_amdgcn_readfirstlane((int)(s + s2)))
KERNEL(valu_dep_plus_salu, double x = a + threadIdx.x; uint32_t s = (uint32_t)(uint64_t)cyc, asm volatile(S16("v_fma_f64 %0, %0, %2, %3\n s_add_u32 %1, %1, 3" "\n") : "+v"(x), "+s"(s) : "v"(a), "v"(b) : "scc");, x + (int)__builtin_amdgcn_readfirstlane((int)s))
KERNEL(valu_dep_plus_2salu, double x = a + threadIdx.x; uint32_t s = (uint32_t)(uint64_t)cyc; uint32_t s2 = s + 1, asm volatile(S16("v_fma_f64 %0, %0, %3, %4\n s_add_u32 %1, %1, 3\n s_add_u32 %2, %2, 5" "\n") : "+v"(x), "+s"(s), "+s"(s2) : "v"(a), "v"(b) : "scc");, x + (int)__builtin_amdgcn_readfirstlane((int)(s + s2)))
KERNEL(rfl_salu_valu, uint32_t x = threadIdx.x; uint32_t s = 0, asm volatile(S16("v_readfirstlane_b32 %1, %0\n s_add_u32 %1, %1, 1\n v_add_u32 %0, %0, %1" "\n") : "+v"(x), "+s"(s) :: "scc");, x)
KERNEL(cmp_cndmask, double x = a + threadIdx.x; uint32_t lo = threadIdx.x; uint32_t hi = 7, asm volatile(S16("v_cmp_gt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" "\n") : "+v"(lo) : "v"(hi) : "vcc");, lo)
KERNEL(cmp_sgpr_branchless, uint32_t lo = threadIdx.x; uint32_t hi = 7, asm volatile(S16("v_cmp_gt_u32 vcc, %0, %1\n s_and_b64 vcc, vcc, exec\n v_cndmask_b32 %0, %0, %1, vcc" "\n") : "+v"(lo) : "v"(hi) : "vcc", "scc");, lo)
KERNEL(dpp_step, uint32_t x = threadIdx.x; uint32_t t = 0, asm volatile(S16("s_nop 1
 v_mov_b32_dpp %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf
 v_add_u32 %0, %0, %1" "
") : "+v"(x), "+v"(t));, x)
KERNEL(dpp_step_nonop, uint32_t x = threadIdx.x; uint32_t t = 0; uint32_t y = x + 5, asm volatile(S16("v_mov_b32_dpp %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf
 v_add_u32 %0, %0, %1" "
") : "+v"(x), "+v"(t) : "v"(y));, x)
KERNEL(ldexp_chain, double x = a + threadIdx.x; int k = 1, asm volatile(S16("v_ldexp_f64 %0, %0, %1" "\n") : "+v"(x) : "v"(k));, x)
KERNEL(rcp64_chain, double x = a + threadIdx.x, asm volatile(S16("v_rcp_f64 %0, %0" "\n") : "+v"(x));, x)
KERNEL(cvt_chain, double x = a + threadIdx.x; int k = 1, asm volatile(S16("v_cvt_i32_f64 %1, %0\n v_cvt_f64_i32 %0, %1" "\n") : "+v"(x), "+v"(k));, x)
KERNEL(rndne_chain, double x = a + threadIdx.x, asm volatile(S16("v_rndne_f64 %0, %0" "\n") : "+v"(x));, x)
KERNEL(accvgpr_roundtrip, uint32_t x = threadIdx.x; uint32_t acc_, asm volatile(S16("v_accvgpr_write_b32 a0, %0\n s_nop 1\n v_accvgpr_read_b32 %0, a0" "\n") : "+v"(x) :: "a0");, x)
KERNEL(branch_taken, uint32_t s = (uint32_t)(uint64_t)cyc | 1u, asm volatile(S16("s_cmp_lg_u32 %0, 0\n s_cbranch_scc1 1f\n s_nop 0\n1:" "\n") : "+s"(s) :: "scc");, (int)__builtin_amdgcn_readfirstlane((int)s))
KERNEL(branch_not_taken, uint32_t s = (uint32_t)(uint64_t)cyc | 1u, asm volatile(S16("s_cmp_eq_u32 %0, 0\n s_cbranch_scc1 1f\n s_nop 0\n1:" "\n") : "+s"(s) :: "scc");, (int)__builtin_amdgcn_readfirstlane((int)s))
KERNEL(valu_cmp_branch, uint32_t x = threadIdx.x + 1, asm volatile(S16("v_cmp_ne_u32 vcc, 0, %0\n s_cbranch_vccz 1f\n v_add_u32 %0, %0, 2\n1:" "\n") : "+v"(x) :: "vcc");, x)
__global__ void lds_chase(double* out, uint64_t* cyc, double a, double b) {
    __shared__ uint32_t sh[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) sh[i] = ((i + 64) & 1023) * 4;
    __syncthreads();
    uint32_t p = threadIdx.x * 4;
    uint64_t t0 = now();
    for (int i = 0; i < ITERS; ++i) { REP4(asm volatile(S16("ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)" "\n") : "+v"(p) :: "memory");) }
    uint64_t t1 = now();
    out[threadIdx.x] = p; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void lds_chase_bcast(double* out, uint64_t* cyc, double a, double b) {     // every lane reads the SAME word (the exp / ln tables, the RNG words)
    __shared__ uint32_t sh[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) sh[i] = ((i + 64) & 1023) * 4;
    __syncthreads();
    uint32_t p = 0;
    uint64_t t0 = now();
    for (int i = 0; i < ITERS; ++i) { REP4(asm volatile(S16("ds_read_b32 %0, %0\n s_waitcnt lgkmcnt(0)" "\n") : "+v"(p) :: "memory");) }
    uint64_t t1 = now();
    out[threadIdx.x] = p; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__device__ __noinline__ double callee(double x) { return x * 1.0000001 + 0.5; }
__global__ void call_chain(double* out, uint64_t* cyc, double a, double b) {
    double x = a + threadIdx.x;
    uint64_t t0 = now();
    for (int i = 0; i < ITERS * 64; ++i) x = callee(x);
    uint64_t t1 = now();
    out[threadIdx.x] = x; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    double* out; uint64_t* cyc; uint64_t c;
    (void)hipMalloc(&out, 64 * 8); (void)hipMalloc(&cyc, 64);
#define RUN(name, per) do { for (int w = 0; w < 2; ++w) { hipLaunchKernelGGL(name, dim3(1), dim3(64), 0, 0, out, cyc, 1.0000001, 0.5); (void)hipDeviceSynchronize(); } (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); \
    printf("%-24s %7.2f cycles per pattern = %6.2f per instruction (%d in the pattern)\n", #name, (double)c / (ITERS * 64.0), (double)c / (ITERS * 64.0) / per, per); } while (0)
    RUN(fma64_chain1, 1); RUN(fma64_chain2, 2); RUN(fma64_chain3, 3); RUN(fma64_chain4, 4); RUN(fma64_dep_plus1ind, 2); RUN(fma64_dep_plus2ind, 3);
    RUN(add64_chain1, 1); RUN(fma32_chain1, 1); RUN(fma32_chain4, 4); RUN(i32_chain1, 1); RUN(i32_chain4, 4);
    RUN(salu_chain1, 1); RUN(salu_chain2, 2); RUN(valu_dep_plus_salu, 2); RUN(valu_dep_plus_2salu, 3); RUN(rfl_salu_valu, 3);
    RUN(cmp_cndmask, 2); RUN(cmp_sgpr_branchless, 3); RUN(dpp_step, 3); RUN(dpp_step_nonop, 2); RUN(ldexp_chain, 1); RUN(rcp64_chain, 1); RUN(cvt_chain, 2); RUN(rndne_chain, 1);
    RUN(accvgpr_roundtrip, 3); RUN(branch_taken, 2); RUN(branch_not_taken, 3); RUN(valu_cmp_branch, 3); RUN(lds_chase, 1); RUN(lds_chase_bcast, 1); RUN(call_chain, 1);
    return 0;
}
