// probe_bw.hip — HBM streaming probes with the engine's own access shape (16 B per lane, 1 KiB per wave instruction,
// buffer-descriptor addressing): the measured denominator of the roofline and the known-size
// kernels the rocprofv3 FETCH_SIZE / WRITE_SIZE counters are calibrated on (MI355X_MICROARCH.md §HBM: FETCH_SIZE reads
// 1/2 of a wide coalesced stream on gfx950, WRITE_SIZE is uncalibrated).
#include <hip/hip_runtime.h>
#include <cstdint>
#include "../../include/nuts_amd.h"

namespace {

typedef unsigned int v4u __attribute__((ext_vector_type(4)));
constexpr int AUX_NT = 2;

// persistent grid-stride over 16-byte elements; every wave instruction moves 1 KiB, consecutive waves consecutive KiB
template <int KIND>
__global__ __launch_bounds__(256) void probe_kernel(const double2* __restrict__ a, const double2* __restrict__ b,
                                                   double2* __restrict__ c, uint64_t n16, double s, double* sink) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double acc = 0.0;
    for (; i + 3 * stride < n16; i += 4 * stride) {            // four independent 16 B accesses in flight per lane
        if (KIND == NM_PROBE_COPY || KIND == NM_PROBE_COPY_NT) {
            const double2 q0 = a[i], q1 = a[i + stride], q2 = a[i + 2 * stride], q3 = a[i + 3 * stride];
            if (KIND == NM_PROBE_COPY) { c[i] = q0; c[i + stride] = q1; c[i + 2 * stride] = q2; c[i + 3 * stride] = q3; }
            else {
                __builtin_nontemporal_store(q0.x, &c[i].x); __builtin_nontemporal_store(q0.y, &c[i].y);
                __builtin_nontemporal_store(q1.x, &c[i + stride].x); __builtin_nontemporal_store(q1.y, &c[i + stride].y);
                __builtin_nontemporal_store(q2.x, &c[i + 2 * stride].x); __builtin_nontemporal_store(q2.y, &c[i + 2 * stride].y);
                __builtin_nontemporal_store(q3.x, &c[i + 3 * stride].x); __builtin_nontemporal_store(q3.y, &c[i + 3 * stride].y);
            }
        } else if (KIND == NM_PROBE_TRIAD) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const double2 x = a[i + u * stride], y = b[i + u * stride];
                c[i + u * stride] = make_double2(__builtin_fma(s, y.x, x.x), __builtin_fma(s, y.y, x.y));
            }
        } else if (KIND == NM_PROBE_READ) {
            const double2 q0 = a[i], q1 = a[i + stride], q2 = a[i + 2 * stride], q3 = a[i + 3 * stride];
            acc += (q0.x + q0.y) + (q1.x + q1.y) + (q2.x + q2.y) + (q3.x + q3.y);
        } else {   // NM_PROBE_WRITE
            const double2 q = make_double2(s, (double)i);
            c[i] = q; c[i + stride] = q; c[i + 2 * stride] = q; c[i + 3 * stride] = q;
        }
    }
    for (; i < n16; i += stride) {
        if (KIND == NM_PROBE_READ) acc += a[i].x + a[i].y;
        else if (KIND == NM_PROBE_TRIAD) { const double2 x = a[i], y = b[i]; c[i] = make_double2(__builtin_fma(s, y.x, x.x), __builtin_fma(s, y.y, x.y)); }
        else if (KIND == NM_PROBE_WRITE) c[i] = make_double2(s, (double)i);
        else c[i] = a[i];
    }
    if (KIND == NM_PROBE_READ && acc == 12345.678) *sink = acc;   // keeps the loads alive
}

// flat variant: every block owns one contiguous tile of 256*U 16-byte elements (U per lane, all in flight)
template <int KIND, int U>
__global__ __launch_bounds__(256) void probe_flat_kernel(const double2* __restrict__ a, const double2* __restrict__ b,
                                                        double2* __restrict__ c, uint64_t n16, double s, double* sink) {
    const uint64_t base = (uint64_t)blockIdx.x * (256 * U) + threadIdx.x;
    double2 q[U], r[U];
    if (KIND != NM_PROBE_WRITE) {
#pragma unroll
        for (int u = 0; u < U; ++u) q[u] = base + u * 256 < n16 ? a[base + u * 256] : make_double2(0., 0.);
    }
    if (KIND == NM_PROBE_TRIAD) {
#pragma unroll
        for (int u = 0; u < U; ++u) r[u] = base + u * 256 < n16 ? b[base + u * 256] : make_double2(0., 0.);
    }
    double acc = 0.0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const uint64_t i = base + u * 256;
        if (i >= n16) break;
        if (KIND == NM_PROBE_COPY) c[i] = q[u];
        else if (KIND == NM_PROBE_COPY_NT) { __builtin_nontemporal_store(q[u].x, &c[i].x); __builtin_nontemporal_store(q[u].y, &c[i].y); }
        else if (KIND == NM_PROBE_TRIAD) c[i] = make_double2(__builtin_fma(s, r[u].x, q[u].x), __builtin_fma(s, r[u].y, q[u].y));
        else if (KIND == NM_PROBE_READ) acc += q[u].x + q[u].y;
        else c[i] = make_double2(s, (double)i);
    }
    if (KIND == NM_PROBE_READ && acc == 12345.678) *sink = acc;
}
template <int U>
hipError_t launch_flat(uint64_t kind, hipStream_t st, const double2* a, const double2* b, double2* c, uint64_t n16, double* sink) {
    const unsigned grid = (unsigned)((n16 + 256 * U - 1) / (256 * U));
    switch (kind) {
    case NM_PROBE_COPY: hipLaunchKernelGGL((probe_flat_kernel<NM_PROBE_COPY, U>), dim3(grid), dim3(256), 0, st, a, b, c, n16, 0.5, sink); break;
    case NM_PROBE_COPY_NT: hipLaunchKernelGGL((probe_flat_kernel<NM_PROBE_COPY_NT, U>), dim3(grid), dim3(256), 0, st, a, b, c, n16, 0.5, sink); break;
    case NM_PROBE_TRIAD: hipLaunchKernelGGL((probe_flat_kernel<NM_PROBE_TRIAD, U>), dim3(grid), dim3(256), 0, st, a, b, c, n16, 0.5, sink); break;
    case NM_PROBE_READ: hipLaunchKernelGGL((probe_flat_kernel<NM_PROBE_READ, U>), dim3(grid), dim3(256), 0, st, a, b, c, n16, 0.5, sink); break;
    case NM_PROBE_WRITE: hipLaunchKernelGGL((probe_flat_kernel<NM_PROBE_WRITE, U>), dim3(grid), dim3(256), 0, st, a, b, c, n16, 0.5, sink); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// variant 0: persistent grid-stride (the engine's shape); 1, 2, 3: flat tiles with 1, 4, 8 accesses in flight per lane
hipError_t launch_variant(int variant, uint64_t kind, unsigned grid, hipStream_t st, const double2* a, const double2* b,
                          double2* c, uint64_t n16, double* sink);

hipError_t launch_probe(uint64_t kind, unsigned grid, hipStream_t st, const double2* a, const double2* b, double2* c,
                        uint64_t n16, double* sink) {
    switch (kind) {
    case NM_PROBE_COPY: hipLaunchKernelGGL(probe_kernel<NM_PROBE_COPY>, dim3(grid), dim3(256), 0, st, a, b, c, n16, 0.5, sink); break;
    case NM_PROBE_COPY_NT: hipLaunchKernelGGL(probe_kernel<NM_PROBE_COPY_NT>, dim3(grid), dim3(256), 0, st, a, b, c, n16, 0.5, sink); break;
    case NM_PROBE_TRIAD: hipLaunchKernelGGL(probe_kernel<NM_PROBE_TRIAD>, dim3(grid), dim3(256), 0, st, a, b, c, n16, 0.5, sink); break;
    case NM_PROBE_READ: hipLaunchKernelGGL(probe_kernel<NM_PROBE_READ>, dim3(grid), dim3(256), 0, st, a, b, c, n16, 0.5, sink); break;
    case NM_PROBE_WRITE: hipLaunchKernelGGL(probe_kernel<NM_PROBE_WRITE>, dim3(grid), dim3(256), 0, st, a, b, c, n16, 0.5, sink); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}
hipError_t launch_variant(int variant, uint64_t kind, unsigned grid, hipStream_t st, const double2* a, const double2* b,
                          double2* c, uint64_t n16, double* sink) {
    switch (variant) {
    case 0: return launch_probe(kind, grid, st, a, b, c, n16, sink);
    case 1: return launch_flat<1>(kind, st, a, b, c, n16, sink);
    case 2: return launch_flat<4>(kind, st, a, b, c, n16, sink);
    default: return launch_flat<8>(kind, st, a, b, c, n16, sink);
    }
}
}  // namespace

// declared in include/nuts_amd.h
extern "C" int nm_probe_bandwidth_impl(uint64_t kind, uint64_t bytes_per_array, uint64_t iters, double* ms_per_iter,
                                       uint64_t* bytes_read, uint64_t* bytes_written, const char** err) {
    *err = nullptr;
    const uint64_t n16 = bytes_per_array / 16;
    if (n16 == 0 || iters == 0) { *err = "empty probe"; return 1; }
    double2 *a = nullptr, *b = nullptr, *c = nullptr;
    double* sink = nullptr;
    hipError_t e = hipSuccess;
    auto fin = [&](hipError_t er) {
        if (a) (void)hipFree(a); if (b) (void)hipFree(b); if (c) (void)hipFree(c); if (sink) (void)hipFree(sink);
        if (er != hipSuccess) { *err = hipGetErrorString(er); return 3; }
        return 0;
    };
    const bool needs_a = kind != NM_PROBE_WRITE, needs_b = kind == NM_PROBE_TRIAD, needs_c = kind != NM_PROBE_READ;
    if (needs_a && (e = hipMalloc(&a, n16 * 16)) != hipSuccess) return fin(e);
    if (needs_b && (e = hipMalloc(&b, n16 * 16)) != hipSuccess) return fin(e);
    if (needs_c && (e = hipMalloc(&c, n16 * 16)) != hipSuccess) return fin(e);
    if ((e = hipMalloc(&sink, 8)) != hipSuccess) return fin(e);
    int cus = 256;
    int dev = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const unsigned grid = (unsigned)cus * 8u;
    hipStream_t st = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if ((e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking)) != hipSuccess) return fin(e);
    // the fills run on the probe's own stream: a null-stream fill is not ordered against a hipStreamNonBlocking stream
    if (a && e == hipSuccess) e = hipMemsetAsync(a, 0x11, n16 * 16, st);
    if (b && e == hipSuccess) e = hipMemsetAsync(b, 0x22, n16 * 16, st);
    if (c && e == hipSuccess) e = hipMemsetAsync(c, 0, n16 * 16, st);
    if (e != hipSuccess) { (void)hipStreamDestroy(st); return fin(e); }
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    double best = -1.0;
    for (int variant = 0; variant < 4 && e == hipSuccess; ++variant) {           // the fastest launch shape is the probe's answer
        e = launch_variant(variant, kind, grid, st, a, b, c, n16, sink);        // warm-up launch
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) break;
        (void)hipEventRecord(e0, st);
        for (uint64_t i = 0; i < iters && e == hipSuccess; ++i) e = launch_variant(variant, kind, grid, st, a, b, c, n16, sink);
        (void)hipEventRecord(e1, st);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        float ms = 0.f;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        const double per = (double)ms / (double)iters;
        if (e == hipSuccess && (best < 0.0 || per < best)) best = per;
    }
    if (ms_per_iter) *ms_per_iter = best;
    if (bytes_read) *bytes_read = (needs_a ? n16 * 16 : 0) + (needs_b ? n16 * 16 : 0);
    if (bytes_written) *bytes_written = needs_c ? n16 * 16 : 0;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipStreamDestroy(st);
    return fin(e);
}

// ---------------------------------------------------------------------------------------------
// Fixed-work issue-rate calibration (round 6): `waves` one-wavefront blocks each run a DEPENDENT chain of
// `chain` v_fma_f64 (64 per loop trip, one asm block).  Every BASELINE kernel of this engine runs one useful wavefront per SIMD and is bound
// by the instructions that wavefront issues, so the time of this loop — nanoseconds per dependent instruction of a lone wavefront —
// is the box-dependent factor of their speed: the bench line prints it beside every config so that a slower box is not read as a regression.
// ---------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(64) void issue_kernel(double* out, uint64_t trips, double a, double b) {
    double x = (double)threadIdx.x * 1e-3;
    for (uint64_t t = 0; t < trips; ++t) {
        asm volatile(
            "v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n"
            "v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n"
            "v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n"
            "v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n"
            "v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n"
            "v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n"
            "v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n"
            "v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n"
            "v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n"
            "v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n"
            "v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n"
            "v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n"
            "v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n"
            "v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n"
            "v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n"
            "v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n"
            : "+v"(x) : "v"(a), "v"(b));
    }
    if (x == 12345.678) out[0] = x;      // keeps the chain alive
}
}  // namespace

// declared in include/nuts_amd.h (nm_probe_issue)
extern "C" int nm_probe_issue_impl(uint64_t waves, uint64_t chain, double* ns_per_instruction, const char** err) {
    *err = nullptr;
    if (chain < 64) { *err = "chain shorter than one loop trip (64 instructions)"; return 1; }
    int cus = 256, dev = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const unsigned grid = waves ? (unsigned)waves : (unsigned)cus * 4u;          // default: one wavefront per SIMD
    const uint64_t trips = chain / 64;
    double* out = nullptr;
    hipError_t e = hipMalloc(&out, 8);
    if (e != hipSuccess) { *err = hipGetErrorString(e); return 3; }
    hipStream_t st = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    if (e != hipSuccess) { (void)hipFree(out); *err = hipGetErrorString(e); return 3; }
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    double best = -1.0;
    for (int rep = 0; rep < 4 && e == hipSuccess; ++rep) {                      // (the first is the warm-up; the fastest of the rest is the answer)
        (void)hipEventRecord(e0, st);
        hipLaunchKernelGGL(issue_kernel, dim3(grid), dim3(64), 0, st, out, trips, 0.999999, 1e-9);
        e = hipGetLastError();
        (void)hipEventRecord(e1, st);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        float ms = 0.f;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        const double ns = (double)ms * 1e6 / (double)(trips * 64);
        if (e == hipSuccess && rep > 0 && (best < 0.0 || ns < best)) best = ns;
    }
    if (ns_per_instruction) *ns_per_instruction = best;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipStreamDestroy(st); (void)hipFree(out);
    if (e != hipSuccess) { *err = hipGetErrorString(e); return 3; }
    return 0;
}

