// math_seam.hip — the per-vector seam of the reference: `trait Math` (src/math/math.rs:15-314) over device vectors, one
// exported function per hot-path method (SURVEY §8(a) rows M1-M15, §8(b) "Per-vector backend").  One `nm_math` stands for one
// `CpuMath<F>` (src/math/cpu_math.rs:19-41): a density, a dimension, a HIP stream; an `nm_vec` is its `M::Vector`
// (opaque, device resident, the engine's tile layout).  Every method is one small launch of ONE block — this seam has
// no batching (one chain per Math), so on a GPU it is launch-latency bound; it exists for API completeness and for
// unit parity of the fused kernels' building blocks: each method executes the arithmetic of nuts_kernels.hpp (same
// operations, FMAs where the reference writes mul_add, the engine's reduction order).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstring>
#include <new>
#include "nuts_kernels.hpp"

using namespace nm;

struct nm_vec { double* d; };
struct nm_math {
    nm_logp_spec spec;
    uint64_t dim = 0, dpad = 0;
    int dpl = 0, w = 1;
    double* d_params = nullptr;
    double* d_zig = nullptr;
    double* d_out = nullptr;       // 8 doubles of results
    hipStream_t stream = nullptr;
};

namespace {
enum Op : int { OP_AXPY_OUT, OP_AXPY, OP_MULT, OP_DOT, OP_PRODS3, OP_GAUSSIAN, OP_UPD_VAR, OP_STD_DRAW_GRAD, OP_STD_GRAD, OP_STD_DRAW,
                OP_SUM_LN, OP_ALL_FINITE, OP_ALL_FINITE_NONZERO, OP_LOGP, OP_SQ_NORM_SUM, OP_FILL, OP_RECIP, OP_COPY,
                OP_STD_NORM_FLOW, OP_STD_NORM_GRAD_FLOW, OP_ESH, OP_NORMALIZE };
struct VArgs {
    int op;
    uint64_t dim;
    double *a, *b, *c, *d, *e;      // vectors (device, padded tile layout)
    double s0, s1, s2, s3;          // scalars
    int flag;
    double* out;                    // results
    const double* params;
    uint32_t key[8];
    uint64_t pos;
    const double *zig_x, *zig_f;
};

template <int DPL, int W, class Dens>
__global__ __launch_bounds__(64 * W) void vec_op_kernel(const VArgs A) {
    dm_init_lds();
    __shared__ double lred[2 * RED_MAX_VALUES * W];
    __shared__ double ldens[Dens::kNeedsLdsVector ? 64 * W * DPL : 2];
    __shared__ uint32_t lcache[RNG_CACHE_WORDS];
    __shared__ double lstage[64 * W * DPL + 72];
    Reducer<W> R;
    R.init(lred);
    const int dim = (int)A.dim;
    Tile<DPL> a, b, c, d, e;
    auto valid = [&](int k) { return elem_index<W>(k) < dim; };
    switch (A.op) {
    case OP_AXPY_OUT:      // M1  out = fma(a, x, y)   (util.rs:448-505)
        load_tile<DPL, W>(a, A.a); load_tile<DPL, W>(b, A.b);
#pragma unroll
        for (int k = 0; k < DPL; ++k) c.a[k] = valid(k) ? __builtin_fma(A.s0, a.a[k], b.a[k]) : 0.0;
        store_tile<DPL, W>(c, A.c);
        break;
    case OP_AXPY:          // M2  y = fma(a, x, y)     (util.rs:402-446)
        load_tile<DPL, W>(a, A.a); load_tile<DPL, W>(b, A.b);
#pragma unroll
        for (int k = 0; k < DPL; ++k) b.a[k] = valid(k) ? __builtin_fma(A.s0, a.a[k], b.a[k]) : 0.0;
        store_tile<DPL, W>(b, A.b);
        break;
    case OP_MULT:          // M3  dest = a * b         (util.rs:21-112)
        load_tile<DPL, W>(a, A.a); load_tile<DPL, W>(b, A.b);
#pragma unroll
        for (int k = 0; k < DPL; ++k) c.a[k] = valid(k) ? a.a[k] * b.a[k] : 0.0;
        store_tile<DPL, W>(c, A.c);
        break;
    case OP_DOT: {         // M4  sum a_i b_i, fma accumulation, the engine's order (util.rs:349-400)
        load_tile<DPL, W>(a, A.a); load_tile<DPL, W>(b, A.b);
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < DPL; ++k) acc = __builtin_fma(a.a[k], b.a[k], acc);
        acc = R.sum(acc);
        if (tid() == 0) A.out[0] = acc;
        break;
    }
    case OP_PRODS3: {      // M5  s = (p1 + p2) - n1 ; (sum s x, sum s y)   (util.rs:221-347)
        load_tile<DPL, W>(a, A.a); load_tile<DPL, W>(b, A.b); load_tile<DPL, W>(c, A.c); load_tile<DPL, W>(d, A.d); load_tile<DPL, W>(e, A.e);
        double t1 = 0.0, t2 = 0.0;
#pragma unroll
        for (int k = 0; k < DPL; ++k) {
            const double s = (a.a[k] + c.a[k]) - b.a[k];
            t1 = __builtin_fma(s, d.a[k], t1);
            t2 = __builtin_fma(s, e.a[k], t2);
        }
        R.sum2(t1, t2);
        if (tid() == 0) { A.out[0] = t1; A.out[1] = t2; }
        break;
    }
    case OP_GAUSSIAN: {    // M6  dest_i = stds_i * N(0,1), stream order = index order (cpu_math.rs:561-577)
        DevRng rng;
        rng.init(A.key, A.pos, lcache);
        ZigTables T = {A.zig_x, A.zig_f};
        fill_standard_normals(rng, lstage, dim, T);
        load_tile<DPL, W>(b, A.b);
        const double2* s2 = reinterpret_cast<const double2*>(lstage) + tid();
#pragma unroll
        for (int m = 0; m < DPL / 2; ++m) {
            const double2 q = s2[m * 64 * W];
            a.a[2 * m] = valid(2 * m) ? b.a[2 * m] * q.x : 0.0;
            a.a[2 * m + 1] = valid(2 * m + 1) ? b.a[2 * m + 1] * q.y : 0.0;
        }
        store_tile<DPL, W>(a, A.a);
        if (tid() == 0) A.out[0] = u2d(rng.pos);      // the stream position after the call (bit pattern)
        break;
    }
    case OP_UPD_VAR:       // M7  d = x - mean; mean += d * scale; var += d * d  (cpu_math.rs:605-631)
        load_tile<DPL, W>(a, A.a); load_tile<DPL, W>(b, A.b); load_tile<DPL, W>(c, A.c);
#pragma unroll
        for (int k = 0; k < DPL; ++k) {
            const double diff = c.a[k] - a.a[k];
            a.a[k] = valid(k) ? a.a[k] + diff * A.s0 : 0.0;
            b.a[k] = valid(k) ? b.a[k] + diff * diff : 0.0;
        }
        store_tile<DPL, W>(a, A.a); store_tile<DPL, W>(b, A.b);
        break;
    case OP_STD_DRAW_GRAD: // M8  (cpu_math.rs:671-708)   a = inv_std, b = std, c = draw_var, d = grad_var
    case OP_STD_DRAW:      // M10 (cpu_math.rs:633-669)   c = draw_var, s3 = scale
        load_tile<DPL, W>(a, A.a); load_tile<DPL, W>(b, A.b); load_tile<DPL, W>(c, A.c);
        if (A.op == OP_STD_DRAW_GRAD) load_tile<DPL, W>(d, A.d);
#pragma unroll
        for (int k = 0; k < DPL; ++k) {
            double val = A.op == OP_STD_DRAW_GRAD ? __builtin_sqrt(c.a[k] / d.a[k]) : c.a[k] * A.s3;
            if (!is_finite(val) | (val == 0.0)) {
                if (A.flag) { b.a[k] = __builtin_sqrt(A.s0); a.a[k] = __builtin_sqrt(1.0 / A.s0); }
            } else {
                val = clampd(val, A.s1, A.s2);
                b.a[k] = __builtin_sqrt(val);
                a.a[k] = __builtin_sqrt(1.0 / val);
            }
            if (!valid(k)) { a.a[k] = 0.0; b.a[k] = 0.0; }
        }
        store_tile<DPL, W>(a, A.a); store_tile<DPL, W>(b, A.b);
        break;
    case OP_STD_GRAD:      // M9  (cpu_math.rs:710-738)   a = inv_std, b = std, c = gradient
        load_tile<DPL, W>(c, A.c);
#pragma unroll
        for (int k = 0; k < DPL; ++k) {
            double val = 1.0 / clampd(__builtin_fabs(c.a[k]), A.s1, A.s2);
            if (!is_finite(val)) val = A.s0;
            b.a[k] = valid(k) ? __builtin_sqrt(val) : 0.0;
            a.a[k] = valid(k) ? __builtin_sqrt(1.0 / val) : 0.0;
        }
        store_tile<DPL, W>(a, A.a); store_tile<DPL, W>(b, A.b);
        break;
    case OP_SUM_LN: {      // M11 sum ln a_i (cpu_math.rs:300-304), the engine's order
        load_tile<DPL, W>(a, A.a);
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < DPL; ++k) acc = acc + (valid(k) ? dlog_impl<false>(valid(k) ? a.a[k] : 1.0) : 0.0);
        acc = R.sum(acc);
        if (tid() == 0) A.out[0] = acc;
        break;
    }
    case OP_ALL_FINITE: case OP_ALL_FINITE_NONZERO: {   // M12 (cpu_math.rs:283-298)
        load_tile<DPL, W>(a, A.a);
        bool ok = true;
#pragma unroll
        for (int k = 0; k < DPL; ++k) ok = ok && (!valid(k) || (is_finite(a.a[k]) && (A.op == OP_ALL_FINITE || a.a[k] != 0.0)));
        ok = R.all(ok);
        if (tid() == 0) A.out[0] = ok ? 1.0 : 0.0;
        break;
    }
    case OP_LOGP: {        // M14 logp_array(position, gradient) (math.rs:46-50 -> CpuLogpFunc::logp)
        Dens dens;
        dens.init(A.params, dim, R);
        dens.set_lds(ldens);
        load_tile<DPL, W>(a, A.a);
        const double lp = dens.template eval<DPL, W>(a, b, dim, R);
        store_tile<DPL, W>(b, A.b);
        if (tid() == 0) { A.out[0] = lp; A.out[1] = 0.0; }
        break;
    }
    case OP_SQ_NORM_SUM: { // M15 sum (x + y)^2 (cpu_math.rs:235-243)
        load_tile<DPL, W>(a, A.a); load_tile<DPL, W>(b, A.b);
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < DPL; ++k) acc = acc + (a.a[k] + b.a[k]) * (a.a[k] + b.a[k]);
        acc = R.sum(acc);
        if (tid() == 0) A.out[0] = acc;
        break;
    }
    case OP_FILL:          // M13 fill_array
#pragma unroll
        for (int k = 0; k < DPL; ++k) a.a[k] = valid(k) ? A.s0 : 0.0;
        store_tile<DPL, W>(a, A.a);
        break;
    case OP_RECIP:         // M13 array_recip (cpu_math.rs:328-330)
        load_tile<DPL, W>(a, A.a);
#pragma unroll
        for (int k = 0; k < DPL; ++k) b.a[k] = valid(k) ? 1.0 / a.a[k] : 0.0;
        store_tile<DPL, W>(b, A.b);
        break;
    case OP_COPY:          // M13 copy_into
        load_tile<DPL, W>(a, A.a);
        store_tile<DPL, W>(a, A.b);
        break;
    case OP_STD_NORM_FLOW: {   // std_norm_flow (util.rs:507-589): pos_out = p cos e + v sin e ; vel = -p sin e + v cos e
        load_tile<DPL, W>(a, A.a); load_tile<DPL, W>(c, A.c);
        const double2 sc = dsincos(A.s0);
        const double es = sc.x, ec = sc.y, nes = -sc.x;
#pragma unroll
        for (int k = 0; k < DPL; ++k) {
            b.a[k] = valid(k) ? __builtin_fma(a.a[k], ec, c.a[k] * es) : 0.0;
            d.a[k] = valid(k) ? __builtin_fma(a.a[k], nes, c.a[k] * ec) : 0.0;
        }
        store_tile<DPL, W>(b, A.b); store_tile<DPL, W>(d, A.c);
        break;
    }
    case OP_STD_NORM_GRAD_FLOW:   // std_norm_grad_flow / _inplace (util.rs:591-741): vel_out = fma(e, pos + grad, vel)
        load_tile<DPL, W>(a, A.a); load_tile<DPL, W>(b, A.b); load_tile<DPL, W>(c, A.c);
#pragma unroll
        for (int k = 0; k < DPL; ++k) d.a[k] = valid(k) ? __builtin_fma(A.s0, a.a[k] + b.a[k], c.a[k]) : 0.0;
        store_tile<DPL, W>(d, A.d);
        break;
    case OP_ESH: {         // esh_momentum_update (cpu_math.rs:505-551)
        load_tile<DPL, W>(a, A.a); load_tile<DPL, W>(b, A.b);
        const double dke = esh_update_core<DPL, W>(a, b, A.s0, dim, R);
        store_tile<DPL, W>(b, A.b);
        if (tid() == 0) A.out[0] = dke;
        break;
    }
    case OP_NORMALIZE:     // array_normalize (cpu_math.rs:496-503)
        load_tile<DPL, W>(a, A.a);
        normalize_tile<DPL, W>(a, R);
        store_tile<DPL, W>(a, A.a);
        break;
    }
}

template <class Dens>
hipError_t launch_dens(int dpl, int w, const VArgs& A, hipStream_t st) {
#define NM_VL(D_, W_) hipLaunchKernelGGL((vec_op_kernel<D_, W_, Dens>), dim3(1), dim3(64 * W_), 0, st, A); return hipGetLastError();
    switch (w * 100 + dpl) {
    case 102: NM_VL(2, 1) case 104: NM_VL(4, 1) case 108: NM_VL(8, 1) case 116: NM_VL(16, 1)
    case 208: NM_VL(8, 2) case 216: NM_VL(16, 2) case 404: NM_VL(4, 4) case 416: NM_VL(16, 4)
    }
#undef NM_VL
    return hipErrorInvalidValue;
}
hipError_t launch_op(const nm_math* m, const VArgs& A) {
    switch (m->spec.kind) {
    case NM_LOGP_DIAG_NORMAL: return launch_dens<DiagNormal>(m->dpl, m->w, A, m->stream);
    case NM_LOGP_FUNNEL: return launch_dens<Funnel>(m->dpl, m->w, A, m->stream);
    case NM_LOGP_MVN_PREC: return launch_dens<MvnPrec>(m->dpl, m->w, A, m->stream);
    case NM_LOGP_EIGHT_SCHOOLS:
        if (m->dpl != 2 || m->w != 1) return hipErrorInvalidValue;
        hipLaunchKernelGGL((vec_op_kernel<2, 1, EightSchools>), dim3(1), dim3(64), 0, m->stream, A);
        return hipGetLastError();
    default: return launch_dens<IidNormal>(m->dpl, m->w, A, m->stream);
    }
}
thread_local std::string g_err;
nm_status mfail(nm_status st, const char* what, hipError_t e = hipSuccess) {
    g_err = what;
    if (e != hipSuccess) { g_err += ": "; g_err += hipGetErrorString(e); }
    return st;
}
nm_status run(nm_math* m, VArgs& A, double* h_out = nullptr, int n_out = 0) {
    A.dim = m->dim; A.out = m->d_out; A.params = m->d_params; A.zig_x = m->d_zig; A.zig_f = m->d_zig + 257;
    hipError_t e = launch_op(m, A);
    if (e == hipSuccess && n_out) e = hipMemcpyAsync(h_out, m->d_out, n_out * sizeof(double), hipMemcpyDeviceToHost, m->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(m->stream);
    return e == hipSuccess ? NM_OK : mfail(NM_ERR_HIP, "nm_vec operation", e);
}
}  // namespace

extern "C" const char* nm_math_last_error(void) { return g_err.c_str(); }
extern "C" void nm_math_destroy(nm_math* m);

#include "zig_tables.hpp"
extern "C" nm_status nm_math_create(const nm_logp_spec* logp, nm_math** out) {
    if (!logp || !out) return mfail(NM_ERR_INVALID_ARG, "null argument");
    *out = nullptr;
    if (logp->kind > NM_LOGP_MVN_PREC) return mfail(NM_ERR_UNSUPPORTED, "nm_math covers the built-in densities");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return mfail(NM_ERR_NO_DEVICE, "no HIP device available; there is no CPU fallback");
    uint64_t dpl = 0, w = 0;
    if (nm_pick_tiling(logp->dim, logp->kind == NM_LOGP_EIGHT_SCHOOLS ? 2 : 0, logp->kind == NM_LOGP_EIGHT_SCHOOLS ? 1 : 0, &dpl, &w) != NM_OK)
        return mfail(NM_ERR_UNSUPPORTED, "no tiling for this dim");
    nm_math* m = new (std::nothrow) nm_math();
    if (!m) return mfail(NM_ERR_HIP, "out of host memory");
    m->spec = *logp; m->dim = logp->dim; m->dpl = (int)dpl; m->w = (int)w; m->dpad = 64ull * dpl * w;
    static const double X[257] = NM_ZIG_NORM_X, F[257] = NM_ZIG_NORM_F;
    // Stream discipline (DESIGN "Stream discipline"): every fill, copy and kernel of this Math runs on m->stream; the null stream
    // is never used (it is not ordered against a hipStreamNonBlocking stream).
    hipError_t e = hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipMalloc(&m->d_params, (logp->n_params ? logp->n_params : 1) * sizeof(double));
    if (e == hipSuccess && logp->n_params) e = hipMemcpyAsync(m->d_params, logp->h_params, logp->n_params * sizeof(double), hipMemcpyHostToDevice, m->stream);
    if (e == hipSuccess) e = hipMalloc(&m->d_zig, 2 * 257 * sizeof(double));
    if (e == hipSuccess) e = hipMemcpyAsync(m->d_zig, X, sizeof X, hipMemcpyHostToDevice, m->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(m->d_zig + 257, F, sizeof F, hipMemcpyHostToDevice, m->stream);
    if (e == hipSuccess) e = hipMalloc(&m->d_out, 8 * sizeof(double));
    if (e == hipSuccess) e = hipStreamSynchronize(m->stream);      // (h_params belongs to the caller: consumed before returning)
    if (e != hipSuccess) { nm_math_destroy(m); return mfail(NM_ERR_HIP, "nm_math_create", e); }
    m->spec.h_params = nullptr;
    *out = m;
    return NM_OK;
}
extern "C" void nm_math_destroy(nm_math* m) {
    if (!m) return;
    if (m->d_params) (void)hipFree(m->d_params);
    if (m->d_zig) (void)hipFree(m->d_zig);
    if (m->d_out) (void)hipFree(m->d_out);
    if (m->stream) (void)hipStreamDestroy(m->stream);
    delete m;
}
extern "C" uint64_t nm_math_dim(const nm_math* m) { return m ? m->dim : 0; }
extern "C" uint64_t nm_math_threads(const nm_math* m) { return m ? 64ull * (uint64_t)m->w : 0; }

extern "C" nm_status nm_vec_new(nm_math* m, nm_vec** out) {          // new_array: zeros (math.rs:24)
    if (!m || !out) return mfail(NM_ERR_INVALID_ARG, "null argument");
    nm_vec* v = new (std::nothrow) nm_vec{nullptr};
    if (!v) return mfail(NM_ERR_HIP, "out of host memory");
    // The zero fill runs on the Math's own stream, like every later operation on the vector.  (Round 2 used a null-stream
    // hipMemset here: not ordered against m->stream, it could land after the first kernel's store into the new vector —
    // reproduced by tools/probes/vec_new_race.py.)  The wait keeps nm_vec_free / another handle's use trivially safe.
    hipError_t e = hipMalloc(&v->d, m->dpad * sizeof(double));
    if (e == hipSuccess) e = hipMemsetAsync(v->d, 0, m->dpad * sizeof(double), m->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(m->stream);
    if (e != hipSuccess) { if (v->d) (void)hipFree(v->d); delete v; return mfail(NM_ERR_HIP, "nm_vec_new", e); }
    *out = v;
    return NM_OK;
}
extern "C" void nm_vec_free(nm_vec* v) { if (v) { if (v->d) (void)hipFree(v->d); delete v; } }
extern "C" nm_status nm_vec_read_from_slice(nm_math* m, nm_vec* dst, const double* h) {
    if (!m || !dst || !h) return mfail(NM_ERR_INVALID_ARG, "null argument");
    hipError_t e = hipMemcpyAsync(dst->d, h, m->dim * sizeof(double), hipMemcpyHostToDevice, m->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(m->stream);
    return e == hipSuccess ? NM_OK : mfail(NM_ERR_HIP, "read_from_slice", e);
}
extern "C" nm_status nm_vec_write_to_slice(nm_math* m, const nm_vec* src, double* h) {
    if (!m || !src || !h) return mfail(NM_ERR_INVALID_ARG, "null argument");
    hipError_t e = hipMemcpyAsync(h, src->d, m->dim * sizeof(double), hipMemcpyDeviceToHost, m->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(m->stream);
    return e == hipSuccess ? NM_OK : mfail(NM_ERR_HIP, "write_to_slice", e);
}
#define NM_CHECK(...) do { const void* ps_[] = {__VA_ARGS__}; for (const void* p_ : ps_) if (!p_) return mfail(NM_ERR_INVALID_ARG, "null argument"); } while (0)
extern "C" nm_status nm_vec_copy_into(nm_math* m, const nm_vec* src, nm_vec* dst) {
    NM_CHECK(m, src, dst); VArgs A{}; A.op = OP_COPY; A.a = src->d; A.b = dst->d; return run(m, A);
}
extern "C" nm_status nm_vec_fill_array(nm_math* m, nm_vec* dst, double value) {
    NM_CHECK(m, dst); VArgs A{}; A.op = OP_FILL; A.a = dst->d; A.s0 = value; return run(m, A);
}
extern "C" nm_status nm_vec_array_recip(nm_math* m, const nm_vec* a, nm_vec* dest) {
    NM_CHECK(m, a, dest); VArgs A{}; A.op = OP_RECIP; A.a = a->d; A.b = dest->d; return run(m, A);
}
extern "C" nm_status nm_vec_axpy_out(nm_math* m, const nm_vec* x, const nm_vec* y, double a, nm_vec* out) {
    NM_CHECK(m, x, y, out); VArgs A{}; A.op = OP_AXPY_OUT; A.a = x->d; A.b = y->d; A.c = out->d; A.s0 = a; return run(m, A);
}
extern "C" nm_status nm_vec_axpy(nm_math* m, const nm_vec* x, nm_vec* y, double a) {
    NM_CHECK(m, x, y); VArgs A{}; A.op = OP_AXPY; A.a = x->d; A.b = y->d; A.s0 = a; return run(m, A);
}
extern "C" nm_status nm_vec_array_mult(nm_math* m, const nm_vec* a, const nm_vec* b, nm_vec* dest) {
    NM_CHECK(m, a, b, dest); VArgs A{}; A.op = OP_MULT; A.a = a->d; A.b = b->d; A.c = dest->d; return run(m, A);
}
extern "C" nm_status nm_vec_array_vector_dot(nm_math* m, const nm_vec* a, const nm_vec* b, double* out) {
    NM_CHECK(m, a, b, out); VArgs A{}; A.op = OP_DOT; A.a = a->d; A.b = b->d; return run(m, A, out, 1);
}
extern "C" nm_status nm_vec_scalar_prods3(nm_math* m, const nm_vec* p1, const nm_vec* n1, const nm_vec* p2, const nm_vec* x,
                                          const nm_vec* y, double out[2]) {
    NM_CHECK(m, p1, n1, p2, x, y, out);
    VArgs A{}; A.op = OP_PRODS3; A.a = p1->d; A.b = n1->d; A.c = p2->d; A.d = x->d; A.e = y->d; return run(m, A, out, 2);
}
extern "C" nm_status nm_vec_array_gaussian(nm_math* m, const uint8_t key[32], uint64_t* stream_pos, nm_vec* dest, const nm_vec* stds) {
    NM_CHECK(m, key, stream_pos, dest, stds);
    if (m->dim > 64ull * m->w * m->dpl) return mfail(NM_ERR_UNSUPPORTED, "dim");
    VArgs A{}; A.op = OP_GAUSSIAN; A.a = dest->d; A.b = stds->d; A.pos = *stream_pos;
    for (int i = 0; i < 8; ++i) A.key[i] = (uint32_t)key[4 * i] | ((uint32_t)key[4 * i + 1] << 8) | ((uint32_t)key[4 * i + 2] << 16) | ((uint32_t)key[4 * i + 3] << 24);
    double o = 0.0;
    nm_status st = run(m, A, &o, 1);
    if (st == NM_OK) memcpy(stream_pos, &o, 8);
    return st;
}
extern "C" nm_status nm_vec_array_update_variance(nm_math* m, nm_vec* mean, nm_vec* variance, const nm_vec* value, double diff_scale) {
    NM_CHECK(m, mean, variance, value);
    VArgs A{}; A.op = OP_UPD_VAR; A.a = mean->d; A.b = variance->d; A.c = value->d; A.s0 = diff_scale; return run(m, A);
}
extern "C" nm_status nm_vec_array_update_var_inv_std_draw_grad(nm_math* m, nm_vec* inv_std, nm_vec* std_, const nm_vec* draw_var, const nm_vec* grad_var,
                                                               uint64_t has_fill, double fill, double clamp_lo, double clamp_hi) {
    NM_CHECK(m, inv_std, std_, draw_var, grad_var);
    VArgs A{}; A.op = OP_STD_DRAW_GRAD; A.a = inv_std->d; A.b = std_->d; A.c = draw_var->d; A.d = grad_var->d; A.flag = (int)has_fill;
    A.s0 = fill; A.s1 = clamp_lo; A.s2 = clamp_hi; return run(m, A);
}
extern "C" nm_status nm_vec_array_update_var_inv_std_draw(nm_math* m, nm_vec* inv_std, nm_vec* std_, const nm_vec* draw_var, double scale,
                                                          uint64_t has_fill, double fill, double clamp_lo, double clamp_hi) {
    NM_CHECK(m, inv_std, std_, draw_var);
    VArgs A{}; A.op = OP_STD_DRAW; A.a = inv_std->d; A.b = std_->d; A.c = draw_var->d; A.flag = (int)has_fill;
    A.s0 = fill; A.s1 = clamp_lo; A.s2 = clamp_hi; A.s3 = scale; return run(m, A);
}
extern "C" nm_status nm_vec_array_update_var_inv_std_grad(nm_math* m, nm_vec* inv_std, nm_vec* std_, const nm_vec* gradient, double fill,
                                                          double clamp_lo, double clamp_hi) {
    NM_CHECK(m, inv_std, std_, gradient);
    VArgs A{}; A.op = OP_STD_GRAD; A.a = inv_std->d; A.b = std_->d; A.c = gradient->d; A.s0 = fill; A.s1 = clamp_lo; A.s2 = clamp_hi; return run(m, A);
}
extern "C" nm_status nm_vec_array_sum_ln(nm_math* m, const nm_vec* a, double* out) {
    NM_CHECK(m, a, out); VArgs A{}; A.op = OP_SUM_LN; A.a = a->d; return run(m, A, out, 1);
}
extern "C" nm_status nm_vec_array_all_finite(nm_math* m, const nm_vec* a, uint64_t nonzero_too, uint64_t* out) {
    NM_CHECK(m, a, out);
    VArgs A{}; A.op = nonzero_too ? OP_ALL_FINITE_NONZERO : OP_ALL_FINITE; A.a = a->d;
    double o = 0.0;
    nm_status st = run(m, A, &o, 1);
    *out = o != 0.0;
    return st;
}
extern "C" nm_status nm_vec_logp_array(nm_math* m, const nm_vec* position, nm_vec* gradient, double* logp, uint64_t* status) {
    NM_CHECK(m, position, gradient, logp);
    VArgs A{}; A.op = OP_LOGP; A.a = position->d; A.b = gradient->d;
    double o[2] = {0.0, 0.0};
    nm_status st = run(m, A, o, 2);
    *logp = o[0];
    if (status) *status = 0;       // the built-in densities cannot fail (LogpErr = never)
    return st;
}
extern "C" nm_status nm_vec_sq_norm_sum(nm_math* m, const nm_vec* x, const nm_vec* y, double* out) {
    NM_CHECK(m, x, y, out); VArgs A{}; A.op = OP_SQ_NORM_SUM; A.a = x->d; A.b = y->d; return run(m, A, out, 1);
}

// the Math methods of the non-Euclidean KineticEnergyKinds (math.rs:155-200)
extern "C" nm_status nm_vec_std_norm_flow(nm_math* m, const nm_vec* pos, nm_vec* pos_out, nm_vec* vel, double epsilon) {
    NM_CHECK(m, pos, pos_out, vel); VArgs A{}; A.op = OP_STD_NORM_FLOW; A.a = pos->d; A.b = pos_out->d; A.c = vel->d; A.s0 = epsilon; return run(m, A);
}
extern "C" nm_status nm_vec_std_norm_grad_flow(nm_math* m, const nm_vec* pos, const nm_vec* grad, const nm_vec* vel, nm_vec* vel_out, double epsilon) {
    NM_CHECK(m, pos, grad, vel, vel_out);
    VArgs A{}; A.op = OP_STD_NORM_GRAD_FLOW; A.a = pos->d; A.b = grad->d; A.c = vel->d; A.d = vel_out->d; A.s0 = epsilon; return run(m, A);
}
extern "C" nm_status nm_vec_esh_momentum_update(nm_math* m, const nm_vec* gradient, nm_vec* momentum, double step_size, double* kinetic_energy_change) {
    NM_CHECK(m, gradient, momentum, kinetic_energy_change);
    if (m->dim < 2) return mfail(NM_ERR_INVALID_ARG, "ESH dynamics requires at least 2 dimensions");
    VArgs A{}; A.op = OP_ESH; A.a = gradient->d; A.b = momentum->d; A.s0 = step_size; return run(m, A, kinetic_energy_change, 1);
}
extern "C" nm_status nm_vec_array_normalize(nm_math* m, nm_vec* v) {
    NM_CHECK(m, v); VArgs A{}; A.op = OP_NORMALIZE; A.a = v->d; return run(m, A);
}
