// nuts_group_impl.hpp — body of the small-chain draw kernels (see nuts_group.hpp), included once per group size:
// NM_GS lanes per chain, namespace NM_GNS.  No include guard.
namespace nm {
namespace NM_GNS {

constexpr int GS = NM_GS;         // lanes per chain: 8 (dim <= 16), 16 (dim <= 32), 32 (dim <= 64)
constexpr int GPW = 64 / GS;      // chains per wavefront
constexpr uint32_t GMASK = GS == 32 ? 0xffffffffu : ((1u << (GS & 31)) - 1u);
using grp::GMAXDEPTH;

// exp / ln / ln_1p INLINED in these kernels (they hide nm::dexp & co.: same operations, same bits): a real call starts by
// waiting for every outstanding memory operation of the wavefront, and here 2 wavefronts per SIMD have little else to hide
// that wait behind — measured on K4: +8 % (8192 chains), +4.5 % (65536); the one-chain kernels keep the calls (K2: inlined -3 %)
NM_DEV double dexp(double x) { return dexp_impl<false>(x); }
NM_DEV double dlog(double x) { return dlog_impl<false>(x); }
NM_DEV double dlog1p(double x) { return dlog1p_impl<false>(x); }
NM_DEV double logaddexp_lane(double a, double b) {       // per-lane logaddexp (reference src/math/util.rs:6-19)
    if (a == b) return a + dlog(2.0);
    double diff = a - b;
    if (diff > 0.) return a + dlog1p(dexp(-diff));
    if (diff < 0.) return b + dlog1p(dexp(diff));
    return diff;
}

NM_DEV int gl() { return (int)(threadIdx.x & (unsigned)(GS - 1)); }
NM_DEV int gg() { return (int)((threadIdx.x & 63u) / (unsigned)GS); }
NM_DEV int gbase() { return (int)(threadIdx.x & (unsigned)(64 - GS)); }     // first lane of my group
NM_DEV double gsum(double x) {       // the first log2(GS) steps of wave_sum: the rest would add the +0.0 of the lanes beyond the chain
    x = x + dpp_mov<0xB1>(x);    // xor 1
    x = x + dpp_mov<0x4E>(x);    // xor 2
    x = x + dpp_mov<0x141>(x);   // row_half_mirror = xor 4 on quad sums
    if (GS >= 16) x = x + dpp_mov<0x140>(x);   // row_mirror = xor 8
    if (GS == 32) {              // the two row sums, lower row first: wave_sum's (r0 + r1)
        const int b = gbase();
        const double r0 = __hiloint2double(__shfl(__double2hiint(x), b), __shfl(__double2loint(x), b));
        const double r1 = __hiloint2double(__shfl(__double2hiint(x), b + 16), __shfl(__double2loint(x), b + 16));
        x = r0 + r1;
    }
    return x;
}
NM_DEV void gsum2(double& a, double& b) { a = gsum(a); b = gsum(b); }
NM_DEV double gbcast(double x, int j) {          // the value lane j of my group holds
    const int src = gbase() | j;
    const int lo = __shfl(__double2loint(x), src), hi = __shfl(__double2hiint(x), src);
    return __hiloint2double(hi, lo);
}
NM_DEV uint64_t gbcast_u64(uint64_t x, int j) {
    const int src = gbase() | j;
    const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)x, src), hi = (uint32_t)__shfl((int)(uint32_t)(x >> 32), src);
    return ((uint64_t)hi << 32) | lo;
}

// what the group form of a USER density (include/nuts_amd.h "User densities") works with: its lane holds elements
// 2 lane(), 2 lane() + 1 of the chain's vectors
struct Lanes {
    static constexpr int kLanes = GS;
    NM_DEV static int lane() { return gl(); }
    NM_DEV static double sum(double x) { return gsum(x); }                 // over the chain's lanes, the engine's order
    NM_DEV static double bcast(double x, int j) { return gbcast(x, j); }   // the value lane j of this chain holds
};

struct alignas(16) GroupShared {
    uint32_t rng_cache[GPW][16 * GS];         // GS ChaCha blocks per chain
    double samp[GPW][2 * GS];                     // stream-ordered normals of the momentum refresh
    double xs[GPW][2 * GS];                   // the position, visible to the group (densities that need all of x)
    double l1z[GPW][2 * GS], l1v[GPW][2 * GS];        // L[1] end point
    PendEntry pend[GPW][GMAXDEPTH + 1];
    ChainScalars sc[GPW];
};

// ---- densities (group forms of IidNormal / DiagNormal / EightSchools: same operations, group-relative lanes) ----
struct GIidNormal {
    double mu;
    NM_DEV void set_lds(double*) {}
    NM_DEV void init(const double* params, int) { mu = params[0]; }
    NM_DEV double eval(const double (&x)[2], double (&gx)[2], int dim) const {
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const bool valid = 2 * gl() + k < dim;
            const double diff = x[k] - mu;
            const double term = -0.5 * diff * diff;
            gx[k] = valid ? -diff : 0.0;
            acc = acc + (valid ? term : 0.0);
        }
        return gsum(acc);
    }
};
struct GDiagNormal {
    NM_DEV void set_lds(double*) {}
    const double* prec;
    double norm;
    NM_DEV void init(const double* params, int dim) {
        prec = params;
        double acc = 0.0;
        for (int j = 0; j < 2; ++j) {
            const int d = 2 * gl() + j;
            acc = acc + (d < dim ? dlog(params[d < dim ? d : 0]) : 0.0);
        }
        const double log_det_p = gsum(acc);
        norm = -0.5 * ((double)dim * dlog(6.283185307179586) - log_det_p);
    }
    NM_DEV double eval(const double (&x)[2], double (&gx)[2], int dim) const {
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int d = 2 * gl() + k;
            const bool valid = d < dim;
            const double p = valid ? prec[d] : 0.0;
            const double px = p * x[k];
            gx[k] = valid ? -px : 0.0;
            acc = acc + (valid ? x[k] * px : 0.0);
        }
        const double quad = -0.5 * gsum(acc);
        return quad + norm;
    }
};
struct GEightSchools {
    NM_DEV void set_lds(double*) {}
    const double* par;
    double yk[2], sk[2];           // this lane's y_i and sigma_i (read once: they were two loads from global memory per element and leapfrog)
    NM_DEV void init(const double* params, int) {
        par = params;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int d = 2 * gl() + j;
            const int i = (d >= 2 && d < 10) ? d - 2 : 0;
            yk[j] = params[i]; sk[j] = params[8 + i];
        }
    }
    NM_DEV double eval(const double (&x)[2], double (&gx)[2], int) const {
        const double mu = gbcast(x[0], 0), lt = gbcast(x[1], 0);
        const double tau = dexp(lt);
        const double t5 = (tau / 5.0) * (tau / 5.0);
        const double prior_tau = lt - dlog1p(t5);
        double term[2], dr[2], drth[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int d = 2 * gl() + j;
            const bool school = d >= 2 && d < 10;
            const int i = school ? d - 2 : 0;
            const double th = x[j];
            (void)i;
            const double sg = sk[j];
            const double r = (yk[j] - (mu + tau * th)) / sg;
            term[j] = -0.5 * th * th - 0.5 * r * r;
            dr[j] = r / sg;
            drth[j] = dr[j] * th;
            if (!school) { term[j] = 0.0; dr[j] = 0.0; drth[j] = 0.0; }
        }
        double gmu = 0.0, gtl = 0.0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int l = (2 + i) >> 1, j = (2 + i) & 1;
            gmu = gmu + gbcast(j ? dr[1] : dr[0], l);
            gtl = gtl + gbcast(j ? drth[1] : drth[0], l);
        }
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int d = 2 * gl() + k;
            double t = 0.0, g = 0.0;
            if (d == 0) { t = -mu * mu / 50.0; g = -mu / 25.0 + gmu; }
            else if (d == 1) { t = prior_tau; g = 1.0 - 2.0 * t5 / (1.0 + t5) + gtl * tau; }
            else if (d < 10) { t = term[k]; g = -x[k] + dr[k] * tau; }
            gx[k] = g;
            acc = acc + t;
        }
        return gsum(acc);
    }
};
struct GFunnel {
    NM_DEV void set_lds(double*) {}
    NM_DEV void init(const double*, int) {}
    NM_DEV double eval(const double (&x)[2], double (&gx)[2], int dim) const {
        const double v = gbcast(x[0], 0);
        const double kk = (double)(dim - 1);
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int d = 2 * gl() + k;
            const bool in = d >= 1 && d < dim;
            acc = acc + (in ? x[k] * x[k] : 0.0);
        }
        const double ss = gsum(acc);
        const double ev = dexp(-v);
        const double g0 = -v / 9.0 - 0.5 * kk + 0.5 * ev * ss;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int d = 2 * gl() + k;
            gx[k] = d == 0 ? g0 : (d < dim ? -ev * x[k] : 0.0);
        }
        return -v * v / 18.0 - 0.5 * kk * v - 0.5 * ev * ss;
    }
};
struct GMvnPrec {            // y_d = sum_j P[j][d] x_j, j ascending, one fma per term; x_j published through the group's LDS row
    const double* P;
    double* xs;
    NM_DEV void set_lds(double* lds) { xs = lds; }
    NM_DEV void init(const double* params, int) { P = params; }
    NM_DEV double eval(const double (&x)[2], double (&gx)[2], int dim) const {
        asm volatile("" ::: "memory");
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int d = 2 * gl() + k;
            xs[d] = d < dim ? x[k] : 0.0;
        }
        asm volatile("" ::: "memory");
        double y[2] = {0.0, 0.0};
        for (int j = 0; j < dim; ++j) {
            const double xj = xs[j];
            const double* row = P + (size_t)j * (size_t)dim;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int d = 2 * gl() + k;
                const double p = d < dim ? row[d] : 0.0;
                y[k] = __builtin_fma(p, xj, y[k]);
            }
        }
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const bool valid = 2 * gl() + k < dim;
            gx[k] = valid ? -y[k] : 0.0;
            acc = acc + (valid ? x[k] * y[k] : 0.0);
        }
        return -0.5 * gsum(acc);
    }
};
template <class Dens> struct GroupDensity { using type = void; };
// KinWrap<D> (the kernels with the non-Euclidean KineticEnergyKinds compiled in, nuts_kernels.hpp): the density's own group form; the kinds
// are switched on in the group code by the marker GKin around it (round 4: VERDICT r03 item 8 / "missing" 4)
template <class D> struct GroupDensity<KinWrap<D>> { using type = typename GroupDensity<D>::type; };
template <class GD> struct GKin : GD {};
template <class GD> struct gkin_trait { static constexpr bool value = false; };
template <class GD> struct gkin_trait<GKin<GD>> { static constexpr bool value = true; };
template <> struct GroupDensity<IidNormal> { using type = GIidNormal; };
template <> struct GroupDensity<DiagNormal> { using type = GDiagNormal; };
template <> struct GroupDensity<Funnel> { using type = GFunnel; };
template <> struct GroupDensity<MvnPrec> { using type = GMvnPrec; };
template <> struct GroupDensity<EightSchools> { using type = std::conditional<GS == 8, GEightSchools, void>::type; };   // dim 10

// ---- the chain's generator, one copy per group (same stream as DevRng) ----
// one refill = GS ChaCha blocks, one per lane of the group; a call, not an inlined copy at each of the generator's uses
// (the block function is ~400 instructions and refills are rare)
// (INLINED on purpose.  As a real call it sat inside divergent control flow — only the groups whose cache ran out call it, the
// other chains of the wavefront wait with their lanes off — and twice a change of unrelated code elsewhere in the kernel turned that
// into wrong results: lanes of the waiting groups came back from the call with live registers changed (round 4: the Euclidean
// kernels broke when the divergence test moved into a helper function; with this function inlined every variant is bit-exact
// again.  DESIGN §22.)  No out-of-line call may sit under a branch that is not uniform over the wavefront.)
static __device__ __forceinline__ void g_refill_blocks(const uint32_t* key, uint64_t first_block, uint32_t* cache) {
    uint32_t k[8], out[16];
#pragma unroll
    for (int i = 0; i < 8; ++i) k[i] = key[i];
    chacha8_block(k, first_block + (uint64_t)gl(), 0ull, out);
    uint4* dst = reinterpret_cast<uint4*>(cache + gl() * 16);
    dst[0] = make_uint4(out[0], out[1], out[2], out[3]);
    dst[1] = make_uint4(out[4], out[5], out[6], out[7]);
    dst[2] = make_uint4(out[8], out[9], out[10], out[11]);
    dst[3] = make_uint4(out[12], out[13], out[14], out[15]);
}
struct GRng {
    const uint32_t* key;   // LDS: the chain's ChaCha key (ChainScalars::key of the group's copy)
    uint64_t pos, base;
    uint32_t* cache;       // LDS [16 * GS], this group's
    NM_DEV bool has(uint64_t n) const { return pos >= base && (pos - base) + n <= (uint64_t)(16 * GS); }
    NM_DEV void refill() {
        base = pos & ~15ull;
        g_refill_blocks(key, base >> 4, cache);
        asm volatile("" ::: "memory");
    }
    // Make sure the next n words are in the cache.  When ANY chain of the wavefront (any that is executing this) has run out, EVERY one
    // re-bases its window at its own stream position (round 5): a refill costs the wavefront the same ~420 instructions however many of
    // its chains take part (the others' lanes would sit masked), the window is only a view of the stream (same words, same results), and
    // chains that refill together stay in step — ~0.3 refills per draw and wavefront instead of ~2.3 on K4.  The test is a ballot and a
    // scalar branch instead of an exec-masked region.
    NM_DEV void need(uint64_t n) { if (__ballot(!has(n)) != 0ull) refill(); }
    NM_DEV uint32_t next_u32() {
        need(1);
        const uint32_t w = cache[pos - base];
        pos += 1;
        return w;
    }
    NM_DEV uint64_t next_u64() {
        need(2);
        const uint64_t lo = cache[pos - base], hi = cache[pos - base + 1];
        pos += 2;
        return (hi << 32) | lo;
    }
    NM_DEV bool random_bool_std() { return (int32_t)next_u32() < 0; }
    NM_DEV double random_f64() { return (double)(next_u64() >> 11) * (1.0 / 9007199254740992.0); }
    NM_DEV int random_bool(double p) {
        if (!(p >= 0.0 && p < 1.0)) return p == 1.0 ? 1 : -1;
        const uint64_t p_int = (uint64_t)(p * 18446744073709551616.0);
        return next_u64() < p_int ? 1 : 0;
    }
};

NM_DEV double g_normal_slow(GRng& rng, uint64_t bits, ZigTables T) {     // normal_slow_path, per group
    for (;;) {
        const int i = (int)(bits & 0xff);
        const double u = u2d((bits >> 12) | 0x4000000000000000ull) - 3.0;
        const double x = u * T.x[i];
        if (__builtin_fabs(x) < T.x[i + 1]) return x;
        if (i == 0) {
            double xx = 1.0, yy = 0.0;
            while (-2.0 * yy < xx * xx) {
                const double a = u2d((rng.next_u64() >> 12) | 0x3ff0000000000000ull) - (1.0 - 2.220446049250313e-16 / 2.0);
                const double b = u2d((rng.next_u64() >> 12) | 0x3ff0000000000000ull) - (1.0 - 2.220446049250313e-16 / 2.0);
                xx = dlog(a) / ZIG_R;
                yy = dlog(b);
            }
            return u < 0.0 ? xx - ZIG_R : ZIG_R - xx;
        }
        if (T.f[i + 1] + (T.f[i] - T.f[i + 1]) * rng.random_f64() < dexp(-x * x / 2.0)) return x;
        bits = rng.next_u64();
    }
}
// `count` (<= 2 GS) StandardNormal variates of the chain's stream into samp[0..count): GS candidates per pass, the samples
// before the first rejected lane are the sequential ones, that lane's sample finishes on the slow path.
NM_DEV void g_fill_normals(GRng& rng, double* samp, int count, ZigTables T) {
    const int l = gl();
    int i = 0;
    while (i < count) {
        rng.need(2 * GS);
        const int nvalid = (count - i) < GS ? (count - i) : GS;
        const uint32_t off = (uint32_t)(rng.pos - rng.base) + 2u * (uint32_t)l;
        const uint64_t bits = ((uint64_t)rng.cache[off + 1] << 32) | rng.cache[off];
        const int zi = (int)(bits & 0xff);
        const double u = u2d((bits >> 12) | 0x4000000000000000ull) - 3.0;
        const double x = u * T.x[zi];
        const bool ok = __builtin_fabs(x) < T.x[zi + 1];
        const uint64_t fm = __ballot(l < nvalid && !ok);
        const uint32_t fail = (uint32_t)(fm >> (GS * gg())) & GMASK;
        const int nacc = fail ? (int)__builtin_ctz(fail) : nvalid;
        if (l < nacc) samp[i + l] = x;
        rng.pos += 2ull * (uint64_t)nacc;
        i += nacc;
        if (fail) {
            const uint64_t fbits = gbcast_u64(bits, nacc);
            rng.pos += 2;
            const double xs = g_normal_slow(rng, fbits, T);
            if (l == 0) samp[i] = xs;
            i += 1;
        }
    }
    asm volatile("" ::: "memory");
}

struct GPt { double z[2], v[2], g[2]; double logp, ke; int64_t idx; };

struct GAccept {             // AcceptCollector with the pending differences kept one per lane of the group
    double initial_energy, sum, sum_sym, max_energy_error;
    uint64_t count;
    double pend_d;
    int npend;
    NM_DEV void register_init(double e0) { initial_energy = e0; sum = 0.; sum_sym = 0.; count = 0; max_energy_error = 0.; pend_d = 0.; npend = 0; }
    NM_DEV void flush() {
        if (npend == 0) return;
        const double d = gl() < npend ? pend_d : 0.0;
        const double e = dexp(fmin_rs(d, 0.));
        const double es = 2. * e / (1. + dexp(d));
        for (int i = 0; i < npend; ++i) {
            sum = sum + gbcast(e, i);
            sum_sym = sum_sym + gbcast(es, i);
        }
        npend = 0;
    }
    NM_DEV void register_divergent() { flush(); sum = sum + 0.; sum_sym = sum_sym + 0.; count += 1; max_energy_error = -__builtin_inf(); }
    NM_DEV void register_ok(double end_energy) {
        const double diff = initial_energy - end_energy;
        if (gl() == npend) pend_d = diff;
        npend += 1;
        count += 1;
        if (__builtin_fabs(diff) > __builtin_fabs(max_energy_error)) max_energy_error = diff;
        if (npend == GS) flush();
    }
    NM_DEV double mean() { flush(); return sum / (double)count; }
    NM_DEV double mean_sym() { flush(); return sum_sym / (double)count; }
};

// ---- one chain of the group: pointers, the mass matrix elements of this lane, generator, density ----
template <class GD>
struct GCtx {
    const KParams& P;
    GD dens;
    GRng rng;
    ZigTables zig;
    double* pv;            // this chain's persistent slots, at this lane's elements
    double* sv;            // this block's scratch, at this chain's and lane's elements
    double* l1z; double* l1v; double* samp;
    PendEntry* pend;
    ChainScalars& sc;
    double sig[2], mu[2];
    int dim, md;
    __device__ GCtx(const KParams& p, ChainScalars& s) : P(p), sc(s) {}
    NM_DEV double* Pp(int slot) const { return pv + (size_t)slot * P.dpad; }
    NM_DEV double* Ss(int slot) const { return sv + (size_t)slot * P.dpad; }
    NM_DEV void ld(double (&t)[2], const double* p) const { const double2 q = *reinterpret_cast<const double2*>(p); t[0] = q.x; t[1] = q.y; }
    NM_DEV void st(const double (&t)[2], double* p) const { *reinterpret_cast<double2*>(p) = make_double2(t[0], t[1]); }
    NM_DEV double* edge_z(int id) const { return id == 0 ? Pp(P_Z) : Ss(EDGE0_Z + 3 * id); }
    NM_DEV double* edge_v(int id) const { return Ss(id == 0 ? (int)STAGE_V : EDGE0_V + 3 * id); }
    NM_DEV double* edge_g(int id) const { return id == 0 ? Pp(P_GZ) : Ss(EDGE0_G + 3 * id); }
};

// array_normalize / esh_momentum_update (reference src/math/cpu_math.rs:496-551) on a chain's lanes: normalize_tile / esh_update_core of
// nuts_kernels.hpp with two elements per lane (padding elements hold 0 and stay 0)
NM_DEV void g_normalize(double (&v)[2]) {
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < 2; ++k) acc = acc + v[k] * v[k];
    const double inv = 1.0 / __builtin_sqrt(gsum(acc));
#pragma unroll
    for (int k = 0; k < 2; ++k) v[k] *= inv;
}
NM_DEV double g_esh_update(const double (&g)[2], double (&p)[2], double step_size, int dim) {
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < 2; ++k) acc = acc + g[k] * g[k];
    const double grad_norm = __builtin_sqrt(gsum(acc));
    const double inv_grad_norm = 1.0 / grad_norm;
    acc = 0.0;
#pragma unroll
    for (int k = 0; k < 2; ++k) acc = acc + p[k] * g[k] * inv_grad_norm;
    const double momentum_proj = gsum(acc);
    const double dims_m1 = (double)(dim - 1);
    const double delta = step_size * grad_norm / dims_m1;
    const double zeta = dexp(-delta);
    const double coeff_g = (1.0 - zeta) * (1.0 + zeta + momentum_proj * (1.0 - zeta));
    const double coeff_p = 2.0 * zeta;
#pragma unroll
    for (int k = 0; k < 2; ++k) p[k] = coeff_g * (g[k] * inv_grad_norm) + coeff_p * p[k];
    g_normalize(p);
    const double arg = momentum_proj + (1.0 - momentum_proj) * zeta * zeta;
    return (delta - 6.93147180559945286227e-01 + dlog1p(arg)) * dims_m1;
}
// leapfrog_kin (nuts_kernels.hpp; KineticEnergyKind::ExactNormal and ::Microcanonical, src/math/util.rs:186-258, :507-741)
template <class GD>
NM_DEV void g_leapfrog_kin(GCtx<GD>& C, const GPt& s, GPt& o, double epsilon, double (&x)[2], double (&gx)[2]) {
    const bool micro = C.sc.kin == NM_TRAJ_MICROCANONICAL;
    const double half = epsilon / 2.;
    const double sqrt_n = __builtin_sqrt((double)C.dim);
    if (micro) {
        o.v[0] = s.v[0]; o.v[1] = s.v[1];
        o.ke = s.ke + g_esh_update(s.g, o.v, sqrt_n * epsilon / 2., C.dim);
        const double eps_n = epsilon * sqrt_n;
#pragma unroll
        for (int k = 0; k < 2; ++k) o.z[k] = __builtin_fma(eps_n, o.v[k], s.z[k]);
    } else {
        const double2 sc2 = dsincos_impl(epsilon);      // (inlined: see g_refill_blocks)
        const double es = sc2.x, ec = sc2.y, nes = -es;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const double vh = __builtin_fma(half, s.z[k] + s.g[k], s.v[k]);
            o.z[k] = __builtin_fma(s.z[k], ec, vh * es);
            o.v[k] = __builtin_fma(s.z[k], nes, vh * ec);
        }
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const double t = o.z[k] * C.sig[k];
        x[k] = __builtin_fma(1.0, C.mu[k], t);
    }
    o.logp = C.dens.eval(x, gx, C.dim);
#pragma unroll
    for (int k = 0; k < 2; ++k) o.g[k] = gx[k] * C.sig[k];
    if (micro) {
        o.ke = o.ke + g_esh_update(o.g, o.v, sqrt_n * epsilon / 2., C.dim);
    } else {
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            o.v[k] = __builtin_fma(half, o.z[k] + o.g[k], o.v[k]);
            acc = __builtin_fma(o.v[k], o.v[k], acc);
        }
        o.ke = 0.5 * gsum(acc);
    }
}
// leapfrog's divergence criterion (transformed_hamiltonian.rs:583-590)
template <class GD>
NM_DEV bool g_bad_energy(const GCtx<GD>& C, double energy_error, double max_energy_error) {
    if constexpr (gkin_trait<GD>::value) {
        if (C.sc.kin == NM_TRAJ_MICROCANONICAL) return (__builtin_fabs(energy_error) >= max_energy_error) | !is_finite(energy_error);
    }
    return (energy_error > max_energy_error) | !is_finite(energy_error);
}
// the kinetic energy a trajectory starts with (initialize_trajectory, transformed_hamiltonian.rs:697-727): initial_kinetic of nuts_kernels.hpp
template <class GD>
NM_DEV double g_initial_kinetic(GCtx<GD>& C, double (&v)[2]) {
    if constexpr (gkin_trait<GD>::value) {
        if (C.sc.kin == NM_TRAJ_MICROCANONICAL) {
            g_normalize(v);
            C.st(v, C.Ss(STAGE_V));
            return 0.0;
        }
    }
    double kacc = 0.0;
#pragma unroll
    for (int k = 0; k < 2; ++k) kacc = __builtin_fma(v[k], v[k], kacc);
    return 0.5 * gsum(kacc);
}
template <class GD>
NM_DEV void g_leapfrog(GCtx<GD>& C, const GPt& s, GPt& o, double epsilon) {
    if constexpr (gkin_trait<GD>::value) {
        if (C.sc.kin != NM_TRAJ_EUCLIDEAN) { double x_[2], gx_[2]; g_leapfrog_kin(C, s, o, epsilon, x_, gx_); return; }
    }
    const double half = epsilon / 2.;
    double x[2], gx[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const double vh = __builtin_fma(half, s.g[k], s.v[k]);
        o.v[k] = vh;
        o.z[k] = __builtin_fma(epsilon, vh, s.z[k]);
        const double t = o.z[k] * C.sig[k];
        x[k] = __builtin_fma(1.0, C.mu[k], t);
    }
    o.logp = C.dens.eval(x, gx, C.dim);
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        o.g[k] = gx[k] * C.sig[k];
        o.v[k] = __builtin_fma(half, o.g[k], o.v[k]);
        acc = __builtin_fma(o.v[k], o.v[k], acc);
    }
    o.ke = 0.5 * gsum(acc);
}

NM_DEV bool g_turning_regs(const GPt& a, const GPt& b, bool fwd) {
    double s1 = 0., s2 = 0.;
#pragma unroll
    for (int k = 0; k < 2; ++k) turn_acc(a.z[k], a.v[k], b.z[k], b.v[k], s1, s2);     // generation order; the direction flips the comparison
    gsum2(s1, s2);                                                                     // (nuts_kernels.hpp turning_regs: the reversed pair is the same sums negated)
    return turn_sign(fwd, s1) | turn_sign(fwd, s2);
}

template <class GD>
NM_DEV bool g_merge_weights(GCtx<GD>& C, double a_log_size, double b_log_size, bool is_main, double& total, bool& fatal) {
    // merge_into's arithmetic as one branch-free sequence (dev_math.hpp merge_math_impl; inlined: no out-of-line call under control flow that
    // is not uniform over the wavefront).  With several chains per wavefront every branch of the general routines was taken by some chain.
    C.rng.need(2);
    const uint32_t off_ = (uint32_t)(C.rng.pos - C.rng.base);
    const MergeOut mo = merge_math_impl(a_log_size, b_log_size, is_main ? 1u : 0u, C.rng.cache[off_], C.rng.cache[off_ + 1]);
    total = mo.total;
    C.rng.pos += (uint64_t)(mo.flags & 2u);
    if (mo.flags & 4u) fatal = true;
    return (mo.flags & 1u) != 0;
}

template <class GD>
NM_DEV int g_cand_to_pool(GCtx<GD>& C, uint32_t& used, const double (&z)[2]) {
    const int p = (int)__builtin_ctz(~used);
    used |= 1u << p;
    C.st(z, C.Ss(slot_C(C.md, p)));
    return p;
}

// nuts::draw for the group's chain: the port of nuts_transition (see there for the slot scheme and the merge order)
template <class GD>
NM_DEV uint64_t g_transition(GCtx<GD>& C, GAccept& col, DrawResult& R, double (&zc)[2]) {
    const nm_settings& s = C.P.s;
    ChainScalars& sc = C.sc;
    const int MD = C.md;
    GPt E, O;
    g_fill_normals(C.rng, C.samp, C.dim, C.zig);
#pragma unroll
    for (int k = 0; k < 2; ++k) E.v[k] = 2 * gl() + k < C.dim ? 1.0 * C.samp[2 * gl() + k] : 0.0;
    C.st(E.v, C.Ss(STAGE_V));
    if (sc.mm_id != sc.transform_id) {        // lazy re-whitening after the last mass-matrix update (diagonal.rs:210-221)
        double x[2], gx[2], isig[2];
        C.ld(x, C.Pp(P_X)); C.ld(gx, C.Pp(P_GX)); C.ld(isig, C.Pp(P_ISIG));
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const double t = __builtin_fma(-1.0, C.mu[k], x[k]);
            E.z[k] = isig[k] * t;
            E.g[k] = gx[k] * C.sig[k];
        }
        C.st(E.z, C.Pp(P_Z)); C.st(E.g, C.Pp(P_GZ));
        sc.logdet = sc.mm_logdet;
        sc.transform_id = sc.mm_id;
    } else {
        C.ld(E.z, C.Pp(P_Z));
        C.ld(E.g, C.Pp(P_GZ));
    }
    const double logdet = sc.logdet;
    const double ke_init = g_initial_kinetic(C, E.v);
    E.ke = ke_init;
    [[maybe_unused]] double left_ke = ke_init, right_ke = ke_init;   // the edges' kinetic_energy: an input of the microcanonical leapfrog only
    const double e0 = ke_init - (sc.logp + logdet);
    R.e0 = e0;
    col.register_init(e0);
    int left_slot = 0, right_slot = 0;
    bool o_is_edge = false;
    int o_edge_sign = 0;
    uint64_t depth = 0;
    double log_size = 0.;
    int64_t left_idx = 0, right_idx = 0;
    CandRef mc = {-1, sc.logp, ke_init, 0};
    uint32_t used = 0;

    uint64_t mindepth = s.mindepth, maxdepth = s.maxdepth;
    if (s.has_target_integration_time) {
        const double q = __builtin_ceil(s.target_integration_time / sc.step_size);
        const uint64_t max_steps = q >= 18446744073709551616.0 ? ~0ull : (q > 0 ? (uint64_t)q : 0ull);
        const uint64_t fl = 63 - __builtin_clzll(max_steps | 1ull);
        const uint64_t ce = ((max_steps & (max_steps - 1)) == 0) ? fl : fl + 1;
        mindepth = fl > s.mindepth ? fl : s.mindepth;
        const uint64_t xd = ce > mindepth ? ce : mindepth;
        maxdepth = xd < s.maxdepth ? xd : s.maxdepth;
    }
    R.diverging = false; R.reached_maxdepth = false; R.has_divergence_energy_error = false;
    R.divergence_energy_error = 0.; R.div_start_idx = 0;
    const bool want_div = C.P.out_div_start || C.P.out_div_start_grad || C.P.out_div_end;
    bool fatal = false;
    bool in_extra = false;
    uint64_t extra_left = 0;
    int sign = 1;

    for (;;) {
        bool check;
        if (!in_extra) {
            if (!(depth < maxdepth)) { R.reached_maxdepth = true; break; }
            sign = C.rng.random_bool_std() ? 1 : -1;
            check = (s.check_turning != 0) && !(depth < mindepth);
        } else {
            if (extra_left == 0) break;
            extra_left -= 1;
            check = false;
        }
        const bool fwd = sign > 0;
        const int64_t edge_idx = fwd ? right_idx : left_idx;
        const uint64_t nleaf = 1ull << depth;
        const uint32_t used_before = used;
        const double epsilon = (double)sign * sc.step_size * 1.0;
        int stop = STOP_NONE;
        double sub_log_size = 0.;
        CandRef sub_cand = {-2, 0., 0., 0};
        const bool reuse_edge = o_is_edge && o_edge_sign == sign;
        o_is_edge = false;

#define NM_G_ACCOUNT(START, PT, WOUT)                                                                      \
        {                                                                                                 \
            const double energy_ = (PT).ke - ((PT).logp + logdet);                                        \
            const double err_ = energy_ - e0;                                                             \
            if (g_bad_energy(C, err_, s.max_energy_error)) {                                              \
                col.register_divergent();                                                                 \
                R.diverging = true; R.has_divergence_energy_error = true; R.divergence_energy_error = err_; \
                R.div_start_idx = (PT).idx - (int64_t)sign;                                               \
                if (want_div) {          /* DivergenceInfo locations: the F[0] scratch pair is dead from here on */ \
                    C.st((START).z, C.Ss(slot_F(0))); C.st((PT).z, C.Ss(slot_F(0) + 1));                  \
                }                                                                                         \
                stop = STOP_DIVERGING;                                                                    \
            } else {                                                                                      \
                col.register_ok(energy_);                                                                 \
                WOUT = -err_;                                                                             \
            }                                                                                             \
        }

        if (depth == 0) {
            g_leapfrog(C, E, O, epsilon);
            O.idx = edge_idx + (int64_t)sign;
            NM_G_ACCOUNT(E, O, sub_log_size)
            sub_cand = {-2, O.logp, O.ke, O.idx};
        } else {
            if (!reuse_edge) {
                const int es = fwd ? right_slot : left_slot;
                C.ld(O.z, C.edge_z(es)); C.ld(O.v, C.edge_v(es)); C.ld(O.g, C.edge_g(es));
                if constexpr (gkin_trait<GD>::value) O.ke = fwd ? right_ke : left_ke;
            }
            for (uint64_t n = 0; n < nleaf; n += 2) {
                double wE = 0., wO = 0.;
                g_leapfrog(C, O, E, epsilon);
                E.idx = edge_idx + (int64_t)sign * (int64_t)(n + 1);
                NM_G_ACCOUNT(O, E, wE)
                if (stop != STOP_NONE) break;
                g_leapfrog(C, E, O, epsilon);
                O.idx = edge_idx + (int64_t)sign * (int64_t)(n + 2);
                NM_G_ACCOUNT(E, O, wO)
                if (stop != STOP_NONE) break;
                const uint64_t nn = n + 1;
                const int t = (int)__builtin_ctzll(~nn);
                uint32_t turn_bits = 0;
                if (check) {
                    if (g_turning_regs(E, O, fwd)) turn_bits |= 2u;
                    for (int k = 2; k <= t && turn_bits == 0; ++k) {
                        const uint64_t a_first = nn + 1 - (1ull << k);
                        const int fa = a_first == 0 ? (int)depth : (int)__builtin_ctzll(a_first);
                        double az[2], av[2], lz[2], lv[2], bz[2], bv[2];
                        C.ld(az, C.Ss(slot_F(fa))); C.ld(av, C.Ss(slot_F(fa) + 1));
                        if (k == 2) {
                            C.ld(lz, C.l1z); C.ld(lv, C.l1v);
                            bz[0] = E.z[0]; bz[1] = E.z[1]; bv[0] = E.v[0]; bv[1] = E.v[1];
                        } else {
                            C.ld(lz, C.Ss(slot_L(MD, k - 1))); C.ld(lv, C.Ss(slot_L(MD, k - 1) + 1));
                            C.ld(bz, C.Ss(slot_F(k - 1))); C.ld(bv, C.Ss(slot_F(k - 1) + 1));
                        }
                        double s1 = 0., s2 = 0., s3 = 0., s4 = 0., s5 = 0., s6 = 0.;
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            turn_acc(az[j], av[j], O.z[j], O.v[j], s1, s2);
                            turn_acc(lz[j], lv[j], O.z[j], O.v[j], s3, s4);
                            turn_acc(az[j], av[j], bz[j], bv[j], s5, s6);
                        }
                        gsum2(s1, s2); gsum2(s3, s4); gsum2(s5, s6);
                        if (turn_any6(fwd, s1, s2, s3, s4, s5, s6)) turn_bits |= 1u << k;
                    }
                }
                {
                    double total;
                    const bool take = g_merge_weights(C, wE, wO, false, total, fatal);
                    sub_cand = take ? CandRef{-2, O.logp, O.ke, O.idx} : CandRef{-3, E.logp, E.ke, E.idx};
                    sub_log_size = total;
                    if (fatal) { stop = STOP_FATAL; break; }
                    if (turn_bits & 2u) { stop = STOP_TURNING; break; }
                }
                for (int k = 2; k <= t; ++k) {
                    const PendEntry A = C.pend[k - 1];
                    double total;
                    const bool take = g_merge_weights(C, A.log_size, sub_log_size, false, total, fatal);
                    if (take) {
                        used &= ~(1u << A.cand_slot);
                    } else {
                        if (sub_cand.slot >= 0) used &= ~(1u << sub_cand.slot);
                        sub_cand = {A.cand_slot, A.cand_logp, A.cand_ke, A.cand_idx};
                    }
                    sub_log_size = total;
                    if (fatal) { stop = STOP_FATAL; break; }
                    if ((turn_bits >> k) & 1u) { stop = STOP_TURNING; break; }
                }
                if (stop != STOP_NONE) break;
                if ((n & 3) == 0 && depth > 1) {
                    const int fs = slot_F(n == 0 ? (int)depth : (int)__builtin_ctzll(n));
                    C.st(E.z, C.Ss(fs));
                    C.st(E.v, C.Ss(fs + 1));
                }
                if (n + 2 < nleaf) {
                    if (t == 1) { C.st(O.z, C.l1z); C.st(O.v, C.l1v); }
                    else { C.st(O.z, C.Ss(slot_L(MD, t))); C.st(O.v, C.Ss(slot_L(MD, t) + 1)); }
                    if (sub_cand.slot == -2) sub_cand.slot = g_cand_to_pool(C, used, O.z);
                    else if (sub_cand.slot == -3) sub_cand.slot = g_cand_to_pool(C, used, E.z);
                    PendEntry e;
                    e.log_size = sub_log_size; e.cand_logp = sub_cand.logp; e.cand_ke = sub_cand.ke;
                    e.cand_idx = sub_cand.idx; e.cand_slot = sub_cand.slot; e.pad = 0;
                    C.pend[t] = e;
                }
            }
        }
#undef NM_G_ACCOUNT
        if (stop == STOP_FATAL) { fatal = true; break; }
        if (stop == STOP_DIVERGING) { used = used_before; break; }
        if (stop == STOP_TURNING) {
            used = used_before;
            if (!in_extra) { in_extra = true; extra_left = s.extra_doublings; }
            continue;
        }
        // top-level U-turn tests of the finished sub-tree (last leaf O) against the main tree (src/nuts.rs:143-161)
        bool turning = false;
        if (check) {
            if (depth == 0) turning = g_turning_regs(E, O, fwd);
            else {
                double lz[2], lv[2], rz[2], rv[2], oz[2], ov[2];
                C.ld(lz, C.edge_z(left_slot)); C.ld(lv, C.edge_v(left_slot));
                C.ld(rz, C.edge_z(right_slot)); C.ld(rv, C.edge_v(right_slot));
                if (depth == 1) { oz[0] = E.z[0]; oz[1] = E.z[1]; ov[0] = E.v[0]; ov[1] = E.v[1]; }
                else { C.ld(oz, C.Ss(slot_F((int)depth))); C.ld(ov, C.Ss(slot_F((int)depth) + 1)); }
                double s1 = 0., s2 = 0., s3 = 0., s4 = 0., s5 = 0., s6 = 0.;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    // generation order (nuts_kernels.hpp top_level_turning): (left, O) (right, O) (X, first leaf), X = the tree's end FAR from `other`
                    turn_acc(lz[j], lv[j], O.z[j], O.v[j], s1, s2);
                    turn_acc(rz[j], rv[j], O.z[j], O.v[j], s3, s4);
                    turn_acc(fwd ? lz[j] : rz[j], fwd ? lv[j] : rv[j], oz[j], ov[j], s5, s6);
                }
                gsum2(s1, s2); gsum2(s3, s4); gsum2(s5, s6);
                turning = turn_any6(fwd, s1, s2, s3, s4, s5, s6);
            }
        }
        double total;
        const bool take = g_merge_weights(C, log_size, sub_log_size, true, total, fatal);
        if (fatal) break;
        if (take) {
            if (mc.slot >= 0) used &= ~(1u << mc.slot);
            if (sub_cand.slot == -2) sub_cand.slot = g_cand_to_pool(C, used, O.z);
            else if (sub_cand.slot == -3) sub_cand.slot = g_cand_to_pool(C, used, E.z);
            mc = sub_cand;
        } else if (sub_cand.slot >= 0) {
            used &= ~(1u << sub_cand.slot);
        }
        const bool more = in_extra ? extra_left > 0 : (turning ? s.extra_doublings > 0 : depth + 1 < maxdepth);
        if (more) {
            int ns = fwd ? right_slot : left_slot;
            const int other_side = fwd ? left_slot : right_slot;
            if (ns == 0) ns = other_side == 1 ? 2 : 1;
            C.st(O.z, C.edge_z(ns)); C.st(O.v, C.edge_v(ns)); C.st(O.g, C.edge_g(ns));
            if (fwd) right_slot = ns; else left_slot = ns;
            o_is_edge = true; o_edge_sign = sign;
        }
        if (fwd) right_idx = O.idx; else left_idx = O.idx;
        if constexpr (gkin_trait<GD>::value) { if (fwd) right_ke = O.ke; else left_ke = O.ke; }
        depth += 1;
        log_size = total;
        if (turning && !in_extra) { in_extra = true; extra_left = s.extra_doublings; }
    }
    R.depth = depth;
    R.chosen = mc;
    if (fatal) return NM_CHAIN_LOGP_FATAL;
    if (mc.slot >= 0) C.ld(zc, C.Ss(slot_C(MD, mc.slot)));
    return NM_CHAIN_OK;
}

// ---- warm-up (group forms of the adaptation in nuts_kernels.hpp: the same operations on per-lane scalars) ----
NM_DEV bool gall(bool ok) {
    const uint64_t m = __ballot(ok);
    return ((uint32_t)(m >> (GS * gg())) & GMASK) == GMASK;
}
// DualAverage::new (dual_avg.rs:44-53) or Adam::new (adam.rs:56-64)
NM_DEV void g_stepsize_adapt_reset(ChainScalars& sc, const nm_settings& s, double initial_step) {
    sc.log_step = dlog(initial_step);
    if (s.step_size_method == NM_STEP_ADAM) { sc.adam_m = 0.; sc.adam_v = 0.; sc.adam_t = 0; return; }
    sc.log_step_adapted = sc.log_step;
    sc.hbar = 0.;
    sc.mu = dlog(10. * initial_step);
    sc.da_count = 1;
}
// update_stepsize (reference src/stepsize/adapt.rs:235-267)
template <class GD>
NM_DEV void g_update_stepsize(GCtx<GD>& C, bool use_best_guess) {
    const nm_settings& s = C.P.s;
    const double step = s.step_size_method == NM_STEP_FIXED ? s.fixed_step_size
                      : s.step_size_method == NM_STEP_ADAM ? dexp(C.sc.log_step)
                      : (use_best_guess ? dexp(C.sc.log_step_adapted) : dexp(C.sc.log_step));
    if (s.has_jitter) {
        const double v12 = u2d((C.rng.next_u64() >> 12) | 0x3ff0000000000000ull);
        const double j = (v12 - 1.0) * C.P.jitter_scale + C.P.jitter_low;
        C.sc.step_size = step * j;
    } else {
        C.sc.step_size = step;
    }
}
// DualAverage::advance (dual_avg.rs:55-64) / Adam::advance (adam.rs:70-98)
template <class GD>
NM_DEV void g_update_estimator(GCtx<GD>& C, bool late) {
    const nm_settings& s = C.P.s;
    if (s.step_size_method == NM_STEP_FIXED) return;
    ChainScalars& sc = C.sc;
    const double accept_stat = late ? sc.last_sym_mean_tree_accept : sc.last_mean_tree_accept;
    if (s.step_size_method == NM_STEP_ADAM) {
        const double gradient = accept_stat - s.target_accept;
        sc.adam_t += 1;
        sc.adam_m = s.adam_beta1 * sc.adam_m + (1.0 - s.adam_beta1) * gradient;
        sc.adam_v = s.adam_beta2 * sc.adam_v + (1.0 - s.adam_beta2) * gradient * gradient;
        const double m_hat = sc.adam_m / (1.0 - powi_rs(s.adam_beta1, (int32_t)sc.adam_t));
        const double v_hat = sc.adam_v / (1.0 - powi_rs(s.adam_beta2, (int32_t)sc.adam_t));
        sc.log_step += s.adam_learning_rate * m_hat / (__builtin_sqrt(v_hat) + s.adam_epsilon);
        return;
    }
    const double w = 1. / ((double)sc.da_count + s.da_t0);
    sc.hbar = (1. - w) * sc.hbar + w * (s.target_accept - accept_stat);
    sc.log_step = sc.mu - sc.hbar * __builtin_sqrt((double)sc.da_count) / s.da_gamma;
    sc.log_step = fmin_rs(sc.log_step, C.P.ln_max_step);
    const double mk = dexp(-s.da_k * dlog((double)sc.da_count));
    sc.log_step_adapted = mk * sc.log_step + (1. - mk) * sc.log_step_adapted;
    sc.da_count += 1;
}
// RunningVariance::add_sample (adapt/diagonal.rs:31-44, cpu_math.rs:605-631)
NM_DEV void g_running_variance_add(double (&mean)[2], double (&var)[2], uint64_t new_count, const double (&value)[2]) {
    if (new_count == 1) { mean[0] = value[0]; mean[1] = value[1]; return; }
    const double diff_scale = 1.0 / (double)new_count;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const double diff = value[k] - mean[k];
        mean[k] = mean[k] + diff * diff_scale;
        var[k] = var[k] + diff * diff;
    }
}
template <class GD>
NM_DEV void g_commit_mass_matrix(GCtx<GD>& C, const double (&sig)[2], const double (&isig)[2], const double (&mu)[2]) {
    C.st(sig, C.Pp(P_SIG)); C.st(isig, C.Pp(P_ISIG)); C.st(mu, C.Pp(P_MU));
    C.sig[0] = sig[0]; C.sig[1] = sig[1]; C.mu[0] = mu[0]; C.mu[1] = mu[1];
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const bool valid = 2 * gl() + k < C.dim;
        acc = acc + (valid ? dlog(valid ? isig[k] : 1.0) : 0.0);
    }
    C.sc.mm_logdet = gsum(acc);
    C.sc.mm_id += 1;
}
// Strategy::adapt -> update_diag_draw_grad / update_diag_draw (adapt/diagonal.rs:161-196, diagonal.rs:85-131)
template <class GD>
NM_DEV bool g_mass_matrix_adapt(GCtx<GD>& C, const double (&dm)[2], const double (&dv)[2], const double (&gm)[2], const double (&gv)[2]) {
    if (C.sc.cnt_fg < 3) return false;
    double sig[2] = {C.sig[0], C.sig[1]}, isig[2], mu[2];
    C.ld(isig, C.Pp(P_ISIG));
    if (C.P.s.use_grad_based_estimate) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const bool valid = 2 * gl() + k < C.dim;
            double val = __builtin_sqrt(dv[k] / gv[k]);
            double sd = sig[k], isd = isig[k];
            if (!(!is_finite(val) | (val == 0.0))) {
                val = clampd(val, 1e-20, 1e20);
                sd = __builtin_sqrt(val);
                isd = __builtin_sqrt(1.0 / val);
            }
            const double var = sd * sd;
            double mean = var * gm[k];
            mean = __builtin_fma(1.0, dm[k], mean);
            sig[k] = valid ? sd : 0.0;
            isig[k] = valid ? isd : 0.0;
            mu[k] = valid ? mean : 0.0;
        }
    } else {
        const double scale = 1.0 / (double)C.sc.cnt_fg;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const bool valid = 2 * gl() + k < C.dim;
            const double d = dv[k] * scale;
            double sd = sig[k], isd = isig[k];
            if (!(!is_finite(d) | (d == 0.0))) {
                const double val = clampd(d, 1e-20, 1e20);
                sd = __builtin_sqrt(val);
                isd = __builtin_sqrt(1.0 / val);
            }
            sig[k] = valid ? sd : 0.0;
            isig[k] = valid ? isd : 0.0;
            mu[k] = valid ? dm[k] : 0.0;
        }
    }
    g_commit_mass_matrix(C, sig, isig, mu);
    return true;
}
// stepsize::Strategy::init (src/stepsize/adapt.rs:91-199): the step-size search at x, after the first mass-matrix update
template <class GD>
NM_DEV uint64_t g_stepsize_init(GCtx<GD>& C, const double (&x)[2]) {
    const nm_settings& s = C.P.s;
    if (s.step_size_method == NM_STEP_FIXED) { C.sc.step_size = s.fixed_step_size; return NM_CHAIN_OK; }
    GPt st;
    {   // Hamiltonian::init_state (transformed_hamiltonian.rs:640-661, check_all :310-324)
        double gx[2], isig[2];
        st.logp = C.dens.eval(x, gx, C.dim);
        C.ld(isig, C.Pp(P_ISIG));
        bool ok = true;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const double t = __builtin_fma(-1.0, C.mu[k], x[k]);
            st.z[k] = isig[k] * t;
            st.g[k] = gx[k] * C.sig[k];
            const bool valid = 2 * gl() + k < C.dim;
            ok = ok && (!valid || (is_finite(st.z[k]) && is_finite(st.g[k]) && st.g[k] != 0.0 && is_finite(gx[k]) && is_finite(x[k])));
        }
        if (!gall(ok)) return NM_CHAIN_BAD_INIT;
    }
    const double logdet = C.sc.mm_logdet;
    g_fill_normals(C.rng, C.samp, C.dim, C.zig);
#pragma unroll
    for (int k = 0; k < 2; ++k) st.v[k] = 2 * gl() + k < C.dim ? 1.0 * C.samp[2 * gl() + k] : 0.0;
    const double ke0 = g_initial_kinetic(C, st.v);
    st.ke = ke0;
    const double e0 = ke0 - (st.logp + logdet);
    GAccept col;
    C.sc.step_size = s.initial_step;
    int dir = 0;
    for (int it = 0; it < 101; ++it) {
        GPt o;
        const int sign = it == 0 ? 1 : dir;
        col.register_init(e0);
        g_leapfrog(C, st, o, (double)sign * C.sc.step_size * 1.0);
        const double energy = o.ke - (o.logp + logdet);
        const double err = energy - e0;
        if (g_bad_energy(C, err, 1000.0)) {
            if (it > 0) C.sc.step_size = s.initial_step;
            return NM_CHAIN_OK;
        }
        col.register_ok(energy);
        const double accept = col.mean();
        if (it == 0) { dir = accept > s.target_accept ? 1 : -1; continue; }
        if (dir > 0) {
            if ((accept <= s.target_accept) | (C.sc.step_size > 1e5)) { g_stepsize_adapt_reset(C.sc, s, C.sc.step_size); return NM_CHAIN_OK; }
            C.sc.step_size *= 2.;
        } else {
            if ((accept >= s.target_accept) | (C.sc.step_size < 1e-10)) { g_stepsize_adapt_reset(C.sc, s, C.sc.step_size); return NM_CHAIN_OK; }
            C.sc.step_size /= 2.;
        }
    }
    C.sc.step_size = s.initial_step;
    return NM_CHAIN_OK;
}
// GlobalStrategy::adapt (src/adapt_strategy.rs:121-222); x, gx = the chosen draw
template <bool TUNE, class GD>
NM_DEV uint64_t g_adapt(GCtx<GD>& C, GAccept& col, bool is_good, const double (&x)[2], const double (&gx)[2]) {
    const nm_settings& s = C.P.s;
    ChainScalars& sc = C.sc;
    const uint64_t draw = sc.draw_count;
    sc.last_mean_tree_accept = col.mean();
    sc.last_sym_mean_tree_accept = col.mean_sym();
    sc.last_n_steps = col.count;
    sc.last_max_energy_error = col.max_energy_error;
    if (!TUNE || draw >= s.num_tune) {     // the sampling kernel (TUNE = false) is only launched once every chain is there
        g_update_stepsize(C, true);
        sc.tuning = 0;
        return NM_CHAIN_OK;
    }
    if (draw < C.P.final_step_size_window) {
        const bool is_early = draw < C.P.early_end;
        if (!is_early && draw == C.P.early_end)
            sc.current_window_size = sc.current_window_size > sc.cnt_bg ? sc.current_window_size : sc.cnt_bg;
        const uint64_t switch_freq = is_early ? s.early_mass_matrix_switch_freq : sc.current_window_size;
        double fdm[2], fdv[2], fgm[2], fgv[2], bdm[2], bdv[2], bgm[2], bgv[2];
        C.ld(fdm, C.Pp(E_DM)); C.ld(fdv, C.Pp(E_DV)); C.ld(fgm, C.Pp(E_GM)); C.ld(fgv, C.Pp(E_GV));
        C.ld(bdm, C.Pp(B_DM)); C.ld(bdv, C.Pp(B_DV)); C.ld(bgm, C.Pp(B_GM)); C.ld(bgv, C.Pp(B_GV));
        bool dirty = false;
        if (is_good) {
            sc.cnt_fg += 1;
            sc.cnt_bg += 1;
            g_running_variance_add(fdm, fdv, sc.cnt_fg, x);
            g_running_variance_add(fgm, fgv, sc.cnt_fg, gx);
            g_running_variance_add(bdm, bdv, sc.cnt_bg, x);
            g_running_variance_add(bgm, bgv, sc.cnt_bg, gx);
            dirty = true;
        }
        const bool could_switch = sc.cnt_bg >= switch_freq;
        uint64_t next_window_size;
        if (is_early) next_window_size = s.early_mass_matrix_switch_freq;
        else {
            const double gv = (double)sc.current_window_size * s.mass_matrix_window_growth;
            const double fl = __builtin_floor(gv);
            const uint64_t grown = (uint64_t)((gv - fl >= 0.5) ? fl + 1.0 : fl);
            next_window_size = sc.current_window_size + 1 > grown ? sc.current_window_size + 1 : grown;
        }
        const bool is_late = next_window_size + draw > C.P.final_step_size_window;
        bool force_update = false;
        if (could_switch && !is_late) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                fdm[k] = bdm[k]; fdv[k] = bdv[k]; fgm[k] = bgm[k]; fgv[k] = bgv[k];
                bdm[k] = 0.0; bdv[k] = 0.0; bgm[k] = 0.0; bgv[k] = 0.0;
            }
            sc.cnt_fg = sc.cnt_bg;
            sc.cnt_bg = 0;
            force_update = true;
            dirty = true;
            if (!is_early) sc.current_window_size = next_window_size;
        }
        if (dirty) {
            C.st(bdm, C.Pp(B_DM)); C.st(bdv, C.Pp(B_DV)); C.st(bgm, C.Pp(B_GM)); C.st(bgv, C.Pp(B_GV));
            C.st(fdm, C.Pp(E_DM)); C.st(fdv, C.Pp(E_DV)); C.st(fgm, C.Pp(E_GM)); C.st(fgv, C.Pp(E_GV));
        }
        bool did_change = false;
        if (force_update | (draw - sc.last_update >= s.mass_matrix_update_freq)) did_change = g_mass_matrix_adapt(C, fdm, fdv, fgm, fgv);
        if (did_change) sc.last_update = draw;
        g_update_estimator(C, is_late);
        if (did_change & (sc.has_initial_mass_matrix != 0)) {
            sc.has_initial_mass_matrix = 0;
            return g_stepsize_init(C, x);
        }
        g_update_stepsize(C, false);
        return NM_CHAIN_OK;
    }
    g_update_estimator(C, true);
    g_update_stepsize(C, draw == s.num_tune - 1);
    return NM_CHAIN_OK;
}

template <class GD>
NM_DEV void g_write_row(GCtx<GD>& C, double* base, size_t row, const double (&t)[2]) {
    if (!base) return;
    double* dst = base + row;
#pragma unroll
    for (int k = 0; k < 2; ++k) { const int d = 2 * gl() + k; if (d < C.dim) dst[d] = t[k]; }
}
// DivergenceInfo.{start_location, start_gradient, end_location} (transformed_hamiltonian.rs:590-604), as emit_divergence_vectors
template <class GD>
NM_DEV void g_emit_divergence_vectors(GCtx<GD>& C, int64_t start_idx, size_t row) {
    const KParams& P = C.P;
    double x[2], gx[2], zt[2];
    if (start_idx == 0) {
        C.ld(x, C.Pp(P_X)); C.ld(gx, C.Pp(P_GX));
    } else {
        C.ld(zt, C.Ss(slot_F(0)));
#pragma unroll
        for (int k = 0; k < 2; ++k) x[k] = __builtin_fma(1.0, C.mu[k], zt[k] * C.sig[k]);
        (void)C.dens.eval(x, gx, C.dim);
    }
    g_write_row(C, P.out_div_start, row, x);
    g_write_row(C, P.out_div_start_grad, row, gx);
    C.ld(zt, C.Ss(slot_F(0) + 1));
#pragma unroll
    for (int k = 0; k < 2; ++k) x[k] = __builtin_fma(1.0, C.mu[k], zt[k] * C.sig[k]);
    g_write_row(C, P.out_div_end, row, x);
}

// ---- MclmcChain::draw (reference src/mclmc.rs:219-400; chain_draw_mclmc of nuts_kernels.hpp) for the chains of a group: every chain of a
// wavefront takes the same number of base steps (the step size is fixed), so the lockstep is free.  KinWrap instantiations only. ----
template <class GD>
NM_DEV void g_leapfrog_xg(GCtx<GD>& C, const GPt& s, GPt& o, double epsilon, double (&x)[2], double (&gx)[2]) {
    if (C.sc.kin != NM_TRAJ_EUCLIDEAN) { g_leapfrog_kin(C, s, o, epsilon, x, gx); return; }
    const double half = epsilon / 2.;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const double vh = __builtin_fma(half, s.g[k], s.v[k]);
        o.v[k] = vh;
        o.z[k] = __builtin_fma(epsilon, vh, s.z[k]);
        const double t = o.z[k] * C.sig[k];
        x[k] = __builtin_fma(1.0, C.mu[k], t);
    }
    o.logp = C.dens.eval(x, gx, C.dim);
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        o.g[k] = gx[k] * C.sig[k];
        o.v[k] = __builtin_fma(half, o.g[k], o.v[k]);
        acc = __builtin_fma(o.v[k], o.v[k], acc);
    }
    o.ke = 0.5 * gsum(acc);
}
template <class GD>
NM_DEV void g_sample_velocity(GCtx<GD>& C, double (&v)[2]) {
    g_fill_normals(C.rng, C.samp, C.dim, C.zig);
#pragma unroll
    for (int k = 0; k < 2; ++k) v[k] = 2 * gl() + k < C.dim ? 1.0 * C.samp[2 * gl() + k] : 0.0;
}
NM_DEV double g_kinetic(const double (&v)[2]) {
    double kacc = 0.0;
#pragma unroll
    for (int k = 0; k < 2; ++k) kacc = __builtin_fma(v[k], v[k], kacc);
    return 0.5 * gsum(kacc);
}
// partial_momentum_refresh (transformed_hamiltonian.rs:770-825) with the noise the last g_fill_normals left in C.samp
template <class GD>
NM_DEV void g_partial_refresh(GCtx<GD>& C, GPt& p, double factor) {
    const double half_step = C.sc.step_size * factor / 2.0;
    const double L = C.P.s.momentum_decoherence_length;
    double nz[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) nz[k] = 2 * gl() + k < C.dim ? 1.0 * C.samp[2 * gl() + k] : 0.0;
    if (C.sc.kin == NM_TRAJ_MICROCANONICAL) {       // isokinetic Langevin on the unit sphere
        const double nu = __builtin_sqrt(dexpm1_impl<true>(2.0 * half_step / L) / (double)C.dim);
#pragma unroll
        for (int k = 0; k < 2; ++k) p.v[k] = __builtin_fma(nu, nz[k], p.v[k]);
        g_normalize(p.v);
    } else {                                        // Ornstein-Uhlenbeck: alpha p + sqrt(1 - alpha^2) z
        const double alpha = dexp(-half_step / L);
        const double beta = __builtin_sqrt(1.0 - alpha * alpha);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const double nv = __builtin_fma(alpha, p.v[k], 0.0);
            p.v[k] = __builtin_fma(beta, nz[k], nv);
        }
        p.ke = g_kinetic(p.v);
    }
}
template <bool TUNE, class GD>
NM_DEV void g_chain_draw_mclmc(GCtx<GD>& C, uint64_t chain, uint64_t t_out) {
    const KParams& P = C.P;
    const nm_settings& s = P.s;
    ChainScalars& sc = C.sc;
    nm_draw_stats out = {};
    out.draw = sc.draw_count; out.chain = P.chain_id_offset + chain;
    bool resample_velocity = false;                 // Euclidean -> Microcanonical switch (mclmc.rs:490-504)
    if (s.mclmc_trajectory_kind == NM_MCLMC_EUCLIDEAN_EARLY_THEN_MICROCANONICAL && sc.draw_count == P.mclmc_switch_draw &&
        sc.kin != NM_TRAJ_MICROCANONICAL) {
        sc.kin = NM_TRAJ_MICROCANONICAL;
        resample_velocity = true;
    }
    const double base_step_size = sc.step_size;
    uint64_t num_base_steps;
    {
        const double q = s.subsample_frequency * s.momentum_decoherence_length / base_step_size;
        double r = __builtin_round(q);
        r = r != r ? 1.0 : (r > 1.0 ? r : 1.0);
        r = r < 1e6 ? r : 1e6;
        num_base_steps = (uint64_t)r;
    }
    const int max_halvings = s.dynamic_step_size ? 10 : 0;
    GPt cur, nxt;
    double x[2], gx[2];
    C.ld(x, C.Pp(P_X)); C.ld(gx, C.Pp(P_GX));
    if (resample_velocity) g_sample_velocity(C, cur.v); else C.ld(cur.v, C.Pp(P_V));
    if (sc.mm_id != sc.transform_id) {              // inv_transform_normalize (diagonal.rs:210-221)
        double isig[2];
        C.ld(isig, C.Pp(P_ISIG));
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const double t = __builtin_fma(-1.0, C.mu[k], x[k]);
            cur.z[k] = isig[k] * t;
            cur.g[k] = gx[k] * C.sig[k];
        }
        C.st(cur.z, C.Pp(P_Z)); C.st(cur.g, C.Pp(P_GZ));
        sc.logdet = sc.mm_logdet;
        sc.transform_id = sc.mm_id;
    } else {
        C.ld(cur.z, C.Pp(P_Z)); C.ld(cur.g, C.Pp(P_GZ));
    }
    const double logdet = sc.logdet;
    cur.ke = resample_velocity ? g_initial_kinetic(C, cur.v) : (sc.kin == NM_TRAJ_MICROCANONICAL ? 0.0 : g_kinetic(cur.v));
    cur.logp = sc.logp; cur.idx = 0;
    const double initial_energy = cur.ke - (cur.logp + logdet);
    g_fill_normals(C.rng, C.samp, C.dim, C.zig);    // sample_noise
    const double draw_start_energy = initial_energy;
    GAccept col;
    col.register_init(0.0);
    bool diverged = false, div_has_energy = false, div_has_end = false;
    double div_energy_error = 0.0;
    uint64_t steps_taken = 0, remaining = num_base_steps, stack0 = 0;
    uint32_t stack_bits = 0;
    int stack_len = 0;
    double factor = 1.0, time = 0.0;
    double xn[2] = {0.0, 0.0}, gxn[2] = {0.0, 0.0};
    const int tmp_slot = slot_F(0);                 // tmp_velocity (mclmc.rs:272-274)
    while (remaining > 0) {
        C.st(cur.v, C.Ss(tmp_slot));
        g_partial_refresh(C, cur, factor);
        const double step_baseline = cur.ke - (cur.logp + logdet);
        g_leapfrog_xg(C, cur, nxt, base_step_size * factor, xn, gxn);
        nxt.idx = cur.idx + 1;
        const double energy = nxt.ke - (nxt.logp + logdet);
        const double err = energy - step_baseline;
        bool div_now = false;
        if (g_bad_energy(C, err, s.max_energy_error * factor / (double)num_base_steps)) {
            col.register_divergent(); div_now = true; div_has_energy = true; div_has_end = true; div_energy_error = err;
        } else col.register_ok(energy);
        if (!div_now) {
            g_fill_normals(C.rng, C.samp, C.dim, C.zig);
            g_partial_refresh(C, nxt, factor);
            g_fill_normals(C.rng, C.samp, C.dim, C.zig);
            cur = nxt; x[0] = xn[0]; x[1] = xn[1]; gx[0] = gxn[0]; gx[1] = gxn[1];
            steps_taken += 1;
            remaining -= 1;
            time += factor * base_step_size;
            while (remaining == 0) {
                if (stack_len == 0) break;
                stack_len -= 1;
                remaining = (stack_len == 0 ? stack0 : (((stack_bits >> stack_len) & 1u) ? 2ull : 1ull)) - 1;
                factor *= 2.0;
            }
        } else {
            if (stack_len >= max_halvings) { diverged = true; break; }
            factor *= 0.5;
            if (stack_len == 0) stack0 = remaining;
            else stack_bits = (stack_bits & ~(1u << stack_len)) | ((remaining == 2 ? 1u : 0u) << stack_len);
            stack_len += 1;
            remaining = 2;
            C.ld(cur.v, C.Ss(tmp_slot));
        }
    }
    const size_t row = (size_t)(t_out * P.n_chains + chain) * P.dim;
    const bool is_good = diverged ? (cur.idx > 4) : (cur.idx != 0);
    const double cur_energy = cur.ke - (cur.logp + logdet);
    double energy_change, state_energy, state_energy_error, state_logp;
    int64_t state_idx;
    double sx[2], sgx[2], sz[2], sgz[2];
    if (diverged) {                                 // stay at the pre-trajectory position with a fresh momentum (mclmc.rs:361-388)
        g_write_row(C, P.out_div_start, row, x);
        g_write_row(C, P.out_div_start_grad, row, gx);
        if (div_has_end) g_write_row(C, P.out_div_end, row, xn);
        double v[2];
        g_sample_velocity(C, v);
        const double ke_new = g_initial_kinetic(C, v);
        C.st(v, C.Pp(P_V));
        C.ld(sx, C.Pp(P_X)); C.ld(sgx, C.Pp(P_GX)); C.ld(sz, C.Pp(P_Z)); C.ld(sgz, C.Pp(P_GZ));
        energy_change = cur_energy - draw_start_energy;
        state_logp = sc.logp;
        state_energy = ke_new - (state_logp + logdet);
        state_energy_error = state_energy - state_energy;
        state_idx = 0;
    } else {
        C.st(x, C.Pp(P_X)); C.st(gx, C.Pp(P_GX)); C.st(cur.z, C.Pp(P_Z)); C.st(cur.g, C.Pp(P_GZ)); C.st(cur.v, C.Pp(P_V));
        sc.logp = cur.logp;
#pragma unroll
        for (int k = 0; k < 2; ++k) { sx[k] = x[k]; sgx[k] = gx[k]; sz[k] = cur.z[k]; sgz[k] = cur.g[k]; }
        energy_change = cur_energy - initial_energy;
        state_logp = cur.logp; state_energy = cur_energy; state_energy_error = energy_change; state_idx = cur.idx;
    }
    sc.px_stale = 0;
    g_write_row(C, P.out_positions, row, sx);
    g_write_row(C, P.out_gradient, row, sgx);
    g_write_row(C, P.out_tpos, row, sz);
    g_write_row(C, P.out_tgrad, row, sgz);
    double fd = 0.0;
#pragma unroll
    for (int k = 0; k < 2; ++k) fd = fd + (sz[k] + sgz[k]) * (sz[k] + sgz[k]);
    fd = gsum(fd);
    const int64_t trans_id = sc.transform_id;
    sc.total_steps += col.count;
    const uint64_t ast = g_adapt<TUNE>(C, col, is_good, x, gx);
    if (ast != NM_CHAIN_OK) sc.status = ast;
    out.depth = steps_taken; out.maxdepth_reached = 0; out.diverging = diverged;
    out.tuning = sc.tuning; out.n_steps = sc.last_n_steps;
    out.index_in_trajectory = state_idx; out.transformation_index = trans_id;
    out.step_size = sc.step_size; out.step_size_bar = s.fixed_step_size;
    out.mean_tree_accept = sc.last_mean_tree_accept; out.mean_tree_accept_sym = sc.last_sym_mean_tree_accept;
    out.max_energy_error = sc.last_max_energy_error;
    out.logp = state_logp; out.energy = state_energy; out.energy_error = state_energy_error;
    out.fisher_distance = fd;
    out.divergence_energy_error = (diverged && div_has_energy) ? div_energy_error : __builtin_nan("");
    out.chain_status = ast;
    out.transformation_update_id = -1;
    out.num_eigenvalues = 0;
    out.energy_change = energy_change; out.average_step_size = time / (double)steps_taken;
    if (sc.mm_id != sc.stats_last_id) {
        out.transformation_update_id = sc.mm_id;
        g_write_row(C, P.out_mm_inv, row, C.sig);
        g_write_row(C, P.out_mm_mu, row, C.mu);
    }
    sc.stats_last_id = sc.mm_id;
    if (P.out_stats && gl() == 0) P.out_stats[t_out * P.n_chains + chain] = out;
    sc.draw_count += 1;
}

// NutsChain::draw (reference src/chain.rs:151-188) + the scalar statistics of expanded_draw (:190-232)
template <bool TUNE, class GD>
NM_DEV void g_chain_draw(GCtx<GD>& C, uint64_t chain, uint64_t t_out) {
    const KParams& P = C.P;
    ChainScalars& sc = C.sc;
    if constexpr (gkin_trait<GD>::value) {
        if (P.s.sampler == NM_SAMPLER_MCLMC) { g_chain_draw_mclmc<TUNE>(C, chain, t_out); return; }
    }
    GAccept col;
    DrawResult R;
    double x[2], gx[2], z[2], gz[2];
    const uint64_t st = g_transition(C, col, R, z);
    nm_draw_stats out;
    out.draw = sc.draw_count; out.chain = P.chain_id_offset + chain;
    if (st != NM_CHAIN_OK) {
        sc.status = st;
        if (P.out_stats && gl() == 0) {
            nm_draw_stats zz = {};
            zz.draw = sc.draw_count; zz.chain = P.chain_id_offset + chain; zz.chain_status = st;
            P.out_stats[t_out * P.n_chains + chain] = zz;
        }
        return;
    }
    const size_t row = (size_t)(t_out * P.n_chains + chain) * P.dim;
    if (R.diverging && (P.out_div_start || P.out_div_start_grad || P.out_div_end))
        g_emit_divergence_vectors(C, R.div_start_idx, row);          // before P_X / P_GX take the new draw
    if (R.chosen.slot == -1 && !sc.px_stale) {
        C.ld(x, C.Pp(P_X)); C.ld(gx, C.Pp(P_GX));
        C.ld(z, C.Pp(P_Z)); C.ld(gz, C.Pp(P_GZ));
    } else {
        if (R.chosen.slot == -1) C.ld(z, C.Pp(P_Z));
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const double tt = z[k] * C.sig[k];
            x[k] = __builtin_fma(1.0, C.mu[k], tt);
        }
        (void)C.dens.eval(x, gx, C.dim);
#pragma unroll
        for (int k = 0; k < 2; ++k) gz[k] = gx[k] * C.sig[k];
        const bool need_x = sc.tuning || t_out + 1 == P.n_draws || P.out_div_start || P.out_div_start_grad;
        if (need_x) { C.st(x, C.Pp(P_X)); C.st(gx, C.Pp(P_GX)); }
        sc.px_stale = need_x ? 0 : 1;
        C.st(z, C.Pp(P_Z)); C.st(gz, C.Pp(P_GZ));
        sc.logp = R.chosen.logp;
    }
    const int64_t idx = R.chosen.idx;
    g_write_row(C, P.out_positions, row, x);
    g_write_row(C, P.out_gradient, row, gx);                     // PointStats (transformed_hamiltonian.rs:122-157)
    g_write_row(C, P.out_tpos, row, z);
    g_write_row(C, P.out_tgrad, row, gz);
    double fd = 0.0;
#pragma unroll
    for (int k = 0; k < 2; ++k) fd = fd + (z[k] + gz[k]) * (z[k] + gz[k]);
    fd = gsum(fd);
    const double energy = R.chosen.ke - (R.chosen.logp + sc.logdet);
    const int64_t trans_id = sc.transform_id;
    sc.total_steps += col.count;
    const bool is_good = R.diverging ? ((idx < 0 ? -idx : idx) > 4) : (idx != 0);     // DrawGradCollector (adapt/diagonal.rs:73-83)
    const uint64_t ast = g_adapt<TUNE>(C, col, is_good, x, gx);
    if (ast != NM_CHAIN_OK) sc.status = ast;
    out.depth = R.depth; out.maxdepth_reached = R.reached_maxdepth; out.diverging = R.diverging;
    out.tuning = sc.tuning; out.n_steps = sc.last_n_steps;
    out.index_in_trajectory = idx; out.transformation_index = trans_id;
    out.step_size = sc.step_size;
    out.step_size_bar = P.s.step_size_method == NM_STEP_FIXED ? P.s.fixed_step_size
                      : P.s.step_size_method == NM_STEP_ADAM ? dexp(sc.log_step) : dexp(sc.log_step_adapted);
    out.mean_tree_accept = sc.last_mean_tree_accept; out.mean_tree_accept_sym = sc.last_sym_mean_tree_accept;
    out.max_energy_error = sc.last_max_energy_error;
    out.logp = R.chosen.logp; out.energy = energy; out.energy_error = energy - R.e0;
    out.fisher_distance = fd;
    out.divergence_energy_error = (R.diverging && R.has_divergence_energy_error) ? R.divergence_energy_error : __builtin_nan("");
    out.chain_status = ast;
    out.transformation_update_id = -1;
    out.num_eigenvalues = 0;
    out.energy_change = __builtin_nan(""); out.average_step_size = __builtin_nan("");
    if (sc.mm_id != sc.stats_last_id) {                         // DiagMassMatrix::extract_stats (transform/diagonal.rs:48-70)
        out.transformation_update_id = sc.mm_id;
        g_write_row(C, P.out_mm_inv, row, C.sig);
        g_write_row(C, P.out_mm_mu, row, C.mu);
    }
    sc.stats_last_id = sc.mm_id;
    if (P.out_stats && gl() == 0) P.out_stats[t_out * P.n_chains + chain] = out;
    sc.draw_count += 1;
}

// TUNE = true: the whole adaptation is compiled in (any launch that starts inside the warm-up); TUNE = false: launches
// after it, with the registers the adaptation would cost left to the tree
// ROOMY = true: the same kernel with the register allocation of ONE wavefront per SIMD, for launches that have no second one anyway (a
// grid of at most 4 x CUs blocks: K4's shard of 8192 chains is 1024 wavefronts) — 1.47e9 -> 1.61e9 leapfrogs/s there
template <class Dens, bool TUNE, bool ROOMY = false>
__global__ __launch_bounds__(64, (ROOMY ? 1 : TUNE ? NM_GROUP_OCC_TUNE : NM_GROUP_OCC)) void nuts_group_draw_kernel(const KParams P) {
    using GD0 = typename GroupDensity<Dens>::type;
    using GD = typename std::conditional<kin_trait<Dens>::value, GKin<GD0>, GD0>::type;
    __shared__ GroupShared sh;
    dm_init_lds();
    const int g = gg(), l = gl();
    for (uint64_t base = (uint64_t)blockIdx.x * GPW; base < P.n_chains; base += (uint64_t)gridDim.x * GPW) {
        const uint64_t chain = base + (uint64_t)g;
        if (chain < P.n_chains) {
            GCtx<GD> C(P, sh.sc[g]);
            {   // chain scalars: HBM -> this group's LDS copy
                const uint64_t* src = reinterpret_cast<const uint64_t*>(&P.sc[chain]);
                uint64_t* dst = reinterpret_cast<uint64_t*>(&sh.sc[g]);
                constexpr int NW = (int)(sizeof(ChainScalars) / 8);
                for (int i = l; i < NW; i += GS) dst[i] = src[i];
                asm volatile("" ::: "memory");
            }
            C.dim = (int)P.dim;
            C.md = (int)P.layout_md;
            C.pv = P.pvec + (size_t)chain * NUM_PSLOT * P.dpad + 2 * l;
            C.sv = P.svec + (size_t)blockIdx.x * P.nsslot * P.dpad + 2 * GS * g + 2 * l;
            C.l1z = sh.l1z[g] + 2 * l; C.l1v = sh.l1v[g] + 2 * l; C.samp = sh.samp[g];
            C.pend = sh.pend[g];
            C.zig = {P.zig_x, P.zig_f};
            C.ld(C.sig, C.Pp(P_SIG)); C.ld(C.mu, C.Pp(P_MU));
            C.rng.key = sh.sc[g].key;
            C.rng.pos = C.sc.rng_pos; C.rng.base = C.sc.rng_pos + 16; C.rng.cache = sh.rng_cache[g];
            C.dens.set_lds(sh.xs[g]);
            C.dens.init(P.logp_params, C.dim);
            if (C.sc.status == NM_CHAIN_OK) {
                for (uint64_t t = 0; t < P.n_draws; ++t) {
                    g_chain_draw<TUNE>(C, chain, t);
                    if (C.sc.status != NM_CHAIN_OK) break;
                }
            }
            C.sc.rng_pos = C.rng.pos;
            asm volatile("" ::: "memory");
            {
                const uint64_t* src = reinterpret_cast<const uint64_t*>(&sh.sc[g]);
                uint64_t* dst = reinterpret_cast<uint64_t*>(&P.sc[chain]);
                constexpr int NW = (int)(sizeof(ChainScalars) / 8);
                for (int i = l; i < NW; i += GS) dst[i] = src[i];
            }
        }
    }
}

}  // namespace NM_GNS
}  // namespace nm
