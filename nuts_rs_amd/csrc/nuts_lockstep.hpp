// nuts_lockstep.hpp — the LOCKSTEP matrix-core kernel for chains that share their matrices (round 4; BASELINE config 5:
// N(0, Sigma) with a full Sigma at dim 256 x 4096 chains through a shared low-rank transformation of rank up to dim).
//
// nuts_tile.hpp (rounds 2-3) keeps one wavefront per chain: 16 wavefronts of a block run the whole one-chain NUTS code at 128
// registers each (321 spilled) and meet for the dense products — the matrix cores sat at 39 % busy for two rounds because every
// round waits for the slowest of 16 branchy, spilling chain programs.  Here the mapping is north_star's:
//   * a block of EIGHT wavefronts (two per SIMD, 256 registers each) owns 16 chains and advances ALL of them by one density evaluation
//     per round, in lockstep;
//   * wavefront w owns the two 16-row stripes 2 w, 2 w + 1 of every vector of all 16 chains, in the C / D layout of v_mfma_f64_16x16x4_f64
//     (lane (g, c) = (lane >> 4, lane & 15) holds rows 16 s + g + 4 r, r = 0 .. 3, of chain column c): the products' results land where the
//     leapfrog's elementwise work wants them (a vector is 4 doubles per lane and stripe).  The kernel is NOT spill-free: rocprofv3 reports
//     1224 B of scratch per lane (the state machines' and refresh units' out-of-line code; the product loops themselves keep their
//     operands in registers) — the first build, 16 wavefronts with one stripe each at 128 registers, spilled 245 registers around every product;
//   * the tree — per chain, ragged, data dependent — is a per-chain STATE MACHINE advanced once per round by ONE LANE: wavefronts 0 .. 3 run
//     the machines of four chains each (lane c = chain c: north_star's "one chain per lane" for everything scalar: energies, multinomial
//     merges, U-turn decisions, the generator, the step-size adaptation), on sums the eight wavefronts reduce stripe by stripe;
//   * draws are NOT synchronised between the chains of a block: a chain that finishes its tree recomputes its chosen point,
//     writes its draw and starts the next one while the others continue — every column of every product carries a chain;
//   * the momentum refresh of the next draw (256 normals of the chain's ChaCha stream) is produced by wavefronts 4 .. 7 (four refresh units)
//     while wavefronts 0 .. 3 run the state machines.
// A round = [P3: form the next input column] -> five products (U'z, U s, P x, U't, U s') -> [P1: finish the leapfrog, partial
// sums] -> [stripe reduction] -> [P2: the chains' state machines].
//
// Numerics: the products are the fma chains of nuts_tile.hpp (bit for bit); the reductions over dim (kinetic energy, x'Px,
// the U-turn sums, the Fisher distance) are formed in THIS kernel's order — per lane its 4 rows, (p0 + p1) + (p2 + p3) over
// the four lane-rows of a stripe, stripes in ascending order — which the oracle reproduces (MathCfg.tile_order; the reference's
// own order is its SIMD width's).  Every elementwise operation is the one-chain kernel's.
#pragma once
#include "nuts_tile.hpp"     // tile mode, TileMats, taddr, v4d; nuts_kernels.hpp
#include "nuts_lane.hpp"     // the per-lane scalar toolkit: lexp / llog / llogaddexp, LRng, LAccept

namespace nm {
namespace lock {

using tile::TileMats;
using tile::taddr;
using tile::v4d;
using lane::lexp;
using lane::llog;
using lane::llogaddexp;
using lane::LRng;
using lane::LAccept;

constexpr int LC = 16;             // chains per block = columns of the MFMA
constexpr int LS = 16;             // 16-row stripes of a column tile
constexpr int SPW = 2;             // stripes per wavefront: 8 wavefronts per block = 2 per SIMD = 256 registers each (with one stripe per
                                   // wavefront and 128 registers the first build spilled 245 of them around every product)
constexpr int LWV = LS / SPW;      // wavefronts per block
constexpr int LROWS = 256;         // rows of a column tile (dim, rank <= 256)
constexpr int NRED = 32;           // partial-sum slots per (stripe, chain) and reduction pass
constexpr int GROUPS_PER_PASS = 4; // U-turn test groups (6 sums each) per pass
constexpr int LOCK_MAXDEPTH = 11; // maxdepth + extra_doublings the kernel is laid out for (sums, pend table); the engine checks
constexpr int MAX_GROUPS = LOCK_MAXDEPTH + 1;
constexpr int NSUM = 8 + 6 * MAX_GROUPS;
constexpr int NUNIT = 4;           // momentum-refresh units (wavefronts 4 .. 4 + NUNIT - 1)
constexpr int RF_P = 5;            // passes of the bulk ziggurat per chunk: 320 cells, one chunk for 256 samples and the cells their slow paths swallow

// Development (-DNM_LOCK_PROF=1, tools/bench_k5.py): shader-clock cycles of wavefront 0 of block 0 between the phase marks of a round,
// summed into P.prof[0 .. 7] (logic, unit hand-out, P3, products, P1 + reductions) and the rounds into P.prof[8]
#ifndef NM_LOCK_PROF
#define NM_LOCK_PROF 0
#endif
#if NM_LOCK_PROF
#define NM_LKP(slot) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); lkp_acc[slot] += n_ - lkp_t; lkp_t = n_; }
#else
#define NM_LKP(slot)
#endif

enum Mode : int { M_IDLE = 0, M_LEAF = 1, M_RECOMP = 2, M_KEEP = 3, M_WHITEN = 4, M_START = 5 };
// M_RECOMP: the chosen point's (x, g_x, g_z) from its z;  M_KEEP: the initial point was chosen and P_X / P_GX hold it (no density
// evaluation);  M_WHITEN: lazy re-whitening after a transformation update;  M_START: waits for its momentum refresh
enum RcSrc : int { RC_POOL = 0, RC_PZ = 3 };

// per-chain state, in LDS; read by all wavefronts (control part), written by the chain's logic lane only
struct LkChain {
    // ---- control: what the column does in the coming round (read by P3 / P1 of every stripe)
    int mode;
    int n, depth;             // leaf being computed (0-based within its doubling), depth of the doubling
    int fwd, check;
    int first_leaf;           // this leaf is the first of a draw (P1 also sums v_init^2)
    int load_edge;            // >= 0: the doubling starts from main-tree edge `load_edge` (P3 loads it into cur); -1: cur is the start
    int rc_src, rc_slot;      // M_RECOMP: where the chosen z is
    double eps;
    // stores P3 performs on the points it holds (slot numbers, -1 = none)
    int st_F_prev, st_L_cur, st_cand_cur, st_cand_prev, st_edge_cur;
    int start_draw;           // P3: a draw starts: cur.v <- refresh unit `v_unit`, edge 0 <- cur
    int finish;               // P3: write the finished draw's rows (outputs, persistent slots)
    uint64_t finish_row;
    int v_unit;               // refresh unit that holds / will hold the next draw's normals, -1 none
    int v_ready;
    int was_live;             // took part in this launch (its generator position is written back)
    // ---- tree state (locals of the one-chain transition, src/nuts.rs:281-388)
    double e0, logdet, cur_logp_start;
    int left_slot, right_slot, o_is_edge, o_edge_sign;
    int64_t left_idx, right_idx;
    uint64_t tdepth;
    double log_size;
    CandRef mc;
    uint32_t used, used_before;
    uint64_t mindepth, maxdepth;
    int in_extra, sign;
    uint64_t extra_left;
    int stop;
    double sub_log_size;
    CandRef sub_cand;
    double wE;
    double E_logp, E_ke; int64_t E_idx;
    double O_logp, O_ke; int64_t O_idx;
    LAccept col;
    DrawResult R;
    // ---- generator
    uint64_t rng_pos, rng_filled;
    uint64_t v_pos_start, v_pos_after;
    // ---- draw bookkeeping
    uint64_t chain;           // global chain index (row of the outputs), ~0 = absent
    int live;                 // takes part in this launch
};

struct PendL { double log_size; CandRef c; };

struct alignas(16) LockShared {
    double t[2][LROWS * LC];  // column tiles A, B (products' B operands); between the products and P3 both together: the stripes' partial sums
    double sums[LC][NSUM];    // reduced sums of the round, per chain
    double sig[LROWS], mu[LROWS], mul[LROWS], isig[LROWS];   // shared sigma, mean, mu_lr, 1 / sigma
    double scale[2][LROWS];   // lambda^(1/2) - 1, lambda^(-1/2) - 1
    int which[LC];
    LkChain ch[LC];
    ChainScalars sc[LC];
    PendL pend[LC][LOCK_MAXDEPTH + 2];
    uint32_t ring[32 * 64];   // the logic lanes' ChaCha rings ([word & 31][lane])
    // momentum-refresh units
    uint32_t rf_cache[NUNIT][RNG_CACHE_WORDS];
    uint32_t rf_words[NUNIT][128 * RF_P + 16];
    double rf_samp[NUNIT][LROWS];
    int unit_chain[NUNIT];    // the chain a unit serves in the coming logic phase, -1 none
    int unit_busy[NUNIT];
    int all_idle, npass_dummy;
    double zig[2 * 257];
};
static_assert(sizeof(LockShared) <= 160 * 1024 - 6144, "the block's LDS (dev_math's tables come on top)");

// scratch of a chain in HBM: [slot][256], element (stripe s, lane-row g, r) at s * 16 + g * 4 + r: a lane's 4 rows are 32 contiguous bytes
NM_DEV int sidx(int s, int g) { return s * 16 + g * 4; }

struct Vec4 { double a[4]; };

// ---------------------------------------------------------------------------------------------
// one stripe of a product on the matrix cores: acc += M[16 s .. 16 s + 15][:] B, inner index ascending.
// The A operands of 8 k-pairs (4 KiB per wavefront) are requested ahead of the MFMAs that use them.
// ---------------------------------------------------------------------------------------------
// acc[h] += M[stripe w SPW + h][:] B for the wavefront's SPW stripes; `nst` = stripes the matrix has (beyond: nothing to do).
// Software-pipelined by hand: the A operands of chunk k + 1 (4 k-pairs x 2 stripes = 8 x 16 B per lane) are REQUESTED before the 16 MFMAs
// of chunk k are issued, through a buffer descriptor (one per-lane offset, the chunk's position in an SGPR) and with scheduling barriers
// around the groups — left to itself the compiler rotates the loop so that every chunk waits for its own loads (first build: the
// products ran at half the matrix cores' rate with loads, LDS reads and spills all out of the way).  Requesting a product's FIRST chunk
// before the barrier in front of it (gemm_prefetch; the A operands do not depend on what the barrier guards) was measured and not
// kept: the requests queue in front of P3's own loads and the epilogues' registers — 75 -> 84 us per round.
constexpr int GCH = 4;                    // k-pairs per chunk
struct GemmA { double2 a0[SPW][GCH]; };
NM_DEV void gemm_prefetch(GemmA& G, const double* packed, int w, int nst, int kpairs) {
    if (w * SPW >= nst || kpairs < GCH) return;
    const rsrc_t ra = make_rsrc(packed, (uint64_t)nst * (uint64_t)kpairs * 1024);
    const int lv = lane_id() * 16;
#pragma unroll
    for (int h = 0; h < SPW; ++h) {
        const int st = w * SPW + h;
        const int sb = (st < nst ? st : 0) * kpairs * 1024;
#pragma unroll
        for (int i = 0; i < GCH; ++i) G.a0[h][i] = buf_load2(ra, lv, sb + i * 1024);
    }
}
NM_DEV void gemm_stripes(GemmA& G, const double* packed, int w, int nst, int kpairs, const double* b, v4d (&acc)[SPW]) {
    const int l = lane_id(), kk = l >> 4, c = l & 15;
    bool on[SPW];
    int sbase[SPW];                       // byte offset of my stripes' operands (wave-uniform)
#pragma unroll
    for (int h = 0; h < SPW; ++h) {
        const int st = w * SPW + h;
        on[h] = st < nst;
        sbase[h] = (on[h] ? st : 0) * kpairs * 1024;
    }
    if (!on[0]) return;                   // (stripes are handed out in order: none of mine exists)
    const v4d last_init = acc[SPW - 1];
    const rsrc_t ra = make_rsrc(packed, (uint64_t)nst * (uint64_t)kpairs * 1024);
    const int lv = l * 16;
    constexpr int CH = GCH;
    const int nfull = kpairs / CH;
    double2 a1[SPW][CH];
    auto request = [&](double2 (&a)[SPW][CH], int qq) {           // the A operands of the chunk that starts at k-pair qq
#pragma unroll
        for (int h = 0; h < SPW; ++h)
#pragma unroll
            for (int i = 0; i < CH; ++i) {
#ifdef NM_LOCK_X_NOLOAD        // timing experiment (results are wrong): the products without their A-operand stream
                a[h][i] = double2{(double)(qq + i), 1.0};
#else
                a[h][i] = buf_load2(ra, lv, sbase[h] + (qq + i) * 1024);
#endif
            }
    };
    auto chunk = [&](const double2 (&a)[SPW][CH], int qq) {       // its 16 MFMAs; the B operand of a k-step is read once for both stripes,
        double bv[2 * CH];                                        // and all eight of the chunk are requested before its first MFMA
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            const int r0 = 8 * (qq + i) + kk;
#ifdef NM_LOCK_X_NOLDS         // timing experiment (results are wrong): ... without their B-operand reads
            bv[2 * i] = (double)r0; bv[2 * i + 1] = (double)(r0 + c);
#else
            bv[2 * i] = b[taddr(r0, c)];
            bv[2 * i + 1] = b[taddr(r0 + 4, c)];
#endif
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < CH; ++i) {
#pragma unroll
            for (int h = 0; h < SPW; ++h) acc[h] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[h][i].x, bv[2 * i], acc[h], 0, 0, 0);
#pragma unroll
            for (int h = 0; h < SPW; ++h) acc[h] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[h][i].y, bv[2 * i + 1], acc[h], 0, 0, 0);
        }
    };
    const int qlast = nfull > 0 ? (nfull - 1) * CH : 0;           // (requests beyond the last chunk re-request it: no branch around loads)
    if (nfull > 0) request(G.a0, 0);
    int q = 0;
    for (int ch = 0; ch < nfull; ch += 2) {                       // two chunks per trip: the buffers swap roles, nothing is copied
        request(a1, q + CH <= qlast ? q + CH : qlast);
        __builtin_amdgcn_sched_barrier(0);
        chunk(G.a0, q);
        __builtin_amdgcn_sched_barrier(0);
        request(G.a0, q + 2 * CH <= qlast ? q + 2 * CH : qlast);
        __builtin_amdgcn_sched_barrier(0);
        if (ch + 1 < nfull) chunk(a1, q + CH);
        __builtin_amdgcn_sched_barrier(0);
        q += 2 * CH;
    }
    q = nfull * CH;
    for (; q < kpairs; ++q) {                                     // kpairs not a multiple of the chunk (dim or rank not a multiple of 32)
        const int r0 = 8 * q + kk;
        const double b0 = b[taddr(r0, c)], b1 = b[taddr(r0 + 4, c)];
#pragma unroll
        for (int h = 0; h < SPW; ++h) {
            const double2 a = buf_load2(ra, lv, sbase[h] + q * 1024);
            acc[h] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, b0, acc[h], 0, 0, 0);
            acc[h] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.y, b1, acc[h], 0, 0, 0);
        }
    }
    if (!on[SPW - 1]) {                                           // (a matrix with an odd number of stripes: the last wavefront's second one)
#pragma unroll
        for (int h = 1; h < SPW; ++h) if (!on[h]) acc[h] = last_init;   // (SPW == 2: only the last stripe can be missing on its own)
    }
}

// The block's wavefronts talk through LDS only (the tree scratch in HBM is private to a lane, output rows are read by nobody): a barrier
// has to wait for the wavefront's LDS operations, NOT for its global loads and stores — a workgroup release fence would (s_waitcnt vmcnt(0)),
// and with it for the A operands requested ahead for the next product.
NM_DEV void blk_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// the wavefront's stripe of a chain vector <-> a column tile in LDS
NM_DEV void put_stripe(double* t, int s, const Vec4& v) {
    const int l = lane_id(), g = l >> 4, c = l & 15;
#pragma unroll
    for (int r = 0; r < 4; ++r) t[taddr(16 * s + g + 4 * r, c)] = v.a[r];
}

// wavefront total of a partial over the four lane-rows of a stripe: (p0 + p1) + (p2 + p3), in every lane of the column.
// gfx950's v_permlane16_swap / v_permlane32_swap exchange rows of 16 / halves of 32 lanes between two registers on the VALU: with the
// same value in both, the two results are "my row pair's even row" and "its odd row" (resp. the two halves) — their sum is x + xor16(x)
// (resp. x + xor32(x)) without the LDS crossbar round trip a ds_bpermute shuffle costs (two per sum, five to thirty sums per round).
NM_DEV double stripe_sum(double p) {
    int lo = __double2loint(p), hi = __double2hiint(p);
    auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    const double q = __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
    lo = __double2loint(q); hi = __double2hiint(q);
    a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    return __hiloint2double((int)b[0], (int)a[0]) + __hiloint2double((int)b[1], (int)a[1]);
}

// ---------------------------------------------------------------------------------------------
// the chain's state machine (one lane per chain)
// ---------------------------------------------------------------------------------------------
struct Logic {
    const KParams& P;
    LockShared& S;
    int c;
    LkChain& K;
    ChainScalars& sc;
    LRng rng;
    __device__ Logic(const KParams& p, LockShared& s, int col) : P(p), S(s), c(col), K(s.ch[col]), sc(s.sc[col]) {}

    NM_DEV void rng_begin() {
        rng.init(sc.key, K.rng_pos, S.ring + lane_id());
        rng.filled = K.rng_filled;
    }
    NM_DEV void rng_end() { K.rng_pos = rng.pos; K.rng_filled = rng.filled; }

    NM_DEV bool merge_weights(double a_log_size, double b_log_size, bool is_main, double& total, bool& fatal) {
        total = llogaddexp(a_log_size, b_log_size);
        const double self_log_size = is_main ? a_log_size : total;
        if (b_log_size >= self_log_size) return true;
        const int b = rng.random_bool(lexp(b_log_size - self_log_size));
        if (b < 0) { fatal = true; return false; }
        return b == 1;
    }
    NM_DEV int pool_take() {
        const int p = (int)__builtin_ctz(~K.used);
        K.used |= 1u << p;
        return p;
    }

    // the start of a draw's tree: everything nuts::draw sets up before its first doubling (src/nuts.rs:281-330), then the first
    // doubling's head.  ke_init is not known yet (its sum arrives with the first leaf): e0 and the collector are completed there.
    NM_DEV void tree_begin() {
        const nm_settings& s = P.s;
        K.logdet = sc.logdet;
        K.left_slot = 0; K.right_slot = 0; K.o_is_edge = 0; K.o_edge_sign = 0;
        K.tdepth = 0; K.log_size = 0.; K.left_idx = 0; K.right_idx = 0;
        K.mc = CandRef{-1, sc.logp, 0.0, 0};
        K.used = 0;
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
        K.mindepth = mindepth; K.maxdepth = maxdepth;
        DrawResult& R = K.R;
        R.diverging = false; R.reached_maxdepth = false; R.has_divergence_energy_error = false; R.has_div_end = true;
        R.divergence_energy_error = 0.; R.div_start_idx = 0;
        K.in_extra = 0; K.extra_left = 0; K.sign = 1;
        K.first_leaf = 1;
        doubling_begin();
    }

    // head of the doubling loop (src/nuts.rs:330-348): decides whether another doubling starts; sets up its first leaf, or ends the tree
    NM_DEV void doubling_begin() {
        const nm_settings& s = P.s;
        bool check;
        if (!K.in_extra) {
            if (!(K.tdepth < K.maxdepth)) { K.R.reached_maxdepth = true; tree_end(); return; }
            K.sign = rng.random_bool_std() ? 1 : -1;
            check = (s.check_turning != 0) && !(K.tdepth < K.mindepth);
        } else {
            if (K.extra_left == 0) { tree_end(); return; }
            K.extra_left -= 1;
            check = false;
        }
        K.check = check ? 1 : 0;
        K.fwd = K.sign > 0 ? 1 : 0;
        K.depth = (int)K.tdepth;
        K.n = 0;
        K.used_before = K.used;
        K.eps = (double)K.sign * sc.step_size * 1.0;
        K.stop = STOP_NONE;
        K.sub_log_size = 0.;
        K.sub_cand = CandRef{-2, 0., 0., 0};
        const bool reuse_edge = K.o_is_edge && K.o_edge_sign == K.sign;
        K.o_is_edge = 0;
        // the doubling starts from the main tree's edge on this side; at depth 0 that is the initial point, which cur holds
        K.load_edge = (K.depth == 0 || reuse_edge) ? -1 : (K.fwd ? K.right_slot : K.left_slot);
        K.mode = M_LEAF;
    }

    NM_DEV void tree_end() {
        K.R.depth = K.tdepth;
        K.R.chosen = K.mc;
        K.first_leaf = 0;
        if (K.mc.slot == -1 && !sc.px_stale) { K.mode = M_KEEP; return; }
        K.mode = M_RECOMP;
        if (K.mc.slot == -1) { K.rc_src = RC_PZ; K.rc_slot = 0; }
        else { K.rc_src = RC_POOL; K.rc_slot = slot_C((int)P.layout_md, K.mc.slot); }
    }

    // after the leaf's sums are in: the leaf's energy test and accounting, the sub-tree merges it completes, the end of its doubling
    NM_DEV void leaf_done() {
        const nm_settings& s = P.s;
        const double* sm = S.sums[c];
        const int MD = (int)P.layout_md;
        K.st_F_prev = -1; K.st_L_cur = -1; K.st_cand_cur = -1; K.st_cand_prev = -1; K.st_edge_cur = -1;
        K.load_edge = -1;                                         // (the doubling's first leaf has taken its start point)
        if (K.first_leaf) {                                       // initialize_trajectory's scalars, now that sum v_init^2 is known
            const double ke_init = 0.5 * sm[2];
            K.e0 = ke_init - (sc.logp + K.logdet);
            K.R.e0 = K.e0;
            K.col.register_init(K.e0);
            K.mc.ke = ke_init;
            K.first_leaf = 0;
        }
        const int64_t edge_idx = K.fwd ? K.right_idx : K.left_idx;
        const int n = K.n;
        const uint64_t nleaf = 1ull << K.depth;
        const double p_logp = -0.5 * sm[1], p_ke = 0.5 * sm[0];
        const int64_t p_idx = edge_idx + (int64_t)K.sign * (int64_t)(n + 1);
        // ---- energy test + AcceptanceRateCollector (transformed_hamiltonian.rs:524-615, dual_avg.rs:112-166)
        const double energy = p_ke - (p_logp + K.logdet);
        const double err = energy - K.e0;
        double w = 0.;
        bool fatal = false;
        if ((err > s.max_energy_error) | !is_finite(err)) {
            K.col.register_divergent();
            K.R.diverging = true; K.R.has_divergence_energy_error = true; K.R.divergence_energy_error = err;
            K.R.div_start_idx = p_idx - (int64_t)K.sign;
            K.stop = STOP_DIVERGING;
        } else {
            K.col.register_ok(energy);
            w = -err;
        }
        const bool even = (n & 1) == 0;
        if (K.depth == 0) {
            K.O_logp = p_logp; K.O_ke = p_ke; K.O_idx = p_idx;
            if (K.stop == STOP_NONE) { K.sub_log_size = w; K.sub_cand = CandRef{-2, p_logp, p_ke, p_idx}; }
        } else if (even) {
            K.E_logp = p_logp; K.E_ke = p_ke; K.E_idx = p_idx; K.wE = w;
            if (K.stop == STOP_NONE) { K.n = n + 1; return; }     // the odd leaf follows
        } else {
            K.O_logp = p_logp; K.O_ke = p_ke; K.O_idx = p_idx;
            if (K.stop == STOP_NONE) {
                const uint64_t nn = (uint64_t)n;                  // odd
                const int t = (int)__builtin_ctzll(~nn);
                uint32_t turn_bits = 0;
                if (K.check) {
                    // level 1: (E, O) = (prev, cur); sums formed the forward way: backward = the negated sums (nuts_lane.hpp l_merge_turning)
                    const bool f = K.fwd != 0;
                    if (f ? ((sm[4] < 0.) | (sm[5] < 0.)) : ((sm[4] > 0.) | (sm[5] > 0.))) turn_bits |= 2u;
                    for (int k = 2; k <= t && turn_bits == 0; ++k) {
                        const double* q = sm + 8 + 6 * (k - 2);
                        bool tk = false;
#pragma unroll
                        for (int j = 0; j < 6; ++j) tk = tk | (f ? (q[j] < 0.) : (q[j] > 0.));
                        if (tk) turn_bits |= 1u << k;
                    }
                }
                {
                    double total;
                    const bool take = merge_weights(K.wE, w, false, total, fatal);
                    K.sub_cand = take ? CandRef{-2, K.O_logp, K.O_ke, K.O_idx} : CandRef{-3, K.E_logp, K.E_ke, K.E_idx};
                    K.sub_log_size = total;
                    if (fatal) K.stop = STOP_FATAL;
                    else if (turn_bits & 2u) K.stop = STOP_TURNING;
                }
                for (int k = 2; k <= t && K.stop == STOP_NONE; ++k) {
                    const PendL A = S.pend[c][k - 1];
                    double total;
                    const bool take = merge_weights(A.log_size, K.sub_log_size, false, total, fatal);
                    if (take) {
                        K.used &= ~(1u << A.c.slot);
                    } else {
                        if (K.sub_cand.slot >= 0) K.used &= ~(1u << K.sub_cand.slot);
                        K.sub_cand = A.c;
                    }
                    K.sub_log_size = total;
                    if (fatal) K.stop = STOP_FATAL;
                    else if ((turn_bits >> k) & 1u) K.stop = STOP_TURNING;
                }
                if (K.stop == STOP_NONE) {
                    const int ne = n - 1;                         // the pair's even leaf
                    if ((ne & 3) == 0 && K.depth > 1) K.st_F_prev = slot_F(ne == 0 ? K.depth : (int)__builtin_ctz((unsigned)ne));
                    if ((uint64_t)(n + 1) < nleaf) {
                        K.st_L_cur = slot_L(MD, t);
                        if (K.sub_cand.slot == -2) { K.sub_cand.slot = pool_take(); K.st_cand_cur = slot_C(MD, K.sub_cand.slot); }
                        else if (K.sub_cand.slot == -3) { K.sub_cand.slot = pool_take(); K.st_cand_prev = slot_C(MD, K.sub_cand.slot); }
                        S.pend[c][t] = PendL{K.sub_log_size, K.sub_cand};
                        K.n = n + 1;
                        return;                                   // the doubling goes on
                    }
                }
            }
        }
        // ---- the doubling is over: stopped, or its last leaf is done
        if (K.stop == STOP_FATAL || fatal) { sc.status = NM_CHAIN_LOGP_FATAL; K.mode = M_IDLE; K.live = 0; return; }
        if (K.stop == STOP_DIVERGING) { K.used = K.used_before; tree_end(); return; }
        if (K.stop == STOP_TURNING) {
            K.used = K.used_before;
            if (!K.in_extra) { K.in_extra = 1; K.extra_left = s.extra_doublings; }
            doubling_begin();
            return;
        }
        // top-level U-turn tests of the finished sub-tree against the main tree (src/nuts.rs:143-161)
        bool turning = false;
        if (K.check) {
            const bool f = K.fwd != 0;
            if (K.depth == 0) turning = f ? ((sm[4] < 0.) | (sm[5] < 0.)) : ((sm[4] > 0.) | (sm[5] > 0.));
            else {
                const int t = K.depth;                            // the last leaf closes levels 2 .. depth; the top-level group follows them
                const double* q = sm + 8 + 6 * (t - 1);
#pragma unroll
                for (int j = 0; j < 6; ++j) turning = turning | (f ? (q[j] < 0.) : (q[j] > 0.));
            }
        }
        double total;
        const bool take = merge_weights(K.log_size, K.sub_log_size, true, total, fatal);
        if (fatal) { sc.status = NM_CHAIN_LOGP_FATAL; K.mode = M_IDLE; K.live = 0; return; }
        if (take) {
            if (K.mc.slot >= 0) K.used &= ~(1u << K.mc.slot);
            if (K.sub_cand.slot == -2) { K.sub_cand.slot = pool_take(); K.st_cand_cur = slot_C(MD, K.sub_cand.slot); }
            else if (K.sub_cand.slot == -3) { K.sub_cand.slot = pool_take(); K.st_cand_prev = slot_C(MD, K.sub_cand.slot); }
            K.mc = K.sub_cand;
        } else if (K.sub_cand.slot >= 0) {
            K.used &= ~(1u << K.sub_cand.slot);
        }
        const bool more = K.in_extra ? K.extra_left > 0 : (turning ? s.extra_doublings > 0 : K.tdepth + 1 < K.maxdepth);
        if (more) {
            int ns = K.fwd ? K.right_slot : K.left_slot;
            const int other_side = K.fwd ? K.left_slot : K.right_slot;
            if (ns == 0) ns = other_side == 1 ? 2 : 1;
            K.st_edge_cur = ns;
            if (K.fwd) K.right_slot = ns; else K.left_slot = ns;
            K.o_is_edge = 1; K.o_edge_sign = K.sign;
        }
        if (K.fwd) K.right_idx = K.O_idx; else K.left_idx = K.O_idx;
        K.tdepth += 1;
        K.log_size = total;
        if (turning && !K.in_extra) { K.in_extra = 1; K.extra_left = s.extra_doublings; }
        doubling_begin();
    }

    NM_DEV void update_estimator(bool late) {                     // DualAverage::advance / Adam::advance (nuts_lane.hpp l_update_estimator)
        const nm_settings& s = P.s;
        if (s.step_size_method == NM_STEP_FIXED) return;
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
        sc.log_step = fmin_rs(sc.log_step, P.ln_max_step);
        const double mk = lexp(-s.da_k * llog((double)sc.da_count));
        sc.log_step_adapted = mk * sc.log_step + (1. - mk) * sc.log_step_adapted;
        sc.da_count += 1;
    }
    NM_DEV void update_stepsize(bool use_best_guess) {
        const nm_settings& s = P.s;
        const double step = s.step_size_method == NM_STEP_FIXED ? s.fixed_step_size
                          : s.step_size_method == NM_STEP_ADAM ? lexp(sc.log_step)
                          : (use_best_guess ? lexp(sc.log_step_adapted) : lexp(sc.log_step));
        if (s.has_jitter) {
            const double v12 = u2d((rng.next_u64() >> 12) | 0x3ff0000000000000ull);
            const double j = (v12 - 1.0) * P.jitter_scale + P.jitter_low;
            sc.step_size = step * j;
        } else {
            sc.step_size = step;
        }
    }
    // GlobalStrategy::adapt with a FROZEN transformation (adapt_lr of nuts_kernels.hpp with frozen = true): the schedule's
    // is_late, the step-size estimator and the new step size
    NM_DEV void adapt_frozen() {
        const nm_settings& s = P.s;
        const uint64_t draw = sc.draw_count;
        sc.last_mean_tree_accept = K.col.mean();
        sc.last_sym_mean_tree_accept = K.col.mean_sym();
        sc.last_n_steps = K.col.count;
        sc.last_max_energy_error = K.col.max_energy_error;
        if (draw >= s.num_tune) { update_stepsize(true); sc.tuning = 0; return; }
        if (draw < P.final_step_size_window) {
            const bool is_early = draw < P.early_end;
            uint64_t next_window_size;
            if (is_early) next_window_size = s.early_mass_matrix_switch_freq;
            else {
                const double gv = (double)sc.current_window_size * s.mass_matrix_window_growth;
                const double fl = __builtin_floor(gv);
                const uint64_t grown = (uint64_t)((gv - fl >= 0.5) ? fl + 1.0 : fl);
                next_window_size = sc.current_window_size + 1 > grown ? sc.current_window_size + 1 : grown;
            }
            const bool is_late = next_window_size + draw > P.final_step_size_window;
            update_estimator(is_late);
            update_stepsize(false);
            return;
        }
        update_estimator(true);
        update_stepsize(draw == s.num_tune - 1);
    }

    // the chosen point is in the stripes' registers (x, g_x, z, g_z), sum (z + g_z)^2 in sums[0]: statistics, adaptation, next draw
    NM_DEV void draw_done() {
        const double fd = S.sums[c][0];
        const DrawResult& R = K.R;
        const uint64_t row_idx = sc.draw_count - P.row_base;
        nm_draw_stats out = {};
        out.draw = sc.draw_count; out.chain = P.chain_id_offset + K.chain;
        sc.px_stale = 0;
        if (K.mode != M_KEEP) sc.logp = R.chosen.logp;
        const int64_t idx = R.chosen.idx;
        const double energy = R.chosen.ke - (R.chosen.logp + sc.logdet);
        out.depth = R.depth; out.maxdepth_reached = R.reached_maxdepth; out.diverging = R.diverging;
        out.index_in_trajectory = idx; out.transformation_index = sc.transform_id;
        out.logp = R.chosen.logp; out.energy = energy; out.energy_error = energy - R.e0;
        out.fisher_distance = fd;
        out.divergence_energy_error = (R.diverging && R.has_divergence_energy_error) ? R.divergence_energy_error : __builtin_nan("");
        sc.total_steps += K.col.count;
        adapt_frozen();
        out.tuning = sc.tuning; out.n_steps = sc.last_n_steps;
        out.step_size = sc.step_size;
        out.step_size_bar = P.s.step_size_method == NM_STEP_FIXED ? P.s.fixed_step_size
                          : P.s.step_size_method == NM_STEP_ADAM ? lexp(sc.log_step) : lexp(sc.log_step_adapted);
        out.mean_tree_accept = sc.last_mean_tree_accept; out.mean_tree_accept_sym = sc.last_sym_mean_tree_accept;
        out.max_energy_error = sc.last_max_energy_error;
        out.chain_status = NM_CHAIN_OK;
        out.transformation_update_id = -1;
        out.num_eigenvalues = 0;
        out.energy_change = __builtin_nan(""); out.average_step_size = __builtin_nan("");
        if (sc.mm_id != sc.stats_last_id) {
            out.transformation_update_id = sc.mm_id;
            out.num_eigenvalues = sc.lr_has_inner ? sc.lr_rank : 0;
        }
        sc.stats_last_id = sc.mm_id;
        if (P.out_stats) P.out_stats[row_idx * P.n_chains + K.chain] = out;
        K.finish = 1; K.finish_row = row_idx;
        sc.draw_count += 1;
        K.mode = sc.draw_count < P.draw_end ? M_START : M_IDLE;
        if (K.mode == M_IDLE) K.live = 0;
    }
};

// Out-of-line entry points: the state machines run on wavefront 0 only and the refreshes on wavefronts 1 .. NUNIT; inlined, their
// register needs (the generator's block function, the statistics row, exp / ln) would be charged to the products' loops of every
// wavefront (first build: 361 registers spilled inside the kernel's hot path).
__device__ __attribute__((noinline)) void logic_round(const KParams& P, LockShared& S, int l, int round_mode, int jitter_words) {
    LkChain& K = S.ch[l];
    if (K.start_draw) { S.unit_busy[K.v_unit] = 0; K.v_unit = -1; K.v_ready = 0; }      // P3 took the normals
    K.finish = 0; K.start_draw = 0;
    if (!K.live) return;
    Logic L(P, S, l);
    L.rng_begin();
    if (round_mode == M_LEAF) L.leaf_done();
    else if (round_mode == M_RECOMP || round_mode == M_KEEP) L.draw_done();
    else if (round_mode == M_WHITEN) {
        ChainScalars& q = S.sc[l];
        q.logdet = q.mm_logdet; q.transform_id = q.mm_id;
        K.mode = M_START;
    }
    L.rng_end();
    // where the NEXT draw's momentum refresh starts in the stream: behind this draw's step-size jitter
    if (K.v_unit < 0) {
        if (K.mode == M_START) K.v_pos_start = K.rng_pos;
        else if (K.mode == M_RECOMP || K.mode == M_KEEP) K.v_pos_start = K.rng_pos + (uint64_t)jitter_words;
    }
}
__device__ __attribute__((noinline)) void logic_start(const KParams& P, LockShared& S, int l) {
    LkChain& K = S.ch[l];
    Logic L(P, S, l);
    K.rng_pos = K.v_pos_after; K.rng_filled = K.v_pos_after >> 4;
    L.rng_begin();
    L.tree_begin();
    L.rng_end();
    K.start_draw = 1;                                             // P3 takes the normals out of the unit
}

// ---------------------------------------------------------------------------------------------
// momentum refresh: one wavefront produces `dim` standard normals of a chain's stream, in stream order, into samp[0 .. dim)
// (fill_standard_normals_bulk: the sequential ziggurat's samples and final stream position, bit for bit)
// ---------------------------------------------------------------------------------------------
__device__ __attribute__((noinline)) void refresh_unit(const KParams& P, LockShared& S, int u) {
    const int c = S.unit_chain[u];
    if (c < 0) return;
    LkChain& K = S.ch[c];
    DevRng rng;
    rng.init(S.sc[c].key, K.v_pos_start, S.rf_cache[u]);
    unsigned long long pt = 0;
    ZigTables Z{S.zig, S.zig + 257};
    fill_standard_normals_bulk<RF_P>(rng, S.rf_words[u], S.rf_samp[u], (int)P.dim, Z, 64, P.prof, pt);
    if (lane_id() == 0) { K.v_pos_after = rng.pos; K.v_ready = 1; }
}

// ---------------------------------------------------------------------------------------------
// the kernel
// ---------------------------------------------------------------------------------------------
struct Pt4 { Vec4 z, v, g; };

NM_DEV void ld_vec(Vec4& v, const double* base) {                // a lane's 4 contiguous rows of a scratch vector
    const double2 a = reinterpret_cast<const double2*>(base)[0], b = reinterpret_cast<const double2*>(base)[1];
    v.a[0] = a.x; v.a[1] = a.y; v.a[2] = b.x; v.a[3] = b.y;
}
NM_DEV void st_vec(const Vec4& v, double* base) {
    reinterpret_cast<double2*>(base)[0] = double2{v.a[0], v.a[1]};
    reinterpret_cast<double2*>(base)[1] = double2{v.a[2], v.a[3]};
}
// a lane's rows of a vector stored in natural element order (persistent slots, output rows): rows 16 s + g + 4 r
NM_DEV void ld_nat(Vec4& v, const double* base, int s, int g, int dim) {
#pragma unroll
    for (int r = 0; r < 4; ++r) { const int d = 16 * s + g + 4 * r; v.a[r] = d < dim ? base[d] : 0.0; }
}
NM_DEV void st_nat(const Vec4& v, double* base, int s, int g, int dim) {
#pragma unroll
    for (int r = 0; r < 4; ++r) { const int d = 16 * s + g + 4 * r; if (d < dim) base[d] = v.a[r]; }
}

NM_DEV void turn_group(const Vec4& az, const Vec4& av, const Vec4& lz, const Vec4& lv, const Vec4& bz, const Vec4& bv,
                       const Vec4& oz, const Vec4& ov, double (&a)[6]) {
#pragma unroll
    for (int j = 0; j < 6; ++j) a[j] = 0.;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        turn_acc(az.a[r], av.a[r], oz.a[r], ov.a[r], a[0], a[1]);
        turn_acc(lz.a[r], lv.a[r], oz.a[r], ov.a[r], a[2], a[3]);
        turn_acc(az.a[r], av.a[r], bz.a[r], bv.a[r], a[4], a[5]);
    }
}

// the LR_SET_TRANSFORM commit (LowRankMassMatrix::update's scalars; nuts_tile.hpp did it inside its draw kernel): one wavefront per
// chain, the sum of ln(1 / sigma) in the wave kernels' order — the transformation is set before this kernel's order applies
__global__ __launch_bounds__(64) void lock_commit_kernel(const KParams P) {
    dm_init_lds();
    const uint64_t chain = blockIdx.x;
    if (chain >= P.n_chains) return;
    ChainScalars* q = P.sc + chain;
    if (q->status != NM_CHAIN_OK || q->lr_pending != LR_SET_TRANSFORM) return;
    const int t = (int)threadIdx.x;
    const double* isig = P.pvec + (size_t)chain * NUM_PSLOT * P.dpad + (size_t)P_ISIG * P.dpad;
    double acc = 0.0;
    const int dpl = (int)(P.dpad / 64);
    for (int k = 0; k < dpl; ++k) {
        const int d = 2 * ((k >> 1) * 64 + t) + (k & 1);
        const bool valid = d < (int)P.dim;
        acc = acc + (valid ? dlog_impl<false>(valid ? isig[d] : 1.0) : 0.0);
    }
    const double diag_logdet = wave_sum(acc);
    if (t == 0) {
        if (q->lr_upd_ok) {
            q->mm_logdet = q->lr_upd_logdet + diag_logdet;
            q->mm_id += 1; q->lr_has_inner = 1; q->lr_rank = q->lr_upd_rank;
        }
        q->lr_pending = LR_IDLE;
    }
}

__global__ __launch_bounds__(64 * LWV, LWV / 4) void nuts_lockstep_kernel(const KParams P, const TileMats M) {
    __shared__ LockShared S;
    dm_init_lds();
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));     // my wavefront: stripes w SPW .. w SPW + SPW - 1
    const int l = lane_id(), g = l >> 4, c = l & 15;
    const int dim = (int)P.dim;
    const int MD = (int)P.layout_md;
    constexpr int NT = 64 * LWV;
    for (int i = (int)threadIdx.x; i < 257; i += NT) { S.zig[i] = P.zig_x[i]; S.zig[257 + i] = P.zig_f[i]; }
    const uint64_t n_tiles = (P.n_chains + LC - 1) / LC;
    const int jitter_words = P.s.has_jitter ? 2 : 0;
    // the block's tree scratch through a buffer descriptor: [chain][slot][256], a lane's 4 rows of a stripe are 32 contiguous bytes
    const rsrc_t rscr = make_rsrc(P.svec + (size_t)blockIdx.x * LC * (size_t)P.nsslot * LROWS, (uint64_t)LC * P.nsslot * LROWS * 8);
    const int nss = (int)P.nsslot;
    for (uint64_t tile_i = blockIdx.x; tile_i < n_tiles; tile_i += gridDim.x) {
        __syncthreads();
        // ---- the tile's shared data and the chains' records
        {
            const uint64_t c0 = tile_i * LC;
            const double* lv = P.lrval + (size_t)c0 * 2 * P.lr_rmax;
            for (int i = (int)threadIdx.x; i < 2 * LROWS; i += NT) {
                const int which = i / LROWS, k = i % LROWS;
                S.scale[which][k] = k < M.rank ? lv[(size_t)which * P.lr_rmax + k] - 1.0 : 0.0;
            }
            const double* pv = P.pvec + (size_t)c0 * NUM_PSLOT * P.dpad;
            const double* mlr = P.lrvec + (size_t)c0 * (1 + P.lr_rmax) * P.dpad;
            for (int i = (int)threadIdx.x; i < LROWS; i += NT) {
                const bool in = i < (int)P.dpad;
                S.sig[i] = in ? pv[(size_t)P_SIG * P.dpad + i] : 0.0;
                S.isig[i] = in ? pv[(size_t)P_ISIG * P.dpad + i] : 0.0;
                S.mu[i] = in ? pv[(size_t)P_MU * P.dpad + i] : 0.0;
                S.mul[i] = in ? mlr[i] : 0.0;
            }
            for (int i = (int)threadIdx.x; i < 2 * LROWS * LC; i += NT) S.t[0][i] = 0.0;
            if (threadIdx.x < LC) {
                const int cc = (int)threadIdx.x;
                const uint64_t chain = c0 + (uint64_t)cc;
                LkChain& K = S.ch[cc];
                K.chain = chain; K.live = 0; K.was_live = 0; K.mode = M_IDLE; K.v_unit = -1; K.v_ready = 0; K.finish = 0; K.start_draw = 0;
                K.st_F_prev = -1; K.st_L_cur = -1; K.st_cand_cur = -1; K.st_cand_prev = -1; K.st_edge_cur = -1; K.load_edge = -1;
                K.first_leaf = 0; K.n = 0; K.depth = 0; K.fwd = 1; K.check = 0; K.eps = 0.; K.rc_src = RC_PZ; K.rc_slot = 0;
                K.left_slot = 0; K.right_slot = 0;
                S.which[cc] = 0;
                if (chain < P.n_chains) {
                    S.sc[cc] = P.sc[chain];
                    const ChainScalars& q = S.sc[cc];
                    if (q.status == NM_CHAIN_OK && q.lr_pending == LR_IDLE && q.lr_has_inner != 0 && q.draw_count >= P.row_base && q.draw_count < P.draw_end) {
                        K.live = 1; K.was_live = 1;
                        K.rng_pos = q.rng_pos; K.rng_filled = q.rng_pos >> 4;
                        K.mode = q.mm_id != q.transform_id ? M_WHITEN : M_START;
                        K.v_pos_start = q.rng_pos;
                    }
                }
            }
            if (threadIdx.x < NUNIT) { S.unit_chain[threadIdx.x] = -1; S.unit_busy[threadIdx.x] = 0; }
        }
        __syncthreads();
        const uint64_t my_chain = S.ch[c].chain < P.n_chains ? S.ch[c].chain : 0;
        double* const my_pv = P.pvec + (size_t)my_chain * NUM_PSLOT * P.dpad;
        // byte offset of (slot, stripe h) of my chain in the block's scratch
        auto soff = [&](int slot, int h) { return ((c * nss + slot) * LROWS + sidx(w * SPW + h, g)) * 8; };
        auto ldS = [&](Vec4& v, int slot, int h) {
            const int o = soff(slot, h);
            const double2 a = buf_load2(rscr, o, 0), b = buf_load2(rscr, o + 16, 0);
            v.a[0] = a.x; v.a[1] = a.y; v.a[2] = b.x; v.a[3] = b.y;
        };
        auto stS = [&](const Vec4& v, int slot, int h) {
            const int o = soff(slot, h);
            buf_store2(rscr, o, 0, v.a[0], v.a[1]); buf_store2(rscr, o + 16, 0, v.a[2], v.a[3]);
        };
        Vec4 cz[SPW], cv[SPW], cg[SPW], pz[SPW], pvv[SPW], zin[SPW], vh[SPW];
#pragma unroll
        for (int h = 0; h < SPW; ++h)
#pragma unroll
            for (int r = 0; r < 4; ++r) { cz[h].a[r] = cv[h].a[r] = cg[h].a[r] = pz[h].a[r] = pvv[h].a[r] = zin[h].a[r] = vh[h].a[r] = 0.; }
        int round_mode = M_IDLE;          // what my column submitted in the round in flight
#if NM_LOCK_PROF
        unsigned long long lkp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, lkp_t = __builtin_amdgcn_s_memtime(), lkp_rounds = 0;
#endif
        for (;;) {
            // =====================================================================================================
            // P2 (wavefronts 0 .. 3: the state machines of four chains each; wavefronts 4 .. 7: momentum refreshes)
            // =====================================================================================================
            if (w < 4) {                                           // (lane = chain, as everywhere: the chain's ring column and record)
                if (l < LC && (l >> 2) == w) logic_round(P, S, l, round_mode, jitter_words);
            } else if (w < 4 + NUNIT) {
                refresh_unit(P, S, w - 4);
            }
            NM_LKP(0)
            blk_barrier();
            NM_LKP(1)
            // ---- P2b (wavefront 0): chains whose normals are ready begin their tree; free refresh units are handed out
            if (w == 0) {
                if (l < NUNIT && S.unit_chain[l] >= 0) S.unit_chain[l] = -1;       // served in the phase above
                int want = 0;
                if (l < LC) {
                    LkChain& K = S.ch[l];
                    if (K.live && K.mode == M_START && K.v_unit >= 0 && K.v_ready) logic_start(P, S, l);
                    const bool more_draws = S.sc[l].draw_count + 1 < P.draw_end;
                    want = (K.live && K.v_unit < 0 && (K.mode == M_START || K.mode == M_WHITEN || ((K.mode == M_RECOMP || K.mode == M_KEEP) && more_draws))) ? 1 : 0;
                }
                const unsigned long long wmask = __ballot(want);
                if (l == 0) {
                    unsigned long long wm = wmask;
                    for (int u = 0; u < NUNIT && wm; ++u) {
                        if (S.unit_busy[u]) continue;
                        const int cc = (int)__builtin_ctzll(wm);
                        wm &= wm - 1;
                        S.unit_chain[u] = cc; S.unit_busy[u] = 1;
                        S.ch[cc].v_unit = u; S.ch[cc].v_ready = 0;
                    }
                    int any = 0;
                    for (int i = 0; i < LC; ++i) any |= S.ch[i].live | S.ch[i].finish;
                    S.all_idle = any ? 0 : 1;
                }
            }
            blk_barrier();
            NM_LKP(2)
            if (S.all_idle) break;
            // =====================================================================================================
            // P3 (every wavefront, its stripes): the stores the state machines asked for, the finished draws' rows, the next input column
            // =====================================================================================================
            GemmA G;
            {
                const LkChain& K = S.ch[c];
                const int mode = K.mode;
                const int st_F = K.st_F_prev, st_L = K.st_L_cur, st_cc = K.st_cand_cur, st_cp = K.st_cand_prev, st_e = K.st_edge_cur;
                const int fin = K.finish, sd = K.start_draw, le = K.load_edge, vunit = K.v_unit, rsrc_ = K.rc_src, rslot = K.rc_slot;
                const double eps = K.eps, half = eps / 2.;
                const size_t row = (size_t)(K.finish_row * P.n_chains + my_chain) * P.dim;
#pragma unroll
                for (int h = 0; h < SPW; ++h) {
                    const int s = w * SPW + h;
                    if (round_mode == M_LEAF) {
                        if (st_F >= 0) { stS(pz[h], st_F, h); stS(pvv[h], st_F + 1, h); }
                        if (st_L >= 0) { stS(cz[h], st_L, h); stS(cv[h], st_L + 1, h); }
                        if (st_cc >= 0) stS(cz[h], st_cc, h);
                        if (st_cp >= 0) stS(pz[h], st_cp, h);
                        if (st_e >= 0) { stS(cz[h], EDGE0_Z + 3 * st_e, h); stS(cv[h], EDGE0_V + 3 * st_e, h); stS(cg[h], EDGE0_G + 3 * st_e, h); }
                    }
                    if (fin) {                                     // the draw's transformed point (its x and g_x rows left the registers in the products)
                        if (P.out_tpos) st_nat(cz[h], P.out_tpos + row, s, g, dim);
                        if (P.out_tgrad) st_nat(cg[h], P.out_tgrad + row, s, g, dim);
                        if (round_mode == M_RECOMP) {
                            st_nat(cz[h], my_pv + (size_t)P_Z * P.dpad, s, g, dim); st_nat(cg[h], my_pv + (size_t)P_GZ * P.dpad, s, g, dim);
                        } else {                                   // M_KEEP: the stored point is the draw
                            Vec4 t;
                            if (P.out_positions) { ld_nat(t, my_pv + (size_t)P_X * P.dpad, s, g, dim); st_nat(t, P.out_positions + row, s, g, dim); }
                            if (P.out_gradient) { ld_nat(t, my_pv + (size_t)P_GX * P.dpad, s, g, dim); st_nat(t, P.out_gradient + row, s, g, dim); }
                        }
                    }
                    if (round_mode == M_WHITEN) {                  // the re-whitened point is the chain's current point from now on
                        st_nat(cz[h], my_pv + (size_t)P_Z * P.dpad, s, g, dim); st_nat(cg[h], my_pv + (size_t)P_GZ * P.dpad, s, g, dim);
                    }
                    if (sd) {                                      // initialize_trajectory: fresh momentum, the initial point is edge 0
                        const double* samp = S.rf_samp[vunit];
#pragma unroll
                        for (int r = 0; r < 4; ++r) { const int d = 16 * s + g + 4 * r; cv[h].a[r] = d < dim ? 1.0 * samp[d] : 0.0; }
                        // (the chain's current point comes from its persistent slots, not from registers kept alive across its waiting rounds)
                        ld_nat(cz[h], my_pv + (size_t)P_Z * P.dpad, s, g, dim); ld_nat(cg[h], my_pv + (size_t)P_GZ * P.dpad, s, g, dim);
                        stS(cz[h], EDGE0_Z, h); stS(cv[h], EDGE0_V, h); stS(cg[h], EDGE0_G, h);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) zin[h].a[r] = 0.0;
                    if (mode == M_LEAF) {
                        // (keeping the far edge in registers from the top-level test to here — half of all doublings start there — was
                        // measured: 48 more live registers spill in the products, 76 -> 87 us per round)
                        if (le >= 0) { ldS(cz[h], EDGE0_Z + 3 * le, h); ldS(cv[h], EDGE0_V + 3 * le, h); ldS(cg[h], EDGE0_G + 3 * le, h); }
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            vh[h].a[r] = __builtin_fma(half, cg[h].a[r], cv[h].a[r]);
                            zin[h].a[r] = __builtin_fma(eps, vh[h].a[r], cz[h].a[r]);
                        }
                    } else if (mode == M_RECOMP) {
                        if (rsrc_ == RC_POOL) ldS(zin[h], rslot, h);
                        else ld_nat(zin[h], my_pv + (size_t)P_Z * P.dpad, s, g, dim);
                    } else if (mode == M_WHITEN) {                 // compute_transformed_position's diagonal part (low_rank.rs:325-347)
                        Vec4 x0;
                        ld_nat(x0, my_pv + (size_t)P_X * P.dpad, s, g, dim);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int d = 16 * s + g + 4 * r;
                            const double t = __builtin_fma(-1.0, S.mu[d], x0.a[r]);
                            const double zz = S.isig[d] * t;
                            zin[h].a[r] = d < dim ? __builtin_fma(-1.0, S.mul[d], zz) : 0.0;
                        }
                    }
                    put_stripe(S.t[0], s, zin[h]);
                }
                if (w == 0 && g == 0) S.which[c] = mode == M_WHITEN ? 1 : 0;
                round_mode = mode;
            }
            NM_LKP(3)
            blk_barrier();
            NM_LKP(4)
            // =====================================================================================================
            // the five products
            // =====================================================================================================
            v4d acc[SPW];
            {   // S = U' zin, scaled (lambda^w - 1) on its way out
#pragma unroll
                for (int h = 0; h < SPW; ++h) acc[h] = v4d{0.0, 0.0, 0.0, 0.0};
                gemm_stripes(G, M.ut, w, M.rank_st, M.dim_kp, S.t[0], acc);
                const double* scl = S.scale[S.which[c] & 1];
#pragma unroll
                for (int h = 0; h < SPW; ++h) {
                    Vec4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o.a[r] = acc[h][r] * scl[16 * (w * SPW + h) + g + 4 * r];
                    put_stripe(S.t[1], w * SPW + h, o);
                }
            }
            blk_barrier();
            // the output row of a column that recomputes its chosen point (the draw that ends with this round)
            const size_t out_row = (size_t)((S.sc[c].draw_count - P.row_base) * P.n_chains + my_chain) * P.dim;
            Vec4 x[SPW];
            double xyp[SPW];
            {   // xt = zin + U S ; x = sigma (xt + mu_lr) + mean  (compute_untransformed_position: low_rank.rs:349-375, diagonal.rs:248-257)
#pragma unroll
                for (int h = 0; h < SPW; ++h) acc[h] = v4d{zin[h].a[0], zin[h].a[1], zin[h].a[2], zin[h].a[3]};
                gemm_stripes(G, M.u, w, M.dim_st, M.rank_kp, S.t[1], acc);
#pragma unroll
                for (int h = 0; h < SPW; ++h) {
                    Vec4 xs;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int d = 16 * (w * SPW + h) + g + 4 * r;
                        if (round_mode == M_WHITEN) zin[h].a[r] = acc[h][r];       // the whitened z itself
                        const double t = __builtin_fma(1.0, S.mul[d], acc[h][r]);
                        x[h].a[r] = __builtin_fma(1.0, S.mu[d], t * S.sig[d]);
                        xs.a[r] = (d < dim && (round_mode == M_LEAF || round_mode == M_RECOMP)) ? x[h].a[r] : 0.0;
                    }
                    put_stripe(S.t[0], w * SPW + h, xs);           // (every wavefront has passed the barrier behind the first product: tile A is free)
                }
            }
            blk_barrier();
            {   // y = P x
#pragma unroll
                for (int h = 0; h < SPW; ++h) acc[h] = v4d{0.0, 0.0, 0.0, 0.0};
                gemm_stripes(G, M.p, w, M.dim_st, M.dim_kp, S.t[0], acc);
            }
            // y is used up here: the stripe's part of x'Px, g_x = -y, t3 = sigma g_x — and a recomputing column's x and g_x rows go
            // straight to the draw's output row and the chain's persistent slots (nothing of x, y, g_x stays live across the last two
            // products: with them the products spilled)
            Vec4 t3[SPW];
#pragma unroll
            for (int h = 0; h < SPW; ++h) {
                const int s = w * SPW + h;
                Vec4 gx;
                double xy = 0.;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int d = 16 * s + g + 4 * r;
                    gx.a[r] = 0.0; t3[h].a[r] = 0.0;
                    xy = xy + (d < dim ? x[h].a[r] * acc[h][r] : 0.0);
                }
                xyp[h] = xy;
                if (round_mode == M_WHITEN) {
                    Vec4 g0;
                    ld_nat(g0, my_pv + (size_t)P_GX * P.dpad, s, g, dim);
#pragma unroll
                    for (int r = 0; r < 4; ++r) t3[h].a[r] = g0.a[r] * S.sig[16 * s + g + 4 * r];
                } else if (round_mode == M_LEAF || round_mode == M_RECOMP) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) { const int d = 16 * s + g + 4 * r; gx.a[r] = d < dim ? -acc[h][r] : 0.0; t3[h].a[r] = gx.a[r] * S.sig[d]; }
                    if (round_mode == M_RECOMP) {
                        if (P.out_positions) st_nat(x[h], P.out_positions + out_row, s, g, dim);
                        if (P.out_gradient) st_nat(gx, P.out_gradient + out_row, s, g, dim);
                        st_nat(x[h], my_pv + (size_t)P_X * P.dpad, s, g, dim); st_nat(gx, my_pv + (size_t)P_GX * P.dpad, s, g, dim);
                    }
                }
                put_stripe(S.t[1], s, t3[h]);
            }
            blk_barrier();
            {   // S' = U' t3, scaled with lambda^(1/2) - 1 (the gradient's application, every mode)
#pragma unroll
                for (int h = 0; h < SPW; ++h) acc[h] = v4d{0.0, 0.0, 0.0, 0.0};
                gemm_stripes(G, M.ut, w, M.rank_st, M.dim_kp, S.t[1], acc);
#pragma unroll
                for (int h = 0; h < SPW; ++h) {
                    Vec4 o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o.a[r] = acc[h][r] * S.scale[0][16 * (w * SPW + h) + g + 4 * r];
                    put_stripe(S.t[0], w * SPW + h, o);
                }
            }
            blk_barrier();
            Vec4 gz[SPW];
            {   // g_z = t3 + U S'
#pragma unroll
                for (int h = 0; h < SPW; ++h) acc[h] = v4d{t3[h].a[0], t3[h].a[1], t3[h].a[2], t3[h].a[3]};
                gemm_stripes(G, M.u, w, M.dim_st, M.rank_kp, S.t[0], acc);
#pragma unroll
                for (int h = 0; h < SPW; ++h)
#pragma unroll
                    for (int r = 0; r < 4; ++r) gz[h].a[r] = acc[h][r];
            }
            blk_barrier();                                         // the tiles become the partial-sum buffer
            NM_LKP(5)
            // =====================================================================================================
            // P1 (every wavefront, its stripes): finish the leapfrog, partial sums of the round (passes of NRED slots)
            // =====================================================================================================
            double* const red = S.t[0];                            // [stripe][slot][chain]: 16 x 32 x 16 doubles = both tiles
            auto emit = [&](int s, int j, double p) {
                const double tot = stripe_sum(p);
                if (g == 0) red[((size_t)s * NRED + j) * LC + c] = tot;
            };
            const LkChain& K1 = S.ch[c];
            const int k_n = K1.n, k_depth = K1.depth, k_fwd = K1.fwd, k_check = K1.check, k_ls = K1.left_slot, k_rs = K1.right_slot;
            const double half1 = K1.eps / 2.;
            int ngroups = 0;                                       // U-turn test groups of my column's leaf
            if (round_mode == M_LEAF && k_check && (k_n & 1)) {
                const int t = (int)__builtin_ctz(~(unsigned)k_n);
                ngroups = t - 1;                                   // levels 2 .. t
                if ((k_n + 1) == (1 << k_depth)) ngroups += 1;     // + the top-level tests of the finished doubling
            }
#pragma unroll
            for (int h = 0; h < SPW; ++h) {
                const int s = w * SPW + h;
                pz[h] = cz[h]; pvv[h] = cv[h];
                cz[h] = zin[h]; cg[h] = gz[h];                     // every column: nothing of the old point stays live across the products
                if (round_mode == M_LEAF) {
                    double ke = 0., ki = 0., t1 = 0., t2 = 0.;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        cv[h].a[r] = __builtin_fma(half1, gz[h].a[r], vh[h].a[r]);
                        ke = __builtin_fma(cv[h].a[r], cv[h].a[r], ke);
                        ki = __builtin_fma(pvv[h].a[r], pvv[h].a[r], ki);
                        turn_acc(pz[h].a[r], pvv[h].a[r], cz[h].a[r], cv[h].a[r], t1, t2);
                    }
                    emit(s, 0, ke); emit(s, 1, xyp[h]); emit(s, 2, ki); emit(s, 4, t1); emit(s, 5, t2);
                } else if (round_mode == M_KEEP) {                 // the stored point
                    ld_nat(cz[h], my_pv + (size_t)P_Z * P.dpad, s, g, dim); ld_nat(cg[h], my_pv + (size_t)P_GZ * P.dpad, s, g, dim);
                }
                if (round_mode == M_RECOMP || round_mode == M_KEEP) {
                    double fd = 0.;
#pragma unroll
                    for (int r = 0; r < 4; ++r) fd = fd + (cz[h].a[r] + cg[h].a[r]) * (cz[h].a[r] + cg[h].a[r]);
                    emit(s, 0, fd);
                }
            }
            int npass = (ngroups + GROUPS_PER_PASS - 1) / GROUPS_PER_PASS;
            for (int o = 1; o < 64; o <<= 1) { const int q = __shfl_xor(npass, o); npass = q > npass ? q : npass; }
            if (npass < 1) npass = 1;
            npass = __builtin_amdgcn_readfirstlane(npass);
            for (int pass = 0; pass < npass; ++pass) {
                // the groups of this pass: group i (0-based over the leaf's list) = level k = i + 2 for i < t - 1, then the top level
                if (ngroups > pass * GROUPS_PER_PASS) {
                    const int t = (int)__builtin_ctz(~(unsigned)k_n);
                    for (int gi = 0; gi < GROUPS_PER_PASS; ++gi) {
                        const int i = pass * GROUPS_PER_PASS + gi;
                        if (i < ngroups) {
                            int za, zl, zb;
                            if (i < t - 1) {                       // sub-tree merge of level k (src/nuts.rs:143-161; nuts_lane.hpp l_transition)
                                const int k = i + 2;
                                const unsigned a_first = (unsigned)k_n + 1u - (1u << k);
                                za = slot_F(a_first == 0 ? k_depth : (int)__builtin_ctz(a_first));
                                zl = slot_L(MD, k - 1);
                                zb = k == 2 ? -1 : slot_F(k - 1);
                            } else {                               // the main tree against the finished sub-tree
                                za = EDGE0_Z + 3 * (k_fwd ? k_ls : k_rs);
                                zl = EDGE0_Z + 3 * (k_fwd ? k_rs : k_ls);
                                zb = k_depth == 1 ? -1 : slot_F(k_depth);
                            }
#pragma unroll
                            for (int h = 0; h < SPW; ++h) {
                                Vec4 az, av, lz, lvv, bz, bv;
                                ldS(az, za, h); ldS(av, za + 1, h);
                                ldS(lz, zl, h); ldS(lvv, zl + 1, h);
                                if (zb >= 0) { ldS(bz, zb, h); ldS(bv, zb + 1, h); } else { bz = pz[h]; bv = pvv[h]; }
                                double a6[6];
                                turn_group(az, av, lz, lvv, bz, bv, cz[h], cv[h], a6);
#pragma unroll
                                for (int j = 0; j < 6; ++j) emit(w * SPW + h, 8 + 6 * gi + j, a6[j]);
                            }
                        }
                    }
                }
                blk_barrier();
                {   // thread (slot j, chain cc) adds the 16 stripes in stripe order
                    const int T = (int)threadIdx.x;
                    static_assert(NT == NRED * LC, "one thread per (slot, chain)");
                    const int j = T / LC, cc = T % LC;
                    double tot = red[(size_t)j * LC + cc];
#pragma unroll
                    for (int st = 1; st < LS; ++st) tot = tot + red[((size_t)st * NRED + j) * LC + cc];
                    const int dst = j < 8 ? (pass == 0 ? j : -1) : 8 + 6 * GROUPS_PER_PASS * pass + (j - 8);
                    if (dst >= 0 && dst < NSUM) S.sums[cc][dst] = tot;
                }
                blk_barrier();
            }
            NM_LKP(6)
#if NM_LOCK_PROF
            lkp_rounds += 1;
#endif
        }
#if NM_LOCK_PROF
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            for (int i = 0; i < 8; ++i) (void)__hip_atomic_fetch_add(&P.prof[i], lkp_acc[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            (void)__hip_atomic_fetch_add(&P.prof[8], lkp_rounds, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#endif
        // ---- the tile is done: the chains' records go back
        __syncthreads();
        if (threadIdx.x < LC) {
            const int cc = (int)threadIdx.x;
            const uint64_t chain = S.ch[cc].chain;
            if (chain < P.n_chains) {
                ChainScalars& q = S.sc[cc];
                if (S.ch[cc].was_live) q.rng_pos = S.ch[cc].rng_pos;
                P.sc[chain] = q;
            }
        }
    }
}

}  // namespace lock
}  // namespace nm
