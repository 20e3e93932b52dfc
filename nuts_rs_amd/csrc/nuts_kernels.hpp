// nuts_kernels.hpp — the many-chain NUTS kernels for gfx950 (MI355X).
//
// ONE BLOCK OF W WAVEFRONTS (W = 1, 2 or 4) = ONE CHAIN AT A TIME.  The block keeps two live phase-space points (ping-pong: even leaf E,
// odd leaf O; each z, v, g_z) and the chain's mass matrix (sigma, mu) in VGPRs (DPL doubles per lane per
// vector), runs the whole NUTS transition — momentum refresh, every doubling of the tree with the fused
// leapfrog + logp/grad, the U-turn / divergence tests, the multinomial merges, then the per-draw adaptation —
// and loops over draws and over its share of the chains without returning to the host.
// All per-chain control flow (tree depth, termination, RNG consumption) is wave-uniform, so ragged trees cost
// no lane divergence, and a chain that stops early lets its wave move on (no host-side active mask, no stream
// compaction between doublings).
// HBM is touched only for what the tree must remember: sub-tree end points that are needed again after the
// next leaf (U-turn tests between non-adjacent states), one z-vector per multinomial candidate, and the two
// edges of the main tree.  That scratch belongs to the WAVE, not to the chain (a launch uses one wave per
// resident slot and strides over the chains), so its footprint is small enough to live in L2 / Infinity Cache.
//
// Everything here is force-inlined into the kernels: the register tiles are passed by reference, and a real
// call would force them through scratch memory.
//
// Reference semantics (pymc-devs/nuts-rs 0.18.3), cited per function below:
//   src/nuts.rs:94-388, src/dynamics/transformed_hamiltonian.rs:161-262,:310-351,:524-736,
//   src/transform/diagonal.rs:85-265, src/transform/adapt/diagonal.rs:17-236, src/adapt_strategy.rs:77-222,
//   src/stepsize/adapt.rs:52-272, src/stepsize/dual_avg.rs:34-166, src/chain.rs:137-188.
#pragma once
#include <type_traits>
#include "dev_math.hpp"
#include "../../include/nuts_amd.h"

namespace nm {

constexpr int MAX_MAXDEPTH = 20;

// LDS residency switches (tuning): which tree end points of the resident chain stay on the CU
#ifndef NM_TRIM_FIRST
#define NM_TRIM_FIRST 1   // first doubling / depth-1 top-level tests use the points still held in E
#endif
#ifndef NM_LDS_L1
#define NM_LDS_L1 1      // L[1]: last leaf of the pending level-1 sub-tree
#endif
// U-turn operand loops: at most this many unrolled iterations of operand loads may be in flight (a scheduling barrier
// follows each group) — unbounded hoisting of the 6 operand streams costs ~190 VGPRs at DPL 16 and ends in scratch spills
#ifndef NM_CHECK_GROUP
#define NM_CHECK_GROUP 2
#endif
#define NM_GROUP_BARRIER(m) do { if (((m) + 1) % NM_CHECK_GROUP == 0) __builtin_amdgcn_sched_barrier(0); } while (0)

// ---------------------------------------------------------------------------------------------
// HBM layout.
//   pvec[chain][PSlot][DP]  : what a chain owns between draws
//   svec[wave][scratch][DP] : tree scratch of a resident wave
// ---------------------------------------------------------------------------------------------
enum PSlot : int {
    P_X = 0, P_GX, P_Z, P_GZ,          // current point (TransformedPoint, transformed_hamiltonian.rs:56-77)
    P_SIG, P_ISIG, P_MU,               // DiagMassMatrix stds / inv_stds / mean (transform/diagonal.rs:9-17)
    E_DM, E_DV, E_GM, E_GV,            // foreground RunningVariance of draws / grads (adapt/diagonal.rs:108-115)
    B_DM, B_DV, B_GM, B_GV,            // background
    P_V,                               // MclmcChain: the momentum survives from draw to draw (src/mclmc.rs:551)
    NUM_PSLOT
};
// main-tree edges: `left` and `right` are edge ids 0/1 (both start as the initial point's id 0; a successful doubling
// writes the new edge to the id this side owns alone, or to id 1 while both sides still share id 0).  Only the g_z of
// an edge lives here (EDGEn_G); its (z, v) stay in LDS (BlockShared::edge_z / edge_v).
enum SSlot : int { EDGE0_Z = 0, EDGE0_V, EDGE0_G, EDGE1_Z, EDGE1_V, EDGE1_G, EDGE2_Z, EDGE2_V, EDGE2_G, STAGE_V, S_DYN };
__host__ __device__ inline int slot_edge(int id) { return EDGE0_Z + 3 * id; }
// dynamic scratch: F[k] (z,v), L[k] (z,v) for k in 0..=maxdepth, then the candidate pool C[p] (z), p in 0..maxdepth+2
__host__ __device__ inline int slot_F(int k) { return S_DYN + 2 * k; }
__host__ __device__ inline int slot_L(int maxdepth, int k) { return S_DYN + 2 * (maxdepth + 1) + 2 * k; }
__host__ __device__ inline int slot_C(int maxdepth, int p) { return S_DYN + 4 * (maxdepth + 1) + p; }
__host__ __device__ inline int num_sslots(int maxdepth) { return S_DYN + 4 * (maxdepth + 1) + (maxdepth + 3); }
// Batched merges (DPL <= 4, one wave per chain; see resolve_chunk): the z of the last NM_RING leaves, one slot per leaf (leaf n of
// a doubling lives in slot n mod NM_RING), behind the slots above.  The host adds NM_RING to nsslot for these tilings.
constexpr int NM_RING = 64;
__host__ __device__ inline int slot_R(int maxdepth, int i) { return num_sslots(maxdepth) + i; }
// The depth every slot family (F, L, candidate pool, pend tables) is laid out for.  A tree that turned at depth d <= maxdepth is extended by
// `extra_doublings` further doublings (src/nuts.rs:350-371), so sub-trees of level up to maxdepth + extra_doublings - 1 are built; with the
// layout of `maxdepth` alone slot_L(MD, MD + 1) is slot_C(MD, 0) and the candidate pool overflows.
// (Computed once by the host into KParams::layout_md: reading both settings fields in ctx_begin changed the register allocation of the
// 16-wavefront matrix-core kernel enough to break it — the last doubling of trees deeper than 6 stopped after one leaf)
__host__ __device__ inline int layout_depth(const nm_settings& s) { return (int)(s.maxdepth + s.extra_doublings); }
#ifndef NM_REG_EDGES
#define NM_REG_EDGES 1            // the main tree's two end points live in registers (0: in the HBM scratch, rounds 1-4; bisecting builds)
#endif
#ifndef NM_EDGES_IN_ACC
#define NM_EDGES_IN_ACC 0
#endif
#ifndef NM_LF_FMA_FORM
#define NM_LF_FMA_FORM 0          // the fused leapfrog's two fused multiply-adds: 0 inline asm at the call sites of NM_X_ASM_SITES, 2 __builtin_fma everywhere
#endif
#ifndef NM_FUSED_LEAPFROG
#define NM_FUSED_LEAPFROG 1        // 0: the leapfrog as three loops over the tile for every density (rounds 1-4; bisecting builds)
#endif
#ifndef NM_MERGE_MATH_ROUTINE
#define NM_MERGE_MATH_ROUTINE 1   // 0: merge_weights through the general-purpose exp / ln_1p (rounds 1-4; bisecting builds)
#endif
#ifndef NM_BATCH_MERGES
#define NM_BATCH_MERGES 1        // 0: every merge evaluated where the reference evaluates it (tuning / bisecting builds)
#endif
#ifndef NM_BATCH_IN_TILES
#define NM_BATCH_IN_TILES 0
#endif
// Main-tree end points in registers (round 5).  The tree's left and right ends (z, v, g_z each) were scratch slots in HBM: written after
// every doubling that is followed by another, read back by every top-level U-turn test (four tiles) and when the trajectory is extended on
// the other side (three) — 30 of the ~64 tiles of 8 KiB a K2 draw moved, every read a round trip of a microsecond or two under load on the
// wavefront's critical path.  A block of one wavefront per SIMD owns 512 registers per lane; two live points and two end points are
// 12 tiles of 32 = 384 — which the register allocator does not manage without spilling to scratch memory (measured: 1088 B per lane), so
// the end points' (z, v) — what every test reads — are registers and their g_z, read only when the trajectory is extended on the other
// side, stays in its slot: 10 tiles.  Not for the register-capped tile / cluster builds.
// Measured (profiles/r05f_*): (16 doubles, 1 wave) K2 2.02e11 -> 2.13e11, (2, 1) K3 +7 %; the register-capped tilings in between lose
// ((8, 1) at two wavefronts per SIMD: -33 %, (4, 1): -7 %: the end points go to scratch memory) and keep the slots.
template <int DPL, int W> constexpr bool reg_edges() { return bool(NM_REG_EDGES) & bool(NM_TRIM_FIRST) & !bool(NM_TILE_MODE) & !bool(NM_CLUSTER_MODE) & (W == 1) & (DPL == 16 || DPL == 2); }
// timing experiment only (results are wrong): the U-turn tests of levels >= 2 and the top-level test read registers instead of their scratch slots
#ifdef NM_X_NO_TEST_LOADS
#define NM_TLD(r, so, m, alt) make_double2((alt).a[2 * (m)], (alt).a[2 * (m) + 1])
#else
#define NM_TLD(r, so, m, alt) C.ld2(r, so, m)
#endif
// Round 6 ("NOG" + "FD", the (16 doubles, 1 wavefront) tiling with an element-wise density and the diagonal transformation: K2).  The launch is bound by the
// bytes the tree's end points move (21 B per step x dim against 3.7 necessary; every phase that touches memory stalls), so:
//   NOG  a point is (z, v): its transformed gradient g_z is NOT kept.  For an element-wise density g_z is three operations per element away from z
//        (x = sigma z + mu, g_x = density'(x), g_z = sigma g_x — the very operations that produced it, hence the same bits), so the leapfrog
//        recomputes the start point's g_z on the fly (+4 instructions per element pair) instead of holding two more tiles of 32 registers and an
//        HBM slot per main-tree edge.  The one gradient that is NOT a function of its z — the trajectory's initial point, whose z a
//        re-whitening may have recomputed from x (transformed_hamiltonian.rs:687-736, diagonal.rs:210-221) — is streamed from its slot P_GZ.
//   FD   the 64 registers that frees hold (z, v) of the FIRST leaf of the doubling in progress (F[depth]): the operand of every level-k test of
//        a sub-tree that starts at leaf 0 and of the top-level test's third pair (src/nuts.rs:143-161) — written once and read two or three
//        times per doubling before, now never in memory: the top-level tests read registers only.
#ifndef NM_NOG
#define NM_NOG 1
#endif
#ifndef NM_NOG_82
#define NM_NOG_82 0   // the gradient-free points on the (8 doubles, 2 wavefronts) tiling too (measured: profiles/r06s_*)
#endif
#ifndef NM_TEST_CHUNK
#define NM_TEST_CHUNK 4      // (16-doubles tiling) iterations the level-k / top-level U-turn tests request their slots' rows ahead: measured 0 / 2 / 4 / 8 (profiles/r06u_*)
#endif
#ifndef NM_TEST_CHUNK3_MAX
#define NM_TEST_CHUNK3_MAX 2
#endif
#ifndef NM_FD
#define NM_FD 0      // measured (profiles/r06n_k2_nog_variants.txt): the 64 registers cost more in spills than the loads they save
#endif
#ifndef NM_GTILE
#define NM_GTILE 1     // 1: one shared gradient tile instead of recomputing the start point's g_z in every leapfrog (32 registers for 64 instructions per leapfrog)
#endif
template <int DPL, int W, class Dens> constexpr bool nog_mode();
#ifndef NM_BATCH_MAX_DPL
#define NM_BATCH_MAX_DPL 4      // the widest one-wavefront tiling with the batched merges (resolve_chunk); the host's scratch layout follows it
#endif
template <int DPL, int W> constexpr bool batched_merges() { return NM_BATCH_MERGES && DPL <= NM_BATCH_MAX_DPL && W == 1 && (!NM_TILE_MODE || NM_BATCH_IN_TILES) && !NM_CLUSTER_MODE; }

// Per-chain scalars (everything of NutsChain / GlobalStrategy / stepsize::Strategy / DualAverage that is not a vector)
struct ChainScalars {
    uint32_t key[8];
    uint64_t rng_pos;
    // current point
    double logp, logdet;
    int64_t transform_id;
    // mass matrix
    double mm_logdet;
    int64_t mm_id;
    // hamiltonian
    double step_size;
    // GlobalStrategy (adapt_strategy.rs:24-39)
    uint64_t draw_count, last_update, current_window_size, tuning, has_initial_mass_matrix;
    // RunningVariance counts (draw and grad estimators always have equal counts)
    uint64_t cnt_fg, cnt_bg;
    // DualAverage (dual_avg.rs:34-41)
    double log_step, log_step_adapted, hbar, mu;
    uint64_t da_count;
    // Adam (stepsize/adam.rs:42-53); its log_step shares `log_step` above
    double adam_m, adam_v;
    uint64_t adam_t;
    // stepsize::Strategy last_* (stepsize/adapt.rs:58-64)
    double last_mean_tree_accept, last_sym_mean_tree_accept, last_max_energy_error;
    uint64_t last_n_steps;
    uint64_t status;       // NM_CHAIN_*
    uint64_t total_steps;  // leapfrogs since creation (metric)
    uint64_t px_stale;     // 1: P_X / P_GX do not hold the current point (they equal what P_Z recomputes to)
    int64_t stats_last_id; // mass-matrix id at the previous statistics extraction (chain.rs:195-200), starts at -1
    // ---- low-rank transformation and its adaptation (NM_ADAPT_LOW_RANK; LrWrap kernels only)
    uint64_t lr_has_inner;        // LowRankMassMatrix.inner.is_some() (low_rank.rs:112)
    uint64_t lr_rank;             // InnerMatrix.num_eigenvalues
    // LowRankMassMatrixStrategy (adapt/low_rank.rs:14-21): the deque is rows [lr_start, lr_start + lr_len) of the chain's
    // window buffer; lr_split = background_split
    uint64_t lr_start, lr_len, lr_split;
    // hand-off with the host's estimator: 1 = GlobalStrategy::adapt paused at `mass_matrix_adapt.adapt(..)`, waiting;
    // 2 = the host answered (lr_upd_*); 3 = a transformation from nm_engine_set_transform waits to be committed
    uint64_t lr_pending;
    uint64_t lr_upd_ok, lr_upd_rank;   // the answer: a finite update was uploaded (P_SIG / P_ISIG / P_MU, lrvec, lrval), its rank
    double lr_upd_logdet;              // its -1/2 sum ln lambda (InnerMatrix::new, low_rank.rs:57)
    uint64_t lr_is_late;               // is_late of the paused adapt call
    uint64_t lr_row;                   // output row of the paused draw
    // KinWrap kernels: the Hamiltonian's current KineticEnergyKind (NutsSettings::trajectory_kind; MclmcChain switches it)
    uint64_t kin;
};
enum { LR_IDLE = 0, LR_WAIT_HOST = 1, LR_ANSWERED = 2, LR_SET_TRANSFORM = 3 };

struct KParams {
    nm_settings s;
    uint64_t n_chains, dim, dpad, chain_id_offset, nsslot;
    double* pvec;
    double* svec;
    ChainScalars* sc;
    const double* zig_x;
    const double* zig_f;
    const double* logp_params;
    // derived schedule constants (GlobalStrategy::new, adapt_strategy.rs:77-98)
    uint64_t early_end, final_step_size_window;
    uint64_t mclmc_switch_draw;        // MclmcChain::switch_draw (sampler.rs:441)
    // chains wider than one block (NM_CLUSTER_MODE kernels): cl_k members of cl_slice elements each per chain, their mailboxes
    uint64_t cl_k, cl_slice;
    uint64_t cl_general;               // 1: always the general (release / acquire) exchange, never the same-XCD one (NM_CLUSTER_GENERAL=1: tests)
    unsigned long long* cl_box;        // [clusters][2][cl_k][RED_MAX_VALUES]
    unsigned long long* cl_cnt;        // [clusters], zeroed before every launch
    double ln_max_step;                // ln(da_max_step_size), dual_avg.rs:59
    double jitter_low, jitter_scale;   // Uniform::new(1-j, 1+j) (stepsize/adapt.rs:259-261)
    // outputs of the draw kernel
    double* out_positions;       // [n_draws][n_chains][dim] or null
    nm_draw_stats* out_stats;    // [n_draws][n_chains] or null
    // vector-valued statistics (nm_draw_outputs), [n_draws][n_chains][dim] or null
    double *out_gradient, *out_tpos, *out_tgrad, *out_mm_inv, *out_mm_mu;
    double *out_div_start, *out_div_start_grad, *out_div_end;
    uint64_t n_draws;
    unsigned long long* prof;    // NM_PROF builds: cycle counters of block 0 (tools/prof_phases.py)
    const double* x0;            // init kernel: [n_chains][dim]
    // low-rank transformation (LrWrap kernels): per chain [1 + lr_rmax][dpad] (row 0 = mu_lr, row 1 + k = eigenvector k,
    // tile layout like every chain vector), [2][lr_rmax] (lambda^1/2, lambda^-1/2), and the window [lr_cap][2][dim]
    double* lrvec;
    double* lrval;
    double* lrwin;
    uint64_t lr_rmax, lr_cap;
    uint64_t draw_end, row_base;  // LrWrap draw kernel: chains draw until draw_count == draw_end; output row = draw_count - row_base
    double* out_mm_eigvals;
    const uint8_t* init_mask;     // init kernel: chains with mask 0 are left untouched (null = all)
    // NM_LOGP_HOST_CALLBACK: one mailbox per chain in pinned host memory (CbMail header, then x[dim], grad[dim])
    unsigned char* cb_mail;
    uint64_t cb_stride;           // bytes per mailbox
    uint64_t layout_md;           // layout_depth(s) = maxdepth + extra_doublings: the depth the tree's slot families are laid out for
};

// Phase timing for development (-DNM_PROF=1): block 0 accumulates shader-clock cycles between marks into P.prof[].
#ifndef NM_PROF
#define NM_PROF 0
#endif
#if NM_PROF
#define NM_MARK(C, slot)                                                                  \
    {                                                                                     \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();                     \
        if (blockIdx.x == 0 && threadIdx.x == 0)                                          \
            (void)__hip_atomic_fetch_add(&(C).P.prof[slot], now_ - (C).prof_t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    \
        (C).prof_t = now_;                                                                \
    }
#else
#define NM_MARK(C, slot)
#endif

// ---------------------------------------------------------------------------------------------
// register tiles
// ---------------------------------------------------------------------------------------------
template <int DPL>
struct Tile {
    double a[DPL];
};

// Tiling of a chain vector over the T = 64*W threads of its block: thread t holds, for m = 0..DPL/2-1, the PAIR of
// elements d = 2*(m*T + t) + {0,1} (16 B per thread, consecutive threads contiguous).
template <int DPL, int W>
NM_DEV void load_tile(Tile<DPL>& t, const double* base) {
    const double2* p = reinterpret_cast<const double2*>(base) + tid();
#pragma unroll
    for (int m = 0; m < DPL / 2; ++m) {
        double2 q = p[m * 64 * W];
        t.a[2 * m] = q.x;
        t.a[2 * m + 1] = q.y;
    }
}
template <int DPL, int W>
NM_DEV void store_tile(const Tile<DPL>& t, double* base) {
    double2* p = reinterpret_cast<double2*>(base) + tid();
#pragma unroll
    for (int m = 0; m < DPL / 2; ++m) p[m * 64 * W] = make_double2(t.a[2 * m], t.a[2 * m + 1]);
}
// element index held in register k of this thread
// Where the end points live.  Plain values: the allocator keeps them in the accumulation registers of the (16, 1) kernel (all 512
// registers of its SIMD) at ~150 more vector instructions per leapfrog.  Placing them there BY HAND (NM_EDGES_IN_ACC = 1: v_accvgpr_write /
// read, one per 32 bits, a write per doubling, a read per test) was built and is worse: the 128 accumulation registers it pins are the
// allocator's spill space, and what no longer fits goes to scratch memory (544 B per lane against 384).
template <int DPL, bool ACC> struct EdgeTile;
template <int DPL> struct EdgeTile<DPL, false> {
    Tile<DPL> t;
    NM_DEV void put(const Tile<DPL>& s) { t = s; }
    NM_DEV void get(Tile<DPL>& d) const { d = t; }
    NM_DEV double2 pair(int m) const { return make_double2(t.a[2 * m], t.a[2 * m + 1]); }
};
template <int DPL> struct EdgeTile<DPL, true> {
    int w[2 * DPL];
    NM_DEV void put(const Tile<DPL>& s) {
#pragma unroll
        for (int k = 0; k < DPL; ++k) {
            asm("v_accvgpr_write_b32 %0, %1" : "=a"(w[2 * k]) : "v"(__double2loint(s.a[k])));
            asm("v_accvgpr_write_b32 %0, %1" : "=a"(w[2 * k + 1]) : "v"(__double2hiint(s.a[k])));
        }
    }
    NM_DEV double at(int k) const {
        int lo, hi;
        asm("v_accvgpr_read_b32 %0, %1" : "=v"(lo) : "a"(w[2 * k]));
        asm("v_accvgpr_read_b32 %0, %1" : "=v"(hi) : "a"(w[2 * k + 1]));
        return __hiloint2double(hi, lo);
    }
    NM_DEV void get(Tile<DPL>& d) const {
#pragma unroll
        for (int k = 0; k < DPL; ++k) d.a[k] = at(k);
    }
    NM_DEV double2 pair(int m) const { return make_double2(at(2 * m), at(2 * m + 1)); }
};
template <int W>
NM_DEV int elem_index(int k) { return 2 * ((k >> 1) * 64 * W + tid()) + (k & 1); }

// ---------------------------------------------------------------------------------------------
// HBM access through buffer descriptors (guide T8/T20).  A chain vector is addressed as
//   SRD(region base, wave-uniform)  +  soffset = slot * DP * 8 (SGPR)  +  voffset = tid*16 + m*T*16 (one VGPR + immediate)
// so no 64-bit per-lane address is ever formed: with plain pointers the compiler hoists one such address per
// (slot, m) out of the draw loop — hundreds of VGPRs that end up spilled to scratch and reloaded in the hot loop.
// ---------------------------------------------------------------------------------------------
typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
typedef unsigned int v2u __attribute__((ext_vector_type(2)));

// A wave-uniform int the compiler cannot second-guess.  A plain __builtin_amdgcn_readfirstlane is folded away when
// the IR uniformity analysis already calls its input uniform, yet instruction selection may still have computed that
// input on the VALU — and then every buffer op taking it as soffset gets wrapped in a waterfall loop (T20), which
// serialises the loads.  Laundering the value through an empty asm (opaque, "divergent" to the analysis) keeps the
// readfirstlane, so the result is a real SGPR; the compiler still sees the builtin and pads its hazards.
// a wave-uniform value the compiler must keep in a register (not re-load from where it came: kernel arguments are "rematerialisable")
NM_DEV double pin_scalar(double x) {
    int lo = __builtin_amdgcn_readfirstlane(__double2loint(x)), hi = __builtin_amdgcn_readfirstlane(__double2hiint(x));
    asm volatile("" : "+s"(lo), "+s"(hi));
    return __hiloint2double(hi, lo);
}
NM_DEV int force_sgpr(int x) {
    asm volatile("" : "+v"(x));
    return __builtin_amdgcn_readfirstlane(x);
}

NM_DEV rsrc_t make_rsrc(const void* base, uint64_t bytes) {
    // the inputs ARE wave-uniform; readfirstlane makes that provable so no waterfall loop is generated (T20)
    const uint64_t b = (uint64_t)base;
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(b >> 32));
    const int n = __builtin_amdgcn_readfirstlane((int)(uint32_t)bytes);
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), (short)0, n, 0x00020000);
}
NM_DEV double2 buf_load2(rsrc_t r, int voff, int soff) {
    v4u q = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
    return make_double2(__hiloint2double((int)q.y, (int)q.x), __hiloint2double((int)q.w, (int)q.z));
}
// cache-policy bits of the raw buffer intrinsics on gfx940+: sc0 = 1, nt = 2, sc1 = 16
#ifndef NM_NT_STORES
#define NM_NT_STORES 1
#endif
constexpr int NM_AUX_NT = NM_NT_STORES ? 2 : 0;
// Round 6: the one-wavefront 16-doubles tiling (K2) stores its candidates / per-draw state with the default policy: a lone wavefront waits for its
// stores wherever the next s_waitcnt vmcnt happens to be (loads and stores share the counter), and a streaming store is acknowledged later than a
// write-back one (measured, profiles/r06n_k2_nog_variants.txt: 2.10 -> 2.15e11 on the round-5 kernel, 2.24 -> 2.27e11 with gradient-free points).
#ifndef NM_NT_MAX_DPL
#define NM_NT_MAX_DPL 8
#endif
template <int DPL, int W> constexpr int aux_nt() { return (NM_NT_STORES && (DPL <= NM_NT_MAX_DPL || W > 1)) ? 2 : 0; }
// Output rows of a draw: plain masked 8-byte stores (0), a per-row buffer descriptor with 16-byte stores for every vector output (1) or for the
// position row only (2), and the cache policy of those stores.  Measured on K2 with every draw recorded (tools/gpu_wrb.sh,
// profiles/r04zz_write_row_variants.txt): 0: 75.8 ms per 200 draws, 1: 73.4, 2: 72.4, 2 + non-temporal: 72.1 (without recording: 71 - 72 ms in
// every form) — the position row is the one output every caller takes; the kernel is at its register cap and the form that touches the least
// code wins.
#ifndef NM_WRITE_ROW_BUF
#define NM_WRITE_ROW_BUF 2
#endif
#ifndef NM_ROW_AUX
#define NM_ROW_AUX 2
#endif
// The data registers of a 128-bit buffer store must not be overwritten right behind it.  LLVM's hazard recogniser knows this
// hazard ("VMEM store of more than 64 bits, then a VALU write of its data VGPRs": 1 wait state) but exempts MUBUF stores whose
// soffset is an SGPR — the addressing used here — and gfx950 does show it: `buffer_store_dwordx4 v[4:7], .., s8 offen` directly
// followed by `v_fma_f64 v[4:5], ..` stored the fma's result in lanes 12-15 of every row (first seen as a background variance
// estimator that came out of set_position as 1e-9 instead of 0 in a few chains; tools/probes/determinism2.py).  Every such
// store is therefore followed by an `s_nop 1` that names its data registers as inputs: they stay live for two wait states
// (tools/check_store_hazard.py scans the generated assembly for unguarded instances).
template <int AUX>
NM_DEV void buf_store2_aux(rsrc_t r, int voff, int soff, double a, double b) {
    v4u q;
    q.x = (unsigned)__double2loint(a); q.y = (unsigned)__double2hiint(a);
    q.z = (unsigned)__double2loint(b); q.w = (unsigned)__double2hiint(b);
    __builtin_amdgcn_raw_buffer_store_b128(q, r, voff, soff, AUX);
#ifndef NM_X_NO_STORE_GUARD            // (defined only by tests/test_build_guards.py: the build must REJECT the unguarded form)
    asm volatile("s_nop 1" ::"v"(q));     // the store-data hazard above
#endif
}
NM_DEV void buf_store2(rsrc_t r, int voff, int soff, double a, double b) {
    v4u q;
    q.x = (unsigned)__double2loint(a); q.y = (unsigned)__double2hiint(a);
    q.z = (unsigned)__double2loint(b); q.w = (unsigned)__double2hiint(b);
    __builtin_amdgcn_raw_buffer_store_b128(q, r, voff, soff, 0);
    asm volatile("s_nop 1" ::"v"(q));     // the store-data hazard above
}

// ---------------------------------------------------------------------------------------------
// densities: eval(x, gx, dim) -> logp (wave-uniform), fills gx; padded elements (index >= dim) must give
// zero terms and zero gradient.
// ---------------------------------------------------------------------------------------------
struct IidNormal {
    static constexpr bool kNeedsLdsVector = false;
    // element form of eval (leapfrog's fused loop): gradient element and, through `term`, what eval adds to its sum for this element
    static constexpr bool kElementwise = true;
    template <int W, bool FULL>
    NM_DEV double elem(double xk, int k, int dim, double& term) const {
        const double diff = xk - mu;
        const double t = -0.5 * diff * diff;
        if constexpr (FULL) { term = t; return -diff; }
        else { const bool valid = elem_index<W>(k) < dim; term = valid ? t : 0.0; return valid ? -diff : 0.0; }
    }
    NM_DEV double finish(double sum) const { return sum; }
    NM_DEV void set_lds(double*) {}   // reference benches/sample.rs:49-62
    double mu;
    template <int W>
    NM_DEV void init(const double* params, int, Reducer<W>&) { mu = params[0]; }
    template <int W>
    NM_DEV void init_slice(const double* params, int, int, int, Reducer<W>&) { mu = params[0]; }   // cluster mode: this block holds one slice
    template <int DPL, int W>
    NM_DEV double eval(const Tile<DPL>& x, Tile<DPL>& gx, int dim, Reducer<W>& R) const {
        double acc = 0.0;
#if !NM_CLUSTER_MODE
        if (dim == DPL * 64 * W) {          // a full tile has no padding: no validity selects (4 v_cndmask + a compare per element; uniform branch)
#pragma unroll
            for (int k = 0; k < DPL; ++k) {
                double diff = x.a[k] - mu;
                double term = -0.5 * diff * diff;
                gx.a[k] = -diff;
                acc = acc + term;
            }
            return R.sum(acc);
        }
#endif
#pragma unroll
        for (int k = 0; k < DPL; ++k) {
            bool valid = elem_index<W>(k) < dim;
            double diff = x.a[k] - mu;
            double term = -0.5 * diff * diff;
            gx.a[k] = valid ? -diff : 0.0;
            acc = acc + (valid ? term : 0.0);
        }
        return R.sum(acc);
    }
};

struct DiagNormal {
    static constexpr bool kNeedsLdsVector = false;
    static constexpr bool kElementwise = true;
    template <int W, bool FULL>
    NM_DEV double elem(double xk, int k, int dim, double& term) const {
        const int d = elem_index<W>(k);
        const bool valid = FULL || d < dim;
        const double p = valid ? prec[d] : 0.0;
        const double px = p * xk;
        term = valid ? xk * px : 0.0;
        return valid ? -px : 0.0;
    }
    NM_DEV double finish(double sum) const { return -0.5 * sum + norm; }
    NM_DEV void set_lds(double*) {}  // diagonal-P case of the MvNormal fixture, reference src/transform/mod.rs:98-112
    const double* prec;
    double norm;
    template <int W>
    NM_DEV void init(const double* params, int dim, Reducer<W>& R) {
        prec = params;
        double acc = 0.0;
        for (int m = 0; m < (dim + 128 * W - 1) / (128 * W); ++m)
            for (int j = 0; j < 2; ++j) {
                int d = 2 * (m * 64 * W + tid()) + j;
                acc = acc + (d < dim ? dlog(params[d < dim ? d : 0]) : 0.0);
            }
        double log_det_p = R.sum(acc);
        norm = -0.5 * ((double)dim * ulog(6.283185307179586) - log_det_p);
    }
    // cluster mode: this block holds elements [goff, goff + dim) of a chain of gdim elements (R.sum spans the whole chain)
    template <int W>
    NM_DEV void init_slice(const double* params, int dim, int gdim, int goff, Reducer<W>& R) {
        prec = params + goff;
        double acc = 0.0;
        for (int m = 0; m < (dim + 128 * W - 1) / (128 * W); ++m)
            for (int j = 0; j < 2; ++j) {
                int d = 2 * (m * 64 * W + tid()) + j;
                acc = acc + (d < dim ? dlog(prec[d < dim ? d : 0]) : 0.0);
            }
        double log_det_p = R.sum(acc);
        norm = -0.5 * ((double)gdim * ulog(6.283185307179586) - log_det_p);
    }
    template <int DPL, int W>
    NM_DEV double eval(const Tile<DPL>& x, Tile<DPL>& gx, int dim, Reducer<W>& R) const {
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < DPL; ++k) {
            int d = elem_index<W>(k);
            bool valid = d < dim;
            double p = valid ? prec[d] : 0.0;
            double px = p * x.a[k];
            gx.a[k] = valid ? -px : 0.0;
            acc = acc + (valid ? x.a[k] * px : 0.0);
        }
        double quad = -0.5 * R.sum(acc);
        return quad + norm;
    }
};

// Neal's funnel (defined by this repo, absent from the reference): x[0] = v ~ N(0, 3^2),
// x[i] | v ~ N(0, e^v), i = 1..dim-1.  logp = -v^2/18 - (k/2) v - e^{-v}/2 * sum x_i^2  (k = dim - 1).
// Same operation order as oracle/nmo_nuts.hpp LOGP_FUNNEL.
struct Funnel {
    static constexpr bool kNeedsLdsVector = false;
    NM_DEV void set_lds(double*) {}
    template <int W>
    NM_DEV void init(const double*, int, Reducer<W>&) {}
    template <int DPL, int W>
    NM_DEV double eval(const Tile<DPL>& x, Tile<DPL>& gx, int dim, Reducer<W>& R) const {
        // element 0 lives in thread 0, register 0; for W > 1 it reaches the other waves as a sum with zeros (exact)
        const double v = W == 1 ? readlane_f64(x.a[0], 0) : R.sum(tid() == 0 ? x.a[0] : 0.0);
        const double kk = (double)(dim - 1);
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < DPL; ++k) {
            int d = elem_index<W>(k);
            bool in = d >= 1 && d < dim;
            acc = acc + (in ? x.a[k] * x.a[k] : 0.0);
        }
        const double ss = R.sum(acc);
        const double ev = uexp(-v);
        const double g0 = -v / 9.0 - 0.5 * kk + 0.5 * ev * ss;
#pragma unroll
        for (int k = 0; k < DPL; ++k) {
            int d = elem_index<W>(k);
            gx.a[k] = d == 0 ? g0 : (d < dim ? -ev * x.a[k] : 0.0);
        }
        return -v * v / 18.0 - 0.5 * kk * v - 0.5 * ev * ss;
    }
};

// Non-centered eight schools (defined by this repo): x = (mu, log tau, theta~[8]), params = y[8], sigma[8].
// mu ~ N(0,5^2), tau ~ HalfCauchy(5) (+ log-Jacobian), theta~ ~ N(0,1), y_i ~ N(mu + tau theta~_i, sigma_i^2).
// Same operation order as oracle/nmo_nuts.hpp LOGP_EIGHT_SCHOOLS (dim = 10: elements 2l, 2l+1 in lane l).
// Normal with a full (symmetric) precision matrix P: logp = -1/2 x'Px, grad = -Px (BASELINE config K5; the reference's
// MvNormal fixture, src/transform/mod.rs:98-112, has this form but only diagonal P — the definition with a full P and
// no normalising constant is this repo's).  One chain's gradient is a GEMV; it is computed column by column,
// y_d = sum_j P[j][d] x_j with j ascending and one fma per term (P symmetric, so row j read contiguously IS column j):
// every lane streams its own elements of row j (coalesced, L2-resident: P is shared by all chains) and x_j is an LDS
// broadcast.  MFMA does not apply: one chain has one right-hand side, and chains do not run in lockstep.
struct MvnPrec {
    static constexpr bool kNeedsLdsVector = true;
    const double* P;
    double* xs;          // LDS [64*W*DPL]: the position, visible to all lanes
    template <int W>
    NM_DEV void init(const double* params, int, Reducer<W>&) { P = params; }
    NM_DEV void set_lds(double* lds) { xs = lds; }
    template <int DPL, int W>
    NM_DEV double eval(const Tile<DPL>& x, Tile<DPL>& gx, int dim, Reducer<W>& R) const {
        block_sync(W == 1);                                // the previous evaluation's readers are done
#pragma unroll
        for (int k = 0; k < DPL; ++k) {
            const int d = elem_index<W>(k);
            xs[d] = d < dim ? x.a[k] : 0.0;
        }
        block_sync(W == 1);
        double y[DPL];
#pragma unroll
        for (int k = 0; k < DPL; ++k) y[k] = 0.0;
        for (int j = 0; j < dim; ++j) {
            const double xj = xs[j];
            const double* row = P + (size_t)j * (size_t)dim;
#pragma unroll
            for (int k = 0; k < DPL; ++k) {
                const int d = elem_index<W>(k);
                const double p = d < dim ? row[d] : 0.0;
                y[k] = __builtin_fma(p, xj, y[k]);
            }
        }
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < DPL; ++k) {
            const bool valid = elem_index<W>(k) < dim;
            gx.a[k] = valid ? -y[k] : 0.0;
            acc = acc + (valid ? x.a[k] * y[k] : 0.0);
        }
        return -0.5 * R.sum(acc);
    }
};

struct EightSchools {
    static constexpr bool kNeedsLdsVector = false;
    NM_DEV void set_lds(double*) {}
    const double* par;
    template <int W>
    NM_DEV void init(const double* params, int, Reducer<W>&) { par = params; }
    template <int DPL, int W>
    NM_DEV double eval(const Tile<DPL>& x, Tile<DPL>& gx, int dim, Reducer<W>& R) const {
        static_assert(W == 1, "dim 10 fits one wave");
        const double mu = readlane_f64(x.a[0], 0), lt = readlane_f64(x.a[1], 0);
        const double tau = uexp(lt);
        const double t5 = (tau / 5.0) * (tau / 5.0);
        const double prior_tau = lt - ulog1p(t5);
        // per-lane school terms: lane l (1..4) holds theta~_{2l-2}, theta~_{2l-1}
        double term[2], dr[2], drth[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int d = elem_index<W>(j);
            bool school = d >= 2 && d < 10;
            int i = school ? d - 2 : 0;
            double th = x.a[j];
            double sg = par[8 + i];
            double r = (par[i] - (mu + tau * th)) / sg;
            term[j] = -0.5 * th * th - 0.5 * r * r;
            dr[j] = r / sg;
            drth[j] = dr[j] * th;
            if (!school) { term[j] = 0.0; dr[j] = 0.0; drth[j] = 0.0; }
        }
        // gmu = sum_i dr_i and gtau_lin = sum_i dr_i theta_i, sequentially in i (the oracle's loop order)
        double gmu = 0.0, gtl = 0.0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int l = (2 + i) >> 1, j = (2 + i) & 1;
            gmu = gmu + readlane_f64(j ? dr[1] : dr[0], l);
            gtl = gtl + readlane_f64(j ? drth[1] : drth[0], l);
        }
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < DPL; ++k) {
            int d = elem_index<W>(k);
            double t = 0.0, g = 0.0;
            if (k < 2) {
                if (d == 0) { t = -mu * mu / 50.0; g = -mu / 25.0 + gmu; }
                else if (d == 1) { t = prior_tau; g = 1.0 - 2.0 * t5 / (1.0 + t5) + gtl * tau; }
                else if (d < 10) { t = term[k]; g = -x.a[k] + dr[k] * tau; }
            }
            gx.a[k] = g;
            acc = acc + t;
        }
        (void)dim;
        return R.sum(acc);
    }
};

// The slow path of the boundary: a density evaluated by a HOST function pointer with the reference's own shape,
// `CpuLogpFunc::logp(&mut self, position: &[f64], gradient: &mut [f64]) -> Result<f64, E>` (src/math/cpu_math.rs:885-891),
// errors classified by `LogpError::is_recoverable` (src/math/math.rs:9-13).  A kernel cannot call the host, so every
// evaluation is a round trip through the chain's mailbox in pinned, fine-grained host memory: the block writes x,
// publishes a sequence number (system-scope release), and polls for the answer that the engine's service threads write
// after calling the user's function (nuts_engine.hip).  Latency ~ PCIe round trip + the callback; it exists so that any
// CpuLogpFunc drops in, not for throughput.
struct CbMail { uint64_t req, resp; int64_t status; double logp; };
struct HostCb {
    static constexpr bool kNeedsLdsVector = false;
    static constexpr bool kCanFail = true;
    NM_DEV void set_lds(double*) {}
    CbMail* mail = nullptr;
    double *mx = nullptr, *mg = nullptr;
    int status = 0;              // of the last evaluation: 0 ok, 1 recoverable error, 2 unrecoverable (or the host never answered)
    template <int W>
    NM_DEV void init(const double*, int, Reducer<W>&) {}
    NM_DEV void bind(const KParams& P, uint64_t chain) {
        unsigned char* base = P.cb_mail + (size_t)chain * P.cb_stride;
        mail = reinterpret_cast<CbMail*>(base);
        mx = reinterpret_cast<double*>(base + sizeof(CbMail));
        mg = mx + P.dim;
#if NM_CLUSTER_MODE
        next_seq = __hip_atomic_load(&mail->req, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) + 1;    // no request is in flight here
#endif
    }
#if NM_CLUSTER_MODE
    // a chain wider than one block: every member writes its slice of the position into the chain's mailbox, the first member
    // rings, all members wait for the answer and read their slice of the gradient
    int goff = 0, member = 0;
    uint64_t next_seq = 0;
    template <int W>
    NM_DEV void init_slice(const double*, int, int, int goff_, Reducer<W>& R) { goff = goff_; member = R.cl.member; }
    template <int DPL, int W>
    NM_DEV double eval(const Tile<DPL>& x, Tile<DPL>& gx, int dim, Reducer<W>& R) {
#pragma unroll
        for (int k = 0; k < DPL; ++k) {
            const int d = elem_index<W>(k);
            if (d < dim) __hip_atomic_store(&mx[goff + d], x.a[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");            // system scope: this member's slice is in host memory ...
        (void)R.sum(0.0);                                         // ... and so is every other member's, once all have met here
        int failed = 0;
        const uint64_t seq = next_seq;
        if (tid() == 0) {
            if (member == 0) __hip_atomic_store(&mail->req, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            const unsigned long long t0 = wall_clock64();           // 100 MHz
            while (__hip_atomic_load(&mail->resp, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
                __builtin_amdgcn_s_sleep(64);
                if (wall_clock64() - t0 > 6000000000ull) { failed = 1; break; }      // 60 s without an answer: the chain fails
            }
        }
        next_seq = seq + 1;
        failed = R.sum(failed ? 1.0 : 0.0) != 0.0 ? 1 : 0;         // no member rings again before all have read this answer
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
        const int64_t st = failed ? 2 : __hip_atomic_load(&mail->status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        status = __builtin_amdgcn_readfirstlane((int)st);
        const double lp = __hip_atomic_load(&mail->logp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#pragma unroll
        for (int k = 0; k < DPL; ++k) {
            const int d = elem_index<W>(k);
            gx.a[k] = (d < dim && status == 0) ? __hip_atomic_load(&mg[goff + d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0.0;
        }
        return uniform_f64(status == 0 ? lp : __builtin_nan(""));
    }
#else
    template <int DPL, int W>
    NM_DEV double eval(const Tile<DPL>& x, Tile<DPL>& gx, int dim, Reducer<W>& R) {
#pragma unroll
        for (int k = 0; k < DPL; ++k) {
            const int d = elem_index<W>(k);
            if (d < dim) __hip_atomic_store(&mx[d], x.a[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");            // system scope: x is in host memory before the request is
        block_sync(W == 1);
        int failed = 0;
        if (tid() == 0) {
            const uint64_t seq = __hip_atomic_load(&mail->req, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) + 1;
            __hip_atomic_store(&mail->req, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            const unsigned long long t0 = wall_clock64();           // 100 MHz
            while (__hip_atomic_load(&mail->resp, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
                __builtin_amdgcn_s_sleep(64);
                if (wall_clock64() - t0 > 6000000000ull) { failed = 1; break; }      // 60 s without an answer: the chain fails
            }
        }
        failed = R.sum(failed ? 1.0 : 0.0) != 0.0 ? 1 : 0;         // also the block-wide synchronisation point
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
        const int64_t st = failed ? 2 : __hip_atomic_load(&mail->status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        status = __builtin_amdgcn_readfirstlane((int)st);
        const double lp = __hip_atomic_load(&mail->logp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#pragma unroll
        for (int k = 0; k < DPL; ++k) {
            const int d = elem_index<W>(k);
            gx.a[k] = (d < dim && status == 0) ? __hip_atomic_load(&mg[d], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0.0;
        }
        return uniform_f64(status == 0 ? lp : __builtin_nan(""));
    }
#endif
};
template <class D, class = void> struct elementwise_trait { static constexpr bool value = false; };
template <class D> struct elementwise_trait<D, typename std::enable_if<D::kElementwise>::type> { static constexpr bool value = true; };

template <class D, class = void> struct can_fail { static constexpr bool value = false; };
template <class D> struct can_fail<D, typename std::enable_if<D::kCanFail>::type> { static constexpr bool value = true; };
// status of the density's last evaluation (0 for densities that cannot fail)
template <class Ctx_> NM_DEV int dens_status(Ctx_& C) {
    if constexpr (can_fail<decltype(C.dens)>::value) return C.dens.status; else return 0;
}

// A density wrapped in LrWrap selects the kernels that carry the low-rank transformation (LowRankMassMatrix, reference
// src/transform/low_rank.rs) and its adaptation protocol; plain densities compile the diagonal-only code they always had.
template <class D>
struct LrWrap : D {};
template <class D> struct lr_trait { static constexpr bool value = false; };
template <class D> struct lr_trait<LrWrap<D>> { static constexpr bool value = true; };
// KinWrap<D>: the same density with the non-Euclidean KineticEnergyKinds compiled into the leapfrog (nm_settings.trajectory_kind,
// reference src/dynamics/transformed_hamiltonian.rs:27-50); the plain kernels carry only the Euclidean integrator.
template <class D>
struct KinWrap : D {};
template <class D> struct kin_trait { static constexpr bool value = false; };
template <class D> struct kin_trait<KinWrap<D>> { static constexpr bool value = true; };
// the kernels with the low-rank transformation carry the kinds too (LowRankNutsSettings::trajectory_kind): they are bound by
// their matrix traffic, not by code size
template <class D> struct kin_trait<LrWrap<D>> { static constexpr bool value = true; };
// densities of the tile kernel (nuts_tile.hpp) set kTile: products with the SHARED matrices (U', U, P) are rendezvous
// GEMMs of the block's 16 chains on the matrix cores
template <class D, class = void> struct tile_trait { static constexpr bool value = false; };
template <class D> struct tile_trait<D, typename std::enable_if<D::kTile>::type> { static constexpr bool value = true; };
template <int DPL, int W, class Dens> constexpr bool nog_mode() {
    return bool(NM_NOG) && ((DPL == 16 && W == 1) || (bool(NM_NOG_82) && DPL == 8 && W == 2)) && bool(NM_TRIM_FIRST) && !bool(NM_TILE_MODE) && !bool(NM_CLUSTER_MODE) && bool(NM_FUSED_LEAPFROG) && !batched_merges<DPL, W>() &&
           elementwise_trait<Dens>::value && !lr_trait<Dens>::value && !kin_trait<Dens>::value && !tile_trait<Dens>::value;
}

// ---------------------------------------------------------------------------------------------
// Per-wave context: everything a chain keeps in registers / SGPRs while its kernel runs
// ---------------------------------------------------------------------------------------------
struct PendEntry {        // a completed sub-tree of `other` waiting for its sibling (one per level); lives in LDS
    double log_size;
    double cand_logp, cand_ke;
    int64_t cand_idx;
    int cand_slot;        // >= 0: pool slot C[slot]
    int pad;
};

template <int DPL, int W, class Dens>
struct alignas(16) BlockShared {      // LDS of one block (one block = W waves = one resident chain); the vectors are read and written 16 bytes at a time
    alignas(16) uint32_t rng_cache[RNG_CACHE_WORDS];
    alignas(16) double sig[NM_TILE_MODE ? 2 : 64 * W * DPL];     // DiagMassMatrix stds of the resident chain, tile order
    alignas(16) double mu[NM_TILE_MODE ? 2 : 64 * W * DPL];      // DiagMassMatrix mean   (tile mode: one shared copy per block, nuts_tile.hpp)
    alignas(16) double red[2 * RED_MAX_VALUES * W + (NM_CLUSTER_MODE ? RED_MAX_VALUES + 1 + 2 * RED_MAX_VALUES * CL_MAX_MEMBERS : 0)];
    alignas(16) double l1_z[NM_LDS_L1 ? 64 * W * DPL : 2];    // L[1]: (z, v) of the last leaf of the pending level-1 sub-tree — the hottest
    alignas(16) double l1_v[NM_LDS_L1 ? 64 * W * DPL + 72 : 2];   // end point (written every 4th leaf, read two leaves later) never leaves the CU
    // Between trees both arrays are free: the momentum refresh uses l1_v as its ChaCha word buffer (hence the 72
    // extra doubles: 64 spare cells + one block of alignment) and l1_z as the stream-ordered sample vector.
    // wave-uniform state is kept once PER WAVE: every wave computes the same values, so private copies need no
    // synchronisation (a shared copy would be a read-modify-write race between the waves)
    PendEntry pend[W][MAX_MAXDEPTH + 1];
    alignas(16) double dens_lds[Dens::kNeedsLdsVector ? 64 * W * DPL : 2];   // a density's block-visible vector (MvnPrec: the position)
    ChainScalars sc[W];       // the resident chain's scalars (copied in at ctx_begin; wave 0's copy goes back at ctx_end)
};

template <int DPL, int W, class Dens>
struct ChainCtx {
    const KParams& P;
    Dens dens;
    Reducer<W> red;
    DevRng rng;
    ZigTables zig;
    double* pv;         // this chain's persistent slots
    double* sv;         // this wave's tree scratch
    double* l1z;        // LDS: L[1] end point
    double* l1v;
    double* lsig;       // LDS [64*DPL]: sigma of the resident chain (tile order: lane l reads its own elements)
    double* lmu;        // LDS [64*DPL]: mu
    PendEntry* pend;    // LDS
    int dim;            // elements this block holds (the chain's dim; in cluster mode this member's slice)
    int gdim;           // the chain's dim
    int goff;           // cluster mode: index of this member's first element in the chain's vectors, else 0
    int maxdepth_cfg;
    ChainScalars& sc;   // LDS
    unsigned long long prof_t;

    __device__ ChainCtx(const KParams& p, ChainScalars& lds_sc) : P(p), sc(lds_sc) {}
    rsrc_t rp, rs;      // buffer descriptors of this chain's persistent slots / this block's tree scratch
    rsrc_t rl;          // LrWrap: this chain's low-rank vectors (mu_lr, eigenvectors)
    const double* lvals;   // LrWrap: this chain's [2][lr_rmax] eigenvalue arrays
    int voff;           // tid * 16: byte offset of this thread's first pair inside a chain vector
    int slot_bytes;     // DP * 8
    NM_DEV double* slot(int s) const { return pv + (size_t)s * P.dpad; }
    NM_DEV double* sslot(int s) const { return sv + (size_t)s * P.dpad; }
    // tile <-> HBM slot of the persistent (P) or scratch (S) region
    NM_DEV void loadR(Tile<DPL>& t, rsrc_t r, int s) const {
        const int so = force_sgpr(s * slot_bytes);
#pragma unroll
        for (int m = 0; m < DPL / 2; ++m) {
            double2 q = buf_load2(r, voff + m * (64 * W * 16), so);
            t.a[2 * m] = q.x; t.a[2 * m + 1] = q.y;
        }
    }
    NM_DEV void storeR(const Tile<DPL>& t, rsrc_t r, int s) const {
        const int so = force_sgpr(s * slot_bytes);
#pragma unroll
        for (int m = 0; m < DPL / 2; ++m) buf_store2(r, voff + m * (64 * W * 16), so, t.a[2 * m], t.a[2 * m + 1]);
    }
    NM_DEV void loadP(Tile<DPL>& t, int s) const { loadR(t, rp, s); }
    NM_DEV void storeP(const Tile<DPL>& t, int s) const { storeR(t, rp, s); }
    // streaming variants: data that is read back late or never (candidates, per-draw state, edge gradients) should
    // not push the soon-to-be-re-read end points (F, L) out of the 4 MiB L2 that 128 resident chains share
    NM_DEV void storeR_nt(const Tile<DPL>& t, rsrc_t r, int so) const {
#pragma unroll
        for (int m = 0; m < DPL / 2; ++m) buf_store2_aux<aux_nt<DPL, W>()>(r, voff + m * (64 * W * 16), so, t.a[2 * m], t.a[2 * m + 1]);
    }
    NM_DEV void storeP_nt(const Tile<DPL>& t, int s) const { storeR_nt(t, rp, force_sgpr(s * slot_bytes)); }
    NM_DEV void storeS_nt(const Tile<DPL>& t, int s) const { storeR_nt(t, rs, force_sgpr(s * slot_bytes)); }
    NM_DEV void loadS(Tile<DPL>& t, int s) const { loadR(t, rs, s); }
    NM_DEV void storeS(const Tile<DPL>& t, int s) const { storeR(t, rs, s); }
    // one pair of a slot (U-turn operand streams); so = slot byte offset (wave-uniform)
    NM_DEV int soS(int s) const { return force_sgpr(s * slot_bytes); }
    NM_DEV double2 ld2(rsrc_t r, int so, int m) const { return buf_load2(r, voff + m * (64 * W * 16), so); }
    // main-tree edges.  Edge id 0 is the trajectory's initial point and costs no store: its z / g_z are the chain's
    // P_Z / P_GZ slots and its v is the staged normals buffer (stream order == memory order of a chain vector);
    // ids 1, 2 are scratch slots.
    struct SlotRef { rsrc_t r; int so; };
    // (the edge id is wave-uniform; readfirstlane makes the descriptor / offset selection provably so — T20)
    NM_DEV SlotRef edge_z(int id) const { id = force_sgpr(id); return id == 0 ? SlotRef{rp, soS(P_Z)} : SlotRef{rs, soS(EDGE0_Z + 3 * id)}; }
    NM_DEV SlotRef edge_v(int id) const { id = force_sgpr(id); return SlotRef{rs, soS(id == 0 ? (int)STAGE_V : EDGE0_V + 3 * id)}; }
    NM_DEV SlotRef edge_g(int id) const { id = force_sgpr(id); return id == 0 ? SlotRef{rp, soS(P_GZ)} : SlotRef{rs, soS(EDGE0_G + 3 * id)}; }
    NM_DEV void loadRef(Tile<DPL>& t, SlotRef f) const {
        const int so = f.so;
#pragma unroll
        for (int m = 0; m < DPL / 2; ++m) {
            double2 q = buf_load2(f.r, voff + m * (64 * W * 16), so);
            t.a[2 * m] = q.x; t.a[2 * m + 1] = q.y;
        }
    }
    NM_DEV void storeRef_nt(const Tile<DPL>& t, SlotRef f) const { storeR_nt(t, f.r, f.so); }
    NM_DEV void storeRef(const Tile<DPL>& t, SlotRef f) const {
        const int so = f.so;
#pragma unroll
        for (int m = 0; m < DPL / 2; ++m) buf_store2(f.r, voff + m * (64 * W * 16), so, t.a[2 * m], t.a[2 * m + 1]);
    }
    NM_DEV void load(Tile<DPL>& t, const double* base) const { load_tile<DPL, W>(t, base); }
    NM_DEV void store(const Tile<DPL>& t, double* base) const { store_tile<DPL, W>(t, base); }
    NM_DEV int elem(int k) const { return elem_index<W>(k); }
    NM_DEV const double2* tptr(const double* base) const { return reinterpret_cast<const double2*>(base) + tid(); }
};

// the thread that writes a draw's nm_draw_stats row (cluster mode: of the chain's first member)
#if NM_CLUSTER_MODE
#define NM_STAT_WRITER(C) (tid() == 0 && (C).red.cl.member == 0)
#else
#define NM_STAT_WRITER(C) (tid() == 0)
#endif
template <int DPL, int W, class Dens>
NM_DEV void ctx_begin(ChainCtx<DPL, W, Dens>& C, BlockShared<DPL, W, Dens>& sh, uint64_t chain, uint64_t wave) {
#ifndef NM_PACKED_MAX_DPL
#define NM_PACKED_MAX_DPL 4
#endif
#ifndef NM_PACKED_TESTS_16
#define NM_PACKED_TESTS_16 1     // the six sums of a level-k / top-level test through one transposed butterfly on the 16-doubles tiling too (2.48 -> 2.555e11: profiles/r06n_*)
#endif
    C.red.packed_tests = bool(NM_PACKED_TESTS_16) && DPL == 16 && W == 1 && !NM_TILE_MODE && !NM_CLUSTER_MODE;
    C.red.packed = DPL <= NM_PACKED_MAX_DPL && !NM_TILE_MODE;        // (dev_math.hpp Reducer::packed: per tiling, from the measurements of round 5)
    const KParams& P = C.P;
    C.dim = (int)P.dim; C.gdim = (int)P.dim; C.goff = 0;
#if NM_CLUSTER_MODE
    // `chain` arrives as the sub-chain index chain * cl_k + member (its own persistent vectors and copy of the scalars);
    // C.red.cl.{box, cnt, k, member, epoch} were set by the kernel
    C.goff = C.red.cl.member * (int)P.cl_slice;
    C.dim = (int)(P.dim - (uint64_t)C.goff < P.cl_slice ? P.dim - (uint64_t)C.goff : P.cl_slice);
#endif
    C.maxdepth_cfg = (int)P.layout_md;       // slots are laid out for the deepest tree a draw can grow: maxdepth + extra_doublings
    C.pv = P.pvec + (size_t)chain * NUM_PSLOT * P.dpad;
    C.sv = P.svec + (size_t)wave * P.nsslot * P.dpad;
    C.slot_bytes = (int)(P.dpad * 8);
    C.voff = tid() * 16;
    C.rp = make_rsrc(C.pv, (uint64_t)NUM_PSLOT * P.dpad * 8);
    C.rs = make_rsrc(C.sv, (uint64_t)P.nsslot * P.dpad * 8);
    if constexpr (lr_trait<Dens>::value) {
        C.rl = make_rsrc(P.lrvec + (size_t)chain * (1 + P.lr_rmax) * P.dpad, (uint64_t)(1 + P.lr_rmax) * P.dpad * 8);
        C.lvals = P.lrval + (size_t)chain * 2 * P.lr_rmax;
    }
    C.lsig = sh.sig;
    C.l1z = NM_LDS_L1 ? sh.l1_z : C.sslot(slot_L(C.maxdepth_cfg, 1));
    C.l1v = NM_LDS_L1 ? sh.l1_v : C.sslot(slot_L(C.maxdepth_cfg, 1) + 1);
    C.lmu = sh.mu;
    C.pend = sh.pend[W == 1 ? 0 : wave_id()];
    C.zig = {P.zig_x, P.zig_f};
    {   // chain scalars: HBM -> LDS
        const uint64_t* src = reinterpret_cast<const uint64_t*>(&P.sc[chain]);
        uint64_t* dst = reinterpret_cast<uint64_t*>(&sh.sc[W == 1 ? 0 : wave_id()]);
        constexpr int NW = (int)(sizeof(ChainScalars) / 8);
        static_assert(sizeof(ChainScalars) % 8 == 0 && NW <= 64, "ChainScalars must be <= 64 u64 words");
        chain_sync();
        if (lane_id() < NW) dst[lane_id()] = src[lane_id()];
        chain_sync();
    }
    C.red.init(sh.red);
#if NM_CLUSTER_MODE
#endif
    C.rng.init(C.sc.key, C.sc.rng_pos, sh.rng_cache);
#if NM_CLUSTER_MODE
    C.dens.init_slice(P.logp_params, C.dim, C.gdim, C.goff, C.red);     // element-wise densities only (IidNormal, DiagNormal)
#else
    C.dens.init(P.logp_params, C.dim, C.red);
#endif
    C.dens.set_lds(sh.dens_lds);
#if NM_CLUSTER_MODE
    if constexpr (can_fail<Dens>::value) C.dens.bind(P, chain / P.cl_k);      // `chain` is the sub-chain index here
#else
    if constexpr (can_fail<Dens>::value) C.dens.bind(P, chain);
#endif
}
template <int DPL, int W, class Dens>
NM_DEV void ctx_end(ChainCtx<DPL, W, Dens>& C, uint64_t chain) {
    C.sc.rng_pos = C.rng.pos;
    chain_sync();
    {
        const uint64_t* src = reinterpret_cast<const uint64_t*>(&C.sc);
        uint64_t* dst = reinterpret_cast<uint64_t*>(&C.P.sc[chain]);
        constexpr int NW = (int)(sizeof(ChainScalars) / 8);
        if (tid() < NW) dst[tid()] = src[tid()];   // threads 0..NW-1 are in wave 0: its copy
    }
}

// a phase-space point in registers
template <int DPL>
struct Pt {
    Tile<DPL> z, v, g;   // g = transformed gradient g_z
    double logp, ke;
    int64_t idx;
};

// apply_lowrank_transform_inplace (reference src/math/cpu_math.rs:383-425): v += U ((vals - 1) . (U' v)).
// One pass over the eigenvectors: row k (a chain vector in tile layout, 16 B / lane loads) is dotted with the ORIGINAL v
// (per-lane serial fma in register order, then the block reduction: the oracle's vector_dot order), scaled, and
// accumulated into the result with one fma per element while it is still in registers — U is read once per call.
// `which` = 0: lambda^1/2, 1: lambda^-1/2.
template <int DPL, int W, class Dens>
NM_DEV void lr_apply(ChainCtx<DPL, W, Dens>& C, int which, Tile<DPL>& v) {
    if constexpr (tile_trait<Dens>::value) { C.dens.tile_apply(which, v); return; }
    const int r = (int)C.sc.lr_rank;
    if (r == 0) return;
    constexpr int NB = DPL >= 8 ? 2 : 4;          // eigenvectors in flight (register budget)
    const double* vals = C.lvals + (size_t)which * C.P.lr_rmax;
    Tile<DPL> out = v;
    for (int k0 = 0; k0 < r; k0 += NB) {
        double u[NB][DPL];
        double p[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            p[j] = 0.0;
            if (k0 + j < r) {
                const int so = force_sgpr((1 + k0 + j) * C.slot_bytes);
#pragma unroll
                for (int m = 0; m < DPL / 2; ++m) {
                    const double2 q = buf_load2(C.rl, C.voff + m * (64 * W * 16), so);
                    u[j][2 * m] = q.x; u[j][2 * m + 1] = q.y;
                }
#pragma unroll
                for (int k = 0; k < DPL; ++k) p[j] = __builtin_fma(u[j][k], v.a[k], p[j]);
            } else {
#pragma unroll
                for (int k = 0; k < DPL; ++k) u[j][k] = 0.0;
            }
        }
        C.red.sum_n(p);
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            if (k0 + j < r) {
                const double sc_ = p[j] * (vals[k0 + j] - 1.0);
#pragma unroll
                for (int k = 0; k < DPL; ++k) out.a[k] = __builtin_fma(u[j][k], sc_, out.a[k]);
            }
        }
    }
    v = out;
}
template <int DPL, int W, class Dens>
NM_DEV void lr_load_mu(ChainCtx<DPL, W, Dens>& C, Tile<DPL>& mu_lr) {
#pragma unroll
    for (int m = 0; m < DPL / 2; ++m) {
        const double2 q = buf_load2(C.rl, C.voff + m * (64 * W * 16), 0);
        mu_lr.a[2 * m] = q.x; mu_lr.a[2 * m + 1] = q.y;
    }
}
// compute_untransformed_position: x = F(z)  (diagonal.rs:248-257; low_rank.rs:349-375)
template <int DPL, int W, class Dens>
NM_DEV void transform_to_x(ChainCtx<DPL, W, Dens>& C, const Tile<DPL>& z, Tile<DPL>& x) {
    Tile<DPL> t = z;
    bool inner = false;
    if constexpr (lr_trait<Dens>::value) inner = C.sc.lr_has_inner != 0;
    if (inner) {
        lr_apply(C, 0, t);
        Tile<DPL> ml;
        lr_load_mu(C, ml);
#pragma unroll
        for (int k = 0; k < DPL; ++k) t.a[k] = __builtin_fma(1.0, ml.a[k], t.a[k]);
    }
    const double2* sg2 = C.tptr(C.lsig);
    const double2* mu2 = C.tptr(C.lmu);
#pragma unroll
    for (int m = 0; m < DPL / 2; ++m) {
        const double2 sg = sg2[m * 64 * W], mm = mu2[m * 64 * W];
        x.a[2 * m] = __builtin_fma(1.0, mm.x, t.a[2 * m] * sg.x);
        x.a[2 * m + 1] = __builtin_fma(1.0, mm.y, t.a[2 * m + 1] * sg.y);
    }
}
// compute_transformed_gradient: g_z = J_F' g_x  (diagonal.rs:258-265; low_rank.rs:377-393)
template <int DPL, int W, class Dens>
NM_DEV void transform_to_gz(ChainCtx<DPL, W, Dens>& C, const Tile<DPL>& gx, Tile<DPL>& gz) {
    const double2* sg2 = C.tptr(C.lsig);
#pragma unroll
    for (int m = 0; m < DPL / 2; ++m) {
        const double2 sg = sg2[m * 64 * W];
        gz.a[2 * m] = gx.a[2 * m] * sg.x;
        gz.a[2 * m + 1] = gx.a[2 * m + 1] * sg.y;
    }
    if constexpr (lr_trait<Dens>::value) { if (C.sc.lr_has_inner) lr_apply(C, 0, gz); }
}
// compute_transformed_position: z = F^-1(x)  (diagonal.rs:233-246; low_rank.rs:325-347)
template <int DPL, int W, class Dens>
NM_DEV void transform_to_z(ChainCtx<DPL, W, Dens>& C, const Tile<DPL>& x, Tile<DPL>& z) {
    Tile<DPL> isig, mu;
    C.loadP(isig, P_ISIG);
    C.load(mu, C.lmu);
#pragma unroll
    for (int k = 0; k < DPL; ++k) {
        const double t = __builtin_fma(-1.0, mu.a[k], x.a[k]);
        z.a[k] = isig.a[k] * t;
    }
    if constexpr (lr_trait<Dens>::value) {
        if (C.sc.lr_has_inner) {
            Tile<DPL> ml;
            lr_load_mu(C, ml);
#pragma unroll
            for (int k = 0; k < DPL; ++k) z.a[k] = __builtin_fma(-1.0, ml.a[k], z.a[k]);
            lr_apply(C, 1, z);
        }
    }
}

// ---- the non-Euclidean KineticEnergyKinds (KinWrap kernels only) -----------------------------------------------------
// `v.iter().map(|x| x * x).sum()` in the engine's reduction order (oracle Ctx::sum_sq, REDUCE_GPU)
template <int DPL, int W>
NM_DEV double sum_sq_tile(const Tile<DPL>& v, Reducer<W>& R) {
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < DPL; ++k) acc = acc + v.a[k] * v.a[k];
    return R.sum(acc);
}
// array_normalize (reference src/math/cpu_math.rs:496-503)
template <int DPL, int W>
NM_DEV void normalize_tile(Tile<DPL>& v, Reducer<W>& R) {
    const double inv = 1.0 / __builtin_sqrt(sum_sq_tile(v, R));
#pragma unroll
    for (int k = 0; k < DPL; ++k) v.a[k] *= inv;
}
// esh_momentum_update (reference src/math/cpu_math.rs:505-551): the ESH momentum step on the unit sphere; returns the
// change of the kinetic energy.  Padding elements hold 0 and stay 0.
template <int DPL, int W>
NM_DEV double esh_update_core(const Tile<DPL>& g, Tile<DPL>& p, double step_size, int dim, Reducer<W>& R) {
    const double grad_norm = __builtin_sqrt(sum_sq_tile(g, R));
    const double inv_grad_norm = 1.0 / grad_norm;
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < DPL; ++k) acc = acc + p.a[k] * g.a[k] * inv_grad_norm;
    const double momentum_proj = R.sum(acc);
    const double dims_m1 = (double)(dim - 1);
    const double delta = step_size * grad_norm / dims_m1;
    const double zeta = uexp(-delta);
    const double coeff_g = (1.0 - zeta) * (1.0 + zeta + momentum_proj * (1.0 - zeta));
    const double coeff_p = 2.0 * zeta;
#pragma unroll
    for (int k = 0; k < DPL; ++k) p.a[k] = coeff_g * (g.a[k] * inv_grad_norm) + coeff_p * p.a[k];
    normalize_tile(p, R);
    const double arg = momentum_proj + (1.0 - momentum_proj) * zeta * zeta;
    return (delta - 6.93147180559945286227e-01 + ulog1p(arg)) * dims_m1;
}
template <int DPL, int W, class Dens>
NM_DEV double esh_update(ChainCtx<DPL, W, Dens>& C, const Tile<DPL>& g, Tile<DPL>& p, double step_size) {
    return esh_update_core<DPL, W>(g, p, step_size, C.gdim, C.red);
}
// leapfrog's divergence criterion (transformed_hamiltonian.rs:583-590)
template <int DPL, int W, class Dens>
NM_DEV bool bad_energy(const ChainCtx<DPL, W, Dens>& C, double energy_error, double max_energy_error) {
    if constexpr (kin_trait<Dens>::value) {
        if (C.sc.kin == NM_TRAJ_MICROCANONICAL)
            return (__builtin_fabs(energy_error) >= max_energy_error) | !is_finite(energy_error);
    }
    return (energy_error > max_energy_error) | !is_finite(energy_error);
}
// The leapfrog of KineticEnergyKind::ExactNormal (std_norm_grad_flow / std_norm_flow, src/math/util.rs:507-741) and
// ::Microcanonical (two ESH momentum half-steps around the position step, :186-226, :235-258) with the diagonal
// transformation; `o.ke` of the microcanonical kind is the accumulated kinetic-energy change (s.ke + both half-steps).
template <int DPL, int W, class Dens>
NM_DEV void leapfrog_kin(ChainCtx<DPL, W, Dens>& C, const Pt<DPL>& s, Pt<DPL>& o, double epsilon, Tile<DPL>* x_out, Tile<DPL>* gx_out) {
    const bool micro = C.sc.kin == NM_TRAJ_MICROCANONICAL;
    const double half = epsilon / 2.;
    const double sqrt_n = __builtin_sqrt((double)C.gdim);          // the chain's dim (cluster mode: not the slice's)
    Tile<DPL> x, gx;
    if (micro) {
        o.v = s.v;
        o.ke = s.ke + esh_update(C, s.g, o.v, sqrt_n * epsilon / 2.);
        const double eps_n = epsilon * sqrt_n;
#pragma unroll
        for (int k = 0; k < DPL; ++k) o.z.a[k] = __builtin_fma(eps_n, o.v.a[k], s.z.a[k]);
    } else {
        const double2 sc2 = dsincos(epsilon);
        const double es = uniform_f64(sc2.x), ec = uniform_f64(sc2.y);
        const double nes = -es;
#pragma unroll
        for (int k = 0; k < DPL; ++k) {
            const double vh = __builtin_fma(half, s.z.a[k] + s.g.a[k], s.v.a[k]);
            o.z.a[k] = __builtin_fma(s.z.a[k], ec, vh * es);
            o.v.a[k] = __builtin_fma(s.z.a[k], nes, vh * ec);
        }
    }
    bool lr_inner = false;
    if constexpr (lr_trait<Dens>::value) lr_inner = C.sc.lr_has_inner != 0;
    if (lr_inner) {                               // F = the low-rank transformation (low_rank.rs:286-300), as in the Euclidean leapfrog
        if constexpr (lr_trait<Dens>::value) {
            transform_to_x(C, o.z, x);
            o.logp = C.dens.template eval<DPL, W>(x, gx, C.dim, C.red);
            transform_to_gz(C, gx, o.g);
        }
    } else {
    const double2* sg2 = C.tptr(C.lsig);
    const double2* mu2 = C.tptr(C.lmu);
#pragma unroll
    for (int m = 0; m < DPL / 2; ++m) {
        const double2 sg = sg2[m * 64 * W], mm = mu2[m * 64 * W];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int k = 2 * m + j;
            const double t = o.z.a[k] * (j ? sg.y : sg.x);
            x.a[k] = __builtin_fma(1.0, (j ? mm.y : mm.x), t);
        }
    }
    o.logp = C.dens.template eval<DPL, W>(x, gx, C.dim, C.red);
#pragma unroll
    for (int m = 0; m < DPL / 2; ++m) {
        const double2 sg = sg2[m * 64 * W];
        o.g.a[2 * m] = gx.a[2 * m] * sg.x;
        o.g.a[2 * m + 1] = gx.a[2 * m + 1] * sg.y;
    }
    }
    if (micro) {
        o.ke = o.ke + esh_update(C, o.g, o.v, sqrt_n * epsilon / 2.);
    } else {
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < DPL; ++k) {
            o.v.a[k] = __builtin_fma(half, o.z.a[k] + o.g.a[k], o.v.a[k]);
            acc = __builtin_fma(o.v.a[k], o.v.a[k], acc);
        }
        o.ke = 0.5 * C.red.sum(acc);
    }
    if (x_out) *x_out = x;
    if (gx_out) *gx_out = gx;
}

// One leapfrog, registers to registers (reference transformed_hamiltonian.rs:524-615 + diagonal.rs:196-209, :248-265):
//   v½ = fma(ε/2, g_z, v); z' = fma(ε, v½, z); x' = z'·σ + μ; (logp, g_x) = density(x'); g_z' = g_x·σ;
//   v' = fma(ε/2, g_z', v½); KE' = ½ Σ fma(v', v', ·)
#ifndef NM_X_ASM_SITES
#define NM_X_ASM_SITES 4          // call sites whose fused leapfrog uses the inline-asm form of its two fused multiply-adds: bit 0 MCLMC (x_out / gx_out), 1 the
                                  // step-size search, 2 the tree.  Tree only: it is where the instruction pays (K2 +5 %: profiles/r05u), and every failing
                                  // build of DESIGN §22's fourth incident had the asm form at the MCLMC site (never executed by the failing runs)
#endif
// GMODE (round 6, nog_mode): where the start point's g_z comes from — 0: its tile s.g; 1: recomputed from s.z (element-wise densities: the operations that
// produced it); 2: streamed from the slot `gsrc` (the trajectory's initial point: P_GZ).  With GMODE != 0 neither s.g nor o.g is touched.
template <int DPL, int W, class Dens, int SITE = 0, int GMODE = 0>
NM_DEV void leapfrog(ChainCtx<DPL, W, Dens>& C, const Pt<DPL>& s, Pt<DPL>& o, double epsilon, Tile<DPL>* x_out, Tile<DPL>* gx_out,
                     typename ChainCtx<DPL, W, Dens>::SlotRef gsrc = typename ChainCtx<DPL, W, Dens>::SlotRef{}, Tile<DPL>* gshared = nullptr) {
    // GMODE 3 (NM_GTILE): ONE gradient tile shared by the chain of leapfrogs — *gshared holds g_z of the point the previous leapfrog produced (= this one's
    // start point), is read element by element and overwritten with the end point's; modes 1 / 2 fill it when the chain of leapfrogs restarts elsewhere.
    static_assert(GMODE == 0 || (elementwise_trait<Dens>::value && !lr_trait<Dens>::value && !kin_trait<Dens>::value && NM_FUSED_LEAPFROG && !NM_CLUSTER_MODE && !NM_TILE_MODE),
                  "the gradient-free point needs the fused element-wise leapfrog");
#ifdef NM_X_NO_LEAPFROG           // timing experiment only: the tree without its integrator
    if (!x_out && !gx_out) { o.z = s.z; o.v = s.v; o.g = s.g; o.logp = s.logp + epsilon * 1e-6; o.ke = s.ke; return; }
#endif
    if constexpr (kin_trait<Dens>::value) {
        if (C.sc.kin != NM_TRAJ_EUCLIDEAN) { leapfrog_kin(C, s, o, epsilon, x_out, gx_out); return; }
    }
    const double half = epsilon / 2.;
    Tile<DPL> x, gx;
    if constexpr (lr_trait<Dens>::value) {
        if (C.sc.lr_has_inner) {              // the same steps with F = the low-rank transformation (low_rank.rs:286-300)
#pragma unroll
            for (int k = 0; k < DPL; ++k) {
                const double vh = __builtin_fma(half, s.g.a[k], s.v.a[k]);
                o.v.a[k] = vh;
                o.z.a[k] = __builtin_fma(epsilon, vh, s.z.a[k]);
            }
            transform_to_x(C, o.z, x);
            o.logp = C.dens.template eval<DPL, W>(x, gx, C.dim, C.red);
            transform_to_gz(C, gx, o.g);
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < DPL; ++k) {
                o.v.a[k] = __builtin_fma(half, o.g.a[k], o.v.a[k]);
                acc = __builtin_fma(o.v.a[k], o.v.a[k], acc);
            }
            o.ke = 0.5 * C.red.sum(acc);
            if (x_out) *x_out = x;
            if (gx_out) *gx_out = gx;
            return;
        }
    }
    const double2* sg2 = C.tptr(C.lsig);
    const double2* mu2 = C.tptr(C.lmu);
#if NM_FUSED_LEAPFROG
    if constexpr (elementwise_trait<Dens>::value && !NM_CLUSTER_MODE && !NM_TILE_MODE) {
        // Element-wise densities: everything the step does to element k — half kick, drift, x = sigma z + mu, the density's gradient
        // element and logp term, g_z = sigma g_x, second half kick, v^2 — depends on element k alone; only the two sums couple the
        // elements, and nothing reads them before the end.  ONE pass, element pair by element pair, instead of three loops over the
        // tile: the intermediate tiles (x, g_x, the half-kicked v) never exist, so the two points the tree keeps (96 + 96 registers at
        // 16 doubles per lane) stay in the 256 architectural registers instead of commuting to the accumulation registers
        // (v_accvgpr_read / write: ~190 of a leapfrog's ~390 instructions in round 4's code; a lone wavefront is bound by the NUMBER of
        // instructions it issues).  Same operations on the same operands in the same order per element; the sums accumulate in
        // element order as before and go through one reduction of two values (the same butterfly per value): same bits.
        double acc = 0.0, kacc = 0.0;
        auto pass = [&](auto full_tile) __attribute__((always_inline)) {
            constexpr bool FULL = decltype(full_tile)::value;
            double2 sg = sg2[0], mm = mu2[0];
            [[maybe_unused]] double2 gg = make_double2(0., 0.);
            if constexpr (GMODE == 2) gg = C.ld2(gsrc.r, gsrc.so, 0);
#pragma unroll
            for (int m = 0; m < DPL / 2; ++m) {
                const double2 sg_c = sg, mm_c = mm;
                [[maybe_unused]] const double2 gg_c = gg;
                if (m + 1 < DPL / 2) { sg = sg2[(m + 1) * 64 * W]; mm = mu2[(m + 1) * 64 * W]; }      // the next pair's sigma / mu are in flight during this pair
                if constexpr (GMODE == 2) { if (m + 1 < DPL / 2) gg = C.ld2(gsrc.r, gsrc.so, m + 1); }
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int k = 2 * m + j;
                    const double sgk = j ? sg_c.y : sg_c.x, muk = j ? mm_c.y : mm_c.x;
                    double gsk;                      // the start point's g_z element
                    if constexpr (GMODE == 0) gsk = s.g.a[k];
                    else if constexpr (GMODE == 3) gsk = gshared->a[k];
                    else if constexpr (GMODE == 2) gsk = j ? gg_c.y : gg_c.x;
                    else {                           // the operations that produced it when this point was a leapfrog's end point (below): same bits
                        const double ts = s.z.a[k] * sgk;
                        const double xs = __builtin_fma(1.0, muk, ts);
                        double unused_term;
                        const double gxs = C.dens.template elem<W, FULL>(xs, k, C.dim, unused_term);
                        gsk = gxs * sgk;
                    }
                    // (v_fma_f64 spelled out: the source point's v and z stay live, and the compiler's two-address form — v_mov_b64 + v_fmac_f64 —
                    // costs an instruction more per fma; the same IEEE fused multiply-add)
                    double vh, zk;
                    if constexpr (NM_LF_FMA_FORM == 0 && ((NM_X_ASM_SITES >> SITE) & 1)) {
                        asm("v_fma_f64 %0, %1, %2, %3" : "=v"(vh) : "v"(half), "v"(gsk), "v"(s.v.a[k]));
                        asm("v_fma_f64 %0, %1, %2, %3" : "=v"(zk) : "v"(epsilon), "v"(vh), "v"(s.z.a[k]));
                    } else {
                        vh = __builtin_fma(half, gsk, s.v.a[k]);
                        zk = __builtin_fma(epsilon, vh, s.z.a[k]);
                    }
                    o.z.a[k] = zk;
                    const double t = zk * sgk;
                    const double xk = __builtin_fma(1.0, muk, t);
                    double term;
                    const double gxk = C.dens.template elem<W, FULL>(xk, k, C.dim, term);
                    acc = acc + term;
                    const double gk = gxk * sgk;
                    if constexpr (GMODE == 0) o.g.a[k] = gk;
                    else if constexpr (NM_GTILE != 0) gshared->a[k] = gk;
                    const double vk = __builtin_fma(half, gk, vh);
                    o.v.a[k] = vk;
                    kacc = __builtin_fma(vk, vk, kacc);
                    if (x_out) x_out->a[k] = xk;
                    if (gx_out) gx_out->a[k] = gxk;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        if (C.dim == DPL * 64 * W) pass(std::true_type{}); else pass(std::false_type{});
        C.red.sum2(acc, kacc);
        o.logp = C.dens.finish(acc);
        o.ke = 0.5 * kacc;
        return;
    }
#endif
#pragma unroll
    for (int m = 0; m < DPL / 2; ++m) {
        const double2 sg = sg2[m * 64 * W], mm = mu2[m * 64 * W];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int k = 2 * m + j;
            double vh = __builtin_fma(half, s.g.a[k], s.v.a[k]);
            o.v.a[k] = vh;
            o.z.a[k] = __builtin_fma(epsilon, vh, s.z.a[k]);
            double t = o.z.a[k] * (j ? sg.y : sg.x);
            x.a[k] = __builtin_fma(1.0, (j ? mm.y : mm.x), t);
        }
    }
    o.logp = C.dens.template eval<DPL, W>(x, gx, C.dim, C.red);
    double acc = 0.0;
#pragma unroll
    for (int m = 0; m < DPL / 2; ++m) {
        const double2 sg = sg2[m * 64 * W];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int k = 2 * m + j;
            o.g.a[k] = gx.a[k] * (j ? sg.y : sg.x);
            o.v.a[k] = __builtin_fma(half, o.g.a[k], o.v.a[k]);
            acc = __builtin_fma(o.v.a[k], o.v.a[k], acc);
        }
    }
    o.ke = 0.5 * C.red.sum(acc);
    if (x_out) *x_out = x;
    if (gx_out) *gx_out = gx;
}

// AcceptanceRateCollector (reference src/stepsize/dual_avg.rs:112-166).
// The two running means need exp(min(d,0)) and exp(d) of every leaf's energy difference d.  Evaluating them when the
// leaf is produced would put two dependent exp chains on the (wave-uniform, latency-bound) critical path of every
// leaf; instead lane (i mod 64) keeps the i-th pending d in a VGPR and `flush` evaluates up to 64 of them at once
// (one exp per lane) and then adds them to the sums sequentially in leaf order — the same additions in the same
// order as the reference's `RunningMean::add`, hence the same bits.
struct AcceptCollector {
    double initial_energy, sum, sum_sym, max_energy_error;
    uint64_t count;
    double pend_d;     // per-lane: the pending d of leaf (count - npend + lane) ...
    int npend;         // ... for lane < npend
    NM_DEV void register_init(double e0) {
        initial_energy = e0; sum = 0.; sum_sym = 0.; count = 0; max_energy_error = 0.;
        pend_d = 0.; npend = 0;
    }
    NM_DEV void flush() {
        if (npend == 0) return;
        // per-lane evaluation (lanes >= npend compute on 0 and are ignored)
        const double d = lane_id() < npend ? pend_d : 0.0;
        const double e = dexp(fmin_rs(d, 0.));
        const double es = 2. * e / (1. + dexp(d));
        for (int i = 0; i < npend; ++i) {
            sum = sum + readlane_f64(e, i);
            sum_sym = sum_sym + readlane_f64(es, i);
        }
        npend = 0;
    }
    NM_DEV void register_divergent() {
        flush();
        sum = sum + 0.; sum_sym = sum_sym + 0.; count += 1; max_energy_error = -__builtin_inf();
    }
    NM_DEV void register_ok(double end_energy) {
        const double diff = initial_energy - end_energy;
        if (lane_id() == npend) pend_d = diff;
        npend += 1;
        count += 1;
        if (__builtin_fabs(diff) > __builtin_fabs(max_energy_error)) max_energy_error = diff;
        if (npend == 64) flush();
    }
    NM_DEV double mean() { flush(); return sum / (double)count; }
    NM_DEV double mean_sym() { flush(); return sum_sym / (double)count; }
};

// is_turning partial sums (reference transformed_hamiltonian.rs:617-638, scalar_prods3 util.rs:221-347):
// s = (z_end + 0) - z_start; t1 += s*v_start; t2 += s*v_end
// (The reference's `+ 0` — scalar_prods3's third operand, the zeros vector — turns a -0.0 of z_end into +0.0, which can change the
// SIGN of a zero difference and nothing else; a zero term leaves a non-zero sum alone and a sum of zeros compares false with zero
// whatever its sign.  The sums are only ever compared with zero, so the addition is not executed: one instruction in four.)
NM_DEV void turn_acc(double zs, double vs, double ze, double ve, double& t1, double& t2) {
    double s = ze - zs;
    t1 = __builtin_fma(s, vs, t1);
    t2 = __builtin_fma(s, ve, t2);
}

// momentum refresh (array_gaussian, reference src/math/cpu_math.rs:561-577).  The stream-ordered samples are produced
// in LDS (fill_standard_normals_bulk) and read back in tile order; STAGE_V (HBM) keeps a copy because the initial
// point's velocity is the v of main-tree edge id 0.
template <int DPL, int W, class Dens>
NM_DEV void sample_velocity(ChainCtx<DPL, W, Dens>& C, Tile<DPL>& v, bool stage = true) {
#if NM_LDS_L1
    // at most 17 passes (1088 cells) per chunk: wider tilings take several chunks, which keeps the refresh's register
    // footprint (4 values per pass and lane in flight) the same for every kernel.  A vector that fits one 64-lane
    // pass with room to spare is cheaper with the simple pass-at-a-time routine (K4: +10 %).
#if NM_CLUSTER_MODE
    // every member draws the chain's whole vector, slice after slice in stream order (the members' streams stay identical),
    // and keeps its own slice
    for (int j = 0; j < C.red.cl.k; ++j) {
        const int off = j * (int)C.P.cl_slice;
        const int cnt = C.gdim - off < (int)C.P.cl_slice ? C.gdim - off : (int)C.P.cl_slice;
        fill_standard_normals_bulk<(DPL * W + 1 < 17 ? DPL * W + 1 : 17)>(C.rng, reinterpret_cast<uint32_t*>(C.l1v), C.l1z, cnt, C.zig, 64 * W, C.P.prof, C.prof_t);
        if (j == C.red.cl.member) {
            const double2* s2 = C.tptr(C.l1z);
#pragma unroll
            for (int m = 0; m < DPL / 2; ++m) {
                const double2 q = s2[m * 64 * W];
                v.a[2 * m] = C.elem(2 * m) < C.dim ? 1.0 * q.x : 0.0;
                v.a[2 * m + 1] = C.elem(2 * m + 1) < C.dim ? 1.0 * q.y : 0.0;
            }
        }
        block_sync(false);
    }
    C.storeS(v, STAGE_V);
#else
    if (DPL == 2 && C.dim <= 48) {
        fill_standard_normals(C.rng, C.l1z, C.dim, C.zig);
    } else {
        fill_standard_normals_bulk<(DPL * W + 1 < 17 ? DPL * W + 1 : 17)>(C.rng, reinterpret_cast<uint32_t*>(C.l1v), C.l1z, C.dim, C.zig, 64 * W, C.P.prof, C.prof_t);
    }
    const double2* s2 = C.tptr(C.l1z);
#pragma unroll
    for (int m = 0; m < DPL / 2; ++m) {
        const double2 q = s2[m * 64 * W];
        v.a[2 * m] = C.elem(2 * m) < C.dim ? 1.0 * q.x : 0.0;
        v.a[2 * m + 1] = C.elem(2 * m + 1) < C.dim ? 1.0 * q.y : 0.0;
    }
    if (stage) C.storeS(v, STAGE_V);
#endif
#else
    fill_standard_normals(C.rng, C.sslot(STAGE_V), C.dim, C.zig);
    const int so = C.soS(STAGE_V);
#pragma unroll
    for (int m = 0; m < DPL / 2; ++m) {
        double2 q = C.ld2(C.rs, so, m);
        v.a[2 * m] = C.elem(2 * m) < C.dim ? 1.0 * q.x : 0.0;
        v.a[2 * m + 1] = C.elem(2 * m + 1) < C.dim ? 1.0 * q.y : 0.0;
    }
#endif
    block_sync(W == 1);
}

template <int DPL, int W>
NM_DEV double kinetic(const Tile<DPL>& v, Reducer<W>& R) {
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < DPL; ++k) acc = __builtin_fma(v.a[k], v.a[k], acc);
    return 0.5 * R.sum(acc);
}
// the kinetic energy a trajectory starts with (initialize_trajectory, transformed_hamiltonian.rs:697-727): the
// microcanonical kind first puts the fresh momentum on the unit sphere (and re-stages it: the initial point's v is an
// edge operand) and starts its accumulated kinetic-energy change at 0
template <int DPL, int W, class Dens>
NM_DEV double initial_kinetic(ChainCtx<DPL, W, Dens>& C, Tile<DPL>& v, bool stage = true) {
    if constexpr (kin_trait<Dens>::value) {
        if (C.sc.kin == NM_TRAJ_MICROCANONICAL) {
            normalize_tile(v, C.red);
            if (stage) C.storeS(v, STAGE_V);
            return 0.0;
        }
    }
    return kinetic(v, C.red);
}

// Σ ln(t) over valid elements (array_sum_ln, cpu_math.rs:300-304)
template <int DPL, int W, class Dens>
NM_DEV double sum_ln_tile(ChainCtx<DPL, W, Dens>& C, const Tile<DPL>& t) {
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < DPL; ++k) {
        bool valid = C.elem(k) < C.dim;
        acc = acc + (valid ? dlog_impl<false>(valid ? t.a[k] : 1.0) : 0.0);   // inlined: the DPL evaluations interleave
    }
    return C.red.sum(acc);
}


// DualAverage::new (dual_avg.rs:44-53) or Adam::new (adam.rs:56-64), as stepsize::Strategy::{new, init} pick them
NM_DEV void stepsize_adapt_reset(ChainScalars& sc, const nm_settings& s, double initial_step) {
    sc.log_step = ulog(initial_step);
    if (s.step_size_method == NM_STEP_ADAM) {
        sc.adam_m = 0.; sc.adam_v = 0.; sc.adam_t = 0;
        return;
    }
    sc.log_step_adapted = sc.log_step;
    sc.hbar = 0.;
    sc.mu = ulog(10. * initial_step);
    sc.da_count = 1;
}

// f64::powi = compiler-builtins' __powidf2: square-and-multiply from the low bit of the exponent
NM_DEV double powi_rs(double a, int32_t b) {
    uint32_t pw = b < 0 ? 0u - (uint32_t)b : (uint32_t)b;
    double mul = 1.0;
    for (;;) {
        if (pw & 1u) mul *= a;
        pw >>= 1;
        if (pw == 0) break;
        a *= a;
    }
    return b < 0 ? 1.0 / mul : mul;
}

// Hamiltonian::init_state at x with the current mass matrix (reference transformed_hamiltonian.rs:640-661,
// check_all :310-324).  Fills st.z, st.g, gx, st.logp; returns false for BadInitGrad.
template <int DPL, int W, class Dens>
NM_DEV bool init_state(ChainCtx<DPL, W, Dens>& C, const Tile<DPL>& x, Pt<DPL>& st, Tile<DPL>& gx) {
    st.logp = C.dens.template eval<DPL, W>(x, gx, C.dim, C.red);
    if (dens_status(C) != 0) return false;                       // NutsError::LogpFailure: the caller reads dens_status
    if constexpr (lr_trait<Dens>::value) {
        transform_to_z(C, x, st.z);
        transform_to_gz(C, gx, st.g);
        bool ok_lr = true;
#pragma unroll
        for (int k = 0; k < DPL; ++k) {
            const bool valid = C.elem(k) < C.dim;
            ok_lr = ok_lr && (!valid || (is_finite(st.z.a[k]) && is_finite(st.g.a[k]) && st.g.a[k] != 0.0 &&
                                         is_finite(gx.a[k]) && is_finite(x.a[k])));
        }
        return C.red.all(ok_lr);
    }
    Tile<DPL> isig, sig, mu;
    C.loadP(isig, P_ISIG);
    C.load(sig, C.lsig);
    C.load(mu, C.lmu);
    bool ok = true;
#pragma unroll
    for (int k = 0; k < DPL; ++k) {
        double t = __builtin_fma(-1.0, mu.a[k], x.a[k]);       // compute_transformed_position diagonal.rs:233-246
        st.z.a[k] = isig.a[k] * t;
        st.g.a[k] = gx.a[k] * sig.a[k];                       // compute_transformed_gradient :258-265
        bool valid = C.elem(k) < C.dim;
        ok = ok && (!valid || (is_finite(st.z.a[k]) && is_finite(st.g.a[k]) && st.g.a[k] != 0.0 &&
                               is_finite(gx.a[k]) && is_finite(x.a[k])));
    }
    return C.red.all(ok);
}

// stepsize::Strategy::init (reference src/stepsize/adapt.rs:91-199): step-size search at `x`.
template <int DPL, int W, class Dens>
NM_DEV uint64_t stepsize_init(ChainCtx<DPL, W, Dens>& C, const Tile<DPL>& x) {
    const nm_settings& s = C.P.s;
    if (s.step_size_method == NM_STEP_FIXED) { C.sc.step_size = s.fixed_step_size; return NM_CHAIN_OK; }
    Pt<DPL> st;
    {
        Tile<DPL> gx;
        if (!init_state(C, x, st, gx)) return dens_status(C) != 0 ? NM_CHAIN_LOGP_FATAL : NM_CHAIN_BAD_INIT;
    }
    const double logdet = C.sc.mm_logdet;
    sample_velocity(C, st.v);                    // initialize_trajectory(resample) :687-736
    const double ke0 = initial_kinetic(C, st.v);
    st.ke = ke0;
    const double e0 = ke0 - (st.logp + logdet);
    AcceptCollector col;
    C.sc.step_size = s.initial_step;
    int dir = 0;
    for (int it = 0; it < 101; ++it) {
        Pt<DPL> o;
        const int sign = it == 0 ? 1 : dir;
        col.register_init(e0);
        leapfrog<DPL, W, Dens, 1>(C, st, o, (double)sign * C.sc.step_size * 1.0, (Tile<DPL>*)nullptr, (Tile<DPL>*)nullptr);
        const double energy = o.ke - (o.logp + logdet);
        const double err = energy - e0;
        if (dens_status(C) != 0) {                              // `let LeapfrogResult::Ok(_) = .. else { .. return Ok(()) }` (adapt.rs:122-150)
            if (it > 0) C.sc.step_size = s.initial_step;
            return NM_CHAIN_OK;
        }
        if (bad_energy(C, err, 1000.0)) {                       // hard-coded 1000.0 (adapt.rs:118, :142)
            if (it > 0) C.sc.step_size = s.initial_step;
            return NM_CHAIN_OK;
        }
        col.register_ok(energy);
        const double accept = col.mean();
        if (it == 0) { dir = accept > s.target_accept ? 1 : -1; continue; }
        if (dir > 0) {
            if ((accept <= s.target_accept) | (C.sc.step_size > 1e5)) { stepsize_adapt_reset(C.sc, s, C.sc.step_size); return NM_CHAIN_OK; }
            C.sc.step_size *= 2.;
        } else {
            if ((accept >= s.target_accept) | (C.sc.step_size < 1e-10)) { stepsize_adapt_reset(C.sc, s, C.sc.step_size); return NM_CHAIN_OK; }
            C.sc.step_size /= 2.;
        }
    }
    C.sc.step_size = s.initial_step;
    return NM_CHAIN_OK;
}

// update_stepsize (reference src/stepsize/adapt.rs:235-267)
template <int DPL, int W, class Dens>
NM_DEV void update_stepsize(ChainCtx<DPL, W, Dens>& C, bool use_best_guess) {
    const nm_settings& s = C.P.s;
    double step = s.step_size_method == NM_STEP_FIXED ? s.fixed_step_size
                : s.step_size_method == NM_STEP_ADAM ? uexp(C.sc.log_step)          // Adam has no averaged iterate
                : (use_best_guess ? uexp(C.sc.log_step_adapted) : uexp(C.sc.log_step));
    if (s.has_jitter) {
        double v12 = u2d((C.rng.next_u64() >> 12) | 0x3ff0000000000000ull);
        double j = (v12 - 1.0) * C.P.jitter_scale + C.P.jitter_low;
        C.sc.step_size = step * j;
    } else {
        C.sc.step_size = step;
    }
}
// DualAverage::advance (reference src/stepsize/dual_avg.rs:55-64) / Adam::advance
template <int DPL, int W, class Dens>
NM_DEV void update_estimator(ChainCtx<DPL, W, Dens>& C, bool late) {
    const nm_settings& s = C.P.s;
    if (s.step_size_method == NM_STEP_FIXED) return;
    ChainScalars& sc = C.sc;
    const double accept_stat = late ? sc.last_sym_mean_tree_accept : sc.last_mean_tree_accept;
    if (s.step_size_method == NM_STEP_ADAM) {                    // Adam::advance (stepsize/adam.rs:70-98)
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
    const double mk = uexp(-s.da_k * ulog((double)sc.da_count));
    sc.log_step_adapted = mk * sc.log_step + (1. - mk) * sc.log_step_adapted;
    sc.da_count += 1;
}

// RunningVariance::add_sample (reference adapt/diagonal.rs:31-44, array_update_variance cpu_math.rs:605-631) on
// estimators held in registers
template <int DPL>
NM_DEV void running_variance_add_regs(Tile<DPL>& mean, Tile<DPL>& var, uint64_t new_count, const Tile<DPL>& value) {
    if (new_count == 1) { mean = value; return; }
    const double diff_scale = 1.0 / (double)new_count;
#pragma unroll
    for (int k = 0; k < DPL; ++k) {
        double diff = value.a[k] - mean.a[k];
        mean.a[k] = mean.a[k] + diff * diff_scale;
        var.a[k] = var.a[k] + diff * diff;
    }
}

// writes sigma / inv_sigma / mu (HBM + the LDS copy of the resident chain), logdet, id
template <int DPL, int W, class Dens>
NM_DEV void commit_mass_matrix(ChainCtx<DPL, W, Dens>& C, const Tile<DPL>& sig, const Tile<DPL>& isig, const Tile<DPL>& mu) {
    C.storeP(sig, P_SIG);
    C.storeP(isig, P_ISIG);
    C.storeP(mu, P_MU);
    C.store(sig, C.lsig);
    C.store(mu, C.lmu);
#if NM_TILE_MODE
    // the matrix-core kernel of DiagNutsSettings reads sigma / mu of the leapfrog straight from these slots (generic loads of a
    // line that is hot in the vector L1): make the new values the ones every later load sees
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
#endif
    C.sc.mm_logdet = sum_ln_tile(C, isig);
    C.sc.mm_id += 1;
}

// DiagMassMatrix::update_diag_grad (reference diagonal.rs:133-154, cpu_math.rs:710-738)
template <int DPL, int W, class Dens>
NM_DEV void mass_matrix_from_grad(ChainCtx<DPL, W, Dens>& C, const Tile<DPL>& x, const Tile<DPL>& gx) {
    Tile<DPL> sig, isig, mu;
#pragma unroll
    for (int k = 0; k < DPL; ++k) {
        bool valid = C.elem(k) < C.dim;
        double val = 1.0 / clampd(__builtin_fabs(gx.a[k]), 1e-20, 1e20);
        if (!is_finite(val)) val = 1.0;
        double sd = __builtin_sqrt(val), isd = __builtin_sqrt(1.0 / val);
        double var = sd * sd;
        double mean = var * gx.a[k];
        mean = __builtin_fma(1.0, x.a[k], mean);
        sig.a[k] = valid ? sd : 0.0;
        isig.a[k] = valid ? isd : 0.0;
        mu.a[k] = valid ? mean : 0.0;
    }
    commit_mass_matrix(C, sig, isig, mu);
    if constexpr (lr_trait<Dens>::value) { C.sc.lr_has_inner = 0; C.sc.lr_rank = 0; }   // update_from_grad: inner = None (low_rank.rs:147)
}

// Strategy::adapt -> update_diag_draw_grad / update_diag_draw (reference adapt/diagonal.rs:161-196,
// diagonal.rs:85-131, cpu_math.rs:633-708).  Returns did_change.
template <int DPL, int W, class Dens>
NM_DEV bool mass_matrix_adapt(ChainCtx<DPL, W, Dens>& C, const Tile<DPL>& dm, const Tile<DPL>& dv, const Tile<DPL>& gm,
                              const Tile<DPL>& gv) {     // the foreground estimators (draw mean / var, grad mean / var)
    if (C.sc.cnt_fg < 3) return false;
    Tile<DPL> sig, isig, mu;
    C.load(sig, C.lsig);
    C.loadP(isig, P_ISIG);
    if (C.P.s.use_grad_based_estimate) {
#pragma unroll
        for (int k = 0; k < DPL; ++k) {
            bool valid = C.elem(k) < C.dim;
            double val = __builtin_sqrt(dv.a[k] / gv.a[k]);
            double sd = sig.a[k], isd = isig.a[k];
            if (!(!is_finite(val) | (val == 0.0))) {
                val = clampd(val, 1e-20, 1e20);
                sd = __builtin_sqrt(val);
                isd = __builtin_sqrt(1.0 / val);
            }
            double var = sd * sd;
            double mean = var * gm.a[k];
            mean = __builtin_fma(1.0, dm.a[k], mean);
            sig.a[k] = valid ? sd : 0.0;
            isig.a[k] = valid ? isd : 0.0;
            mu.a[k] = valid ? mean : 0.0;
        }
    } else {
        const double scale = 1.0 / (double)C.sc.cnt_fg;
#pragma unroll
        for (int k = 0; k < DPL; ++k) {
            bool valid = C.elem(k) < C.dim;
            double d = dv.a[k] * scale;
            double sd = sig.a[k], isd = isig.a[k];
            if (!(!is_finite(d) | (d == 0.0))) {
                double val = clampd(d, 1e-20, 1e20);
                sd = __builtin_sqrt(val);
                isd = __builtin_sqrt(1.0 / val);
            }
            sig.a[k] = valid ? sd : 0.0;
            isig.a[k] = valid ? isd : 0.0;
            mu.a[k] = valid ? dm.a[k] : 0.0;
        }
    }
    commit_mass_matrix(C, sig, isig, mu);
    return true;
}

// GlobalStrategy::adapt (reference src/adapt_strategy.rs:121-222).  x, gx = chosen draw.
template <int DPL, int W, class Dens>
NM_DEV uint64_t adapt(ChainCtx<DPL, W, Dens>& C, AcceptCollector& col, bool is_good,
                      const Tile<DPL>& x, const Tile<DPL>& gx) {
    const nm_settings& s = C.P.s;
    ChainScalars& sc = C.sc;
    const uint64_t draw = sc.draw_count;
    sc.last_mean_tree_accept = col.mean();                       // step_size.update (stepsize/adapt.rs:201-209)
    sc.last_sym_mean_tree_accept = col.mean_sym();
    sc.last_n_steps = col.count;
    sc.last_max_energy_error = col.max_energy_error;
    if (draw >= s.num_tune) {
        update_stepsize(C, true);
        sc.tuning = 0;
        return NM_CHAIN_OK;
    }
    if (draw < C.P.final_step_size_window) {
        const bool is_early = draw < C.P.early_end;
        if (!is_early && draw == C.P.early_end)
            sc.current_window_size = sc.current_window_size > sc.cnt_bg ? sc.current_window_size : sc.cnt_bg;
        const uint64_t switch_freq = is_early ? s.early_mass_matrix_switch_freq : sc.current_window_size;
        // The four estimators (foreground / background x draws / gradients: 8 vectors) are read once, updated,
        // switched and used for the new mass matrix in registers, and written back once: one HBM round trip per
        // draw instead of one per add_sample / copy / adapt step.
        Tile<DPL> fdm, fdv, fgm, fgv, bdm, bdv, bgm, bgv;
        C.loadP(fdm, E_DM); C.loadP(fdv, E_DV); C.loadP(fgm, E_GM); C.loadP(fgv, E_GV);
        C.loadP(bdm, B_DM); C.loadP(bdv, B_DV); C.loadP(bgm, B_GM); C.loadP(bgv, B_GV);
        bool dirty = false;
        const bool frozen = s.freeze_transform != 0;            // engine knob: the transformation is given, no estimator
        if (is_good && !frozen) {                                // update_estimators (adapt/diagonal.rs:134-141)
            sc.cnt_fg += 1;
            sc.cnt_bg += 1;
            running_variance_add_regs(fdm, fdv, sc.cnt_fg, x);
            running_variance_add_regs(fgm, fgv, sc.cnt_fg, gx);
            running_variance_add_regs(bdm, bdv, sc.cnt_bg, x);
            running_variance_add_regs(bgm, bgv, sc.cnt_bg, gx);
            dirty = true;
        }
        const bool could_switch = !frozen && sc.cnt_bg >= switch_freq;
        uint64_t next_window_size;
        if (is_early) next_window_size = s.early_mass_matrix_switch_freq;
        else {
            // (current_window_size as f64 * growth).round() as u64 : round half away from zero
            double gv = (double)sc.current_window_size * s.mass_matrix_window_growth;
            double fl = __builtin_floor(gv);
            uint64_t grown = (uint64_t)((gv - fl >= 0.5) ? fl + 1.0 : fl);
            next_window_size = sc.current_window_size + 1 > grown ? sc.current_window_size + 1 : grown;
        }
        const bool is_late = next_window_size + draw > C.P.final_step_size_window;
        bool force_update = false;
        if (could_switch && !is_late) {                          // switch (adapt/diagonal.rs:143-148)
            fdm = bdm; fdv = bdv; fgm = bgm; fgv = bgv;
            sc.cnt_fg = sc.cnt_bg;
            sc.cnt_bg = 0;
#pragma unroll
            for (int k = 0; k < DPL; ++k) { bdm.a[k] = 0.0; bdv.a[k] = 0.0; bgm.a[k] = 0.0; bgv.a[k] = 0.0; }
            force_update = true;
            dirty = true;
            if (!is_early) sc.current_window_size = next_window_size;
        }
        if (dirty) {                                             // written back before the mass matrix needs registers
            C.storeP(bdm, B_DM); C.storeP(bdv, B_DV); C.storeP(bgm, B_GM); C.storeP(bgv, B_GV);
            C.storeP(fdm, E_DM); C.storeP(fdv, E_DV); C.storeP(fgm, E_GM); C.storeP(fgv, E_GV);
        }
        bool did_change = false;
        if (!frozen && (force_update | (draw - sc.last_update >= s.mass_matrix_update_freq))) did_change = mass_matrix_adapt(C, fdm, fdv, fgm, fgv);
        if (did_change) sc.last_update = draw;
        update_estimator(C, is_late);
        if (did_change & (sc.has_initial_mass_matrix != 0)) {
            sc.has_initial_mass_matrix = 0;
            return stepsize_init(C, x);
        }
        update_stepsize(C, false);
        return NM_CHAIN_OK;
    }
    update_estimator(C, true);
    update_stepsize(C, draw == s.num_tune - 1);
    return NM_CHAIN_OK;
}

// ---------------------------------------------------------------------------------------------
// GlobalStrategy<LowRankMassMatrixStrategy> (reference src/adapt_strategy.rs:121-222 over src/transform/adapt/low_rank.rs).
// The schedule, the window bookkeeping and the step-size part run here; `mass_matrix_adapt.adapt()` itself — the dense
// linear algebra of compute_update — is the host's: the chain pauses (LR_WAIT_HOST) and the next launch resumes it.
// ---------------------------------------------------------------------------------------------
template <int DPL, int W, class Dens>
NM_DEV void lr_push(ChainCtx<DPL, W, Dens>& C, uint64_t chain, const Tile<DPL>& x, const Tile<DPL>& gx) {
    const KParams& P = C.P;
    const uint64_t idx = C.sc.lr_start + C.sc.lr_len;            // draws.push_back / grads.push_back (adapt/low_rank.rs:323-333)
    if (idx < P.lr_cap) {
        double* base = P.lrwin + ((size_t)chain * P.lr_cap + idx) * 2 * P.dim;
#pragma unroll
        for (int k = 0; k < DPL; ++k) {
            const int d = C.elem(k);
            if (d < C.dim) { base[d] = x.a[k]; base[P.dim + d] = gx.a[k]; }
        }
    }
    C.sc.lr_len += 1;
}
// what follows `mass_matrix_adapt.adapt()` in GlobalStrategy::adapt (adapt_strategy.rs:190-213)
template <int DPL, int W, class Dens>
NM_DEV uint64_t adapt_tail(ChainCtx<DPL, W, Dens>& C, bool did_change, bool is_late, const Tile<DPL>& x) {
    ChainScalars& sc = C.sc;
    if (did_change) sc.last_update = sc.draw_count;
    update_estimator(C, is_late);
    if (did_change & (sc.has_initial_mass_matrix != 0)) {
        sc.has_initial_mass_matrix = 0;
        return stepsize_init(C, x);
    }
    update_stepsize(C, false);
    return NM_CHAIN_OK;
}
template <int DPL, int W, class Dens>
NM_DEV uint64_t adapt_lr(ChainCtx<DPL, W, Dens>& C, uint64_t chain, AcceptCollector& col, bool is_good,
                         const Tile<DPL>& x, const Tile<DPL>& gx) {
    const nm_settings& s = C.P.s;
    ChainScalars& sc = C.sc;
    const uint64_t draw = sc.draw_count;
    sc.last_mean_tree_accept = col.mean();
    sc.last_sym_mean_tree_accept = col.mean_sym();
    sc.last_n_steps = col.count;
    sc.last_max_energy_error = col.max_energy_error;
    if (draw >= s.num_tune) {
        update_stepsize(C, true);
        sc.tuning = 0;
        return NM_CHAIN_OK;
    }
    if (draw < C.P.final_step_size_window) {
        const bool frozen = s.freeze_transform != 0;
        const bool is_early = draw < C.P.early_end;
        const uint64_t bg0 = frozen ? 0 : sc.lr_len - sc.lr_split;          // background_count (adapt/low_rank.rs:343-345)
        if (!is_early && draw == C.P.early_end) sc.current_window_size = sc.current_window_size > bg0 ? sc.current_window_size : bg0;
        const uint64_t switch_freq = is_early ? s.early_mass_matrix_switch_freq : sc.current_window_size;
        if (is_good && !frozen) lr_push(C, chain, x, gx);                  // update_estimators
        const bool could_switch = !frozen && (sc.lr_len - sc.lr_split) >= switch_freq;
        uint64_t next_window_size;
        if (is_early) next_window_size = s.early_mass_matrix_switch_freq;
        else {
            double gv = (double)sc.current_window_size * s.mass_matrix_window_growth;
            double fl = __builtin_floor(gv);
            uint64_t grown = (uint64_t)((gv - fl >= 0.5) ? fl + 1.0 : fl);
            next_window_size = sc.current_window_size + 1 > grown ? sc.current_window_size + 1 : grown;
        }
        const bool is_late = next_window_size + draw > C.P.final_step_size_window;
        bool force_update = false;
        if (could_switch && !is_late) {                                    // switch (adapt/low_rank.rs:335-342)
            sc.lr_start += sc.lr_split;
            sc.lr_len -= sc.lr_split;
            sc.lr_split = sc.lr_len;
            force_update = true;
            if (!is_early) sc.current_window_size = next_window_size;
        }
        if (!frozen && (force_update | (draw - sc.last_update >= s.mass_matrix_update_freq)) && sc.lr_len >= 3) {
            sc.lr_pending = LR_WAIT_HOST;                                   // adapt() -> update(): the host's part
            sc.lr_is_late = is_late ? 1 : 0;
            return NM_CHAIN_OK;
        }
        return adapt_tail(C, false, is_late, x);
    }
    update_estimator(C, true);
    update_stepsize(C, draw == s.num_tune - 1);
    return NM_CHAIN_OK;
}
// LowRankMassMatrix::update (low_rank.rs:155-186) once the host has written sigma / 1/sigma / mean (P_SIG, P_ISIG, P_MU),
// mu_lr and the eigenvectors (lrvec), lambda^(+-1/2) (lrval) and -1/2 sum ln lambda (lr_upd_logdet)
template <int DPL, int W, class Dens>
NM_DEV void lr_commit_update(ChainCtx<DPL, W, Dens>& C) {
    Tile<DPL> t;
    C.loadP(t, P_SIG); C.store(t, C.lsig);
    C.loadP(t, P_MU); C.store(t, C.lmu);
    C.loadP(t, P_ISIG);
    const double diag_logdet = sum_ln_tile(C, t);                 // DiagMassMatrix::set_transform (diagonal.rs:155-161)
    C.sc.mm_logdet = C.sc.lr_upd_logdet + diag_logdet;            // inner.logdet() + diag.logdet()
    C.sc.mm_id += 1;
    C.sc.lr_has_inner = 1;
    C.sc.lr_rank = C.sc.lr_upd_rank;
}

// ---------------------------------------------------------------------------------------------
// The tree.  Iterative restatement of NutsTree::extend (reference src/nuts.rs:108-170).
//
// A doubling of the main tree at depth j generates 2^j leaves in one direction.  Leaf n (0-based) closes
// the sub-trees of levels 1..t, t = number of trailing one bits of n; each closing is a merge of the pending
// level-(k-1) sub-tree A with the just-completed level-(k-1) sub-tree B (whose last leaf is the newest point).
// Leaves are produced in pairs into two register-resident points, E (even leaf) and O (odd leaf), so the
// level-1 merge and the B.first operand of the level-2 merge never touch memory.  What must survive longer is
// addressed by level, with no copies:
//   leaf n = 0 mod 4 -> F[tz(n)] (F[j] for n = 0): first leaf of sub-trees of level >= 2 that start at n
//   odd  leaf n      -> L[to(n)]                  : last leaf of the pending level-to(n) sub-tree
//   merge at level k >= 2 at odd leaf n: A.first = F[tz(n+1-2^k)] (F[j] if that leaf is 0), A.last = L[k-1],
//                                        B.first = E (k = 2) or F[k-1] (k > 2), B.last = O.
// Candidates are renamed, never copied: a pool slot index travels with the (log_size, candidate) scalars, and a
// candidate is written (its z only; g_x is recomputed once per draw for the winner) when its registers are
// about to be reused.
// ---------------------------------------------------------------------------------------------
struct CandRef { int slot; double logp, ke; int64_t idx; };   // slot: -3 live in E, -2 live in O, -1 initial point, >=0 pool
enum TreeStop { STOP_NONE = 0, STOP_TURNING = 1, STOP_DIVERGING = 2, STOP_FATAL = 3 };

// multinomial merge weights (reference merge_into, src/nuts.rs:172-207).  Returns take_B.
template <int DPL, int W, class Dens>
NM_DEV bool merge_weights(ChainCtx<DPL, W, Dens>& C, double a_log_size, double b_log_size, bool is_main, double& total, bool& fatal) {
#ifdef NM_X_NO_MERGE_MATH      // timing experiment only (results are wrong): what do the merges' special functions + Bernoulli cost?
    total = (a_log_size > b_log_size ? a_log_size : b_log_size) + 0.5; (void)is_main; (void)fatal;
    return false;
#endif
    NM_MARK(C, 13)
#if NM_MERGE_MATH_ROUTINE
    // the whole of merge_into's arithmetic in one branch-free routine (dev_math.hpp merge_math).  The chain's next u64 is read here and
    // consumed only if the routine says random_bool drew it; refilling the word cache early changes nothing (it is a window onto the stream).
    if (!C.rng.has(2)) C.rng.refill();
    const uint32_t off_ = (uint32_t)(C.rng.pos - C.rng.base);
    const MergeOut mo = merge_math(a_log_size, b_log_size, is_main ? 1u : 0u, C.rng.cache[off_], C.rng.cache[off_ + 1]);
    total = uniform_f64(mo.total);
    const uint32_t mf = (uint32_t)__builtin_amdgcn_readfirstlane((int)mo.flags);
    C.rng.pos += (uint64_t)(mf & 2u);
    NM_MARK(C, 29)
    if (mf & 4u) fatal = true;
    return (mf & 1u) != 0;
#else
    total = logaddexp(a_log_size, b_log_size);
    NM_MARK(C, 14)
    const double self_log_size = is_main ? a_log_size : total;
    if (b_log_size >= self_log_size) return true;
    const double p_ = uexp(b_log_size - self_log_size);
    NM_MARK(C, 15)
    int b = C.rng.random_bool(p_);
    NM_MARK(C, 29)
    if (b < 0) { fatal = true; return false; }
    return b == 1;
#endif
}

struct DrawResult {
    uint64_t depth;
    bool diverging, reached_maxdepth, has_divergence_energy_error;
    bool has_div_end;            // DivergenceInfo.end_location is Some (None when the divergence is a recoverable logp error)
    double divergence_energy_error;
    int64_t div_start_idx;       // index_in_trajectory of the point the divergent leapfrog started from
    CandRef chosen;
    double e0;
};


// is_turning and the direction of integration.  The reference orders the two states by index_in_trajectory
// (transformed_hamiltonian.rs:617-638): integrating backwards, the LATER generated point b is `start` and the earlier one a is
// `end`.  With s = (z_end + 0) - z_start, t1 += s v_start, t2 += s v_end (turn_acc), swapping the roles negates s exactly
// (x - y = -(y - x) in IEEE arithmetic; both are +0 when x == y), hence every product, every partial sum and every step of the
// reduction: (t1, t2)_backward = (-t2, -t1)_forward, up to the sign of zeros, which `< 0` cannot see.  So the sums are always
// accumulated in generation order (a = start) and the direction only flips the comparison: fwd: t < 0, backward: t > 0 (NaN: false
// both ways).  Per-element operand selects by `fwd` were 128 v_cndmask per level-1 test at 16 doubles per lane (round 5: a lone
// wavefront is bound by the number of instructions it issues, tools/probes/ubench_issue.hip).
NM_DEV bool turn_sign(bool fwd, double t) { return fwd ? t < 0. : t > 0.; }
NM_DEV bool turn_any6(bool fwd, double s1, double s2, double s3, double s4, double s5, double s6) {
    return fwd ? ((s1 < 0.) | (s2 < 0.) | (s3 < 0.) | (s4 < 0.) | (s5 < 0.) | (s6 < 0.))
               : ((s1 > 0.) | (s2 > 0.) | (s3 > 0.) | (s4 > 0.) | (s5 > 0.) | (s6 > 0.));
}
// is_turning(first-generated a, later-generated b) for two register points
template <int DPL, int W>
NM_DEV bool turning_regs(const Pt<DPL>& a, const Pt<DPL>& b, bool fwd, Reducer<W>& R) {
    double s1 = 0., s2 = 0.;
#pragma unroll
    for (int k = 0; k < DPL; ++k) turn_acc(a.z.a[k], a.v.a[k], b.z.a[k], b.v.a[k], s1, s2);
    double s2v[2] = {s1, s2};
    return R.any_sign(s2v, fwd);
}

// the candidate's z goes to a fresh pool slot
template <int DPL, int W, class Dens>
NM_DEV int cand_to_pool(ChainCtx<DPL, W, Dens>& C, uint32_t& used, const Tile<DPL>& z) {
    const int p = (int)__builtin_ctz(~used);
    used |= 1u << p;
    C.storeS_nt(z, slot_C(C.maxdepth_cfg, p));
    return p;
}

// ---------------------------------------------------------------------------------------------
// Batched multinomial merges (round 3).  Measured on one funnel chain (tools/leaf_latency.py, profiles/r03*): of the 2.23 us a
// leaf costs a lone wavefront, 0.93 us are the merges' scalar arithmetic — logaddexp (exp + ln_1p), exp, a Bernoulli word — a
// dependent chain evaluated 64 lanes wide for ONE value, once per leaf.  But a merge needs nothing from the leapfrogs that
// follow it, and nothing that follows needs the merge before the doubling ends: U-turn tests read end points, divergence reads
// energies, and the multinomial candidate is consumed by the merge into the main tree.  So the merges of levels 1..6 are
// deferred and evaluated per aligned chunk of 64 leaves, VECTORISED: lane n keeps the log-weight of leaf n (the AcceptCollector
// trick), level k of the merge tree is ONE evaluation of the special functions across the lanes that close a level-k sub-tree
// (32, 16, .. 1 merges at once), six rounds for 63 merges.  Each merge computes exactly what merge_into computes
// (src/nuts.rs:172-207) on the same operands, so the bits are the same; the random words are handed out afterwards in the
// reference's post-order (by closing leaf, then by level: a prefix count over the lanes of the merges that consume a word), so
// the stream position and every Bernoulli outcome are the sequential ones.  A doubling that stops early (U-turn at level k of
// leaf n, divergence) performed, in the reference, exactly the merges that close at leaves before n plus levels 1..k at n: the
// same routine counts their words with a mask (their outcomes die with the discarded sub-tree).  Levels >= 7 (one merge per 64
// leaves) stay sequential, with the chunk as their "leaf".  The candidate of an unresolved sub-tree is not known when its
// registers are reused, so every leaf's z goes to a ring of NM_RING scratch slots (one streaming store per leaf instead of one
// candidate store per pair); the winner is copied to the candidate pool when its chunk becomes a pending sub-tree.
// ---------------------------------------------------------------------------------------------
struct ChunkResult { double log_size; int cand_lane; bool fatal; };
NM_DEV double shfl_f64(double x, int src_lane) {
    const int lo = __shfl(__double2loint(x), src_lane), hi = __shfl(__double2hiint(x), src_lane);
    return __hiloint2double(hi, lo);
}
// Eighteen inlined special-function evaluations: the transition has exactly ONE call site for it (three of them pushed the leaf
// loop's registers into scratch: 503 spilled SGPRs; as a real call it corrupted the LrWrap<HostCb> kernel's state).
// wv: lane l holds the log-weight of the chunk's leaf l.  The merges performed are those closing at lanes < n_last (every
// level up to m) and levels 1..k_last at lane n_last (packed: m | n_last << 8 | k_last << 16).  `words` points at the chain's
// next random word in the LDS cache (the caller made sure 128 words are there).  Returns the level-m node's log size
// (meaningful when the chunk is complete: n_last = 2^m - 1, k_last = m) and, packed, its candidate lane | fatal << 8 |
// words consumed << 16.
struct ChunkOut { double log_size; uint32_t packed; };
typedef const __attribute__((address_space(3))) uint32_t* lds_words_t;       // an LDS pointer (ds_read, not a flat load)
NM_DEV ChunkOut resolve_chunk_core(double wv, uint32_t shape, lds_words_t words) {
    const int m = (int)(shape & 0xff), n_last = (int)((shape >> 8) & 0xff), k_last = (int)((shape >> 16) & 0xff);
    const int l = lane_id();
    const double ln2 = 0x1.62e42fefa39efp-1;    // = dlog(2.0) (tests/cpp/merge_math_check.hip)
    double ls = wv;                      // log size of the sub-tree this lane currently closes (level 0: its own leaf)
    uint64_t pint[6];                    // Bernoulli thresholds of this lane's merges, by level
    uint32_t need = 0, sure = 0, bad = 0, perf = 0;
#pragma unroll
    for (int k = 1; k <= 6; ++k) {
        pint[k - 1] = 0;
        if (k <= m) {                    // (m is wave-uniform)
            const int h = 1 << (k - 1);
            const bool part = ((l + 1) & ((1 << k) - 1)) == 0;                    // this lane closes a level-k sub-tree
            const bool done = part && (l < n_last || (l == n_last && k <= k_last));
            const double a = shfl_f64(ls, l - h), b = ls;                         // A = the sibling that closed h leaves earlier
            const double diff = a - b;
            const double e = exp_sl(diff > 0. ? -diff : diff);                      // (branch-free forms of the same operation sequences, dev_math.hpp)
            const double lp = log1p_unit(e);
            const double total = a == b ? a + ln2 : (diff > 0. ? a + lp : (diff < 0. ? b + lp : diff));   // logaddexp (util.rs:6-19)
            const bool ge = b >= total;                                           // self.log_size = the merged size (not the main tree)
            const double p_ = exp_sl(b - total);
            const bool in01 = p_ >= 0.0 && p_ < 1.0;                              // random_bool(p): p outside [0, 1) draws nothing
            pint[k - 1] = in01 ? (uint64_t)(p_ * 18446744073709551616.0) : 0ull;
            if (done) {
                perf |= 1u << k;
                if (ge || (!in01 && p_ == 1.0)) sure |= 1u << k;
                else if (in01) need |= 1u << k;
                else bad |= 1u << k;
            }
            if (part) ls = total;
        }
    }
    const uint32_t fatal = __ballot(bad != 0) != 0ull ? 1u : 0u;
    // the words, in post-order: by closing lane, then by level
    uint32_t before = 0, total_words = 0;
#pragma unroll
    for (int k = 1; k <= 6; ++k) {
        const uint64_t bm = __ballot((need >> k) & 1u);
        before += __builtin_amdgcn_mbcnt_hi((uint32_t)(bm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)bm, 0u));
        total_words += 2u * (uint32_t)__builtin_popcountll(bm);
    }
    uint32_t take = sure;
#pragma unroll
    for (int k = 1; k <= 6; ++k) {
        if ((need >> k) & 1u) {
            const uint32_t q = before + (uint32_t)__builtin_popcount(need & ((1u << k) - 1u));
            const uint64_t w = ((uint64_t)words[2u * q + 1u] << 32) | words[2u * q];
            if (w < pint[k - 1]) take |= 1u << k;
        }
    }
    // candidates: the merged sub-tree keeps A's draw unless B's is taken (src/nuts.rs:199-205)
    int cand = l;
#pragma unroll
    for (int k = 1; k <= 6; ++k) {
        if (k <= m) {
            const int from = __shfl(cand, l - (1 << (k - 1)));
            if ((perf >> k) & 1u) cand = ((take >> k) & 1u) ? cand : from;
        }
    }
    const int top = (1 << m) - 1;
    ChunkOut R;
    R.log_size = readlane_f64(ls, top);
    R.packed = (uint32_t)__builtin_amdgcn_readlane(cand, top) | (fatal << 8) | ((uint32_t)__builtin_amdgcn_readfirstlane((int)total_words) << 16);
    return R;
}
template <int DPL, int W, class Dens>
NM_DEV ChunkResult resolve_chunk(ChainCtx<DPL, W, Dens>& C, double wv, int m, int n_last, int k_last) {
    if (!C.rng.has(128)) C.rng.refill();                                          // at most 63 merges x 2 words
    const ChunkOut o = resolve_chunk_core(wv, (uint32_t)m | ((uint32_t)n_last << 8) | ((uint32_t)k_last << 16),
                                          (lds_words_t)(C.rng.cache + (uint32_t)(C.rng.pos - C.rng.base)));
    C.rng.pos += (uint64_t)(o.packed >> 16);
    ChunkResult R;
    R.log_size = uniform_f64(o.log_size);
    R.cand_lane = __builtin_amdgcn_readfirstlane((int)(o.packed & 0xff));
    R.fatal = ((o.packed >> 8) & 1u) != 0;
    return R;
}
// the chunk's winner leaves the ring: its z goes to a pool slot (the ring is overwritten by the next 64 leaves)
template <int DPL, int W, class Dens>
NM_DEV int ring_to_pool(ChainCtx<DPL, W, Dens>& C, uint32_t& used, int ring_lane) {
    Tile<DPL> z;
    C.loadS(z, slot_R(C.maxdepth_cfg, ring_lane));
    return cand_to_pool(C, used, z);
}

// nuts::draw (reference src/nuts.rs:281-388).  On entry the chain's current point is in its slots P_*.
// On exit, if R.chosen.slot >= 0, zc holds the chosen point's z.
template <int DPL, int W, class Dens>
NM_DEV uint64_t nuts_transition(ChainCtx<DPL, W, Dens>& C, AcceptCollector& col, DrawResult& R, Tile<DPL>& zc) {
    const nm_settings& s = C.P.s;
    ChainScalars& sc = C.sc;
    const int MD = C.maxdepth_cfg;
    Pt<DPL> E, O;
    constexpr bool RE = reg_edges<DPL, W>();
    constexpr bool NOG = nog_mode<DPL, W, Dens>();       // points are (z, v): no g_z tile (see nog_mode)
    constexpr bool FD = NOG && bool(NM_FD);                // the first leaf of the doubling in progress lives in FDz / FDv
    constexpr int TCH = (DPL == 16 && W == 1 && !batched_merges<DPL, W>()) ? NM_TEST_CHUNK : 0;     // iterations a U-turn test requests its slots' rows ahead (16-doubles tiling)
    [[maybe_unused]] Tile<DPL> Gsh;                        // NM_GTILE: g_z of the point the last leapfrog produced
    [[maybe_unused]] Tile<DPL>* const gsh = NM_GTILE ? &Gsh : nullptr;
    [[maybe_unused]] Tile<DPL> FDz, FDv;
    [[maybe_unused]] EdgeTile<DPL, NM_EDGES_IN_ACC != 0> MLz, MLv, MRz, MRv;   // RE: (z, v) of the main tree's left / right end point; their g_z (read only when the
                                                          // trajectory is extended on the other side) stays in the scratch slot
    // ---- initialize_trajectory (transformed_hamiltonian.rs:687-736)
    NM_MARK(C, 0)
    sample_velocity(C, E.v, !RE);
    NM_MARK(C, 1)
    if (sc.mm_id != sc.transform_id) {                           // lazy re-whitening (inv_transform_normalize, diagonal.rs:210-221)
        Tile<DPL> x, gx, isig, sig, mu;
        C.loadP(x, P_X);
        C.loadP(gx, P_GX);
        if constexpr (lr_trait<Dens>::value) {                   // low_rank.rs:302-314
            transform_to_z(C, x, E.z);
            if constexpr (tile_trait<Dens>::value) C.dens.tile_skip_density();   // keeps the block's (apply, density, apply) cadence
            transform_to_gz(C, gx, E.g);
        } else {
        C.loadP(isig, P_ISIG);
        C.load(sig, C.lsig);
        C.load(mu, C.lmu);
#pragma unroll
        for (int k = 0; k < DPL; ++k) {
            double t = __builtin_fma(-1.0, mu.a[k], x.a[k]);
            E.z.a[k] = isig.a[k] * t;
            E.g.a[k] = gx.a[k] * sig.a[k];
        }
        }
        C.storeP(E.z, P_Z);
        C.storeP(E.g, P_GZ);
        sc.logdet = sc.mm_logdet;
        sc.transform_id = sc.mm_id;
    } else {
        C.loadP(E.z, P_Z);
        if constexpr (!NOG) C.loadP(E.g, P_GZ);          // (NOG: the initial point's g_z is streamed from P_GZ by the leapfrogs that start there)
    }
    const double logdet = sc.logdet;
    const double ke_init = initial_kinetic(C, E.v, !RE);
    E.ke = ke_init;
    if constexpr (RE) { MLz.put(E.z); MLv.put(E.v); MRz.put(E.z); MRv.put(E.v); }      // both ends = the initial point
    const double e0 = ke_init - (sc.logp + logdet);
    R.e0 = e0;
    col.register_init(e0);
    // main tree = the initial point
    int left_slot = 0, right_slot = 0;   // edge slot ids
    bool o_is_edge = false;              // O still holds the edge written by the last successful doubling ...
    int o_edge_sign = 0;                 // ... in this direction
    uint64_t depth = 0;
    double log_size = 0.;
    int64_t left_idx = 0, right_idx = 0;
    [[maybe_unused]] double left_ke = ke_init, right_ke = ke_init;   // the edges' kinetic_energy: an input of the microcanonical leapfrog only
    CandRef mc = {-1, sc.logp, ke_init, 0};
    uint32_t used = 0;   // candidate-pool occupancy bitmask
    // batched merges (resolve_chunk): lane (n mod 64) keeps leaf n's log-weight, logp and kinetic energy until its chunk is resolved
    constexpr bool BATCH = batched_merges<DPL, W>();
    [[maybe_unused]] double wv = 0., lpv = 0., kev = 0.;

    uint64_t mindepth = s.mindepth, maxdepth = s.maxdepth;
    if (s.has_target_integration_time) {                         // src/nuts.rs:300-320
        // (`as u64` saturates, NaN -> 0; the reference then panics on log2(0) — here that case counts as 1 step)
        double q = __builtin_ceil(s.target_integration_time / sc.step_size);
        uint64_t max_steps = q >= 18446744073709551616.0 ? ~0ull : (q > 0 ? (uint64_t)q : 0ull);
        uint64_t fl = 63 - __builtin_clzll(max_steps | 1ull);
        uint64_t ce = ((max_steps & (max_steps - 1)) == 0) ? fl : fl + 1;
        mindepth = fl > s.mindepth ? fl : s.mindepth;
        uint64_t xd = ce > mindepth ? ce : mindepth;
        maxdepth = xd < s.maxdepth ? xd : s.maxdepth;
    }
    // (read once per draw into a register: as `s.max_energy_error` the compiler re-loaded it from the kernel's argument block after
    // every leapfrog — an s_load and its full ~200-cycle latency, 50 instruction slots of a lone wavefront, on every leaf)
    const double max_energy_error = pin_scalar(s.max_energy_error);
    R.diverging = false; R.reached_maxdepth = false; R.has_divergence_energy_error = false; R.has_div_end = true;
    R.divergence_energy_error = 0.; R.div_start_idx = 0;
    const bool want_div = C.P.out_div_start || C.P.out_div_start_grad || C.P.out_div_end;
    bool fatal = false;
    bool in_extra = false;        // inside the `for _ in 0..extra_doublings` loop of src/nuts.rs:350-371
    uint64_t extra_left = 0;
    int sign = 1;

    for (;;) {
        bool check;
        if (!in_extra) {
            if (C.P.dim == 0) break;                             // src/nuts.rs:322-326: nothing to integrate, the draw is the initial point
            if (!(depth < maxdepth)) { R.reached_maxdepth = true; break; }
            sign = C.rng.random_bool_std() ? 1 : -1;             // src/nuts.rs:334, hamiltonian.rs:111-118
            check = (s.check_turning != 0) && !(depth < mindepth);
        } else {
            if (extra_left == 0) break;
            extra_left -= 1;
            check = false;
        }
        NM_MARK(C, 24)
        const bool fwd = sign > 0;
        const int64_t edge_idx = fwd ? right_idx : left_idx;
        const uint64_t nleaf = 1ull << depth;
        const uint32_t used_before = used;
        const double epsilon = (double)sign * sc.step_size * 1.0;
        int stop = STOP_NONE;
        double sub_log_size = 0.;
        CandRef sub_cand = {-2, 0., 0., 0};
        const bool reuse_edge = o_is_edge && o_edge_sign == sign;
        o_is_edge = false;               // O is about to be overwritten; set again only by a successful merge

        // divergence test + collector for a fresh leaf (transformed_hamiltonian.rs:590-612); returns -energy_error
#define NM_LEAF_ACCOUNT(START, PT, WOUT)                                                                  \
        {                                                                                                 \
            const double energy_ = (PT).ke - ((PT).logp + logdet);                                        \
            const double err_ = energy_ - e0;                                                             \
            const int dst_ = dens_status(C);   /* logp error (transformed_hamiltonian.rs:562-578) */          \
            if (dst_ == 2) { stop = STOP_FATAL; }                                                         \
            else if (dst_ == 1) {              /* recoverable: a divergence without energy error / end point */ \
                col.register_divergent();                                                                 \
                R.diverging = true; R.has_divergence_energy_error = false; R.has_div_end = false;         \
                R.div_start_idx = (PT).idx - (int64_t)sign;                                               \
                if (want_div) C.storeS((START).z, slot_F(0));                                             \
                stop = STOP_DIVERGING;                                                                    \
            } else                                                                                        \
            if (bad_energy(C, err_, max_energy_error)) {                                                  \
                col.register_divergent();                                                                 \
                R.diverging = true; R.has_divergence_energy_error = true; R.divergence_energy_error = err_; \
                R.div_start_idx = (PT).idx - (int64_t)sign;                                               \
                if (want_div) {          /* DivergenceInfo locations: the F[0] scratch pair is dead from here on */ \
                    C.storeS((START).z, slot_F(0)); C.storeS((PT).z, slot_F(0) + 1);                      \
                }                                                                                         \
                stop = STOP_DIVERGING;                                                                    \
            } else {                                                                                      \
                col.register_ok(energy_);                                                                 \
                WOUT = -err_;                                                                             \
            }                                                                                             \
        }

        // top-level U-turn tests of the finished `other` sub-tree against the main tree (src/nuts.rs:143-161); O is its
        // last leaf.  Reads end points only, so it can run before the sub-tree's own last merges (see the pair loop).
        auto top_level_turning = [&]() __attribute__((always_inline)) -> bool {
                double s1 = 0., s2 = 0., s3 = 0., s4 = 0., s5 = 0., s6 = 0.;
                const auto mlz = C.edge_z(left_slot), mlv = C.edge_v(left_slot);
                const auto mrz = C.edge_z(right_slot), mrv = C.edge_v(right_slot);
                if (depth == 0) {
#if NM_TRIM_FIRST
                    return turning_regs(E, O, fwd, C.red);        // (initial point, leaf): both in registers
#else
#pragma unroll
                    for (int m = 0; m < DPL / 2; ++m) {
                        double2 az = fwd ? C.ld2(mlz.r, mlz.so, m) : C.ld2(mrz.r, mrz.so, m);
                        double2 av = fwd ? C.ld2(mlv.r, mlv.so, m) : C.ld2(mrv.r, mrv.so, m);
                        turn_acc(az.x, av.x, O.z.a[2 * m], O.v.a[2 * m], s1, s2); turn_acc(az.y, av.y, O.z.a[2 * m + 1], O.v.a[2 * m + 1], s1, s2);
                    }
                    C.red.sum2(s1, s2);
                    return turn_sign(fwd, s1) | turn_sign(fwd, s2);
#endif
                } else {
                    // other.first (leaf 0 of this doubling): F[depth]; at depth 1 it is still E
                    const int so_ofz = C.soS(slot_F((int)depth)), so_ofv = C.soS(slot_F((int)depth) + 1);
                    const bool of_in_regs = NM_TRIM_FIRST && depth == 1;
                    // The reference's three pairs (src/nuts.rs:143-161), each written (earlier in the trajectory, later):
                    //   forward   (tree.left, other.right) (tree.right, other.right) (tree.left, other.left)     other.right = O, other.left = its first leaf
                    //   backward  (other.left, tree.right) (other.right, tree.right) (other.left, tree.left)     other.left = O, other.right = its first leaf
                    // In generation order (tree side first; turning_regs' note: a pair the other way round is the same sums negated) both
                    // are (left, O) (right, O) (X, first leaf) with X = the tree's edge FAR from `other`: left going forward, right going
                    // backward.  One loop per direction (they differ in one operand; selecting it per element would be 8 v_cndmask per row).
                    auto rows = [&](auto far_is_left, auto of_regs) __attribute__((always_inline)) {
                        constexpr bool OFR = decltype(of_regs)::value;       // other.first is in registers (E at depth 1, FDz / FDv with NM_FD)
                        auto row = [&](int m, double2 oz, double2 ov) __attribute__((always_inline)) {
                            double2 lz, lv, rz, rv;
                            if constexpr (RE) {
                                lz = MLz.pair(m); lv = MLv.pair(m); rz = MRz.pair(m); rv = MRv.pair(m);
                            } else {
                                lz = C.ld2(mlz.r, mlz.so, m); lv = C.ld2(mlv.r, mlv.so, m);
                                rz = C.ld2(mrz.r, mrz.so, m); rv = C.ld2(mrv.r, mrv.so, m);
                            }
                            const double cz0 = O.z.a[2 * m], cz1 = O.z.a[2 * m + 1];
                            const double cv0 = O.v.a[2 * m], cv1 = O.v.a[2 * m + 1];
                            turn_acc(lz.x, lv.x, cz0, cv0, s1, s2); turn_acc(lz.y, lv.y, cz1, cv1, s1, s2);
                            turn_acc(rz.x, rv.x, cz0, cv0, s3, s4); turn_acc(rz.y, rv.y, cz1, cv1, s3, s4);
                            if constexpr (decltype(far_is_left)::value) { turn_acc(lz.x, lv.x, oz.x, ov.x, s5, s6); turn_acc(lz.y, lv.y, oz.y, ov.y, s5, s6); }
                            else { turn_acc(rz.x, rv.x, oz.x, ov.x, s5, s6); turn_acc(rz.y, rv.y, oz.y, ov.y, s5, s6); }
                        };
                        if constexpr (!OFR && TCH > 0) {                     // the slot's rows requested TCH iterations ahead (see the level-k tests)
#pragma unroll
                            for (int m0 = 0; m0 < DPL / 2; m0 += (TCH > 0 ? TCH : 1)) {
                                double2 ozb[TCH > 0 ? TCH : 1], ovb[TCH > 0 ? TCH : 1];
#pragma unroll
                                for (int c = 0; c < TCH; ++c) if (m0 + c < DPL / 2) { ozb[c] = NM_TLD(C.rs, so_ofz, m0 + c, E.z); ovb[c] = NM_TLD(C.rs, so_ofv, m0 + c, E.v); }
                                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                                for (int c = 0; c < TCH; ++c) if (m0 + c < DPL / 2) row(m0 + c, ozb[c], ovb[c]);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        } else {
#pragma unroll
                        for (int m = 0; m < DPL / 2; ++m) {
                            double2 oz, ov;
                            if constexpr (FD) { oz = make_double2(FDz.a[2 * m], FDz.a[2 * m + 1]); ov = make_double2(FDv.a[2 * m], FDv.a[2 * m + 1]); }   // leaf 0 of this doubling
                            else if constexpr (OFR) { oz = make_double2(E.z.a[2 * m], E.z.a[2 * m + 1]); ov = make_double2(E.v.a[2 * m], E.v.a[2 * m + 1]); }
                            else { oz = NM_TLD(C.rs, so_ofz, m, E.z); ov = NM_TLD(C.rs, so_ofv, m, E.v); }
                            row(m, oz, ov);
                            NM_GROUP_BARRIER(m);
                        }
                        }
                    };
                    if (FD || of_in_regs) { if (fwd) rows(std::true_type{}, std::true_type{}); else rows(std::false_type{}, std::true_type{}); }
                    else { if (fwd) rows(std::true_type{}, std::false_type{}); else rows(std::false_type{}, std::false_type{}); }
                    { double sv6[6] = {s1, s2, s3, s4, s5, s6}; return C.red.any_sign(sv6, fwd); }
                }
        };
        if (depth == 0) {
            // a single leaf from the initial point, which E has held since initialize_trajectory: E -> O
#if !NM_TRIM_FIRST
            { const int es = fwd ? right_slot : left_slot;
              C.loadRef(E.z, C.edge_z(es)); C.loadRef(E.v, C.edge_v(es)); C.loadRef(E.g, C.edge_g(es)); }
#endif
            if constexpr (NOG) leapfrog<DPL, W, Dens, 2, 2>(C, E, O, epsilon, (Tile<DPL>*)nullptr, (Tile<DPL>*)nullptr, C.edge_g(0), gsh);
            else leapfrog<DPL, W, Dens, 2>(C, E, O, epsilon, (Tile<DPL>*)nullptr, (Tile<DPL>*)nullptr);
            O.idx = edge_idx + (int64_t)sign;
            NM_LEAF_ACCOUNT(E, O, sub_log_size)
            NM_MARK(C, 27)
            sub_cand = {-2, O.logp, O.ke, O.idx};
        } else {
            [[maybe_unused]] bool g_from_slot = false;      // NOG: the first leapfrog of this doubling starts at the trajectory's initial point
            if (!reuse_edge) {                              // same direction as the last doubling: the edge is still in O
                const int es = fwd ? right_slot : left_slot;
                if constexpr (RE) {
                    if constexpr (NOG) g_from_slot = es == 0;
                    else C.loadRef(O.g, C.edge_g(es));
                    // (a real branch: as selects the choice costs a v_cndmask per 32 bits of both tiles; the empty asm keeps the arms apart)
                    if (fwd) { asm volatile("; reload: right end"); MRz.get(O.z); MRv.get(O.v); } else { asm volatile("; reload: left end"); MLz.get(O.z); MLv.get(O.v); }
                } else {
                C.loadRef(O.z, C.edge_z(es)); C.loadRef(O.v, C.edge_v(es));
                if constexpr (NOG) g_from_slot = es == 0; else C.loadRef(O.g, C.edge_g(es));
                }
                if constexpr (kin_trait<Dens>::value) O.ke = fwd ? right_ke : left_ke;
            }
            for (uint64_t n = 0; n < nleaf; n += 2) {
                if constexpr (!BATCH) {
                // ======== every merge where the reference evaluates it (tilings without the batched merges; unchanged since round 2) ========
                // ---- even leaf n
                double wE = 0., wO = 0.;
                NM_MARK(C, 16)
                if constexpr (NOG) {
                    if (n == 0 && g_from_slot) leapfrog<DPL, W, Dens, 2, 2>(C, O, E, epsilon, (Tile<DPL>*)nullptr, (Tile<DPL>*)nullptr, C.edge_g(0), gsh);
                    else if (NM_GTILE && !(n == 0 && !reuse_edge)) leapfrog<DPL, W, Dens, 2, 3>(C, O, E, epsilon, (Tile<DPL>*)nullptr, (Tile<DPL>*)nullptr, C.edge_g(0), gsh);
                    else leapfrog<DPL, W, Dens, 2, 1>(C, O, E, epsilon, (Tile<DPL>*)nullptr, (Tile<DPL>*)nullptr, C.edge_g(0), gsh);
                } else
                leapfrog<DPL, W, Dens, 2>(C, O, E, epsilon, (Tile<DPL>*)nullptr, (Tile<DPL>*)nullptr);
                NM_MARK(C, 17)
                E.idx = edge_idx + (int64_t)sign * (int64_t)(n + 1);
                NM_LEAF_ACCOUNT(O, E, wE)
                if (stop != STOP_NONE) break;
                // ---- odd leaf n + 1
                NM_MARK(C, 18)
                if constexpr (NOG) leapfrog<DPL, W, Dens, 2, (NM_GTILE ? 3 : 1)>(C, E, O, epsilon, (Tile<DPL>*)nullptr, (Tile<DPL>*)nullptr, C.edge_g(0), gsh);
                else
                leapfrog<DPL, W, Dens, 2>(C, E, O, epsilon, (Tile<DPL>*)nullptr, (Tile<DPL>*)nullptr);
                NM_MARK(C, 19)
                O.idx = edge_idx + (int64_t)sign * (int64_t)(n + 2);
                NM_LEAF_ACCOUNT(E, O, wO)
                if (stop != STOP_NONE) break;
                const uint64_t nn = n + 1;
                const int t = (int)__builtin_ctzll(~nn);           // trailing ones of the odd leaf: merges up to level t
                // ---- every U-turn test this leaf completes (levels 1..t, and the top-level one when it is the last leaf
                // of the doubling) BEFORE any merge arithmetic: the tests only read fixed end points, so their HBM round
                // trips overlap each other instead of alternating with the merges' long scalar chains (exp, ln_1p,
                // Bernoulli), which run afterwards on the recorded bits.  A turning level ends the doubling, so higher
                // levels are not even loaded.  Bit k of turn_bits: the level-k test says "turning".
                uint32_t turn_bits = 0;
                NM_MARK(C, 20)
                if (check) {
                    if (turning_regs(E, O, fwd, C.red)) turn_bits |= 2u;
                    for (int k = 2; k <= t && turn_bits == 0; ++k) {
                        // (A.first,B.last) (A.last,B.last) (A.first,B.first) in generation order  [src/nuts.rs:143-161]
                        const uint64_t a_first = nn + 1 - (1ull << k);
                        const int fa = a_first == 0 ? (int)depth : (int)__builtin_ctzll(a_first);
                        const int so_afz = C.soS(slot_F(fa)), so_afv = C.soS(slot_F(fa) + 1);
                        const int so_alz = C.soS(slot_L(MD, k - 1)), so_alv = C.soS(slot_L(MD, k - 1) + 1);
                        double s1 = 0., s2 = 0., s3 = 0., s4 = 0., s5 = 0., s6 = 0.;
                        // (NOG: A.first is the doubling's first leaf when the sub-tree starts there — registers FDz / FDv instead of the slot F[depth])
                        auto level_rows = [&](auto a_in_fd) __attribute__((always_inline)) {
                        constexpr bool AFD = decltype(a_in_fd)::value;
                        if (k == 2) {
                            const double2* l1z2 = C.tptr(C.l1z);      // A.last = L[1] lives in LDS
                            const double2* l1v2 = C.tptr(C.l1v);
                            auto row2 = [&](int m, double2 az, double2 av) __attribute__((always_inline)) {
                                const double2 lz = l1z2[m * 64 * W], lv = l1v2[m * 64 * W];
#pragma unroll
                                for (int j = 0; j < 2; ++j) {
                                    const double azj = j ? az.y : az.x, avj = j ? av.y : av.x;
                                    const double lzj = j ? lz.y : lz.x, lvj = j ? lv.y : lv.x;
                                    const double cz = O.z.a[2 * m + j], cv = O.v.a[2 * m + j];
                                    const double bz = E.z.a[2 * m + j], bv = E.v.a[2 * m + j];
                                    turn_acc(azj, avj, cz, cv, s1, s2);      // (generation order; the direction flips the comparison: turn_any6)
                                    turn_acc(lzj, lvj, cz, cv, s3, s4);
                                    turn_acc(azj, avj, bz, bv, s5, s6);
                                }
                            };
                            if constexpr (!AFD && TCH > 0) {
                                // (round 6) the slot's rows REQUESTED TCH iterations ahead of their use: left to itself the compiler issues two loads, waits for
                                // both, computes, and issues the next two — eight serialised round trips per test at 16 doubles per lane
#pragma unroll
                                for (int m0 = 0; m0 < DPL / 2; m0 += (TCH > 0 ? TCH : 1)) {
                                    double2 azb[TCH > 0 ? TCH : 1], avb[TCH > 0 ? TCH : 1];
#pragma unroll
                                    for (int c = 0; c < TCH; ++c) if (m0 + c < DPL / 2) { azb[c] = NM_TLD(C.rs, so_afz, m0 + c, E.z); avb[c] = NM_TLD(C.rs, so_afv, m0 + c, E.v); }
                                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                                    for (int c = 0; c < TCH; ++c) if (m0 + c < DPL / 2) row2(m0 + c, azb[c], avb[c]);
                                    __builtin_amdgcn_sched_barrier(0);
                                }
                            } else {
#pragma unroll
                            for (int m = 0; m < DPL / 2; ++m) {
                                double2 az, av;
                                if constexpr (AFD) { az = make_double2(FDz.a[2 * m], FDz.a[2 * m + 1]); av = make_double2(FDv.a[2 * m], FDv.a[2 * m + 1]); }
                                else { az = NM_TLD(C.rs, so_afz, m, E.z); av = NM_TLD(C.rs, so_afv, m, E.v); }
                                row2(m, az, av);
                                NM_GROUP_BARRIER(m);
                            }
                            }
                        } else {
                            const int so_bfz = C.soS(slot_F(k - 1)), so_bfv = C.soS(slot_F(k - 1) + 1);
                            auto row3 = [&](int m, double2 az, double2 av, double2 lz, double2 lv, double2 bz2, double2 bv2) __attribute__((always_inline)) {
#pragma unroll
                                for (int j = 0; j < 2; ++j) {
                                    const double azj = j ? az.y : az.x, avj = j ? av.y : av.x;
                                    const double lzj = j ? lz.y : lz.x, lvj = j ? lv.y : lv.x;
                                    const double bz = j ? bz2.y : bz2.x, bv = j ? bv2.y : bv2.x;
                                    const double cz = O.z.a[2 * m + j], cv = O.v.a[2 * m + j];
                                    turn_acc(azj, avj, cz, cv, s1, s2);      // (generation order; the direction flips the comparison: turn_any6)
                                    turn_acc(lzj, lvj, cz, cv, s3, s4);
                                    turn_acc(azj, avj, bz, bv, s5, s6);
                                }
                            };
                            constexpr int TCH3 = TCH >= 2 ? (TCH / 2 < NM_TEST_CHUNK3_MAX ? TCH / 2 : NM_TEST_CHUNK3_MAX) : 0;      // six loads per iteration here: fewer iterations ahead
                            if constexpr (!AFD && TCH3 > 0) {
#pragma unroll
                                for (int m0 = 0; m0 < DPL / 2; m0 += (TCH3 > 0 ? TCH3 : 1)) {
                                    constexpr int T3 = TCH3 > 0 ? TCH3 : 1;
                                    double2 azb[T3], avb[T3], lzb[T3], lvb[T3], bzb[T3], bvb[T3];
#pragma unroll
                                    for (int c = 0; c < TCH3; ++c) if (m0 + c < DPL / 2) {
                                        azb[c] = NM_TLD(C.rs, so_afz, m0 + c, E.z); avb[c] = NM_TLD(C.rs, so_afv, m0 + c, E.v);
                                        lzb[c] = NM_TLD(C.rs, so_alz, m0 + c, O.z); lvb[c] = NM_TLD(C.rs, so_alv, m0 + c, O.v);
                                        bzb[c] = NM_TLD(C.rs, so_bfz, m0 + c, E.v); bvb[c] = NM_TLD(C.rs, so_bfv, m0 + c, O.z);
                                    }
                                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                                    for (int c = 0; c < TCH3; ++c) if (m0 + c < DPL / 2) row3(m0 + c, azb[c], avb[c], lzb[c], lvb[c], bzb[c], bvb[c]);
                                    __builtin_amdgcn_sched_barrier(0);
                                }
                            } else {
#pragma unroll
                            for (int m = 0; m < DPL / 2; ++m) {
                                double2 az, av;
                                if constexpr (AFD) { az = make_double2(FDz.a[2 * m], FDz.a[2 * m + 1]); av = make_double2(FDv.a[2 * m], FDv.a[2 * m + 1]); }
                                else { az = NM_TLD(C.rs, so_afz, m, E.z); av = NM_TLD(C.rs, so_afv, m, E.v); }
                                const double2 lz = NM_TLD(C.rs, so_alz, m, O.z), lv = NM_TLD(C.rs, so_alv, m, O.v);
                                const double2 bz2 = NM_TLD(C.rs, so_bfz, m, E.v), bv2 = NM_TLD(C.rs, so_bfv, m, O.z);
                                row3(m, az, av, lz, lv, bz2, bv2);
                                NM_GROUP_BARRIER(m);
                            }
                            }
                        }
                        };
                        if constexpr (FD) { if (a_first == 0) level_rows(std::true_type{}); else level_rows(std::false_type{}); }
                        else level_rows(std::false_type{});
                        { double sv6[6] = {s1, s2, s3, s4, s5, s6}; if (C.red.any_sign(sv6, fwd)) turn_bits |= 1u << k; }
                    }
                }
                NM_MARK(C, 21)
                // ---- level-1 merge: A = {E}, B = {O}
                {
                    double total;
                    const bool take = merge_weights(C, wE, wO, false, total, fatal);
                    sub_cand = take ? CandRef{-2, O.logp, O.ke, O.idx} : CandRef{-3, E.logp, E.ke, E.idx};
                    sub_log_size = total;
                    if (fatal) { stop = STOP_FATAL; break; }
                    if (turn_bits & 2u) { stop = STOP_TURNING; break; }
                }
                for (int k = 2; k <= t; ++k) {
                    const PendEntry A = C.pend[k - 1];
                    double total;
                    const bool take = merge_weights(C, A.log_size, sub_log_size, false, total, fatal);
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
                NM_MARK(C, 22)
                // F: the even leaf (still in E) is the first leaf of the sub-trees of level >= 2 that start at n.  Stored here,
                // after the merges: every call of the merge arithmetic waits for all stores in flight, and nothing reads F
                // before the next pair.  (At depth 1 leaf 0 is still in E when the top-level tests need it.)
#ifndef NM_X_NO_SCRATCH_STORES   // (timing experiment only: results are wrong without the stores)
                if (FD && n == 0) { FDz = E.z; FDv = E.v; }          // the doubling's first leaf stays in registers (any depth >= 1)
                else
                if ((n & 3) == 0 && (depth > 1 || !NM_TRIM_FIRST)) {
                    const int fs = slot_F(n == 0 ? (int)depth : (int)__builtin_ctzll(n));
                    C.storeS(E.z, fs);
                    C.storeS(E.v, fs + 1);
                }
#endif
                if (n + 2 < nleaf) {
                    // O is the last leaf of the pending level-t sub-tree; its candidate leaves the registers
#ifndef NM_X_NO_SCRATCH_STORES
                    if (t == 1) { C.store(O.z, C.l1z); C.store(O.v, C.l1v); }
                    else { C.storeS(O.z, slot_L(MD, t)); C.storeS(O.v, slot_L(MD, t) + 1); }
                    if (sub_cand.slot == -2) sub_cand.slot = cand_to_pool(C, used, O.z);
                    else if (sub_cand.slot == -3) sub_cand.slot = cand_to_pool(C, used, E.z);
#else
                    if (sub_cand.slot < 0) sub_cand.slot = 0;
#endif
                    PendEntry e;
                    e.log_size = sub_log_size; e.cand_logp = sub_cand.logp; e.cand_ke = sub_cand.ke;
                    e.cand_idx = sub_cand.idx; e.cand_slot = sub_cand.slot; e.pad = 0;
                    C.pend[t] = e;
                }
                } else {
                // ======== batched merges (resolve_chunk): DPL <= 4, one wave per chain ========
                // ---- even leaf n
                double wE = 0., wO = 0.;
                NM_MARK(C, 16)
                leapfrog<DPL, W, Dens, 2>(C, O, E, epsilon, (Tile<DPL>*)nullptr, (Tile<DPL>*)nullptr);
                NM_MARK(C, 17)
                E.idx = edge_idx + (int64_t)sign * (int64_t)(n + 1);
                NM_LEAF_ACCOUNT(O, E, wE)
                // BATCH: what the single resolve_chunk site below has to do for this pair — the merges closing at lanes < rs_nl
                // and levels 1..rs_kl at lane rs_nl (a divergence: the merges that closed before the divergent leaf drew their
                // words, src/nuts.rs:131-136; the even leaf closes nothing, so both leaves of a pair give lane n - 1)
                [[maybe_unused]] bool rs_do = false;
                [[maybe_unused]] int rs_nl = 0, rs_kl = 0;
                const int mm = depth < 6 ? (int)depth : 6;      // levels the batch covers
                if constexpr (BATCH) {
                    if (stop == STOP_DIVERGING && (n & 63) != 0) {
                        rs_do = true; rs_nl = (int)(n & 63) - 1; rs_kl = (int)__builtin_ctz(~(unsigned)rs_nl);
                    }
                }
                if (stop == STOP_NONE) {
                if constexpr (BATCH) {
                    if (lane_id() == (int)(n & 63)) { wv = wE; lpv = E.logp; kev = E.ke; }
                }
                // ---- odd leaf n + 1
                NM_MARK(C, 18)
                leapfrog<DPL, W, Dens, 2>(C, E, O, epsilon, (Tile<DPL>*)nullptr, (Tile<DPL>*)nullptr);
                NM_MARK(C, 19)
                O.idx = edge_idx + (int64_t)sign * (int64_t)(n + 2);
                NM_LEAF_ACCOUNT(E, O, wO)
                if constexpr (BATCH) {
                    if (stop == STOP_DIVERGING && (n & 63) != 0) {
                        rs_do = true; rs_nl = (int)(n & 63) - 1; rs_kl = (int)__builtin_ctz(~(unsigned)rs_nl);
                    }
                }
                }
                const uint64_t nn = n + 1;
                const int t = (int)__builtin_ctzll(~nn);           // trailing ones of the odd leaf: merges up to level t
                // ---- every U-turn test this leaf completes (levels 1..t, and the top-level one when it is the last leaf
                // of the doubling) BEFORE any merge arithmetic: the tests only read fixed end points, so their HBM round
                // trips overlap each other instead of alternating with the merges' long scalar chains (exp, ln_1p,
                // Bernoulli), which run afterwards on the recorded bits.  A turning level ends the doubling, so higher
                // levels are not even loaded.  Bit k of turn_bits: the level-k test says "turning".
                uint32_t turn_bits = 0;
                bool chunk_end = false;      // BATCH: this leaf closes its chunk (64 leaves, or the whole doubling when it is smaller)
                if (stop == STOP_NONE) {
                if constexpr (BATCH) {
                    if (lane_id() == (int)((n + 1) & 63)) { wv = wO; lpv = O.logp; kev = O.ke; }
                }
                    NM_MARK(C, 20)
                    if (check) {
                        if (turning_regs(E, O, fwd, C.red)) turn_bits |= 2u;
                        for (int k = 2; k <= t && turn_bits == 0; ++k) {
                            // (A.first,B.last) (A.last,B.last) (A.first,B.first) in generation order  [src/nuts.rs:143-161]
                            const uint64_t a_first = nn + 1 - (1ull << k);
                            const int fa = a_first == 0 ? (int)depth : (int)__builtin_ctzll(a_first);
                            const int so_afz = C.soS(slot_F(fa)), so_afv = C.soS(slot_F(fa) + 1);
                            const int so_alz = C.soS(slot_L(MD, k - 1)), so_alv = C.soS(slot_L(MD, k - 1) + 1);
                            double s1 = 0., s2 = 0., s3 = 0., s4 = 0., s5 = 0., s6 = 0.;
                            if (k == 2) {
                                const double2* l1z2 = C.tptr(C.l1z);      // A.last = L[1] lives in LDS
                                const double2* l1v2 = C.tptr(C.l1v);
#pragma unroll
                                for (int m = 0; m < DPL / 2; ++m) {
                                    const double2 az = C.ld2(C.rs, so_afz, m), av = C.ld2(C.rs, so_afv, m);
                                    const double2 lz = l1z2[m * 64 * W], lv = l1v2[m * 64 * W];
#pragma unroll
                                    for (int j = 0; j < 2; ++j) {
                                        const double azj = j ? az.y : az.x, avj = j ? av.y : av.x;
                                        const double lzj = j ? lz.y : lz.x, lvj = j ? lv.y : lv.x;
                                        const double cz = O.z.a[2 * m + j], cv = O.v.a[2 * m + j];
                                        const double bz = E.z.a[2 * m + j], bv = E.v.a[2 * m + j];
                                        turn_acc(azj, avj, cz, cv, s1, s2);      // (generation order; the direction flips the comparison: turn_any6)
                                        turn_acc(lzj, lvj, cz, cv, s3, s4);
                                        turn_acc(azj, avj, bz, bv, s5, s6);
                                    }
                                    NM_GROUP_BARRIER(m);
                                }
                            } else {
                                const int so_bfz = C.soS(slot_F(k - 1)), so_bfv = C.soS(slot_F(k - 1) + 1);
#pragma unroll
                                for (int m = 0; m < DPL / 2; ++m) {
                                    const double2 az = C.ld2(C.rs, so_afz, m), av = C.ld2(C.rs, so_afv, m);
                                    const double2 lz = C.ld2(C.rs, so_alz, m), lv = C.ld2(C.rs, so_alv, m);
                                    const double2 bz2 = C.ld2(C.rs, so_bfz, m), bv2 = C.ld2(C.rs, so_bfv, m);
#pragma unroll
                                    for (int j = 0; j < 2; ++j) {
                                        const double azj = j ? az.y : az.x, avj = j ? av.y : av.x;
                                        const double lzj = j ? lz.y : lz.x, lvj = j ? lv.y : lv.x;
                                        const double bz = j ? bz2.y : bz2.x, bv = j ? bv2.y : bv2.x;
                                        const double cz = O.z.a[2 * m + j], cv = O.v.a[2 * m + j];
                                        turn_acc(azj, avj, cz, cv, s1, s2);      // (generation order; the direction flips the comparison: turn_any6)
                                        turn_acc(lzj, lvj, cz, cv, s3, s4);
                                        turn_acc(azj, avj, bz, bv, s5, s6);
                                    }
                                    NM_GROUP_BARRIER(m);
                                }
                            }
                            { double sv6[6] = {s1, s2, s3, s4, s5, s6}; if (C.red.any_sign(sv6, fwd)) turn_bits |= 1u << k; }
                        }
                    }
                NM_MARK(C, 21)
                if constexpr (BATCH) {
                    const int ln = (int)(nn & 63);
                    chunk_end = ln == 63 || nn + 1 == nleaf;
                    // the pair's z go to the ring HERE, behind the tests' loads (a load waits for every older store on this
                    // hardware: one counter for both) and as far ahead of the next pair's loads as possible; a pair that ends the
                    // doubling early never needs them
                    if (turn_bits == 0) {
                        C.storeS_nt(E.z, slot_R(MD, (int)(n & 63)));
                        C.storeS_nt(O.z, slot_R(MD, ln));
                    }
                    if (turn_bits != 0 || chunk_end) {
                        // a turning level k ends the doubling after its merge (src/nuts.rs:163-169): levels 1..k of this leaf are performed
                        rs_do = true; rs_nl = ln; rs_kl = turn_bits ? (int)__builtin_ctz(turn_bits) : t;
                    }
                }
                }
                int k_seq = 2;               // first level the sequential loop below merges
                if constexpr (BATCH) {
                    k_seq = 7;
                    if (rs_do) {             // the ONE site of the batched merges
                        const ChunkResult cr = resolve_chunk(C, wv, mm, rs_nl, rs_kl < mm ? rs_kl : mm);
                        if (cr.fatal) { fatal = true; stop = STOP_FATAL; }
                        else if (stop == STOP_NONE) {
                            sub_log_size = cr.log_size;
                            const int cl = cr.cand_lane;
                            sub_cand = CandRef{NM_RING + cl, readlane_f64(lpv, cl), readlane_f64(kev, cl),
                                               edge_idx + (int64_t)sign * (int64_t)((nn & ~63ull) + (uint64_t)cl + 1)};
                            if (turn_bits != 0 && rs_kl <= mm) stop = STOP_TURNING;
                        }
                    }
                    if (stop != STOP_NONE) break;
                } else {
                    if (stop != STOP_NONE) break;
                // ---- level-1 merge: A = {E}, B = {O}
                    double total;
                    const bool take = merge_weights(C, wE, wO, false, total, fatal);
                    sub_cand = take ? CandRef{-2, O.logp, O.ke, O.idx} : CandRef{-3, E.logp, E.ke, E.idx};
                    sub_log_size = total;
                    if (fatal) { stop = STOP_FATAL; break; }
                    if (turn_bits & 2u) { stop = STOP_TURNING; break; }
                }
                if (!BATCH || chunk_end)
                for (int k = k_seq; k <= t; ++k) {
                    const PendEntry A = C.pend[k - 1];
                    double total;
                    const bool take = merge_weights(C, A.log_size, sub_log_size, false, total, fatal);
                    if (take) {
                        used &= ~(1u << A.cand_slot);
                    } else {
                        if (sub_cand.slot >= 0 && sub_cand.slot < NM_RING) used &= ~(1u << sub_cand.slot);
                        sub_cand = {A.cand_slot, A.cand_logp, A.cand_ke, A.cand_idx};
                    }
                    sub_log_size = total;
                    if (fatal) { stop = STOP_FATAL; break; }
                    if ((turn_bits >> k) & 1u) { stop = STOP_TURNING; break; }
                }
                if (stop != STOP_NONE) break;
                NM_MARK(C, 22)
                // F: the even leaf (still in E) is the first leaf of the sub-trees of level >= 2 that start at n.  Stored here,
                // after the merges: every call of the merge arithmetic waits for all stores in flight, and nothing reads F
                // before the next pair.  (At depth 1 leaf 0 is still in E when the top-level tests need it.)
#ifndef NM_X_NO_SCRATCH_STORES   // (timing experiment only: results are wrong without the stores)
                if ((n & 3) == 0 && (depth > 1 || !NM_TRIM_FIRST)) {
                    const int fs = slot_F(n == 0 ? (int)depth : (int)__builtin_ctzll(n));
                    C.storeS(E.z, fs);
                    C.storeS(E.v, fs + 1);
                }
#endif
                if (n + 2 < nleaf) {
                    // O is the last leaf of the pending level-t sub-tree; its candidate leaves the registers
#ifndef NM_X_NO_SCRATCH_STORES
                    if (t == 1) { C.store(O.z, C.l1z); C.store(O.v, C.l1v); }
                    else { C.storeS(O.z, slot_L(MD, t)); C.storeS(O.v, slot_L(MD, t) + 1); }
                    if (!BATCH || chunk_end) {
                    if (sub_cand.slot == -2) sub_cand.slot = cand_to_pool(C, used, O.z);
                    else if (sub_cand.slot == -3) sub_cand.slot = cand_to_pool(C, used, E.z);
                    else if (sub_cand.slot >= NM_RING) sub_cand.slot = ring_to_pool(C, used, sub_cand.slot - NM_RING);
                    }
#else
                    if (sub_cand.slot < 0) sub_cand.slot = 0;
#endif
                    if (!BATCH || chunk_end) {       // (batched: only a whole chunk becomes a pending sub-tree, of level t >= 6)
                    PendEntry e;
                    e.log_size = sub_log_size; e.cand_logp = sub_cand.logp; e.cand_ke = sub_cand.ke;
                    e.cand_idx = sub_cand.idx; e.cand_slot = sub_cand.slot; e.pad = 0;
                    C.pend[t] = e;
                    }
                }
                }
                NM_MARK(C, 23)
            }
        }
#undef NM_LEAF_ACCOUNT
        if (stop == STOP_FATAL) { fatal = true; break; }
        if (stop == STOP_DIVERGING) { used = used_before; break; }     // tree unchanged (src/nuts.rs:123, :134-136)
        if (stop == STOP_TURNING) {                                    // `other` discarded (src/nuts.rs:131-133)
            used = used_before;
            if (!in_extra) { in_extra = true; extra_left = s.extra_doublings; }
            continue;
        }
        // ---- `other` is complete (its last leaf is O): top-level turning tests, then merge into the main tree
        NM_MARK(C, 25)
        bool turning = false;
        if (check) turning = top_level_turning();
        NM_MARK(C, 26)
        double total;
        const bool take = merge_weights(C, log_size, sub_log_size, true, total, fatal);
        if (fatal) break;
        if (take) {
            if (mc.slot >= 0) used &= ~(1u << mc.slot);
            if (sub_cand.slot == -2) sub_cand.slot = cand_to_pool(C, used, O.z);
            else if (sub_cand.slot == -3) sub_cand.slot = cand_to_pool(C, used, E.z);
            else if (sub_cand.slot >= NM_RING) sub_cand.slot = ring_to_pool(C, used, sub_cand.slot - NM_RING);
            mc = sub_cand;
        } else if (sub_cand.slot >= 0 && sub_cand.slot < NM_RING) {
            used &= ~(1u << sub_cand.slot);
        }
        // Will there be another doubling?  If not (U-turn with no extra doublings, or the depth limit), nothing reads
        // the new edge any more: its three tiles stay in registers and 24 KiB of HBM writes per draw are saved.
        const bool more = in_extra ? extra_left > 0 : (turning ? s.extra_doublings > 0 : depth + 1 < maxdepth);
        if (more) { // the new edge goes to the slot this side owns alone, or to the free one while both sides share the initial point
            int ns = fwd ? right_slot : left_slot;
            const int other_side = fwd ? left_slot : right_slot;
            if (ns == 0) ns = other_side == 1 ? 2 : 1;          // id 0 (the initial point) is read-only
            if constexpr (RE) {
                if constexpr (!NOG) C.storeRef(O.g, C.edge_g(ns));
                if (fwd) { asm volatile("; new right end"); MRz.put(O.z); MRv.put(O.v); } else { asm volatile("; new left end"); MLz.put(O.z); MLv.put(O.v); }
            } else {
            C.storeRef(O.z, C.edge_z(ns)); C.storeRef(O.v, C.edge_v(ns)); if constexpr (!NOG) C.storeRef(O.g, C.edge_g(ns));
            }
#ifdef NM_EXTRA_TRAFFIC   // development: is the kernel bound by the bytes it moves? (two more tile stores per doubling)
            C.storeS(O.v, slot_F(MD)); C.storeS(O.z, slot_F(MD) + 1);
#endif
            if (fwd) right_slot = ns; else left_slot = ns;
            o_is_edge = true; o_edge_sign = sign;
        }
        NM_MARK(C, 28)
        if (fwd) right_idx = O.idx; else left_idx = O.idx;
        if constexpr (kin_trait<Dens>::value) { if (fwd) right_ke = O.ke; else left_ke = O.ke; }
        depth += 1;
        log_size = total;
#ifdef NM_X_SABOTAGE_LR41   // (tools/runs/gpu_r06p.sh only: ONE deliberately wrong instantiation, to show that the known-answer checks reject it)
        if constexpr (DPL == 4 && W == 1 && lr_trait<Dens>::value) log_size = 0.5 * total;
#endif
        if (turning && !in_extra) { in_extra = true; extra_left = s.extra_doublings; }
    }
    R.depth = depth;
    R.chosen = mc;
    if (fatal) return NM_CHAIN_LOGP_FATAL;
    if (mc.slot >= 0) C.loadS(zc, slot_C(MD, mc.slot));
    return NM_CHAIN_OK;
}

// ---------------------------------------------------------------------------------------------
// NutsChain::draw (reference src/chain.rs:151-188) + the scalar stats of expanded_draw (:190-232)
// ---------------------------------------------------------------------------------------------
// one row of a [n_draws][n_chains][dim] statistics array
template <int DPL, int W, class Dens>
NM_DEV void write_row(ChainCtx<DPL, W, Dens>& C, double* base, size_t row, const Tile<DPL>& t, bool positions = false) {
    if (!base) return;
    double* dst = base + row + C.goff;
#if NM_WRITE_ROW_BUF
  if (NM_WRITE_ROW_BUF == 1 || positions) {
    // One buffer descriptor per row (its base is wave-uniform), a lane's pair of elements (2 t, 2 t + 1) as ONE 16-byte store, the
    // rows beyond dim cut off by the descriptor's range (whole pairs: the range ends on a pair boundary, an odd last element is
    // written by its lane below): DPL / 2 coalesced stores instead of DPL masked 8-byte stores with a 64-bit address each.
    const int whole = C.dim & ~1;
    const rsrc_t r = make_rsrc(dst, (uint64_t)whole * 8);
#pragma unroll
    for (int m = 0; m < DPL / 2; ++m) buf_store2_aux<NM_ROW_AUX>(r, C.voff + m * (64 * W * 16), 0, t.a[2 * m], t.a[2 * m + 1]);
    if (C.dim & 1) {
#pragma unroll
        for (int m = 0; m < DPL / 2; ++m)
            if (C.elem(2 * m) == C.dim - 1) dst[C.dim - 1] = t.a[2 * m];
    }
    return;
  }
#endif
#pragma unroll
    for (int k = 0; k < DPL; ++k) {
        int d = C.elem(k);
        if (d < C.dim) dst[d] = t.a[k];
    }
}

// DivergenceInfo.{start_location, start_gradient, end_location} (transformed_hamiltonian.rs:590-604) of the draw's
// divergent leapfrog, from the z of its two points (stashed in F[0] by NM_LEAF_ACCOUNT): x = z·σ + μ and the density
// gradient are recomputed with the leapfrog's own operations, hence the same bits.  A start point with index 0 is
// the trajectory's initial point, whose x and g_x are still in P_X / P_GX.
template <int DPL, int W, class Dens>
NM_DEV void emit_divergence_vectors(ChainCtx<DPL, W, Dens>& C, int64_t start_idx, size_t row, bool has_end = true) {
    const KParams& P = C.P;
    Tile<DPL> x, gx, zt, sig, mu;
    C.load(sig, C.lsig);
    C.load(mu, C.lmu);
    if (start_idx == 0) {
        C.loadP(x, P_X); C.loadP(gx, P_GX);
    } else {
        C.loadS(zt, slot_F(0));
        if constexpr (lr_trait<Dens>::value) transform_to_x(C, zt, x);
        else {
#pragma unroll
        for (int k = 0; k < DPL; ++k) x.a[k] = __builtin_fma(1.0, mu.a[k], zt.a[k] * sig.a[k]);
        }
        (void)C.dens.template eval<DPL, W>(x, gx, C.dim, C.red);
    }
    write_row(C, P.out_div_start, row, x);
    write_row(C, P.out_div_start_grad, row, gx);
    if (!has_end) return;
    C.loadS(zt, slot_F(0) + 1);
    if constexpr (lr_trait<Dens>::value) transform_to_x(C, zt, x);
    else {
#pragma unroll
    for (int k = 0; k < DPL; ++k) x.a[k] = __builtin_fma(1.0, mu.a[k], zt.a[k] * sig.a[k]);
    }
    write_row(C, P.out_div_end, row, x);
}

// ---------------------------------------------------------------------------------------------
// MclmcChain::draw (reference src/mclmc.rs:488-556) with mclmc_kernel (:212-409): KinWrap kernels, nm_settings.sampler =
// NM_SAMPLER_MCLMC.  The chain's point is (P_X, P_GX, P_Z, P_GZ, P_V); the momentum noise of the partial refresh stays in the
// LDS staging area of the normal sampler (C.l1z: no tree uses it here) from its draw to its two uses.
// ---------------------------------------------------------------------------------------------
template <int DPL, int W, class Dens>
NM_DEV void mclmc_sample_noise(ChainCtx<DPL, W, Dens>& C) {           // array_gaussian(noise, ones) (mclmc.rs:250, :304, :313)
    block_sync(W == 1);
#if NM_CLUSTER_MODE
    // as in sample_velocity: every member draws the whole vector slice after slice; its own slice waits in a scratch slot
    for (int j = 0; j < C.red.cl.k; ++j) {
        const int off = j * (int)C.P.cl_slice;
        const int cnt = C.gdim - off < (int)C.P.cl_slice ? C.gdim - off : (int)C.P.cl_slice;
        fill_standard_normals_bulk<(DPL * W + 1 < 17 ? DPL * W + 1 : 17)>(C.rng, reinterpret_cast<uint32_t*>(C.l1v), C.l1z, cnt, C.zig, 64 * W, C.P.prof, C.prof_t);
        if (j == C.red.cl.member) {
            Tile<DPL> nz;
            const double2* s2 = C.tptr(C.l1z);
#pragma unroll
            for (int m = 0; m < DPL / 2; ++m) {
                const double2 q = s2[m * 64 * W];
                nz.a[2 * m] = C.elem(2 * m) < C.dim ? 1.0 * q.x : 0.0;
                nz.a[2 * m + 1] = C.elem(2 * m + 1) < C.dim ? 1.0 * q.y : 0.0;
            }
            C.storeS(nz, slot_F(1));
        }
        block_sync(false);
    }
    return;
#endif
    if (DPL == 2 && C.dim <= 48) fill_standard_normals(C.rng, C.l1z, C.dim, C.zig);
    else fill_standard_normals_bulk<(DPL * W + 1 < 17 ? DPL * W + 1 : 17)>(C.rng, reinterpret_cast<uint32_t*>(C.l1v), C.l1z, C.dim, C.zig, 64 * W, C.P.prof, C.prof_t);
    block_sync(W == 1);
}
// partial_momentum_refresh (transformed_hamiltonian.rs:770-825)
template <int DPL, int W, class Dens>
NM_DEV void mclmc_partial_refresh(ChainCtx<DPL, W, Dens>& C, Pt<DPL>& p, double factor) {
    const double half_step = C.sc.step_size * factor / 2.0;
    const double L = C.P.s.momentum_decoherence_length;
    Tile<DPL> nz;
#if NM_CLUSTER_MODE
    C.loadR(nz, C.rs, slot_F(1));
#else
    const double2* s2 = C.tptr(C.l1z);
#pragma unroll
    for (int m = 0; m < DPL / 2; ++m) {
        const double2 q = s2[m * 64 * W];
        nz.a[2 * m] = C.elem(2 * m) < C.dim ? 1.0 * q.x : 0.0;
        nz.a[2 * m + 1] = C.elem(2 * m + 1) < C.dim ? 1.0 * q.y : 0.0;
    }
#endif
    if (C.sc.kin == NM_TRAJ_MICROCANONICAL) {       // isokinetic Langevin on the unit sphere
        const double nu = __builtin_sqrt(uniform_f64(dexpm1(2.0 * half_step / L)) / (double)C.gdim);
#pragma unroll
        for (int k = 0; k < DPL; ++k) p.v.a[k] = __builtin_fma(nu, nz.a[k], p.v.a[k]);
        normalize_tile(p.v, C.red);
    } else {                                        // Ornstein-Uhlenbeck: alpha p + sqrt(1 - alpha^2) z
        const double alpha = uexp(-half_step / L);
        const double beta = __builtin_sqrt(1.0 - alpha * alpha);
#pragma unroll
        for (int k = 0; k < DPL; ++k) {
            const double nv = __builtin_fma(alpha, p.v.a[k], 0.0);
            p.v.a[k] = __builtin_fma(beta, nz.a[k], nv);
        }
        p.ke = kinetic(p.v, C.red);
    }
}

template <int DPL, int W, class Dens>
NM_DEV void finish_draw_lr(ChainCtx<DPL, W, Dens>& C, uint64_t chain, nm_draw_stats& out, uint64_t ast, uint64_t row_idx);
// (returns false when the chain cannot go on in this launch: failed, or — low-rank adaptation — paused for the host's estimator)
template <int DPL, int W, class Dens>
NM_DEV bool chain_draw_mclmc(ChainCtx<DPL, W, Dens>& C, uint64_t chain, uint64_t t_out) {
    const KParams& P = C.P;
    const nm_settings& s = P.s;
    ChainScalars& sc = C.sc;
    nm_draw_stats out = {};
    out.draw = sc.draw_count; out.chain = P.chain_id_offset + chain;
    // ---- Euclidean -> Microcanonical switch (mclmc.rs:490-504)
    bool resample_velocity = false;
    if (s.mclmc_trajectory_kind == NM_MCLMC_EUCLIDEAN_EARLY_THEN_MICROCANONICAL && sc.draw_count == P.mclmc_switch_draw &&
        sc.kin != NM_TRAJ_MICROCANONICAL) {
        sc.kin = NM_TRAJ_MICROCANONICAL;
        resample_velocity = true;
    }
    const double base_step_size = sc.step_size;
    uint64_t num_base_steps;
    {   // (subsample_frequency * L / eps).round().max(1.0).min(1e6) as u64 (mclmc.rs:219-232): round half away from zero
        const double q = s.subsample_frequency * s.momentum_decoherence_length / base_step_size;
        double r = __builtin_round(q);
        r = r != r ? 1.0 : (r > 1.0 ? r : 1.0);
        r = r < 1e6 ? r : 1e6;
        num_base_steps = (uint64_t)r;
    }
    const int max_halvings = s.dynamic_step_size ? 10 : 0;
    // ---- current = copy(state); initialize_trajectory(current, resample_velocity) (transformed_hamiltonian.rs:687-736)
    Pt<DPL> cur, nxt;
    Tile<DPL> x, gx;
    C.loadP(x, P_X); C.loadP(gx, P_GX);
    if (resample_velocity) sample_velocity(C, cur.v); else C.loadP(cur.v, P_V);
    if (sc.mm_id != sc.transform_id) {                              // inv_transform_normalize (diagonal.rs:210-221, low_rank.rs:302-314)
        if constexpr (lr_trait<Dens>::value) {
            transform_to_z(C, x, cur.z);
            transform_to_gz(C, gx, cur.g);
        } else {
        Tile<DPL> isig, sig, mu;
        C.loadP(isig, P_ISIG); C.load(sig, C.lsig); C.load(mu, C.lmu);
#pragma unroll
        for (int k = 0; k < DPL; ++k) {
            const double t = __builtin_fma(-1.0, mu.a[k], x.a[k]);
            cur.z.a[k] = isig.a[k] * t;
            cur.g.a[k] = gx.a[k] * sig.a[k];
        }
        }
        C.storeP(cur.z, P_Z); C.storeP(cur.g, P_GZ);
        sc.logdet = sc.mm_logdet;
        sc.transform_id = sc.mm_id;
    } else {
        C.loadP(cur.z, P_Z); C.loadP(cur.g, P_GZ);
    }
    const double logdet = sc.logdet;
    cur.ke = resample_velocity ? initial_kinetic(C, cur.v) : (sc.kin == NM_TRAJ_MICROCANONICAL ? 0.0 : kinetic(cur.v, C.red));
    cur.logp = sc.logp; cur.idx = 0;
    const double initial_energy = cur.ke - (cur.logp + logdet);
    mclmc_sample_noise(C);
    const double draw_start_energy = initial_energy;
    AcceptCollector col;
    col.register_init(0.0);                         // adapt.new_collector(): never register_init'ed, the base energy is 0 (dual_avg.rs:119-127)
    bool diverged = false, div_has_energy = false, div_has_end = false;
    double div_energy_error = 0.0;
    uint64_t steps_taken = 0, remaining = num_base_steps, stack0 = 0;
    uint32_t stack_bits = 0;                        // remaining_stack: the entries above the first are 1 or 2
    int stack_len = 0;
    double factor = 1.0, time = 0.0;
    uint64_t fatal = NM_CHAIN_OK;
    Tile<DPL> xn, gxn;
    const int tmp_slot = slot_F(0);                 // tmp_velocity (mclmc.rs:272-274): a scratch slot no tree uses here
    while (remaining > 0) {
        C.storeS(cur.v, tmp_slot);
        mclmc_partial_refresh(C, cur, factor);
        const double step_baseline = cur.ke - (cur.logp + logdet);
        leapfrog(C, cur, nxt, base_step_size * factor, &xn, &gxn);
        nxt.idx = cur.idx + 1;
        const double energy = nxt.ke - (nxt.logp + logdet);
        const double err = energy - step_baseline;
        const int dst = dens_status(C);
        bool div_now = false;
        if (dst == 2) { fatal = NM_CHAIN_LOGP_FATAL; break; }
        if (dst == 1) { col.register_divergent(); div_now = true; div_has_energy = false; div_has_end = false; }
        else if (bad_energy(C, err, s.max_energy_error * factor / (double)num_base_steps)) {
            col.register_divergent(); div_now = true; div_has_energy = true; div_has_end = true; div_energy_error = err;
        } else col.register_ok(energy);
        if (!div_now) {
            mclmc_sample_noise(C);
            mclmc_partial_refresh(C, nxt, factor);
            mclmc_sample_noise(C);
            cur = nxt; x = xn; gx = gxn;
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
            C.loadR(cur.v, C.rs, tmp_slot);
        }
    }
    if (fatal != NM_CHAIN_OK) {
        sc.status = fatal;
        if (P.out_stats && NM_STAT_WRITER(C)) {
            nm_draw_stats zz = {};
            zz.draw = sc.draw_count; zz.chain = P.chain_id_offset + chain; zz.chain_status = fatal;
            P.out_stats[t_out * P.n_chains + chain] = zz;
        }
        return false;
    }
    const size_t row = (size_t)(t_out * P.n_chains + chain) * P.dim;
    // DrawGradCollector::register_draw(current) (adapt/diagonal.rs:73-83)
    const bool is_good = diverged ? (cur.idx > 4) : (cur.idx != 0);
    const double cur_energy = cur.ke - (cur.logp + logdet);
    double energy_change, state_energy, state_energy_error, state_logp;
    int64_t state_idx;
    Tile<DPL> sx, sgx, sz, sgz;                     // the state the draw returns
    if (diverged) {                                 // stay at the pre-trajectory position with a fresh momentum (mclmc.rs:361-388)
        if (P.out_div_start) write_row(C, P.out_div_start, row, x);
        if (P.out_div_start_grad) write_row(C, P.out_div_start_grad, row, gx);
        if (P.out_div_end && div_has_end) write_row(C, P.out_div_end, row, xn);
        Tile<DPL> v;
        sample_velocity(C, v);
        const double ke_new = initial_kinetic(C, v);
        C.storeP(v, P_V);
        C.loadP(sx, P_X); C.loadP(sgx, P_GX); C.loadP(sz, P_Z); C.loadP(sgz, P_GZ);
        energy_change = cur_energy - draw_start_energy;
        state_logp = sc.logp;
        state_energy = ke_new - (state_logp + logdet);
        state_energy_error = state_energy - state_energy;       // energy() - initial_energy of a freshly initialised point
        state_idx = 0;
    } else {
        C.storeP(x, P_X); C.storeP(gx, P_GX); C.storeP(cur.z, P_Z); C.storeP(cur.g, P_GZ); C.storeP(cur.v, P_V);
        sc.logp = cur.logp;
        sx = x; sgx = gx; sz = cur.z; sgz = cur.g;
        energy_change = cur_energy - initial_energy;             // current.energy_error()
        state_logp = cur.logp; state_energy = cur_energy; state_energy_error = energy_change; state_idx = cur.idx;
    }
    sc.px_stale = 0;
    write_row(C, P.out_positions, row, sx);
    write_row(C, P.out_gradient, row, sgx);
    write_row(C, P.out_tpos, row, sz);
    write_row(C, P.out_tgrad, row, sgz);
    double fd = 0.0;
#pragma unroll
    for (int k = 0; k < DPL; ++k) fd = fd + (sz.a[k] + sgz.a[k]) * (sz.a[k] + sgz.a[k]);
    fd = C.red.sum(fd);
    const int64_t trans_id = sc.transform_id;
    sc.total_steps += col.count;
    out.depth = steps_taken; out.maxdepth_reached = 0; out.diverging = diverged;
    out.index_in_trajectory = state_idx; out.transformation_index = trans_id;
    out.logp = state_logp; out.energy = state_energy; out.energy_error = state_energy_error;
    out.fisher_distance = fd;
    out.divergence_energy_error = (diverged && div_has_energy) ? div_energy_error : __builtin_nan("");
    out.energy_change = energy_change; out.average_step_size = time / (double)steps_taken;
    if constexpr (lr_trait<Dens>::value) {          // LowRankMclmcSettings: the adaptation may hand over to the host's estimator
        const uint64_t ast_lr = adapt_lr(C, chain, col, is_good, x, gx);
        if (sc.lr_pending == LR_WAIT_HOST) {        // the rest of this draw happens in lr_resume
            sc.lr_row = t_out;
            if (P.out_stats && NM_STAT_WRITER(C)) P.out_stats[t_out * P.n_chains + chain] = out;
            return false;
        }
        finish_draw_lr(C, chain, out, ast_lr, t_out);
        return sc.status == NM_CHAIN_OK;
    }
    uint64_t ast = adapt(C, col, is_good, x, gx);    // the collector saw `current`; Fixed step size: the state's position is not used
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
        if (P.out_mm_inv) { C.load(sx, C.lsig); write_row(C, P.out_mm_inv, row, sx); }
        if (P.out_mm_mu) { C.load(sx, C.lmu); write_row(C, P.out_mm_mu, row, sx); }
    }
    sc.stats_last_id = sc.mm_id;
    if (P.out_stats && NM_STAT_WRITER(C)) P.out_stats[t_out * P.n_chains + chain] = out;
    sc.draw_count += 1;
    return sc.status == NM_CHAIN_OK;
}

template <int DPL, int W, class Dens>
NM_DEV void chain_draw(ChainCtx<DPL, W, Dens>& C, uint64_t chain, uint64_t t_out) {
    const KParams& P = C.P;
    ChainScalars& sc = C.sc;
    if constexpr (kin_trait<Dens>::value) {
        if (P.s.sampler == NM_SAMPLER_MCLMC) { (void)chain_draw_mclmc(C, chain, t_out); return; }
    }
    AcceptCollector col;
    DrawResult R;
    Tile<DPL> x, gx, z, gz;
    uint64_t st = nuts_transition(C, col, R, z);
    NM_MARK(C, 2)
    nm_draw_stats out;
    out.draw = sc.draw_count; out.chain = P.chain_id_offset + chain;
    if (st != NM_CHAIN_OK) {
        sc.status = st;
        if (P.out_stats && NM_STAT_WRITER(C)) {
            nm_draw_stats zz = {};
            zz.draw = sc.draw_count; zz.chain = P.chain_id_offset + chain; zz.chain_status = st;
            P.out_stats[t_out * P.n_chains + chain] = zz;
        }
        return;
    }
    const size_t row = (size_t)(t_out * P.n_chains + chain) * P.dim;
    if (R.diverging && (P.out_div_start || P.out_div_start_grad || P.out_div_end))
        emit_divergence_vectors(C, R.div_start_idx, row, R.has_div_end);         // before P_X / P_GX take the new draw
    if (R.chosen.slot == -1 && !sc.px_stale) {                   // the draw is the trajectory's initial point
        C.loadP(x, P_X); C.loadP(gx, P_GX);
        C.loadP(z, P_Z); C.loadP(gz, P_GZ);
    } else {
        if (R.chosen.slot == -1) C.loadP(z, P_Z);                // initial point again, its x / g_x were not written
        // the winner's x, g_x, g_z from its z: the same operations as inside the leapfrog => the same bits
        Tile<DPL> sig, mu;
        C.load(sig, C.lsig);
        C.load(mu, C.lmu);
#pragma unroll
        for (int k = 0; k < DPL; ++k) {
            double tt = z.a[k] * sig.a[k];
            x.a[k] = __builtin_fma(1.0, mu.a[k], tt);
        }
        (void)C.dens.template eval<DPL, W>(x, gx, C.dim, C.red);
#pragma unroll
        for (int k = 0; k < DPL; ++k) gz.a[k] = gx.a[k] * sig.a[k];
        // x and g_x of the current point are consumed from memory only by the warm-up (re-whitening, step-size
        // search), by the divergence statistics and by the host after the launch; in between they are skipped
        // (sc.px_stale) and, should a later draw stay on its initial point, rebuilt from P_Z exactly as here.
        const bool need_x = sc.tuning || t_out + 1 == P.n_draws || P.out_div_start || P.out_div_start_grad;
        if (need_x) { C.storeP_nt(x, P_X); C.storeP_nt(gx, P_GX); }
        sc.px_stale = need_x ? 0 : 1;
        C.storeP(z, P_Z); C.storeP(gz, P_GZ);
        sc.logp = R.chosen.logp;
    }
    // DrawGradCollector::register_draw (adapt/diagonal.rs:73-83)
    const int64_t idx = R.chosen.idx;
    const bool is_good = R.diverging ? ((idx < 0 ? -idx : idx) > 4) : (idx != 0);
    write_row(C, P.out_positions, row, x, true);
    write_row(C, P.out_gradient, row, gx);                       // PointStats (transformed_hamiltonian.rs:122-157)
    write_row(C, P.out_tpos, row, z);
    write_row(C, P.out_tgrad, row, gz);
    double fd = 0.0;                                             // sq_norm_sum (cpu_math.rs:235-243)
#pragma unroll
    for (int k = 0; k < DPL; ++k) fd = fd + (z.a[k] + gz.a[k]) * (z.a[k] + gz.a[k]);
    fd = C.red.sum(fd);
    const double energy = R.chosen.ke - (R.chosen.logp + sc.logdet);
    const int64_t trans_id = sc.transform_id;
    sc.total_steps += col.count;
    NM_MARK(C, 3)
    uint64_t ast = adapt(C, col, is_good, x, gx);
    NM_MARK(C, 4)
    if (ast != NM_CHAIN_OK) sc.status = ast;
    out.depth = R.depth; out.maxdepth_reached = R.reached_maxdepth; out.diverging = R.diverging;
    out.tuning = sc.tuning; out.n_steps = sc.last_n_steps;
    out.index_in_trajectory = idx; out.transformation_index = trans_id;
    out.step_size = sc.step_size;
    out.step_size_bar = P.s.step_size_method == NM_STEP_FIXED ? P.s.fixed_step_size
                      : P.s.step_size_method == NM_STEP_ADAM ? uexp(sc.log_step) : uexp(sc.log_step_adapted);
    out.mean_tree_accept = sc.last_mean_tree_accept; out.mean_tree_accept_sym = sc.last_sym_mean_tree_accept;
    out.max_energy_error = sc.last_max_energy_error;
    out.logp = R.chosen.logp; out.energy = energy; out.energy_error = energy - R.e0;
    out.fisher_distance = fd;
    out.divergence_energy_error = (R.diverging && R.has_divergence_energy_error) ? R.divergence_energy_error : __builtin_nan("");
    out.chain_status = ast;
    // DiagMassMatrix::extract_stats (transform/diagonal.rs:48-70): an event when the version moved since the last draw
    out.transformation_update_id = -1;
    out.num_eigenvalues = 0;
    out.energy_change = __builtin_nan(""); out.average_step_size = __builtin_nan("");
    if (sc.mm_id != sc.stats_last_id) {
        out.transformation_update_id = sc.mm_id;
        if (P.out_mm_inv) { C.load(x, C.lsig); write_row(C, P.out_mm_inv, row, x); }
        if (P.out_mm_mu) { C.load(x, C.lmu); write_row(C, P.out_mm_mu, row, x); }
    }
    sc.stats_last_id = sc.mm_id;
    NM_MARK(C, 5)
    if (P.out_stats && NM_STAT_WRITER(C)) P.out_stats[t_out * P.n_chains + chain] = out;
    sc.draw_count += 1;
}

// ---------------------------------------------------------------------------------------------
// The same draw for the LrWrap kernels: the transformation may carry a low-rank part, the adaptation is
// GlobalStrategy<LowRankMassMatrixStrategy>, and a draw can pause inside `adapt` for the host's estimator (its statistics
// row is completed when the chain resumes).  Output rows are addressed by the chain's own draw counter.
// ---------------------------------------------------------------------------------------------
template <int DPL, int W, class Dens>
NM_DEV void finish_draw_lr(ChainCtx<DPL, W, Dens>& C, uint64_t chain, nm_draw_stats& out, uint64_t ast, uint64_t row_idx) {
    const KParams& P = C.P;
    ChainScalars& sc = C.sc;
    if (ast != NM_CHAIN_OK) sc.status = ast;
    const size_t row = (size_t)(row_idx * P.n_chains + chain) * P.dim;
    out.tuning = sc.tuning; out.n_steps = sc.last_n_steps;
    out.step_size = sc.step_size;
    out.step_size_bar = P.s.step_size_method == NM_STEP_FIXED ? P.s.fixed_step_size
                      : P.s.step_size_method == NM_STEP_ADAM ? uexp(sc.log_step) : uexp(sc.log_step_adapted);
    out.mean_tree_accept = sc.last_mean_tree_accept; out.mean_tree_accept_sym = sc.last_sym_mean_tree_accept;
    out.max_energy_error = sc.last_max_energy_error;
    out.chain_status = ast;
    out.transformation_update_id = -1;                            // LowRankMassMatrix::extract_stats (low_rank.rs:218-262)
    out.num_eigenvalues = 0;
    if (P.s.sampler != NM_SAMPLER_MCLMC) { out.energy_change = __builtin_nan(""); out.average_step_size = __builtin_nan(""); }
    if (sc.mm_id != sc.stats_last_id) {
        out.transformation_update_id = sc.mm_id;
        out.num_eigenvalues = sc.lr_has_inner ? sc.lr_rank : 0;
        Tile<DPL> t;
        if (P.out_mm_inv) { C.load(t, C.lsig); write_row(C, P.out_mm_inv, row, t); }
        if (P.out_mm_mu) { C.load(t, C.lmu); write_row(C, P.out_mm_mu, row, t); }
        if (P.out_mm_eigvals && sc.lr_has_inner) {
            double* dst = P.out_mm_eigvals + row;
            for (int d = tid(); d < C.dim; d += 64 * W) dst[d] = d < (int)sc.lr_rank ? C.lvals[d] : __builtin_nan("");
        }
    }
    sc.stats_last_id = sc.mm_id;
    if (P.out_stats && tid() == 0) P.out_stats[row_idx * P.n_chains + chain] = out;
    sc.draw_count += 1;
}

// returns false when the chain cannot go on in this launch (paused for the host, or failed)
template <int DPL, int W, class Dens>
NM_DEV bool chain_draw_lr(ChainCtx<DPL, W, Dens>& C, uint64_t chain) {
    const KParams& P = C.P;
    ChainScalars& sc = C.sc;
    const uint64_t row_idx = sc.draw_count - P.row_base;
    if (P.s.sampler == NM_SAMPLER_MCLMC) return chain_draw_mclmc(C, chain, row_idx);     // LowRankMclmcSettings
    AcceptCollector col;
    DrawResult R;
    Tile<DPL> x, gx, z, gz;
    uint64_t st = nuts_transition(C, col, R, z);
    nm_draw_stats out = {};
    out.draw = sc.draw_count; out.chain = P.chain_id_offset + chain;
    if (st != NM_CHAIN_OK) {
        sc.status = st;
        out.chain_status = st;
        if (P.out_stats && tid() == 0) P.out_stats[row_idx * P.n_chains + chain] = out;
        return false;
    }
    const size_t row = (size_t)(row_idx * P.n_chains + chain) * P.dim;
    if (R.diverging && (P.out_div_start || P.out_div_start_grad || P.out_div_end))
        emit_divergence_vectors(C, R.div_start_idx, row, R.has_div_end);
    if (R.chosen.slot == -1 && !sc.px_stale) {
        C.loadP(x, P_X); C.loadP(gx, P_GX);
        C.loadP(z, P_Z); C.loadP(gz, P_GZ);
    } else {
        if (R.chosen.slot == -1) C.loadP(z, P_Z);
        transform_to_x(C, z, x);                                  // the leapfrog's own operations => the same bits
        (void)C.dens.template eval<DPL, W>(x, gx, C.dim, C.red);
        transform_to_gz(C, gx, gz);
        const bool need_x = sc.tuning || sc.draw_count + 1 == P.draw_end || P.out_div_start || P.out_div_start_grad;
        if (need_x) { C.storeP_nt(x, P_X); C.storeP_nt(gx, P_GX); }
        sc.px_stale = need_x ? 0 : 1;
        C.storeP(z, P_Z); C.storeP(gz, P_GZ);
        sc.logp = R.chosen.logp;
    }
    const int64_t idx = R.chosen.idx;
    const bool is_good = R.diverging ? ((idx < 0 ? -idx : idx) > 4) : (idx != 0);
    write_row(C, P.out_positions, row, x, true);
    write_row(C, P.out_gradient, row, gx);
    write_row(C, P.out_tpos, row, z);
    write_row(C, P.out_tgrad, row, gz);
    double fd = 0.0;
#pragma unroll
    for (int k = 0; k < DPL; ++k) fd = fd + (z.a[k] + gz.a[k]) * (z.a[k] + gz.a[k]);
    fd = C.red.sum(fd);
    const double energy = R.chosen.ke - (R.chosen.logp + sc.logdet);
    out.depth = R.depth; out.maxdepth_reached = R.reached_maxdepth; out.diverging = R.diverging;
    out.index_in_trajectory = idx; out.transformation_index = sc.transform_id;
    out.logp = R.chosen.logp; out.energy = energy; out.energy_error = energy - R.e0;
    out.fisher_distance = fd;
    out.divergence_energy_error = (R.diverging && R.has_divergence_energy_error) ? R.divergence_energy_error : __builtin_nan("");
    sc.total_steps += col.count;
    const uint64_t ast = adapt_lr(C, chain, col, is_good, x, gx);
    if (sc.lr_pending == LR_WAIT_HOST) {                          // the rest of this draw happens in lr_resume
        sc.lr_row = row_idx;
        if (P.out_stats && tid() == 0) P.out_stats[row_idx * P.n_chains + chain] = out;
        return false;
    }
    finish_draw_lr(C, chain, out, ast, row_idx);
    return sc.status == NM_CHAIN_OK;
}

// the host answered: LowRankMassMatrixStrategy::adapt returned true (adapt/low_rank.rs:347-353); finish GlobalStrategy::adapt
// and the draw's statistics
template <int DPL, int W, class Dens>
NM_DEV void lr_resume(ChainCtx<DPL, W, Dens>& C, uint64_t chain) {
    const KParams& P = C.P;
    ChainScalars& sc = C.sc;
    if (sc.lr_upd_ok) lr_commit_update(C);
    Tile<DPL> x;
    C.loadP(x, P_X);
    sc.lr_pending = LR_IDLE;
    const uint64_t ast = adapt_tail(C, true, sc.lr_is_late != 0, x);
    nm_draw_stats out = {};
    if (P.out_stats) {
        const uint64_t* src = reinterpret_cast<const uint64_t*>(&P.out_stats[sc.lr_row * P.n_chains + chain]);
        uint64_t* dst = reinterpret_cast<uint64_t*>(&out);
        for (int i = 0; i < (int)(sizeof(nm_draw_stats) / 8); ++i) dst[i] = src[i];
    }
    finish_draw_lr(C, chain, out, ast, sc.lr_row);
}

// ---------------------------------------------------------------------------------------------
// kernels: one block = one wave; a wave strides over the chains
// ---------------------------------------------------------------------------------------------
// Minimum waves per SIMD the register allocator must leave room for (second __launch_bounds__ argument): the small
// tilings are latency-bound with few lanes busy, so more resident chains per CU beat a spill-free allocation there.
#ifndef NM_OCC_DPL2
#define NM_OCC_DPL2 2    // <= 256 VGPRs.  (Rounds 1-2: 4 waves / 128 VGPRs, K4 +17 %, K3 +6 %.  Round 3, with the batched merges: at 128 the
                         // leaf loop spills (205 VGPRs) and one funnel chain needs 2.15 us per leapfrog; at 256 nothing spills: 1.56 us, and
                         // 8192 chains run at 9.7e8 leapfrogs/s against 6.7e8 — profiles/r03g_*)
#endif
#ifndef NM_OCC_DPL4
#define NM_OCC_DPL4 2    // the DPL 4 kernels sit at 252..262 VGPRs: pin them below 256 (neutral for the elementwise densities,
#endif                  // x1.6 for the full-precision normal, whose GEMV needs the second wave to hide L2 latency)
#ifndef NM_OCC_DPL8_W2
#define NM_OCC_DPL8_W2 2   // (8 doubles, 2 waves): 1.62e11 on K2 against 1.81e11 for (16, 1) — used only on request
#endif
#ifndef NM_OCC_DPL8
#define NM_OCC_DPL8 2    // <= 256 VGPRs (72 spilled): dims 257..512 +18 % (two chains per SIMD hide each other's latency)
#endif
template <int DPL, int W>
constexpr int draw_min_waves() {
    if (W == 2 && DPL == 8) return NM_OCC_DPL8_W2;
    return W != 1 ? 1 : DPL == 2 ? NM_OCC_DPL2 : DPL == 4 ? NM_OCC_DPL4 : DPL == 8 ? NM_OCC_DPL8 : 1;
}

#if NM_CLUSTER_MODE
// Start of a cluster-mode kernel: the members of a chain tell each other which XCD they run on (one exchange with the safe
// release / acquire protocol); if it is the same one, the exchanges of the launch skip the cache maintenance.  Returns the
// link to start from (epoch 1).
template <int W>
NM_DEV ClusterLink cluster_start(const KParams& P, double* red_lds, unsigned cl_k, unsigned cl_member, uint64_t cl_id) {
    ClusterLink L;
    L.box = P.cl_box + cl_id * (unsigned long long)CL_BOX_WORDS * cl_k * RED_MAX_VALUES; L.cnt = P.cl_cnt + cl_id;
    L.k = (int)cl_k; L.member = (int)cl_member; L.epoch = 0ull; L.same_xcd = 0; L.dead = 0;
    Reducer<W> r;
    r.init(red_lds);
    r.cl = L;
    const double x = (double)xcc_id();
    double v[2] = {x, x * x};
    r.template cluster_combine<2>(v);
    r.cl.same_xcd = (!P.cl_general && (double)cl_k * v[1] == v[0] * v[0]) ? 1 : 0;    // sum of squares = square of the sum / k  <=>  all equal
    return r.cl;
}
#endif

template <int DPL, int W, class Dens>
__global__ __launch_bounds__(64 * W, (draw_min_waves<DPL, W>())) __attribute__((amdgpu_flat_work_group_size(64 * W, 64 * W))) void nuts_draw_kernel(const KParams P) {
    __shared__ BlockShared<DPL, W, Dens> sh;
    dm_init_lds();
#if NM_CLUSTER_MODE
    // block b = member (b / 8) % k of cluster (b / 8k) * 8 + b % 8: with the round-robin placement of blocks on the 8 XCDs
    // the members of a chain share an L2 (a performance matter only: the exchange is agent-scope)
    const unsigned cl_k = (unsigned)P.cl_k, cl_member = (blockIdx.x / 8u) % cl_k;
    const uint64_t cl_id = (uint64_t)(blockIdx.x / (8u * cl_k)) * 8u + blockIdx.x % 8u, n_clusters = gridDim.x / cl_k;
    ClusterLink cl_link = cluster_start<W>(P, sh.red, cl_k, cl_member, cl_id);
    for (uint64_t chain = cl_id; chain < P.n_chains; chain += n_clusters) {
        ChainCtx<DPL, W, Dens> C(P, sh.sc[W == 1 ? 0 : wave_id()]);
        C.red.cl = cl_link;
        const uint64_t sci = chain * cl_k + cl_member;
        ctx_begin(C, sh, sci, blockIdx.x);
        if (C.sc.status == NM_CHAIN_OK) {
            {
                Tile<DPL> t;
                C.loadP(t, P_SIG); C.store(t, C.lsig);
                C.loadP(t, P_MU); C.store(t, C.lmu);
            }
            for (uint64_t t = 0; t < P.n_draws; ++t) {
                chain_draw(C, chain, t);
                if (C.red.cl.dead) C.sc.status = NM_CHAIN_LOGP_FATAL;      // an exchange timed out: stop instead of hanging
                if (C.sc.status != NM_CHAIN_OK) break;
            }
        }
        ctx_end(C, sci);
        cl_link.epoch = C.red.cl.epoch; cl_link.dead = C.red.cl.dead;
        __syncthreads();
    }
    return;
#endif
    for (uint64_t chain = blockIdx.x; chain < P.n_chains; chain += gridDim.x) {
        ChainCtx<DPL, W, Dens> C(P, sh.sc[W == 1 ? 0 : wave_id()]);
        ctx_begin(C, sh, chain, blockIdx.x);
        if (C.sc.status == NM_CHAIN_OK) {
            {
                Tile<DPL> t;
                C.loadP(t, P_SIG); C.store(t, C.lsig);
                C.loadP(t, P_MU); C.store(t, C.lmu);
            }
#if NM_PROF
            C.prof_t = __builtin_amdgcn_s_memtime();
#endif
            if constexpr (lr_trait<Dens>::value) {
                if (C.sc.lr_pending == LR_SET_TRANSFORM) {              // nm_engine_set_transform: LowRankMassMatrix::update
                    if (C.sc.lr_upd_ok) lr_commit_update(C);
                    C.sc.lr_pending = LR_IDLE;
                }
                if (C.sc.lr_pending == LR_ANSWERED) lr_resume(C, chain);
                if (C.sc.lr_pending == LR_IDLE && C.sc.status == NM_CHAIN_OK)
                    while (C.sc.draw_count < P.draw_end) { if (!chain_draw_lr(C, chain)) break; }
            } else {
            for (uint64_t t = 0; t < P.n_draws; ++t) {
                chain_draw(C, chain, t);
                if (C.sc.status != NM_CHAIN_OK) break;
            }
            }
        }
        ctx_end(C, chain);
        __syncthreads();
    }
}

// NutsChain::set_position (reference src/chain.rs:137-149 -> GlobalStrategy::init adapt_strategy.rs:100-119)
template <int DPL, int W, class Dens>
__global__ __launch_bounds__(64 * W) __attribute__((amdgpu_flat_work_group_size(64 * W, 64 * W))) void nuts_init_kernel(const KParams P) {
    __shared__ BlockShared<DPL, W, Dens> sh;
    dm_init_lds();
#if NM_CLUSTER_MODE
    const unsigned cl_k = (unsigned)P.cl_k, cl_member = (blockIdx.x / 8u) % cl_k;
    const uint64_t cl_id = (uint64_t)(blockIdx.x / (8u * cl_k)) * 8u + blockIdx.x % 8u, n_clusters = gridDim.x / cl_k;
    ClusterLink cl_link = cluster_start<W>(P, sh.red, cl_k, cl_member, cl_id);
    for (uint64_t x0_chain = cl_id; x0_chain < P.n_chains; x0_chain += n_clusters) {
        if (P.init_mask && !P.init_mask[x0_chain]) continue;
        const uint64_t chain = x0_chain * cl_k + cl_member;                 // the sub-chain: this member's vectors and scalars
        ChainCtx<DPL, W, Dens> C(P, sh.sc[W == 1 ? 0 : wave_id()]);
        C.red.cl = cl_link;
        ctx_begin(C, sh, chain, blockIdx.x);
        ChainScalars& sc = C.sc;
#else
    for (uint64_t chain = blockIdx.x; chain < P.n_chains; chain += gridDim.x) {
        if (P.init_mask && !P.init_mask[chain]) continue;                  // per-chain Chain::set_position: the others keep their state
        const uint64_t x0_chain = chain;
        ChainCtx<DPL, W, Dens> C(P, sh.sc[W == 1 ? 0 : wave_id()]);
        ctx_begin(C, sh, chain, blockIdx.x);
        ChainScalars& sc = C.sc;
#endif
        // stepsize::Strategy::new (stepsize/adapt.rs:67-72) belongs to the chain's construction: only the first
        // set_position of a chain does it (mm_id is still -1); a retry after BadInitGrad keeps the adaptation state
        if (sc.mm_id < 0) {
            stepsize_adapt_reset(sc, P.s, P.s.initial_step);
            // TransformedHamiltonian::new(.., kind): NutsSettings::trajectory_kind, or MclmcChain's initial_kind (sampler.rs:433-438)
            sc.kin = P.s.sampler == NM_SAMPLER_MCLMC
                   ? (P.s.mclmc_trajectory_kind == NM_MCLMC_MICROCANONICAL ? NM_TRAJ_MICROCANONICAL : NM_TRAJ_EUCLIDEAN)
                   : P.s.trajectory_kind;
        }
        Tile<DPL> x, gx;
#pragma unroll
        for (int k = 0; k < DPL; ++k) {
            int d = C.elem(k);
            x.a[k] = d < C.dim ? P.x0[x0_chain * P.dim + (uint64_t)C.goff + d] : 0.0;
        }
        // init_state_untransformed (transformed_hamiltonian.rs:663-685)
        (void)C.dens.template eval<DPL, W>(x, gx, C.dim, C.red);
        bool ok = true;
#pragma unroll
        for (int k = 0; k < DPL; ++k) ok = ok && is_finite(gx.a[k]) && is_finite(x.a[k]);
        uint64_t status = NM_CHAIN_OK;
        if (dens_status(C) != 0) status = NM_CHAIN_LOGP_FATAL;      // `?` on the logp error (transformed_hamiltonian.rs:668)
        else if (!C.red.all(ok)) status = NM_CHAIN_BAD_INIT;
        if (status == NM_CHAIN_OK) {
            // DiagAdaptStrategy::init (adapt/diagonal.rs:209-231): seed the four estimators, mass matrix from |grad|
            if constexpr (lr_trait<Dens>::value) {
                // LowRankMassMatrixStrategy::init (adapt/low_rank.rs:299-317): add_draw(point), update_from_grad.  (A retry
                // after a failed set_position pushes again: the reference's deque keeps the failed attempt's point.)
                lr_push(C, chain, x, gx);
            } else {
                // DiagAdaptStrategy::init: add_sample on all four estimators.  The first call of a chain sets the means
                // (count 1); a retry after BadInitGrad accumulates, exactly as the reference's second init does.
                Tile<DPL> m_, v_;
                sc.cnt_fg += 1; sc.cnt_bg += 1;
                C.loadP(m_, E_DM); C.loadP(v_, E_DV); running_variance_add_regs(m_, v_, sc.cnt_fg, x); C.storeP(m_, E_DM); C.storeP(v_, E_DV);
                C.loadP(m_, E_GM); C.loadP(v_, E_GV); running_variance_add_regs(m_, v_, sc.cnt_fg, gx); C.storeP(m_, E_GM); C.storeP(v_, E_GV);
                C.loadP(m_, B_DM); C.loadP(v_, B_DV); running_variance_add_regs(m_, v_, sc.cnt_bg, x); C.storeP(m_, B_DM); C.storeP(v_, B_DV);
                C.loadP(m_, B_GM); C.loadP(v_, B_GV); running_variance_add_regs(m_, v_, sc.cnt_bg, gx); C.storeP(m_, B_GM); C.storeP(v_, B_GV);
            }
            mass_matrix_from_grad(C, x, gx);
            status = stepsize_init(C, x);                             // step_size.init (adapt_strategy.rs:117-118)
        }
        if (status == NM_CHAIN_OK) {
            Pt<DPL> st;                                               // hamiltonian.init_state (chain.rs:147)
            Tile<DPL> g2;
            if (!init_state(C, x, st, g2)) status = dens_status(C) != 0 ? NM_CHAIN_LOGP_FATAL : NM_CHAIN_BAD_INIT;
            else {
                C.storeP(x, P_X); C.storeP(g2, P_GX);
                C.storeP(st.z, P_Z); C.storeP(st.g, P_GZ);
                sc.logp = st.logp; sc.logdet = sc.mm_logdet; sc.transform_id = sc.mm_id;
                if constexpr (kin_trait<Dens>::value) {
                    if (P.s.sampler == NM_SAMPLER_MCLMC) {            // MclmcChain::set_position: initialize_trajectory(resample) (mclmc.rs:482-485)
                        sample_velocity(C, st.v);
                        (void)initial_kinetic(C, st.v);
                        C.storeP(st.v, P_V);
                    }
                }
            }
        }
#if NM_CLUSTER_MODE
        if (C.red.cl.dead) status = NM_CHAIN_LOGP_FATAL;
#endif
        sc.status = status;
        ctx_end(C, chain);
#if NM_CLUSTER_MODE
        cl_link.epoch = C.red.cl.epoch; cl_link.dead = C.red.cl.dead;
#endif
        __syncthreads();
    }
}

}  // namespace nm
