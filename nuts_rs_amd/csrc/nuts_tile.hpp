// nuts_tile.hpp — the draw kernel for chains that SHARE their matrices: 16 chains per block, dense products on the
// matrix cores (v_mfma_f64_16x16x4_f64).  BASELINE config 5: N(0, Sigma) with a full Sigma at dim 256 x 4096 chains,
// sampled through a transformation of rank up to dim that all chains share.
//
// With one chain per block (nuts_kernels.hpp) a product with a dim x dim matrix is a GEMV that re-reads the matrix for
// every chain: 1.5 MiB per leapfrog at dim 256, HBM-bound when the matrices are per chain (2.8e6 leapfrogs/s measured)
// and L2-bound when they are shared.  Here a block is 16 wavefronts = 16 chains.  Every wavefront runs the SAME chain
// code as the one-chain kernel (the whole NUTS transition of nuts_kernels.hpp, wave-uniform control flow, its own
// ragged tree), but the three products of a leapfrog —
//     S = U' z ; x = z + U ((lambda^1/2 - 1) . S)        (LowRankMassMatrix, reference src/math/cpu_math.rs:332-425)
//     y = P x                                             (the density's gradient)
//     S = U' t ; g_z = t + U ((lambda^1/2 - 1) . S)
// — are RENDEZVOUS: each chain writes its vector as one column of a dim x 16 LDS tile, the block meets at a barrier,
// the 16 wavefronts split the (rows x 16) output into 16-row stripes and accumulate them with MFMA (A operand: the
// shared matrix, pre-packed in operand order, L2-resident; B operand: the column tile in LDS), meet again, and every
// chain reads its column back.  One matrix element now serves 16 chains.
//
// Ragged trees: a chain that finished its draw keeps attending the rendezvous with an empty column (it still computes
// its stripe for the others) until all 16 are done; then the block starts the next draw together.  All chains issue
// whole (apply, density, apply) cadences only, so the block's phase is common to all waves by construction.
//
// Numerics: v_mfma_f64_16x16x4_f64 is an fma chain over k ascending, bit for bit (tools/probes/mfma_f64_probe.hip), so a
// stripe's entry is the SEQUENTIAL fma dot product over the inner index: y = P x has the bits of the one-chain MvnPrec
// density (already sequential in j), and the two products of the transformation are sequential over d resp. k — the
// oracle's `lr_seq_dots` mode (the one-chain kernels sum U'v in lane order instead; both are documented orders of a
// product whose order the reference leaves to faer).
#pragma once
#define NM_TILE_MODE 1
#ifndef NM_LDS_L1
#define NM_LDS_L1 0          // the per-chain LDS of 16 resident chains must leave room for the column tiles
#endif
#include "nuts_kernels.hpp"

namespace nm {
namespace tile {

#ifndef NM_TILE_CHAINS
#define NM_TILE_CHAINS 16    // chains = wavefronts per block: 16 (all 16 columns of the f64 MFMA carry a chain) or 8 (half of them; tuning builds)
#endif
#ifndef NM_TILE_OCC
#define NM_TILE_OCC 1        // blocks per compute unit the register allocation leaves room for
#endif
constexpr int TC = NM_TILE_CHAINS;  // chains = wavefronts per block
static_assert(TC == 16 || TC == 8, "a column tile holds 16 or 8 chains");
constexpr int TD = 256;             // rows of a column tile: dim and rank <= 256
typedef double v4d __attribute__((ext_vector_type(4)));

// The shared matrices in MFMA operand order.  For a product out[R x 16] = M[R x K] B[K x 16], stripe s (rows 16 s ..),
// k-pair q (inner indices 8 q .. 8 q + 7): packed[(s * kpairs + q) * 64 + lane] = { M[16 s + (lane & 15)][8 q + (lane >> 4)],
// M[16 s + (lane & 15)][8 q + 4 + (lane >> 4)] } — two A operands per 16-byte load, 1 KiB per wave instruction.
struct TileMats {
    const double* ut;        // M = U'  (rank x dim):  M[k][d] = vecs[k][d]
    const double* u;         // M = U   (dim x rank):  M[d][k] = vecs[k][d]
    const double* p;         // M[d][j] = P[j][d]      (the one-chain density's column order)
    int dim, rank;
    int dim_kp, rank_kp;     // k-pairs of inner length dim / rank (padded with zeros)
    int dim_st, rank_st;     // 16-row stripes of dim / rank outputs
};

struct alignas(16) TileShared {
    double zin[TD * TC];     // the chains' input columns; the second product of an apply accumulates in place
    double sbuf[TD * TC];    // S = U' z (first product of an apply) / y = P x
    double sig[TD], mu[TD];  // the shared DiagMassMatrix part (tile order of a one-wave chain)
    double scale[2][TD];     // lambda^(1/2) - 1, lambda^(-1/2) - 1
    int which[TC];           // per column: the scale its current apply uses
    int done[2];             // monotone counts of finished (wave, draw) pairs, one per draw parity (see tile_all_done)
};
// element (row, chain column) of a column tile: 16 doubles per row, columns xor-swizzled so that a wavefront writing its
// own column (rows 2 t, 2 t + 1 per lane) spreads over the banks, while the B-operand read of a row stays a permutation
NM_DEV int taddr(int row, int c) { return row * TC + ((c ^ (row >> 1)) & (TC - 1)); }     // (c may be an MFMA column >= TC: it mirrors column c - TC)

// The end of a draw is a BLOCK-UNIFORM decision: every wave reads the counter of the current draw right after the same barrier and
// must see the same value.  With one counter a wave whose chain is absent or failed could leave draw d and count itself into draw
// d + 1 before a slower wave had read the counter for draw d — that wave then saw 16 (d + 1) + 1, stayed in its waiting loop, and
// the barrier phases of the block parted for good.  Draw d therefore counts in done[d & 1], whose target is
// 16 (d / 2 + 1): a wave of draw d + 1 touches the OTHER counter, and no wave reaches draw d + 2 before every wave has left draw
// d + 1, i.e. long after all reads for draw d.
NM_DEV void tile_count_done(int* done, int draws_done) { if (lane_id() == 0) atomicAdd(&done[draws_done & 1], 1); }
NM_DEV bool tile_all_done(const int* done, int draws_done) {
    return __builtin_amdgcn_readfirstlane(*(volatile const int*)&done[draws_done & 1]) == (NM_TILE_CHAINS) * ((draws_done >> 1) + 1);
}

NM_DEV void tile_barrier() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); }

template <int DPL>
NM_DEV void put_col(double* buf, int c, const Tile<DPL>& t) {
#pragma unroll
    for (int k = 0; k < DPL; ++k) buf[taddr(elem_index<1>(k), c)] = t.a[k];
}
template <int DPL>
NM_DEV void get_col(const double* buf, int c, Tile<DPL>& t) {
#pragma unroll
    for (int k = 0; k < DPL; ++k) t.a[k] = buf[taddr(elem_index<1>(k), c)];
}

// one 16-row stripe: acc += M[stripe rows][:] B, inner index ascending (an fma chain per output entry)
NM_DEV v4d gemm_stripe(const double* packed, int s, int kpairs, const double* b, v4d acc) {
    const int l = lane_id(), kk = l >> 4, c = l & 15;
    const double2* ap = reinterpret_cast<const double2*>(packed) + (size_t)s * (size_t)kpairs * 64 + l;
#pragma unroll 4
    for (int q = 0; q < kpairs; ++q) {
        const double2 a = ap[(size_t)q * 64];
        const int r0 = 8 * q + kk;
        const double b0 = b[taddr(r0, c)], b1 = b[taddr(r0 + 4, c)];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, b0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a.y, b1, acc, 0, 0, 0);
    }
    return acc;
}
// Round 3 measured a hand-pipelined form of this loop (A operands through a buffer descriptor in groups of four k-pairs, the next
// group's loads pinned in front of the current group's MFMAs with scheduling barriers, B operands by ds_read from eight
// precomputed per-lane addresses; the loop above compiles to flat loads and a full `s_waitcnt vmcnt(0) lgkmcnt(0)` in front of
// every pair of MFMAs).  The products' share of a wavefront's cycles fell from 63 % to 55 % (NM_TILE_PROF), but the 32 extra
// registers of the operand ring cost the chain code between the rounds as much: shared low-rank K5 3.80e7 -> 3.85e7 leapfrogs/s,
// DiagNutsSettings K5 1.25e8 -> 1.10e8.  With four wavefronts per SIMD taking turns on the matrix pipe the load latency of one
// is already covered by the others; rejected (profiles/r03h_k5_tile_phases.json, r03j_*).
// C/D layout of the f64 MFMA: lane (g = l >> 4, c = l & 15) holds rows 16 s + g + 4 r, r = 0..3, of column c
NM_DEV void store_stripe(double* buf, int s, v4d acc) {
    const int l = lane_id(), g = l >> 4, c = l & 15;
#pragma unroll
    for (int r = 0; r < 4; ++r) buf[taddr(16 * s + g + 4 * r, c)] = acc[r];
}
// the same for S = U' z, scaled on the way out: S'[k][c] = S[k][c] (lambda_k^w - 1) with w the scale column c asked for —
// the oracle's `sc[k] *= vals[k] - 1` (apply_lowrank_transform, cpu_math.rs:360-364)
NM_DEV void store_stripe_scaled(double* buf, int s, v4d acc, const TileShared& T) {
    const int l = lane_id(), g = l >> 4, c = l & 15;
    const double* scale = T.scale[T.which[c & (TC - 1)] & 1];
#pragma unroll
    for (int r = 0; r < 4; ++r) { const int k = 16 * s + g + 4 * r; buf[taddr(k, c)] = acc[r] * scale[k]; }
}
NM_DEV v4d load_stripe(const double* buf, int s) {
    const int l = lane_id(), g = l >> 4, c = l & 15;
    v4d acc;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = buf[taddr(16 * s + g + 4 * r, c)];
    return acc;
}

// Development (-DNM_TILE_PROF=1, tools/prof_tile.py): where does a wavefront of the block spend its cycles?  Per wavefront: cycles in
// its own chain code between two rendezvous, waiting at the rendezvous' barriers, and in the matrix-core products; summed into
// P.prof[16 ..] at the end of the kernel (all waves of all blocks; [19] counts rounds of wave 0).
#ifndef NM_TILE_PROF
#define NM_TILE_PROF 0
#endif
// what a chain's wavefront knows about its block
struct TileRef {
    TileShared* T;
    TileMats M;
    int c;                   // my column = my wavefront
    int draws_done;          // draws this block has completed (exit test of the waiting loop)
#if NM_TILE_PROF
    unsigned long long t_last = 0, acc_chain = 0, acc_wait = 0, acc_mma = 0, acc_idle_chain = 0, rounds = 0;
    int idle = 0;            // 1 while the wave attends with an empty column (its chain has finished the draw)
#endif
};
#if NM_TILE_PROF
#define NM_TP_ENTER(X) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); if ((X).t_last) { if ((X).idle) (X).acc_idle_chain += n_ - (X).t_last; else (X).acc_chain += n_ - (X).t_last; } (X).t_last = n_; (X).rounds += 1; }
#define NM_TP_ADD(X, acc) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); (X).acc += n_ - (X).t_last; (X).t_last = n_; }
#else
#define NM_TP_ENTER(X)
#define NM_TP_ADD(X, acc)
#endif

// v <- v + U ((lambda^which - 1) . (U' v)) for my chain (v == nullptr: an empty column).  With `may_exit` the call
// returns true — for every wavefront of the block alike — when all 16 chains have finished the current draw.
template <int DPL>
NM_DEV bool apply_round(TileRef& X, int which, Tile<DPL>* v, bool may_exit) {
    TileShared& T = *X.T;
    const int w = X.c;
    NM_TP_ENTER(X)
    if (v) {
        put_col(T.zin, w, *v);
        if (lane_id() == 0) T.which[w] = which;
    }
    tile_barrier();
    NM_TP_ADD(X, acc_wait)
    if (may_exit) {                                   // only ever evaluated true when all 16 are in their waiting loops
        if (tile_all_done(T.done, X.draws_done)) return true;
    }
    for (int s = w; s < X.M.rank_st; s += TC) {
        v4d acc = {0.0, 0.0, 0.0, 0.0};
        acc = gemm_stripe(X.M.ut, s, X.M.dim_kp, T.zin, acc);
        store_stripe_scaled(T.sbuf, s, acc, T);
    }
    NM_TP_ADD(X, acc_mma)
    tile_barrier();
    NM_TP_ADD(X, acc_wait)
    for (int s = w; s < X.M.dim_st; s += TC) {
        v4d acc = load_stripe(T.zin, s);
        acc = gemm_stripe(X.M.u, s, X.M.rank_kp, T.sbuf, acc);
        store_stripe(T.zin, s, acc);
    }
    NM_TP_ADD(X, acc_mma)
    tile_barrier();
    NM_TP_ADD(X, acc_wait)
    if (v) get_col(T.zin, w, *v);
    return false;
}
// y = P x for my chain (x == nullptr: an empty column)
template <int DPL>
NM_DEV void density_round(TileRef& X, const Tile<DPL>* x, Tile<DPL>* y) {
    TileShared& T = *X.T;
    const int w = X.c;
    NM_TP_ENTER(X)
    if (x) put_col(T.zin, w, *x);
    tile_barrier();
    NM_TP_ADD(X, acc_wait)
    for (int s = w; s < X.M.dim_st; s += TC) {
        v4d acc = {0.0, 0.0, 0.0, 0.0};
        acc = gemm_stripe(X.M.p, s, X.M.dim_kp, T.zin, acc);
        store_stripe(T.sbuf, s, acc);
    }
    NM_TP_ADD(X, acc_mma)
    tile_barrier();
    NM_TP_ADD(X, acc_wait)
    if (y) get_col(T.sbuf, w, *y);
}

// y = P x as the ONLY rendezvous of the block (the per-chain diagonal transformation, nuts_tile_diag_kernel): with `may_exit`
// the call returns true — for every wavefront alike — when all 16 chains have finished the current draw.
template <int DPL>
NM_DEV bool density_only_round(TileRef& X, const Tile<DPL>* x, Tile<DPL>* y, bool may_exit) {
    TileShared& T = *X.T;
    const int w = X.c;
    NM_TP_ENTER(X)
    if (x) put_col(T.zin, w, *x);
    tile_barrier();
    NM_TP_ADD(X, acc_wait)
    if (may_exit) {                                   // only ever evaluated true when all 16 are in their waiting loops
        if (tile_all_done(T.done, X.draws_done)) return true;
    }
    for (int s = w; s < X.M.dim_st; s += TC) {
        v4d acc = {0.0, 0.0, 0.0, 0.0};
        acc = gemm_stripe(X.M.p, s, X.M.dim_kp, T.zin, acc);
        store_stripe(T.sbuf, s, acc);
    }
    NM_TP_ADD(X, acc_mma)
    tile_barrier();
    NM_TP_ADD(X, acc_wait)
    if (y) get_col(T.sbuf, w, *y);
    return false;
}

// The same density for chains that keep their OWN diagonal transformation (DiagNutsSettings: every chain adapts alone): the
// precision matrix belongs to the density, so all chains share it whatever their mass matrices are.  Bit for bit the one-chain
// MvnPrec (its y_d is the same sequential fma chain over j).
struct TileMvnDiag {
    static constexpr bool kNeedsLdsVector = false;
    TileRef* X = nullptr;
    template <int W>
    NM_DEV void init(const double*, int, Reducer<W>&) {}
    NM_DEV void set_lds(double*) {}
    template <int DPL, int W>
    NM_DEV double eval(const Tile<DPL>& x, Tile<DPL>& gx, int dim, Reducer<W>& R) const {
        Tile<DPL> xs, y;
#pragma unroll
        for (int k = 0; k < DPL; ++k) xs.a[k] = elem_index<W>(k) < dim ? x.a[k] : 0.0;
        (void)density_only_round(*X, &xs, &y, false);
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < DPL; ++k) {
            const bool valid = elem_index<W>(k) < dim;
            gx.a[k] = valid ? -y.a[k] : 0.0;
            acc = acc + (valid ? x.a[k] * y.a[k] : 0.0);
        }
        return -0.5 * R.sum(acc);
    }
};

// The full-precision normal (MvnPrec of nuts_kernels.hpp: logp = -x'Px/2, grad = -Px, y_d = sum_j fma(P[j][d], x_j, .),
// j ascending) with the product on the matrix cores; the low-rank products of LrWrap go through the same block.
struct TileMvnPrec {
    static constexpr bool kNeedsLdsVector = false;
    static constexpr bool kTile = true;
    TileRef* X = nullptr;
    template <int W>
    NM_DEV void init(const double*, int, Reducer<W>&) {}
    NM_DEV void set_lds(double*) {}
    template <int DPL>
    NM_DEV void tile_apply(int which, Tile<DPL>& v) { (void)apply_round(*X, which, &v, false); }
    NM_DEV void tile_skip_density() { density_round<4>(*X, (const Tile<4>*)nullptr, (Tile<4>*)nullptr); }
    template <int DPL, int W>
    NM_DEV double eval(const Tile<DPL>& x, Tile<DPL>& gx, int dim, Reducer<W>& R) const {
        Tile<DPL> xs, y;
#pragma unroll
        for (int k = 0; k < DPL; ++k) xs.a[k] = elem_index<W>(k) < dim ? x.a[k] : 0.0;
        density_round(*X, &xs, &y);
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < DPL; ++k) {
            const bool valid = elem_index<W>(k) < dim;
            gx.a[k] = valid ? -y.a[k] : 0.0;
            acc = acc + (valid ? x.a[k] * y.a[k] : 0.0);
        }
        return -0.5 * R.sum(acc);
    }
};

// ---------------------------------------------------------------------------------------------
// the kernel: block b owns chain tiles b, b + grid, ...; wavefront w of a tile owns chain 16 tile + w
// ---------------------------------------------------------------------------------------------
template <int DPL, class Dens>
__global__ __launch_bounds__(64 * TC, NM_TILE_OCC * TC / 4) void nuts_tile_draw_kernel(const KParams P, const TileMats M) {
    __shared__ BlockShared<DPL, 1, Dens> sh[TC];
    __shared__ TileShared T;
    dm_init_lds();
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (threadIdx.x == 0) { T.done[0] = 0; T.done[1] = 0; }
    int draws_done = 0;
    const uint64_t n_tiles = (P.n_chains + TC - 1) / TC;
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        __syncthreads();
        const uint64_t chain = tile * TC + (uint64_t)w;
        const bool valid = chain < P.n_chains;
        {   // the tile's shared eigenvalue scales (all chains carry the same transformation: the first chain's copy)
            const double* lv = P.lrval + (size_t)(tile * TC) * 2 * P.lr_rmax;
            for (int i = (int)threadIdx.x; i < 2 * TD; i += 64 * TC) {
                const int which = i / TD, k = i % TD;
                T.scale[which][k] = k < M.rank ? lv[(size_t)which * P.lr_rmax + k] - 1.0 : 0.0;
            }
            const double* pv = P.pvec + (size_t)(tile * TC) * NUM_PSLOT * P.dpad;
            for (int i = (int)threadIdx.x; i < TD; i += 64 * TC) {
                T.sig[i] = i < (int)P.dpad ? pv[(size_t)P_SIG * P.dpad + i] : 0.0;
                T.mu[i] = i < (int)P.dpad ? pv[(size_t)P_MU * P.dpad + i] : 0.0;
            }
            for (int i = (int)threadIdx.x; i < TD * TC; i += 64 * TC) { T.zin[i] = 0.0; T.sbuf[i] = 0.0; }
            if (threadIdx.x < TC) T.which[threadIdx.x] = 0;       // columns of absent chains stay empty, with a valid scale index
        }
        __syncthreads();
        ChainCtx<DPL, 1, Dens> C(P, sh[w].sc[0]);
        TileRef X{&T, M, w, draws_done};
        if (valid) {
            ctx_begin(C, sh[w], chain, (uint64_t)blockIdx.x * TC + (uint64_t)w);
            C.lsig = T.sig; C.lmu = T.mu;
            C.dens.X = &X;
        }
        bool ok = valid && C.sc.status == NM_CHAIN_OK;
        if (ok && C.sc.lr_pending == LR_SET_TRANSFORM) {        // nm_engine_set_transform: LowRankMassMatrix::update
            if (C.sc.lr_upd_ok) {
                Tile<DPL> t;
                C.loadP(t, P_ISIG);
                const double diag_logdet = sum_ln_tile(C, t);
                C.sc.mm_logdet = C.sc.lr_upd_logdet + diag_logdet;
                C.sc.mm_id += 1; C.sc.lr_has_inner = 1; C.sc.lr_rank = C.sc.lr_upd_rank;
            }
            C.sc.lr_pending = LR_IDLE;
        }
        ok = ok && C.sc.lr_pending == LR_IDLE && C.sc.lr_has_inner != 0;
        for (uint64_t t = P.row_base; t < P.draw_end; ++t) {
            X.draws_done = draws_done;
            if (ok && C.sc.draw_count == t) ok = chain_draw_lr(C, chain) && C.sc.status == NM_CHAIN_OK;
            tile_count_done(T.done, draws_done);
#if NM_TILE_PROF
            NM_TP_ENTER(X) X.rounds -= 1; X.idle = 1;
#endif
            // attend the block's rendezvous with an empty column until every chain of the tile has finished this draw
            for (;;) {
                if (apply_round<DPL>(X, 0, (Tile<DPL>*)nullptr, true)) break;
                density_round<DPL>(X, (const Tile<DPL>*)nullptr, (Tile<DPL>*)nullptr);
                (void)apply_round<DPL>(X, 0, (Tile<DPL>*)nullptr, false);
            }
#if NM_TILE_PROF
            X.idle = 0;
#endif
            draws_done += 1;
        }
#if NM_TILE_PROF
        if (lane_id() == 0) {
            atomicAdd(&P.prof[16], X.acc_chain); atomicAdd(&P.prof[17], X.acc_wait); atomicAdd(&P.prof[18], X.acc_mma);
            atomicAdd(&P.prof[20], X.acc_idle_chain); if (w == 0) atomicAdd(&P.prof[19], X.rounds);
        }
#endif
        if (valid) ctx_end(C, chain);
    }
}

// The draw kernel for DiagNutsSettings on the full-precision normal: the same 16 chains per block, each with its own adapting
// diagonal transformation (sigma, mu are read from the chain's persistent slots: 16 x 4 KiB would not fit beside the column
// tiles), ONE rendezvous per density evaluation.  Whatever evaluates the density — leapfrogs, the step-size search inside the
// adaptation, the recomputation of the chosen point — is a round; a chain that has finished its draw attends with an idle column.
template <int DPL, class Dens>
__global__ __launch_bounds__(64 * TC, NM_TILE_OCC * TC / 4) void nuts_tile_diag_kernel(const KParams P, const TileMats M) {
    __shared__ BlockShared<DPL, 1, Dens> sh[TC];
    __shared__ TileShared T;
    dm_init_lds();
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (threadIdx.x == 0) { T.done[0] = 0; T.done[1] = 0; }
    int draws_done = 0;
    const uint64_t n_tiles = (P.n_chains + TC - 1) / TC;
    for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        __syncthreads();
        for (int i = (int)threadIdx.x; i < TD * TC; i += 64 * TC) { T.zin[i] = 0.0; T.sbuf[i] = 0.0; }
        __syncthreads();
        const uint64_t chain = tile * TC + (uint64_t)w;
        const bool valid = chain < P.n_chains;
        ChainCtx<DPL, 1, Dens> C(P, sh[w].sc[0]);
        TileRef X{&T, M, w, draws_done};
        if (valid) {
            ctx_begin(C, sh[w], chain, (uint64_t)blockIdx.x * TC + (uint64_t)w);
            C.lsig = C.slot(P_SIG); C.lmu = C.slot(P_MU);      // the chain's own mass matrix, straight from its slots
            C.dens.X = &X;
        }
        bool ok = valid && C.sc.status == NM_CHAIN_OK;
        for (uint64_t t = 0; t < P.n_draws; ++t) {
            X.draws_done = draws_done;
            if (ok) { chain_draw(C, chain, t); ok = C.sc.status == NM_CHAIN_OK; }
            tile_count_done(T.done, draws_done);
#if NM_TILE_PROF
            NM_TP_ENTER(X) X.rounds -= 1; X.idle = 1;
#endif
            while (!density_only_round<DPL>(X, (const Tile<DPL>*)nullptr, (Tile<DPL>*)nullptr, true)) {}
#if NM_TILE_PROF
            X.idle = 0;
#endif
            draws_done += 1;
        }
#if NM_TILE_PROF
        if (lane_id() == 0) {
            atomicAdd(&P.prof[16], X.acc_chain); atomicAdd(&P.prof[17], X.acc_wait); atomicAdd(&P.prof[18], X.acc_mma);
            atomicAdd(&P.prof[20], X.acc_idle_chain); if (w == 0) atomicAdd(&P.prof[19], X.rounds);
        }
#endif
        if (valid) ctx_end(C, chain);
    }
}

}  // namespace tile
}  // namespace nm
