// A USER density for the module mechanism (include/nuts_amd.h, "User densities"): the diagonal normal again, written
// the way a user of the engine would write a density of their own.  Interface: init / set_lds / eval.
//   thread t of the 64*W threads of a chain owns elements d = 2*(m*64*W + t) + j  (m = 0..DPL/2-1, j = 0,1) = elem_index<W>(k)
//   eval returns logp (a block-wide sum through the Reducer) and fills the gradient tile; padding elements get 0.
#pragma once
#include "nuts_kernels.hpp"

struct MyDiagNormal {
    static constexpr bool kNeedsLdsVector = false;   // no block-visible vector needed
    const double* prec;
    double norm;
    NM_DEV void set_lds(double*) {}
    template <int W>
    NM_DEV void init(const double* params, int dim, nm::Reducer<W>& R) {
        prec = params;
        double acc = 0.0;
        for (int m = 0; m < (dim + 128 * W - 1) / (128 * W); ++m)
            for (int j = 0; j < 2; ++j) {
                const int d = 2 * (m * 64 * W + nm::tid()) + j;
                acc = acc + (d < dim ? nm::dlog(params[d < dim ? d : 0]) : 0.0);
            }
        const double log_det_p = R.sum(acc);
        norm = -0.5 * ((double)dim * nm::ulog(6.283185307179586) - log_det_p);
    }
    // Chains wider than one block (dim > 4096; modules built with -DNM_CLUSTER_MODE=1): this block holds elements
    // [goff, goff + dim) of a chain of gdim elements; `eval` then sees that slice, and every R.sum spans the whole chain.
    template <int W>
    NM_DEV void init_slice(const double* params, int dim, int gdim, int goff, nm::Reducer<W>& R) {
        prec = params + goff;
        double acc = 0.0;
        for (int m = 0; m < (dim + 128 * W - 1) / (128 * W); ++m)
            for (int j = 0; j < 2; ++j) {
                const int d = 2 * (m * 64 * W + nm::tid()) + j;
                acc = acc + (d < dim ? nm::dlog(prec[d < dim ? d : 0]) : 0.0);
            }
        const double log_det_p = R.sum(acc);
        norm = -0.5 * ((double)gdim * nm::ulog(6.283185307179586) - log_det_p);
    }
    template <int DPL, int W>
    NM_DEV double eval(const nm::Tile<DPL>& x, nm::Tile<DPL>& gx, int dim, nm::Reducer<W>& R) const {
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < DPL; ++k) {
            const int d = nm::elem_index<W>(k);
            const bool valid = d < dim;
            const double p = valid ? prec[d] : 0.0;
            const double px = p * x.a[k];
            gx.a[k] = valid ? -px : 0.0;
            acc = acc + (valid ? x.a[k] * px : 0.0);
        }
        return -0.5 * R.sum(acc) + norm;
    }
};

// The same density in GROUP form (optional): for many small chains the engine draws several chains per wavefront, each
// on L::kLanes lanes; this lane holds elements 2 L::lane(), 2 L::lane() + 1.  Same operations, so the same results.
template <class L>
struct MyDiagNormalGroup {
    const double* prec;
    double norm;
    NM_DEV void set_lds(double*) {}
    NM_DEV void init(const double* params, int dim) {
        prec = params;
        double acc = 0.0;
        for (int j = 0; j < 2; ++j) {
            const int d = 2 * L::lane() + j;
            acc = acc + (d < dim ? nm::dlog(params[d < dim ? d : 0]) : 0.0);
        }
        const double log_det_p = L::sum(acc);
        norm = -0.5 * ((double)dim * nm::dlog(6.283185307179586) - log_det_p);
    }
    NM_DEV double eval(const double (&x)[2], double (&gx)[2], int dim) const {
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int d = 2 * L::lane() + k;
            const bool valid = d < dim;
            const double p = valid ? prec[d] : 0.0;
            const double px = p * x[k];
            gx[k] = valid ? -px : 0.0;
            acc = acc + (valid ? x[k] * px : 0.0);
        }
        return -0.5 * L::sum(acc) + norm;
    }
};

// The same density in LANE form (optional): for very many chains with dim <= 16 the engine draws one chain per LANE — the whole
// vector (2 NP elements, zero beyond dim) sits in this lane's registers and nothing crosses lanes.  Sums over dim go through
// nm::lane::pair_tree (pair partials in element order, then a balanced tree: the order every other kernel uses for <= 16 elements).
#ifdef NM_MODULE_LANE_DENSITY
template <int NP>
struct MyDiagNormalLane {
    static constexpr int E = 2 * NP;
    const double* prec;
    double norm;
    NM_DEV void init(const double* params, int dim) {
        prec = params;
        double p[NP];
#pragma unroll
        for (int l = 0; l < NP; ++l) {
            double acc = 0.0;
            for (int j = 0; j < 2; ++j) {
                const int d = 2 * l + j;
                acc = acc + (d < dim ? nm::lane::llog(params[d < dim ? d : 0]) : 0.0);
            }
            p[l] = acc;
        }
        const double log_det_p = nm::lane::pair_tree<NP>(p);
        norm = -0.5 * ((double)dim * nm::lane::llog(6.283185307179586) - log_det_p);
    }
    NM_DEV double eval(const double (&x)[E], double (&gx)[E], int dim) const {
        double p[NP];
#pragma unroll
        for (int l = 0; l < NP; ++l) {
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int d = 2 * l + k;
                const bool valid = d < dim;
                const double pr = valid ? prec[d] : 0.0;
                const double px = pr * x[d];
                gx[d] = valid ? -px : 0.0;
                acc = acc + (valid ? x[d] * px : 0.0);
            }
            p[l] = acc;
        }
        return -0.5 * nm::lane::pair_tree<NP>(p) + norm;
    }
};
#endif
