// A USER density that can FAIL (include/nuts_amd.h, "User densities": `kCanFail`): N(mu, I) on a box — leaving it on the first
// coordinate is a RECOVERABLE error (the reference's `LogpError::is_recoverable() == true`: the leapfrog becomes a divergence
// without an energy error, src/dynamics/transformed_hamiltonian.rs:562-578), leaving it on the second coordinate is not
// (`NutsError::LogpFailure`: the chain stops, src/nuts.rs:231).  params = {mu, wall_recoverable, wall_fatal}.
#pragma once
#include "nuts_kernels.hpp"

struct MyWalledNormal {
    static constexpr bool kNeedsLdsVector = false;
    static constexpr bool kCanFail = true;           // the kernels read `status` after every evaluation
    int status = 0;                                  // 0 ok, 1 recoverable error, 2 unrecoverable error (the same in every thread)
    double mu, wall1, wall2;
    NM_DEV void set_lds(double*) {}
    NM_DEV void bind(const nm::KParams&, uint64_t) {}     // (per-chain resources of a fallible density; none here)
    template <int W>
    NM_DEV void init(const double* params, int, nm::Reducer<W>&) { mu = params[0]; wall1 = params[1]; wall2 = params[2]; }
    template <int DPL, int W>
    NM_DEV double eval(const nm::Tile<DPL>& x, nm::Tile<DPL>& gx, int dim, nm::Reducer<W>& R) {
        double acc = 0.0, out1 = 0.0, out2 = 0.0;
#pragma unroll
        for (int k = 0; k < DPL; ++k) {
            const int d = nm::elem_index<W>(k);
            const bool valid = d < dim;
            const double diff = x.a[k] - mu;
            const double term = -0.5 * diff * diff;
            gx.a[k] = valid ? -diff : 0.0;
            acc = acc + (valid ? term : 0.0);
            out1 = out1 + ((d == 0 && x.a[k] > wall1) ? 1.0 : 0.0);      // the owner of a coordinate sees it; the block sum tells everyone
            out2 = out2 + ((d == 1 && x.a[k] > wall2) ? 1.0 : 0.0);
        }
        double v[3] = {acc, out1, out2};
        R.sum_n(v);                                   // one block reduction for the three
        status = v[2] != 0.0 ? 2 : (v[1] != 0.0 ? 1 : 0);
        return v[0];
    }
};
