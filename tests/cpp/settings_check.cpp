// The other settings types of the C++ host side (LowRankNutsSettings, DiagMclmcSettings, LowRankMclmcSettings): their Default impls
// (src/sampler.rs:342-384, :636-642) through to_c(), and that a ChainBatch can be built from each (fails loudly without a device).
#include "nuts_amd.hpp"
int main() {
    nuts_amd::LowRankNutsSettings a; nuts_amd::LowRankMclmcSettings b; nuts_amd::DiagMclmcSettings c; nuts_amd::DiagNutsSettings d;
    nm_settings sa = a.to_c(), sb = b.to_c(), sc = c.to_c(), sd = d.to_c();
    if (sa.adaptation != NM_ADAPT_LOW_RANK || sa.num_tune != 800 || sa.mass_matrix_update_freq != 20) return 1;
    if (sb.sampler != NM_SAMPLER_MCLMC || sb.adaptation != NM_ADAPT_LOW_RANK || sb.early_mass_matrix_switch_freq != 20 || sb.num_tune != 800) return 2;
    if (sc.sampler != NM_SAMPLER_MCLMC || sc.step_size_method != NM_STEP_FIXED || sd.sampler != NM_SAMPLER_NUTS) return 3;
    try { nuts_amd::ChainBatch cb(a, nuts_amd::LogpSpec::iid_normal(8, 0.0), 2); } catch (const nuts_amd::NutsError&) {}
    try { nuts_amd::ChainBatch cb(b, nuts_amd::LogpSpec::iid_normal(8, 0.0), 2); } catch (const nuts_amd::NutsError&) {}
    return 0;
}
