// The reference's README / src/lib.rs example (10-dim N(3,1), DiagNutsSettings::default(), x0 = zeros) written against
// the C++ host API, plus the Sampler control plane.  Usage: host_api_demo <out.bin>
//   no GPU : prints "NM_ERR_NO_DEVICE" and exits 3 (the engine has no CPU fallback)
//   GPU    : writes [1400][4][10] doubles of draws (chains seeded like the reference's Sampler) and exits 0
#include <cstdio>
#include <chrono>
#include <cstring>
#include <thread>
#include <vector>

#include "nuts_amd.hpp"

int main(int argc, char** argv) {
    using namespace nuts_amd;
    DiagNutsSettings settings;                 // == DiagNutsSettings::default()
    settings.num_chains = 4;
    settings.seed = 0;
    const LogpSpec logp = LogpSpec::iid_normal(10, 3.0);
    const uint64_t total = settings.num_tune + settings.num_draws;
    try {
        ChainBatch sampler(settings, logp, 4);
        sampler.set_position(std::vector<double>(4 * 10, 0.0));
        std::vector<double> trace;
        uint64_t steps = 0;
        for (uint64_t i = 0; i < total; ++i) {
            auto [draw, info] = sampler.draw();        // (positions of all chains, one Progress per chain)
            trace.insert(trace.end(), draw.begin(), draw.end());
            for (const Progress& p : info) steps += p.num_steps;
        }
        // the same run through the control plane, paused and resumed once: chunking must be invisible
        Sampler ctl(settings, logp, std::vector<double>(4 * 10, 0.0), 50);
        ctl.pause();
        ctl.resume();
        Sampler::WaitResult res = ctl.wait_timeout(std::chrono::seconds(120));
        if (res.kind != Sampler::WaitKind::Trace) { std::printf("control plane: %s\n", res.error.c_str()); return 1; }
        if (res.trace.positions.size() != trace.size() ||
            std::memcmp(res.trace.positions.data(), trace.data(), trace.size() * sizeof(double)) != 0) {
            std::printf("control plane trace differs from the draw loop\n");
            return 1;
        }
        const std::vector<ChainProgress> pr = ctl.progress();
        uint64_t steps2 = 0;
        for (const ChainProgress& p : pr) steps2 += p.total_num_steps;
        if (steps2 != steps || pr[0].finished_draws != total || pr[0].tuning) { std::printf("ChainProgress mismatch\n"); return 1; }
        {   // inspect() clones and never consumes (src/sampler.rs:1469-1485): polled after the run has ended it keeps returning the
            // whole trace, and the consuming wait_timeout() afterwards still hands all draws over
            Sampler ctl2(settings, logp, std::vector<double>(4 * 10, 0.0), 200);
            for (int spin = 0; spin < 12000 && ctl2.inspect().second.n_draws != total; ++spin) std::this_thread::sleep_for(std::chrono::milliseconds(10));
            std::this_thread::sleep_for(std::chrono::milliseconds(50));           // (the controller thread has set done_ by now)
            const auto a = ctl2.inspect(), b = ctl2.inspect();
            Sampler::WaitResult r2 = ctl2.wait_timeout(std::chrono::seconds(120));
            if (a.second.n_draws != total || b.second.n_draws != total || r2.kind != Sampler::WaitKind::Trace || r2.trace.n_draws != total ||
                a.second.positions != trace || b.second.positions != trace || r2.trace.positions != trace) {
                std::printf("inspect() after the end of the run lost draws (%llu, %llu, %llu of %llu)\n", (unsigned long long)a.second.n_draws,
                            (unsigned long long)b.second.n_draws, (unsigned long long)r2.trace.n_draws, (unsigned long long)total);
                return 1;
            }
        }
        if (argc > 1) {
            std::FILE* f = std::fopen(argv[1], "wb");
            std::fwrite(trace.data(), sizeof(double), trace.size(), f);
            std::fclose(f);
        }
        std::printf("ok %llu draws x 4 chains, %llu leapfrogs\n", (unsigned long long)total, (unsigned long long)steps);
        return 0;
    } catch (const NutsError& e) {
        if (e.status == NM_ERR_NO_DEVICE) { std::printf("NM_ERR_NO_DEVICE: %s\n", e.what()); return 3; }
        std::printf("NutsError %d: %s\n", (int)e.status, e.what());
        return 2;
    }
}
