// examples/connect_four.rs (src lines 45-80) on the C++ host: the same Coach::setup parameters, the engine behind it.
// Build:  g++ -std=c++17 -O2 -I include examples/connect_four.cpp -o connect_four -L alphazero-rs_amd -laz_engine
//         (and -Wl,-rpath,$PWD/alphazero-rs_amd)
// Run:    ./connect_four ./checkpoint [num_iters] [num_eps] [num_sims] [num_arena_games]
#include <cstdio>
#include <cstdlib>

#include "az_host.hpp"

int main(int argc, char** argv) {
    using namespace az_host;
    const std::string dir = argc > 1 ? argv[1] : "./checkpoint";
    const size_t iters = argc > 2 ? std::strtoul(argv[2], nullptr, 10) : 1;       // num_iters, examples/connect_four.rs:65
    const size_t eps = argc > 3 ? std::strtoul(argv[3], nullptr, 10) : 1;         // num_eps, :66
    const size_t sims = argc > 4 ? std::strtoul(argv[4], nullptr, 10) : 25;       // num_sims, :67
    const size_t arena = argc > 5 ? std::strtoul(argv[5], nullptr, 10) : 40;      // num_arena_games, :64
    try {
        Engine e(0, 8192, 512);
        if (az_net_load(e.raw(), 0, (dir + "/0.aznet").c_str()) != AZ_OK) e.check(az_net_init_random(e.raw(), 0, 0));
        Coach coach = Coach::setup(e, dir, 1000000, 0.6f, 15, 20, 200000, 1, /*concurrent game slots*/ 8192, arena, iters, eps, sims, 1,
                                   1000, 1);
        for (const auto& r : coach.learn(false, /*seed*/ 0))
            std::printf("iteration %zu: %zu samples, new/prev/draw %zu/%zu/%zu, %s, loss (%.4f, %.4f)\n", r.iteration, r.samples, r.nwins,
                        r.pwins, r.draws, r.accepted ? "accepted" : "rejected", r.losses.empty() ? 0.f : r.losses[r.losses.size() - 2],
                        r.losses.empty() ? 0.f : r.losses.back());
        return 0;
    } catch (const std::exception& ex) {
        std::fprintf(stderr, "panic: %s\n", ex.what());
        return 1;
    }
}
