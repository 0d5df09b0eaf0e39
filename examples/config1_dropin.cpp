// BASELINE config 1 as the fine-grained drop-in (examples/connect_four.rs:55-71: 1 episode at a time, 25 sims/move, 1 sim thread,
// inference batch 1): Coach::execute_episode (src/coach.rs:104-157) over ONE az_host::AsyncMcts (one az_tree of n_games = 1 behind
// AsyncMcts::get_action_prob, src/async_mcts.rs:74-115), move by move through the C ABI -- the shape INTEGRATION.md section 4 recommends
// for a host that keeps execute_episode as written.  Prints one JSON line: moves/s with the reference's stub net
// (DumbConnectFourNnet, examples/connect_four.rs:12-43), with the bf16 C = 512 policy+value net, and with that net at
// num_threads = 5 (25 % 5 == 0, src/async_mcts.rs:192).
// Build:  g++ -std=c++17 -O2 -I include examples/config1_dropin.cpp -o config1_dropin -L alphazero-rs_amd -laz_engine -Wl,-rpath,$PWD/alphazero-rs_amd
// Run:    ./config1_dropin [episodes] [sims]
#include <chrono>
#include <cstdio>
#include <cstdlib>

#include "az_host.hpp"

using namespace az_host;

static void run(Engine& e, const char* name, int model_id, int episodes, int sims, int threads, bool last) {
    size_t moves = 0, samples = 0;
    // one warm-up episode (first-touch allocations, graph capture), then the timed ones
    for (int pass = 0; pass < 2; ++pass) {
        const auto t0 = std::chrono::steady_clock::now();
        moves = samples = 0;
        for (int ep = 0; ep < (pass ? episodes : 1); ++ep) {
            AsyncMcts mcts = AsyncMcts::default_(e, 1000000, (size_t)sims, (size_t)threads, 1000, (size_t)model_id, 1);     // src/coach.rs:246-255
            std::vector<uint8_t> mv;
            samples += execute_episode(mcts, 15, (size_t)ep, /*seed*/ 0, &mv).size();
            moves += mv.size();
        }
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (pass) std::printf("\"%s\": {\"episodes\": %d, \"moves\": %zu, \"samples\": %zu, \"seconds\": %.6f, \"moves_per_sec\": %.3f, "
                              "\"sims_per_sec\": %.1f}%s", name, episodes, moves, samples, dt, moves / dt, moves * (double)sims / dt, last ? "" : ", ");
    }
}

int main(int argc, char** argv) {
    const int episodes = argc > 1 ? std::atoi(argv[1]) : 20;
    const int sims = argc > 2 ? std::atoi(argv[2]) : 25;
    try {
        Engine e(0, 64, 512);
        e.check(az_net_set_kind(e.raw(), 0, AZ_NET_STUB, 0));
        e.check(az_net_init_random(e.raw(), 1, 1));
        std::printf("{\"sims_per_move\": %d, ", sims);
        run(e, "stub_net", 0, episodes, sims, 1, false);
        run(e, "conv_net", 1, episodes, sims, 1, false);
        // the reference's own lever for one tree: num_threads simulations in flight (src/async_mcts.rs:191-217), here the
        // engine's deterministic lock-step schedule -- 5 leaves per net call instead of 1
        if (sims % 5 == 0) run(e, "conv_net_5_sim_threads", 1, episodes, sims, 5, true);
        else run(e, "conv_net_1_sim_thread_again", 1, episodes, sims, 1, true);
        std::printf("}\n");
        return 0;
    } catch (const std::exception& ex) {
        std::fprintf(stderr, "panic: %s\n", ex.what());
        return 1;
    }
}
