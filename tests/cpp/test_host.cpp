// test_host.cpp -- drives the engine through the C++ mirror of the reference interface (include/az_host.hpp):
// Coach::execute_episode over AsyncMcts::get_action_prob, and arena::play_game(s) over two closures, exactly as
// src/coach.rs:246-260 and :365-375 do.  Prints one JSON line; tests/test_host_gpu.py compares it with the oracle.
#include <cstdio>
#include <cstdlib>

#include "az_host.hpp"

using namespace az_host;

int main(int argc, char** argv) {
    const int sims = argc > 1 ? std::atoi(argv[1]) : 25;
    const int episodes = argc > 2 ? std::atoi(argv[2]) : 3;
    const uint64_t seed = 17, salt = 1234;
    try {
        Engine e(0, 256, 512);
        e.check(az_net_set_kind(e.raw(), 10, AZ_NET_HASH, salt));
        e.check(az_net_set_kind(e.raw(), 11, AZ_NET_HASH, salt));
        std::printf("{\"episodes\": [");
        for (int ep = 0; ep < episodes; ++ep) {
            AsyncMcts mcts = AsyncMcts::default_(e, 1000000, sims, 1, 1000, 10, 1);      // src/coach.rs:246-255
            std::vector<uint8_t> moves;
            auto samples = execute_episode(mcts, 15, ep, seed, &moves);
            std::printf("%s{\"moves\": [", ep ? ", " : "");
            for (size_t i = 0; i < moves.size(); ++i) std::printf("%s%d", i ? "," : "", moves[i]);
            double zsum = 0, pisum = 0;
            for (auto& s : samples) { zsum += s.v; for (float p : s.pi) pisum += p; }
            std::printf("], \"samples\": %zu, \"zsum\": %.9g, \"pisum\": %.9g}", samples.size(), zsum, pisum);
        }
        // the same with two simulations in flight per tree (num_threads = 2, src/async_mcts.rs:27-36, :191-217)
        std::printf("], \"episodes_t2\": [");
        for (int ep = 0; ep < episodes; ++ep) {
            AsyncMcts mcts = AsyncMcts::default_(e, 1000000, sims + sims % 2, 2, 1000, 10, 1);
            std::vector<uint8_t> moves;
            auto samples = execute_episode(mcts, 15, ep, seed, &moves);
            std::printf("%s[", ep ? ", " : "");
            for (size_t i = 0; i < moves.size(); ++i) std::printf("%s%d", i ? "," : "", moves[i]);
            std::printf("]");
        }
        bool odd_panics = false;
        try { AsyncMcts::default_(e, 1000, 25, 2, 1000, 10, 1); } catch (const Panic&) { odd_panics = true; }    // 25 % 2 != 0, :192
        // arena: one game per seating with a fresh tree pair per game (B8), closures as src/coach.rs:365-372
        std::printf("], \"odd_sims_panic\": %s, \"arena\": [", odd_panics ? "true" : "false");
        for (int g = 0; g < 2; ++g) {
            AsyncMcts nmcts = AsyncMcts::default_(e, 1000000, sims, 1, 1000, 11, 1);
            AsyncMcts pmcts = AsyncMcts::default_(e, 1000000, sims, 1, 1000, 10, 1);
            auto argmax = [](const Policy& p) { size_t b = 0; for (size_t i = 1; i < p.size(); ++i) if (!(p[b] > p[i])) b = i; return (uint8_t)b; };
            PlayerAction newp = [&](const ConnectFourGame& s) { return argmax(nmcts.get_action_prob(s, 0.0f, g, seed)); };
            PlayerAction oldp = [&](const ConnectFourGame& s) { return argmax(pmcts.get_action_prob(s, 0.0f, g, seed)); };
            std::array<const PlayerAction*, 2> seated = g == 0 ? std::array<const PlayerAction*, 2>{&newp, &oldp}
                                                               : std::array<const PlayerAction*, 2>{&oldp, &newp};
            std::printf("%s%d", g ? "," : "", (int)play_game(seated, nullptr));
        }
        // error behaviour: a finished game as root "panics"
        ConnectFourGame b = ConnectFourGame::get_init_board();
        int8_t pl = 1;
        for (uint8_t a : {0, 1, 0, 1, 0, 1, 0}) { auto nx = b.get_next_state(pl, a); b = nx.first; pl = nx.second; }
        bool panicked = false;
        try {
            AsyncMcts m = AsyncMcts::default_(e, 1000, 10, 1, 1000, 10, 1);
            m.get_action_prob(b.get_canonical_form(pl), 1.0f, 0, seed);
        } catch (const Panic&) { panicked = true; }
        std::printf("], \"terminal_root_panics\": %s, \"ended\": %g}\n", panicked ? "true" : "false", b.get_game_ended(pl));
    } catch (const std::exception& ex) {
        std::fprintf(stderr, "FAILED: %s\n", ex.what());
        return 1;
    }
    return 0;
}
