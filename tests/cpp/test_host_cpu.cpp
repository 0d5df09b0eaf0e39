// CPU-only checks of the host mirror (include/az_host.hpp) that need no engine call: the Connect Four rules on
// bitboards and the counter-RNG shuffle.  Prints one JSON line; tests/test_abi_cpu.py compares it with the Python host
// and the oracle.  Compiled with g++ alone (nothing here references libaz_engine.so).
#include <cstdio>
#include <cstdlib>

#include "az_host.hpp"

using namespace az_host;

int main(int argc, char** argv) {
    const size_t n = argc > 1 ? std::strtoul(argv[1], nullptr, 10) : 10;
    const uint64_t seed = argc > 2 ? std::strtoull(argv[2], nullptr, 10) : 3;
    const uint64_t iteration = argc > 3 ? std::strtoull(argv[3], nullptr, 10) : 2;
    std::printf("{\"perm\": [");
    const auto perm = shuffle_permutation(n, seed, iteration);
    for (size_t i = 0; i < perm.size(); ++i) std::printf("%s%lld", i ? "," : "", (long long)perm[i]);
    // the reference's diagonal game (connect_four_game.rs:244-264): player +1 wins on the 11th move
    ConnectFourGame b = ConnectFourGame::get_init_board();
    int8_t pl = 1;
    const uint8_t moves[11] = {0, 1, 1, 2, 2, 3, 2, 3, 3, 6, 3};
    std::printf("], \"ended\": [");
    for (int i = 0; i < 11; ++i) {
        auto nx = b.get_next_state(pl, moves[i]);
        b = nx.first;
        pl = nx.second;
        std::printf("%s%g", i ? "," : "", b.get_game_ended(1));
    }
    const auto valid = b.get_valid_moves(pl);
    std::printf("], \"valid\": [");
    for (int c = 0; c < 7; ++c) std::printf("%s%d", c ? "," : "", (int)valid[c]);
    const ConnectFourGame canon = b.get_canonical_form(pl);
    std::printf("], \"plus\": %llu, \"minus\": %llu, \"canon_plus\": %llu, \"canon_minus\": %llu, \"features_sum\": ",
                (unsigned long long)b.plus, (unsigned long long)b.minus, (unsigned long long)canon.plus, (unsigned long long)canon.minus);
    double fs = 0;
    for (float f : canon.to_features()) fs += f;
    std::printf("%g}\n", fs);
    return 0;
}
