// C++ host: Coach::setup + Coach::learn (include/az_host.hpp) over the C ABI.  Usage: test_coach <dir> <channels> <seed> [comm]
// "comm": the same run as rank 0 of a world of ONE with a communicator (az_comm_unique_id / az_comm_init), so the episode
// batch goes through az_gather_samples and the arena tally through its all-reduce: every file must equal the plain run's.
// Prints one JSON line with the per-iteration report; tests/test_coach_gpu.py compares it and the files written
// under <dir> with the Python host's run of the same configuration.
#include <cstdio>
#include <cstdlib>

#include "az_host.hpp"

using namespace az_host;

int main(int argc, char** argv) {
    if (argc < 4) { std::fprintf(stderr, "usage: test_coach <dir> <channels> <seed>\n"); return 2; }
    const std::string dir = argv[1];
    const int channels = std::atoi(argv[2]);
    const uint64_t seed = std::strtoull(argv[3], nullptr, 10);
    const bool comm = argc > 4 && std::string(argv[4]) == "comm";
    try {
        Engine e(0, 256, channels);
        e.check(az_net_init_random(e.raw(), 0, 3));
        e.check(az_set_option(e.raw(), "train_epochs", 2));
        // the reference's parameter order, src/coach.rs:38-54 (values: a miniature of examples/connect_four.rs:55-71)
        Coach coach = Coach::setup(e, dir, /*mcts_reserve_size*/ 1000000, /*update_threshold*/ 0.55f, /*temp_threshold*/ 15,
                                   /*max_history_length*/ 3, /*max_queue_length*/ 100000, /*inference_batch_size*/ 1,
                                   /*num_episode_threads*/ 64, /*num_arena_games*/ 16, /*num_iters*/ 2, /*num_eps*/ 48,
                                   /*num_sims*/ 25, /*num_sim_threads*/ 1, /*max_depth*/ 1000, /*cpuct*/ 1);
        if (comm) {
            uint8_t id[AZ_COMM_ID_BYTES];
            e.check(az_comm_unique_id(e.raw(), id));
            e.check(az_comm_init(e.raw(), 0, 1, id));
            coach.shard(0, 1);
            coach.use_comm_at_world_1 = true;
        }
        const auto rep = coach.learn(false, seed);
        std::printf("[");
        for (size_t i = 0; i < rep.size(); ++i) {
            const auto& r = rep[i];
            std::printf("%s{\"iteration\": %zu, \"samples\": %zu, \"nwins\": %zu, \"pwins\": %zu, \"draws\": %zu, \"accepted\": %s, "
                        "\"model_id\": %zu, \"losses\": [", i ? ", " : "", r.iteration, r.samples, r.nwins, r.pwins, r.draws,
                        r.accepted ? "true" : "false", r.model_id);
            for (size_t k = 0; k < r.losses.size(); ++k) std::printf("%s%.9g", k ? ", " : "", r.losses[k]);
            std::printf("]}");
        }
        std::printf("]\n");
        if (comm) {
            uint64_t v[3] = {5, 6, 7};
            e.check(az_allreduce_u64(e.raw(), v, 3));
            if (v[0] != 5 || v[1] != 6 || v[2] != 7) { std::fprintf(stderr, "all-reduce at world 1 changed the values\n"); return 1; }
            e.check(az_comm_destroy(e.raw()));
            return 0;
        }
        // resume: a second setup on the same directory continues after the last examples file
        Coach again = Coach::setup(e, dir, 1000000, 0.55f, 15, 3, 100000, 1, 64, 16, 1, 48, 25, 1, 1000, 1);
        if (again.start_iteration != 2 || again.history.size() != 2) { std::fprintf(stderr, "resume failed\n"); return 1; }
        bool panicked = false;
        try { Coach::setup(e, dir, 1000000, 0.55f, 15, 3, 100000, 2, 64, 16, 1, 48, 25, 1, 1000, 1); } catch (const Panic&) { panicked = true; }
        if (!panicked) { std::fprintf(stderr, "num_sims %% inference_batch_size assert missing\n"); return 1; }
        return 0;
    } catch (const std::exception& ex) {
        std::fprintf(stderr, "panic: %s\n", ex.what());
        return 1;
    }
}
