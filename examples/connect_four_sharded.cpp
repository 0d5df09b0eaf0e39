// Config 5 without Python: one process per GPU, each running az_host::Coach::shard(rank, world) -- self-play and the arena
// sharded by global game id (src/coach.rs:241-272, :333-375), ONE az_gather_samples per episode batch and one 3-counter
// all-reduce per arena over RCCL (behind the C ABI), the trainer replicated.
//
// Build:  g++ -std=c++17 -O2 -I include examples/connect_four_sharded.cpp -o connect_four_sharded -L alphazero-rs_amd -laz_engine
//         (and -Wl,-rpath,$PWD/alphazero-rs_amd)
// Run:    for r in 0 1 ... N-1:  ./connect_four_sharded $r N /tmp/az_comm.id ./checkpoint [iters eps sims arena channels slots] &
//
// The communicator's unique id (the "128 bytes shipped by the host" of include/az_host.hpp) travels through the id-file: rank 0
// writes <id-file>.tmp and renames it, every other rank waits for <id-file> to appear (60 s), reads it, and all call
// az_comm_init.  Rank r uses GPU r mod (visible devices).  Rank 0 prints one JSON line per run.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

#include "az_host.hpp"

int main(int argc, char** argv) {
    using namespace az_host;
    if (argc < 5) {
        std::fprintf(stderr, "usage: %s rank world id-file checkpoint-dir [iters eps sims arena channels slots]\n", argv[0]);
        return 2;
    }
    const int rank = std::atoi(argv[1]), world = std::atoi(argv[2]);
    const std::string id_file = argv[3], dir = argv[4];
    const size_t iters = argc > 5 ? std::strtoul(argv[5], nullptr, 10) : 1;
    const size_t eps = argc > 6 ? std::strtoul(argv[6], nullptr, 10) : 64;
    const size_t sims = argc > 7 ? std::strtoul(argv[7], nullptr, 10) : 25;
    const size_t arena = argc > 8 ? std::strtoul(argv[8], nullptr, 10) : 40;
    const int channels = argc > 9 ? std::atoi(argv[9]) : 128;
    const size_t slots = argc > 10 ? std::strtoul(argv[10], nullptr, 10) : 8192;
    if (world < 1 || rank < 0 || rank >= world) { std::fprintf(stderr, "rank %d outside world %d\n", rank, world); return 2; }
    try {
        int ndev = 1;
        if (const char* v = std::getenv("AZ_VISIBLE_GPUS")) ndev = std::max(1, std::atoi(v));
        Engine e(rank % ndev, (int)slots, channels);
        // the communicator: rank 0's id through the id-file
        unsigned char id[AZ_COMM_ID_BYTES];
        if (rank == 0) {
            e.check(az_comm_unique_id(e.raw(), id));
            const std::string tmp = id_file + ".tmp";
            FILE* f = std::fopen(tmp.c_str(), "wb");
            if (!f || std::fwrite(id, 1, sizeof id, f) != sizeof id) throw Panic("cannot write " + tmp);
            std::fclose(f);
            if (std::rename(tmp.c_str(), id_file.c_str()) != 0) throw Panic("cannot publish " + id_file);
        } else {
            const auto t0 = std::chrono::steady_clock::now();
            for (;;) {
                if (FILE* f = std::fopen(id_file.c_str(), "rb")) {
                    const size_t n = std::fread(id, 1, sizeof id, f);
                    std::fclose(f);
                    if (n == sizeof id) break;
                }
                if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60)) throw Panic("no communicator id at " + id_file + " after 60 s");
                std::this_thread::sleep_for(std::chrono::milliseconds(20));
            }
        }
        e.check(az_comm_init(e.raw(), rank, world, id));
        if (az_net_load(e.raw(), 0, (dir + "/0.aznet").c_str()) != AZ_OK) e.check(az_net_init_random(e.raw(), 0, 0));
        Coach coach = Coach::setup(e, dir, 1000000, 0.6f, 15, 20, 200000, 1, slots, arena, iters, eps, sims, 1, 1000, 1);
        coach.shard(rank, world);
        coach.use_comm_at_world_1 = true;                  // world 1 runs the same gather / all-reduce path through a one-rank communicator
        const auto t0 = std::chrono::steady_clock::now();
        const auto reports = coach.learn(false, /*seed*/ 0);
        const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        e.check(az_comm_destroy(e.raw()));
        if (rank == 0) {
            size_t samples = 0, nw = 0, pw = 0, dr = 0, acc = 0;
            for (const auto& r : reports) { samples += r.samples; nw += r.nwins; pw += r.pwins; dr += r.draws; acc += r.accepted ? 1 : 0; }
            std::printf("{\"example\": \"connect_four_sharded\", \"world\": %d, \"iterations\": %zu, \"episodes_per_iteration\": %zu, \"sims\": %zu, "
                        "\"arena_games\": %zu, \"samples\": %zu, \"new_prev_draw\": [%zu, %zu, %zu], \"accepted\": %zu, \"seconds\": %.3f}\n",
                        world, reports.size(), eps, sims, arena, samples, nw, pw, dr, acc, secs);
        }
        return 0;
    } catch (const std::exception& ex) {
        std::fprintf(stderr, "rank %d panic: %s\n", rank, ex.what());
        return 1;
    }
}
