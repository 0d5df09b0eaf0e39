// test_oracle.cpp -- CPU ORACLE (TEST INFRASTRUCTURE ONLY).
//
// Pins the oracle against every known-answer test the reference holds for the
// hot path:
//   src/node.rs:393-655            13 tests on the packed counter and NodeStore
//   connect_four_game.rs:244-264   test_win_diagonal
// plus structural checks of the repairs (69 four-windows, array <-> bitboard).
// Prints one line per test; exit code 0 iff all pass.
#include "az_oracle_games.hpp"

#include <cstdio>
#include <map>
#include <thread>

using namespace azo;

static int g_fail = 0;
#define CHECK(c) do { if (!(c)) { std::printf("  CHECK FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); ++g_fail; return; } } while (0)
static bool similar(float a, float b, float eps) { return std::fabs(a - b) < eps; }

// src/node.rs:393-418
static void test_win() {
    Node<DummyGame> node(10000.0f);
    CHECK(similar(node.get_w(), 0.0f, 1e-3f)); CHECK(node.get_n() == 0); CHECK(node.get_vloss() == 0);
    node.visit();
    CHECK(similar(node.get_w(), 0.0f, 1e-3f)); CHECK(node.get_n() == 1); CHECK(node.get_vloss() == 1);
    node.unvisit(1.0f);
    CHECK(similar(node.get_w(), 1.0f, 1e-3f)); CHECK(node.get_n() == 1); CHECK(node.get_vloss() == 0);
    CHECK(node.get_w() == 10001.0f / 10000.0f);   // C4: literally 1.0001
}
// src/node.rs:420-428
static void test_loss() {
    Node<DummyGame> node(10000.0f);
    node.visit(); node.unvisit(-1.0f);
    CHECK(similar(node.get_w(), -1.0f, 1e-3f)); CHECK(node.get_n() == 1); CHECK(node.get_vloss() == 0);
    CHECK(node.get_w() == -1.0f);
}
// src/node.rs:430-440
static void test_winloss() {
    Node<DummyGame> node(10000.0f);
    node.visit(); node.unvisit(-1.0f); node.visit();
    CHECK(node.get_vloss() == 1);
    node.unvisit(1.0f);
    CHECK(similar(node.get_w(), 0.0f, 1e-3f)); CHECK(node.get_n() == 2); CHECK(node.get_vloss() == 0);
    CHECK(node.get_w() == 1.0f / 10000.0f);       // literally 0.0001
}
// src/node.rs:442-445
static void test_is_lockfree() { std::atomic<uint64_t> a; CHECK(a.is_lock_free()); }
// src/node.rs:447-451
static void test_nodestore_empty() { auto nodes = NodeStore<DummyGame>::empty(2048); CHECK(nodes->size() == 0); }
// src/node.rs:453-467
static void test_nodestore_one() {
    auto nodes = NodeStore<DummyGame>::empty(2048);
    Node<DummyGame> node(10000.0f);
    node.p = std::vector<float>(10, 0.0f); node.v = std::vector<uint8_t>(10, 0); node.s = DummyGame(0);
    size_t idx = nodes->push(node);
    CHECK(nodes->get(idx) != nullptr);
    CHECK(*nodes->get(idx)->s == *node.s);
}
// src/node.rs:469-486
static void test_nodestore_many() {
    auto nodes = NodeStore<DummyGame>::empty(8192);
    std::vector<size_t> idx;
    for (int i = 0; i < 8192; ++i) { Node<DummyGame> n(10000.0f); n.e = (float)i; idx.push_back(nodes->push(n)); }
    CHECK(nodes->size() == 8192);
    for (int i = 0; i < 8192; ++i) CHECK((float)idx[i] == nodes->get(idx[i])->e);
}
static void parallel_push(int n, bool probe, bool* ok) {
    auto nodes = NodeStore<DummyGame>::empty(n);
    std::vector<size_t> idx(n);
    std::atomic<int> next{0};
    std::atomic<bool> bad{false};
    std::vector<std::thread> pool;
    for (int t = 0; t < 8; ++t)
        pool.emplace_back([&] {
            for (int i; (i = next.fetch_add(1)) < n;) {
                Node<DummyGame> node(10000.0f);
                node.e = (float)i;
                if (probe && i > 32 && nodes->state(i - 32) == std::optional<NodeState>(NodeState::PlaceHolder))
                    if (nodes->get(i - 32) == nullptr) bad = true;
                idx[i] = nodes->push(node);
            }
        });
    for (auto& th : pool) th.join();
    *ok = !bad && nodes->size() == (size_t)n;
    for (int i = 0; i < n && *ok; ++i) if ((float)i != nodes->get(idx[i])->e) *ok = false;
}
// src/node.rs:488-505, :507-524, :526-549 (rayon into_par_iter -> 8 std::threads)
static void test_nodestore_some_parallel() { bool ok; parallel_push(1024, false, &ok); CHECK(ok); }
static void test_nodestore_many_parallel() { bool ok; parallel_push(8192, false, &ok); CHECK(ok); }
static void test_nodestore_parallel_push_then_get() { bool ok; parallel_push(8192, true, &ok); CHECK(ok); }
// src/node.rs:551-588
static void test_nodestore_upgrade_many_similar() {
    auto nodes = NodeStore<DummyGame>::empty(8192);
    DummyGame s(0);
    std::map<size_t, std::optional<size_t>> idx;
    for (size_t i = 0; i < 8192; ++i) {
        size_t id = nodes->push(Node<DummyGame>(10000.0f));
        CHECK(nodes->lock(id));
        bool unique = *nodes->upgrade(id, s);
        if (unique) { CHECK(nodes->state(id) == std::optional<NodeState>(NodeState::Locked)); nodes->unlock(id); }
        idx[i] = unique ? std::optional<size_t>(id) : std::nullopt;
    }
    CHECK(nodes->size() == 8192);
    int uniq = 0; size_t root = 99;
    for (auto& kv : idx) if (kv.second) { ++uniq; root = *kv.second; }
    CHECK(uniq == 1); CHECK(root == 0);
}
// src/node.rs:590-632
static void test_nodestore_upgrade() {
    auto nodes = NodeStore<DummyGame>::empty(2048);
    Node<DummyGame> node(10000.0f);
    size_t idx = nodes->push(node);
    DummyGame s(0);
    Node<DummyGame>* const_ref = nodes->get(idx);
    CHECK(!nodes->get(idx)->s);
    CHECK(nodes->lock(0));
    CHECK(*nodes->upgrade(0, s));
    CHECK(nodes->state(0) == std::optional<NodeState>(NodeState::Locked));
    CHECK(const_ref == nodes->get(idx));             // pointer stability
    nodes->unlock(0);
    CHECK(nodes->state(0) == std::optional<NodeState>(NodeState::ExistsTrue));
    CHECK(*nodes->get(idx)->s == s);
    CHECK(nodes->seen.count(s) == 1); CHECK(nodes->seen.size() == 1);
    nodes->push(node);
    CHECK(nodes->lock(1));
    CHECK(!*nodes->upgrade(1, s));
    CHECK(nodes->state(1) == std::optional<NodeState>(NodeState::ExistsFalse));
}
// src/node.rs:634-655
static void test_nodestore_lock() {
    auto nodes = NodeStore<DummyGame>::empty(2048);
    size_t idx = nodes->push(Node<DummyGame>(10000.0f));
    CHECK(nodes->state(idx) == std::optional<NodeState>(NodeState::PlaceHolder));
    CHECK(nodes->lock(idx));
    nodes->upgrade(idx, DummyGame(0));
    CHECK(!nodes->lock(idx));
    CHECK(nodes->state(idx) == std::optional<NodeState>(NodeState::Locked));
    nodes->unlock(idx);
    CHECK(nodes->state(idx) == std::optional<NodeState>(NodeState::ExistsTrue));
}
// connect_four_game.rs:244-264, on both representations
template <class G> static void win_diagonal() {
    G board = G::get_init_board();
    int8_t player = 1;
    for (uint8_t a : {0, 1, 1, 2, 0, 2, 2, 3, 3, 3, 3}) {
        auto nx = board.get_next_state(player, a);
        board = nx.first; player = nx.second;
    }
    CHECK(board.get_game_ended(1) == 1.0f);
    CHECK(board.get_game_ended(-1) == -1.0f);
}
static void test_win_diagonal_array() { win_diagonal<C4Array>(); }
static void test_win_diagonal_bits() { win_diagonal<C4Bits>(); }

// B6: 69 four-windows exist; the literal loops see 56 of them.
static void test_windows() {
    int total = 0, literal_seen = 0;
    auto try_window = [&](int r0, int c0, int dr, int dc) {
        C4Array g, gl; gl.literal_windows = true;
        for (int k = 0; k < 4; ++k) { g.s[r0 + k * dr][c0 + k * dc] = 1; gl.s[r0 + k * dr][c0 + k * dc] = 1; }
        ++total;
        if (g.get_game_ended(1) != 1.0f) ++g_fail;
        if (!C4Bits::has_four(g.to_bits().p1)) ++g_fail;
        if (gl.get_game_ended(1) == 1.0f) ++literal_seen;
    };
    for (int r = 0; r < 6; ++r) for (int c = 0; c + 3 < 7; ++c) try_window(r, c, 0, 1);
    for (int r = 0; r + 3 < 6; ++r) for (int c = 0; c < 7; ++c) try_window(r, c, 1, 0);
    for (int r = 0; r + 3 < 6; ++r) for (int c = 0; c + 3 < 7; ++c) try_window(r, c, 1, 1);
    for (int r = 0; r + 3 < 6; ++r) for (int c = 3; c < 7; ++c) try_window(r, c, 1, -1);
    CHECK(total == 69); CHECK(literal_seen == 56);
    // three in a row / wrap-around across the column sentinel must not count
    C4Bits b; b.p1 = (1ull << 5) | (1ull << 7) | (1ull << 8) | (1ull << 9);
    CHECK(!C4Bits::has_four(b.p1));
}
// random playouts: array and bitboard agree on everything, move by move
static void test_array_vs_bits() {
    uint64_t r = 12345;
    for (int game = 0; game < 2000; ++game) {
        C4Array a; C4Bits b; int8_t player = 1;
        for (;;) {
            CHECK(a.to_bits() == b);
            CHECK(a.get_game_ended(player) == b.get_game_ended(player));
            CHECK(a.get_valid_moves(1) == b.get_valid_moves(1));
            CHECK(a.get_canonical_form(player).to_features() == b.get_canonical_form(player).to_features());
            CHECK(a.flip().to_bits() == b.flip());
            if (a.get_game_ended(player) != 0.0f) break;
            auto v = a.get_valid_moves(1);
            uint8_t mv;
            do { r = mix64(r); mv = (uint8_t)(r % 7); } while (!v[mv]);
            a = a.get_next_state(player, mv).first;
            auto nb = b.get_next_state(player, mv);
            b = nb.first; player = nb.second;
        }
    }
}
// config 1 smoke: 25 sims, stub net, both representations give the same episode
static void test_config1_episode() {
    StubNet net;
    AsyncMcts<C4Bits> m1(1000000 / 16, 25, 1, 1000, 0, 1, &net, 7);
    AsyncMcts<C4Array> m2(1000000 / 16, 25, 1, 1000, 0, 1, &net, 7);
    std::vector<uint8_t> mv1, mv2;
    auto s1 = execute_episode<C4Bits>(m1, 15, 0, 0, &mv1);
    auto s2 = execute_episode<C4Array>(m2, 15, 0, 0, &mv2);
    CHECK(mv1 == mv2); CHECK(s1.size() == s2.size()); CHECK(s1.size() == 2 * mv1.size());
    for (size_t i = 0; i < s1.size(); ++i) { CHECK(s1[i].board == s2[i].board); CHECK(s1[i].pi == s2[i].pi); CHECK(s1[i].v == s2[i].v); }
    CHECK(m1.stats.sims == 25 * mv1.size());
}

int main() {
    struct T { const char* name; void (*fn)(); } tests[] = {
        {"test_win", test_win}, {"test_loss", test_loss}, {"test_winloss", test_winloss},
        {"test_is_lockfree", test_is_lockfree}, {"test_nodestore_empty", test_nodestore_empty},
        {"test_nodestore_one", test_nodestore_one}, {"test_nodestore_many", test_nodestore_many},
        {"test_nodestore_some_parallel", test_nodestore_some_parallel},
        {"test_nodestore_many_parallel", test_nodestore_many_parallel},
        {"test_nodestore_parallel_push_then_get", test_nodestore_parallel_push_then_get},
        {"test_nodestore_upgrade_many_similar", test_nodestore_upgrade_many_similar},
        {"test_nodestore_upgrade", test_nodestore_upgrade}, {"test_nodestore_lock", test_nodestore_lock},
        {"test_win_diagonal_array", test_win_diagonal_array}, {"test_win_diagonal_bits", test_win_diagonal_bits},
        {"test_windows", test_windows}, {"test_array_vs_bits", test_array_vs_bits},
        {"test_config1_episode", test_config1_episode},
    };
    for (auto& t : tests) {
        int before = g_fail;
        t.fn();
        std::printf("%s %s\n", g_fail == before ? "PASS" : "FAIL", t.name);
    }
    std::printf("%d failure(s)\n", g_fail);
    return g_fail ? 1 : 0;
}
