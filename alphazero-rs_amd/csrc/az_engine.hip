// az_engine.hip -- C-ABI implementation (include/az_engine.h): host orchestration of the
// tree kernels (az_tree.hip) and the nets (az_net.hip) on one HIP stream.
//
// There is NO CPU fallback: every entry point runs the HIP kernels or fails with a status.
#include "../../include/az_engine.h"

#include <dlfcn.h>
#include <rccl/rccl.h>      // types only: the library is dlopen'ed by az_comm_* (a host that never shards never loads it)

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "az_net.h"
#include "az_train.h"
#include "az_tree.h"

using namespace az;

// ---------------------------------------------------------------------------------------------
namespace {

struct HipFail { hipError_t code; const char* what; };
#define HIPCHK(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) throw HipFail{_e, #expr}; } while (0)

struct DeviceMem {
    std::vector<void*> ptrs;
    template <class T> T* alloc(size_t n) {
        void* p = nullptr;
        HIPCHK(hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(T)));
        ptrs.push_back(p);
        return (T*)p;
    }
    void release() {
        for (void* p : ptrs) (void)hipFree(p);
        ptrs.clear();
    }
    ~DeviceMem() { release(); }
};

struct NetModel {
    int kind = -1;
    uint64_t salt = 0;
    ConvNet* conv = nullptr;
    uint64_t cache_tag = 0;     // evaluation-cache tag of the current weights (0 = none yet); new tag per upload
    uint64_t generation = 0;    // bumped by every weight upload / kind change of any model of the process: part of a search graph's key
};
uint64_t g_model_generation = 0;

// timed regions (profile mode): event pairs recorded on the engine stream, resolved at sync points
enum Region { RG_TREE = 0, RG_NET = 1, RG_COUNT };   // RG_NET brackets the whole predict (kept for stub nets)
struct Profiler {
    bool on = false;
    std::vector<hipEvent_t> pool;
    struct Rec { hipEvent_t a, b; int region; };
    std::vector<Rec> open;
    hipEvent_t get() {
        if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
        hipEvent_t e;
        HIPCHK(hipEventCreate(&e));
        return e;
    }
    hipEvent_t begin(hipStream_t s) { hipEvent_t a = get(); HIPCHK(hipEventRecord(a, s)); return a; }
    void end(hipEvent_t a, int region, hipStream_t s) {
        hipEvent_t b = get();
        HIPCHK(hipEventRecord(b, s));
        open.push_back({a, b, region});
    }
    // call after the stream has been synchronised
    void resolve(double* ms_by_region) {
        for (auto& r : open) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) ms_by_region[r.region] += ms;
            pool.push_back(r.a);
            pool.push_back(r.b);
        }
        open.clear();
    }
    ~Profiler() { for (auto e : pool) (void)hipEventDestroy(e); }
};

// RCCL entry points, resolved at the first az_comm_* call.  Not a link-time dependency: a process that also hosts another copy
// of RCCL (PyTorch bundles one) must not have two sets of ncclXxx symbols bound into one namespace.
struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool load(std::string* why) {
        if (lib) return true;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) { *why = std::string("cannot load librccl: ") + dlerror(); return false; }
        bool ok = true;
        auto sym = [&](const char* n) { void* p = dlsym(lib, n); if (!p) { ok = false; *why = std::string("librccl lacks ") + n; } return p; };
        GetUniqueId = (decltype(GetUniqueId))sym("ncclGetUniqueId");
        CommInitRank = (decltype(CommInitRank))sym("ncclCommInitRank");
        CommDestroy = (decltype(CommDestroy))sym("ncclCommDestroy");
        AllGather = (decltype(AllGather))sym("ncclAllGather");
        AllReduce = (decltype(AllReduce))sym("ncclAllReduce");
        Send = (decltype(Send))sym("ncclSend");
        Recv = (decltype(Recv))sym("ncclRecv");
        GroupStart = (decltype(GroupStart))sym("ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))sym("ncclGroupEnd");
        GetErrorString = (decltype(GetErrorString))sym("ncclGetErrorString");
        if (!ok) { dlclose(lib); lib = nullptr; }
        return ok;
    }
};
Rccl g_rccl;      // function table only (no per-engine state)

// (s, pi, z) as one 48-byte tuple: the unit of the episode-batch gather (symmetries are regenerated at the destination)
struct PackedSample { unsigned long long s0, s1; float pi[7]; float z; };
static_assert(sizeof(PackedSample) == 48, "48-byte tuples");
__global__ void k_pack_samples(const ulonglong2* st, const float* pi, const float* z, PackedSample* out, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    PackedSample p;
    const ulonglong2 s = st[i];
    p.s0 = s.x; p.s1 = s.y;
#pragma unroll
    for (int a = 0; a < 7; ++a) p.pi[a] = pi[i * 7 + a];
    p.z = z[i];
    out[i] = p;
}
__global__ void k_unpack_samples(const PackedSample* in, ulonglong2* st, float* pi, float* z, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const PackedSample p = in[i];
    st[i] = make_ulonglong2(p.s0, p.s1);
#pragma unroll
    for (int a = 0; a < 7; ++a) pi[i * 7 + a] = p.pi[a];
    z[i] = p.z;
}

inline uint32_t next_pow2_u32(uint64_t x) {
    uint64_t p = 1;
    while (p < x) p <<= 1;
    return (uint32_t)p;
}

struct TreeHost {
    DeviceMem mem;
    TreeDev d{};
    EvalBatch eb{};         // eval batch of even simulations (and of the root evaluation)
    EvalBatch eb2{};        // eval batch of odd simulations: k_backup_select ping-pongs between the two
    // a run of simulation steps as ONE hipGraph launch (run_search): every launch of a search takes the same arguments, so the
    // instantiated graph is reused as long as nothing that shapes a launch changes (graph_key: grids, kernel choices, pointers)
    struct StepGraph { hipGraphExec_t exec = nullptr; std::vector<unsigned char> key; };
    StepGraph step_graph;
    unsigned long long* d_totals = nullptr;   // [ST_COUNT] k_harvest's sums
    uint32_t* d_counts = nullptr;             // [G] NodeStore::len per tree (k_harvest)
    // blocks = child blocks the trees can use (each holds the <= 7 children of one expansion, or a root);
    // reserve_nodes = reserve_space (src/node.rs:146) clamped to what is reachable
    // allocation key (the engine keeps finished calls' arenas for the next call of the same shape)
    int k_G = 0, k_T = 1; uint64_t k_blocks = 0; uint32_t k_H = 0;
    bool in_use = false;
    bool fits(int G, uint64_t blocks, uint32_t H, int T) const { return G == k_G && blocks == k_blocks && H == k_H && T == k_T; }
    // make a kept arena look freshly created (the trees themselves are rebuilt by launch_reset_trees)
    void recycle(uint64_t reserve_nodes, hipStream_t s) {
        d.reserve_nodes = (uint32_t)reserve_nodes;
        HIPCHK(hipMemsetAsync(d.err, 0, ERR_COUNT * sizeof(uint32_t), s));
        HIPCHK(hipMemsetAsync(d_totals, 0, ST_TOTALS * sizeof(unsigned long long), s));
        HIPCHK(hipMemsetAsync(eb.n, 0, sizeof(uint32_t), s));
        HIPCHK(hipMemsetAsync(eb2.n, 0, sizeof(uint32_t), s));
        HIPCHK(hipMemsetAsync(eb.tkey, 0, ((size_t)eb.tmask + 1) * 8, s));       // a call that failed half way may have left keys behind
        HIPCHK(hipMemsetAsync(eb2.tkey, 0, ((size_t)eb2.tmask + 1) * 8, s));
        if (d.thr) HIPCHK(hipMemsetAsync(d.thr, 0, (size_t)d.G * d.T * sizeof(TreeLine), s));
        launch_init_heads(d, s);
    }
    // T = simulations in flight per tree (1: the single-simulation kernels; > 1: per-thread lines + T rows per tree in a leaf batch)
    ~TreeHost() { if (step_graph.exec) (void)hipGraphExecDestroy(step_graph.exec); }
    void create(int G, uint64_t blocks, uint64_t reserve_nodes, uint32_t H, int T, int game) {
        k_G = G; k_blocks = blocks; k_H = H; k_T = T;
        d.game = game;
        d.T = T;
        d.G = G; d.R = (uint32_t)(blocks * BLOCK_SLOTS); d.H = H; d.reserve_nodes = (uint32_t)reserve_nodes;
        size_t slots = (size_t)G * d.R;
        if (d.R > MAX_TREE_SLOTS) throw HipFail{hipErrorInvalidValue, "a tree of more than 2^20 node slots (num_sims x plies too large for the 20-bit links of the node record)"};
        d.node = mem.alloc<uint4>(slots);
        d.key = mem.alloc<unsigned long long>(slots);
        d.hash = mem.alloc<uint32_t>((size_t)G * H);
        d.head = mem.alloc<TreeLine>(G);
        d.path = mem.alloc<uint32_t>((size_t)G * T * PATH_CAP);
        d.err = mem.alloc<uint32_t>(ERR_COUNT);
        d.log_cap = 0;
        if (T > 1) {
            d.thr = mem.alloc<TreeLine>((size_t)G * T);
            HIPCHK(hipMemset(d.thr, 0, (size_t)G * T * sizeof(TreeLine)));
        }
        d_totals = mem.alloc<unsigned long long>(ST_TOTALS);
        d_counts = mem.alloc<uint32_t>(G);
        HIPCHK(hipMemset(d.err, 0, ERR_COUNT * sizeof(uint32_t)));
        HIPCHK(hipMemset(d_totals, 0, ST_TOTALS * sizeof(unsigned long long)));
        launch_init_heads(d, nullptr);
        HIPCHK(hipDeviceSynchronize());
        for (EvalBatch* b : {&eb, &eb2}) {
            const size_t rows = (size_t)G * T;
            b->cap = (int32_t)rows;
            b->n = mem.alloc<uint32_t>(1);
            b->state = mem.alloc<ulonglong2>(rows);
            b->pi = mem.alloc<float>(rows * 8);
            b->v = mem.alloc<float>(rows);
            HIPCHK(hipMemset(b->n, 0, sizeof(uint32_t)));
            // election table of the leaf de-duplication (used when the engine's "eval_dedup" applies to a search's net)
            const uint32_t tsize = next_pow2_u32(4ull * (uint64_t)std::max<size_t>(rows, 16));
            b->tkey = mem.alloc<unsigned long long>(tsize);
            b->tuniq = mem.alloc<uint32_t>(tsize);
            b->tmask = tsize - 1;
            HIPCHK(hipMemset(b->tkey, 0, (size_t)tsize * 8));
        }
    }
};

uint32_t next_pow2(uint64_t x) {
    uint64_t p = 1;
    while (p < x) p <<= 1;
    return (uint32_t)p;
}
// Largest tree an episode of <= `calls` get_action_prob calls can build: the initial root + 7,
// per call one fresh root (S10) and num_sims expansions, 7 placeholders each.
uint64_t reachable_slots(int num_sims, int calls) {
    return 8ull + (uint64_t)calls * ((uint64_t)num_sims * 7ull + 8ull);
}
// child blocks the same tree can use: two for the initial root, per call two for a fresh root (S10) and one per expansion;
// never more than one per pushed node
uint64_t reachable_blocks(int num_sims, int calls, uint64_t reserve_nodes) {
    return std::min<uint64_t>(2ull + (uint64_t)calls * ((uint64_t)num_sims + 2ull), std::max<uint64_t>(reserve_nodes, 2ull) + 1ull);
}
uint32_t hash_entries(int num_sims, int calls) {
    return next_pow2(2ull * ((uint64_t)calls * ((uint64_t)num_sims + 1) + 2));
}

}  // namespace

struct az_engine {
    int device = 0;
    hipStream_t stream = nullptr;
    az_config cfg{};
    std::string err;
    std::map<int, NetModel> nets;
    az_stats stats{};
    Profiler prof;
    NetProfile netprof;
    NetOptions netopt;              // kernel-set switches of the conv net: this engine's, passed down with every forward
    int tree_block4 = 1;            // "tree_block4": k_backup_select as 4-wave workgroups
    // activation workspaces of the conv net: [0] the engine stream, [1] a second concurrent stream (az_arena's old-model
    // search); created on first use, shared by every model id
    NetWorkspace* ws[2] = {nullptr, nullptr};
    // eval log of the last az_selfplay
    std::vector<int32_t> sp_log_count;
    std::vector<uint64_t> sp_log_states;
    std::vector<float> sp_log_pi, sp_log_v;
    int sp_log_cap = 0;
    // eval logs of the last az_arena with record_evals: [0] the new model's trees, [1] the old model's
    struct EvalLog { std::vector<int32_t> count; std::vector<uint64_t> states; std::vector<float> pi, v; };
    EvalLog ar_log[2];
    struct SelfplaySession* sp_session = nullptr;      // az_selfplay_begin .. az_selfplay_end
    int ar_log_cap = 0, ar_log_games = 0;
    std::vector<uint8_t> ar_moves;      // [games][AZ_MAX_PLIES] move record of the last az_arena (az_arena_get_moves)
    std::vector<int32_t> ar_len;
    // NNet::train
    Trainer* trainer = nullptr;
    bool train_open = false;
    TrainHyper hyper;
    int train_epochs = 10, train_batch = 64;       // connect_four_net.py:13-14
    uint64_t train_seed = 0;
    int train_graph = 1;
    int train_gemm = 1;
    int train_gemm3_ring = 1;       // "train_gemm3_ring"
    int train_fwd_x3 = 1;           // "train_fwd_x3"
    int train_wgrad_tr = 1;         // "train_wgrad_tr"
    int train_implicit = 1;         // "train_implicit"
    int train_fork = 0;             // "train_fork": the wgrad chains of a step on a second stream branch (csrc/az_train.hip)
    int train_fwd_dma = 1;
    std::vector<float> train_history;              // (loss_pi, loss_v) mean per epoch of the last az_net_train
    // leaf de-duplication + evaluation cache (az_set_option "eval_dedup", "eval_cache_log2", "eval_cache_max_stones",
    // "eval_cache_persist")
    int profile_every = 1;          // profile mode: bracket every n-th simulation step ("profile_every")
    int dedup_stats = 1;            // "dedup_stats": the leaf-row accounting counters (requested / executed / hits / duplicates)
    uint64_t profile_tick = 0;
    int search_graph = 20;          // "search_graph": simulation steps per hipGraph launch (0 = every kernel launched on its own); not in profile mode
    int selfplay_async = 0;         // "selfplay_async": 1 = az_selfplay with free-running slots (k_async_step) for nets that go through leaf batches; 0 (default) = lock-step
                                    // moves.  Bit-identical games; measured in round 4: larger batches (4000 against 2700 rows per forward), fewer forwards, the same
                                    // time per row -- the step is throughput-bound per row, so it is not the default (profiles/README.md)
    int selfplay_async_launches = 2;   // "selfplay_async_launches": tree launches that share one leaf batch (cache-answered trees go on in the next one)
    int selfplay_async_iters = 6;   // "selfplay_async_iters": stages (backup, root, move, select) a slot may run per launch
    int search_graph_rows = 1024;   // "search_graph_rows": ... for searches whose expected leaf batch has at most this many rows (the arena, a drain, single trees)
    int fused_search = 1;           // stub / hash nets: the whole search in one launch ("fused_search"; 0 = one launch per simulation)
    int eval_dedup = 1;             // 0 off, 1 conv nets (default), 2 every net (lets the hash fixture exercise the machinery)
    int eval_cache_log2 = 30;       // entries = 2^log2 (40 B each: 43 GB); 0 = no cache, in-batch de-duplication only
    int eval_cache_max_stones = 42;
    int eval_cache_persist = 0;     // 0: az_selfplay / az_arena / az_tree_get_action_prob start from an empty cache
    DeviceMem cache_mem;            // the accounting counters
    EvalCache cache{};              // the view the current call uses (prepare_cache); key == nullptr: no cache
    unsigned long long* cache_keys = nullptr;       // the allocation: 2^cache_alloc_log2 entries
    float* cache_pv = nullptr;
    int cache_alloc_log2 = -1;
    uint64_t next_cache_tag = 1;    // 15 bits; wrapping clears the cache
    // tree arenas of finished az_selfplay / az_arena calls, reused by the next call of the same shape (9.7 GB at 8192 x 100:
    // no hipMalloc / hipFree per call); at most three are kept (self-play + the arena's pair)
    std::vector<std::unique_ptr<TreeHost>> tree_pool;
    uint64_t tree_pool_allocs = 0;  // arenas created (a second call of the same shape must not add to it)
    // communicator of the sharded Coach loop (az_comm_init): RCCL on the engine's stream
    ncclComm_t comm = nullptr;
    int comm_rank = 0, comm_world = 1;
    // staging of the collectives (az_gather_samples, az_allreduce_u64): one device allocation that only grows
    struct Scratch {
        void* p = nullptr;
        size_t cap = 0;
        void* ensure(size_t bytes, hipStream_t s) {
            if (bytes <= cap) return p;
            HIPCHK(hipStreamSynchronize(s));
            if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
            const size_t want = std::max<size_t>(bytes + bytes / 4, (size_t)1 << 16);
            HIPCHK(hipMalloc(&p, want));
            cap = want;
            return p;
        }
        ~Scratch() { if (p) (void)hipFree(p); }
    } comm_scratch;
};

struct az_tree {
    az_engine* e = nullptr;
    TreeHost th;
    DeviceMem mem;
    int num_sims = 0, max_depth = 0, model_id = 0, cpuct = 0;
    ulonglong2* d_root_states = nullptr;
    // pinned host block of one get_action_prob call: the roots go up from it and k_root_policy / k_call_readback write the
    // results and counters straight into it, so a call costs one stream synchronisation instead of eight blocking copies
    void* h_io = nullptr;
    CallReadback* h_rb = nullptr;
    ulonglong2* h_states = nullptr;
    float* h_pi = nullptr;
    float* h_q = nullptr;
    uint16_t* h_counts = nullptr;
    ~az_tree() { if (h_io) (void)hipHostFree(h_io); }
};

namespace {

az_status fail(az_engine* e, az_status st, const std::string& msg) {
    if (e) e->err = msg;
    return st;
}
az_status fail_hip(az_engine* e, const HipFail& f) {
    char buf[512];
    std::snprintf(buf, sizeof buf, "HIP error %d (%s) at %s", (int)f.code, hipGetErrorString(f.code), f.what);
    return fail(e, AZ_ERR_HIP, buf);
}

NetWorkspace* workspace_for(az_engine* e, hipStream_t s) {
    const int i = (s == e->stream) ? 0 : 1;
    if (!e->ws[i]) {
        const char* why = nullptr;
        e->ws[i] = netws_create(e->cfg.net_channels, e->cfg.max_batch, &why);
        if (!e->ws[i]) throw HipFail{hipErrorOutOfMemory, why ? why : "netws_create"};
    }
    return e->ws[i];
}

void net_forward(az_engine* e, const NetModel& net, const EvalBatch& eb, int rows_hint, hipStream_t s, int rows_typ = 0, bool timed = true) {
    if (net.kind == AZ_NET_CONV) {
        convnet_forward(net.conv, workspace_for(e, s), eb, rows_hint, rows_typ, s, (e->prof.on && timed) ? &e->netprof : nullptr, e->netopt);
    } else {
        launch_net_fixture(eb, net.kind, net.salt, s);
    }
}

bool dedup_applies(const az_engine* e, const NetModel& net) {
    return e->eval_dedup == 2 || (e->eval_dedup == 1 && net.kind == AZ_NET_CONV);
}

// The engine's evaluation cache.  Sized FROM THE CALL: a search call can insert at most `inserts_bound` distinct states (trees x
// get_action_prob calls x (sims + 1)), so it gets the smallest power of two >= 4 x that bound (8-way buckets stay sparse), at most
// 2^"eval_cache_log2" entries (40 bytes each; 43 GB at the default 30, which only a bench-sized call reaches) -- a 1-tree, 25-sim
// call allocates and clears 40 KB.  The allocation only grows; a smaller call uses (and clears) a prefix of it.  With
// "eval_cache_persist" entries must stay findable across calls, so the cache then has its full configured size from the start.
void prepare_cache(az_engine* e, bool wanted, uint64_t inserts_bound, hipStream_t s) {
    if (!e->cache.stat) {
        e->cache.stat = e->cache_mem.alloc<unsigned long long>((size_t)DD_REPLICAS * DD_STRIDE);
        HIPCHK(hipMemset(e->cache.stat, 0, (size_t)DD_REPLICAS * DD_STRIDE * sizeof(unsigned long long)));
    }
    if (!wanted || e->eval_cache_log2 < 3) { e->cache.key = nullptr; e->cache.pv = nullptr; e->cache.bmask = 0; return; }
    int want = e->eval_cache_log2;
    if (!e->eval_cache_persist) {
        int need = 10;
        while (need < want && (1ull << need) < 4ull * inserts_bound) ++need;
        want = need;
    }
    if (e->cache_alloc_log2 < want) {
        HIPCHK(hipStreamSynchronize(s));
        if (e->cache_keys) { (void)hipFree(e->cache_keys); e->cache_keys = nullptr; }
        if (e->cache_pv) { (void)hipFree(e->cache_pv); e->cache_pv = nullptr; }
        e->cache_alloc_log2 = -1;
        const size_t entries = (size_t)1 << want;
        HIPCHK(hipMalloc((void**)&e->cache_keys, entries * 8));
        HIPCHK(hipMalloc((void**)&e->cache_pv, entries * 8 * sizeof(float)));
        HIPCHK(hipMemset(e->cache_keys, 0, entries * 8));
        e->cache_alloc_log2 = want;
        for (auto& kv : e->nets) kv.second.cache_tag = 0;       // a new allocation holds nobody's entries
        e->next_cache_tag = 1;
    }
    const int view = want;          // persist: the configured size, the same for every call
    e->cache.key = e->cache_keys;
    e->cache.pv = e->cache_pv;
    e->cache.bmask = (uint32_t)(((size_t)1 << view) / 8 - 1);
    if (!e->eval_cache_persist) HIPCHK(hipMemsetAsync(e->cache.key, 0, ((size_t)e->cache.bmask + 1) * 64, s));      // this call's part only
}
// a new tag for a model's new weights: its old entries can never match again
void retag_model(az_engine* e, NetModel& m) {
    if (e->next_cache_tag > 0x7FFFull) {
        if (e->cache_keys) HIPCHK(hipMemset(e->cache_keys, 0, ((size_t)1 << e->cache_alloc_log2) * 8));
        for (auto& kv : e->nets) kv.second.cache_tag = 0;
        e->next_cache_tag = 1;
    }
    m.cache_tag = e->next_cache_tag++;
}
// cache view for one search (key == nullptr when only in-batch de-duplication is wanted or nothing applies)
EvalCache cache_for(az_engine* e, NetModel& net) {
    EvalCache c{};
    if (!dedup_applies(e, net)) return c;
    if (net.cache_tag == 0) retag_model(e, net);
    c = e->cache;
    c.max_stones = (uint32_t)e->eval_cache_max_stones;
    c.tag = (unsigned long long)net.cache_tag << 49;
    if (!e->dedup_stats) c.stat = nullptr;
    return c;
}
// A tree arena for one call: a kept one of the same shape, or a new one (the oldest idle one makes room).
struct TreeLease {
    TreeHost* th = nullptr;
    ~TreeLease() { if (th) th->in_use = false; }
    TreeHost* operator->() const { return th; }
    TreeHost& operator*() const { return *th; }
};
void acquire_trees(az_engine* e, TreeLease& lease, int G, uint64_t blocks, uint64_t reserve_nodes, uint32_t H, int T, hipStream_t s) {
    for (auto& p : e->tree_pool)
        if (!p->in_use && p->fits(G, blocks, H, T)) {
            p->in_use = true;
            lease.th = p.get();
            p->recycle(reserve_nodes, s);
            return;
        }
    // drop idle arenas of other shapes before allocating (two shapes of 10 GB each should not pile up)
    for (size_t i = 0; i < e->tree_pool.size();) {
        if (!e->tree_pool[i]->in_use && (e->tree_pool.size() >= 3 || !e->tree_pool[i]->fits(G, blocks, H, T))) {
            HIPCHK(hipStreamSynchronize(s));
            e->tree_pool.erase(e->tree_pool.begin() + (long)i);
        } else {
            ++i;
        }
    }
    std::unique_ptr<TreeHost> th(new TreeHost());
    th->create(G, blocks, reserve_nodes, H, T, e->cfg.game);
    th->in_use = true;
    lease.th = th.get();
    e->tree_pool.push_back(std::move(th));
    e->tree_pool_allocs += 1;
}

// Eval log of one call (replay parity): rows * cap records in device memory, attached to a tree arena for the call's duration.
struct ScopedEvalLog {
    TreeHost* th = nullptr;
    DeviceMem mem;
    size_t rows = 0;
    int cap = 0;
    void attach(TreeHost& t, size_t rows_, int cap_, const int32_t* log_row) {
        th = &t; rows = rows_; cap = cap_;
        t.d.log_cap = cap;
        t.d.log_state = mem.alloc<ulonglong2>(rows * (size_t)cap);
        t.d.log_pi = mem.alloc<float>(rows * (size_t)cap * 7);
        t.d.log_v = mem.alloc<float>(rows * (size_t)cap);
        t.d.log_row = log_row;
    }
    void copy_out(std::vector<uint64_t>& states, std::vector<float>& pi, std::vector<float>& v) const {
        states.resize(rows * cap * 2); pi.resize(rows * cap * 7); v.resize(rows * cap);
        HIPCHK(hipMemcpy(states.data(), th->d.log_state, states.size() * 8, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(pi.data(), th->d.log_pi, pi.size() * 4, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(v.data(), th->d.log_v, v.size() * 4, hipMemcpyDeviceToHost));
    }
    ~ScopedEvalLog() {          // the arena goes back to the pool without the call's log
        if (th) { th->d.log_cap = 0; th->d.log_state = nullptr; th->d.log_pi = nullptr; th->d.log_v = nullptr; th->d.log_row = nullptr; }
    }
};

// fold the device-side de-duplication counters into the engine stats (after a stream sync)
void fold_dedup(az_engine* e, const unsigned long long* h) {
    e->stats.leaf_rows_requested += h[DD_REQUESTED];
    e->stats.leaf_rows_executed += h[DD_EXECUTED];
    e->stats.eval_cache_hits += h[DD_CACHE_HITS];
    e->stats.eval_batch_dups += h[DD_BATCH_DUPS];
    e->stats.eval_cache_inserts += h[DD_INSERTS];
}
void harvest_dedup(az_engine* e) {
    if (!e->cache.stat) return;
    unsigned long long rep[DD_REPLICAS * DD_STRIDE], h[DD_COUNT] = {};
    HIPCHK(hipMemcpy(rep, e->cache.stat, sizeof rep, hipMemcpyDeviceToHost));
    HIPCHK(hipMemset(e->cache.stat, 0, sizeof rep));
    for (int r = 0; r < DD_REPLICAS; ++r)
        for (int i = 0; i < DD_COUNT; ++i) h[i] += rep[r * DD_STRIDE + i];
    fold_dedup(e, h);
}

// get_action_prob body shared by every entry point: S10/S1 prologue, then num_sims x
// {select+expand (+ leaf request), predict, mask+store+backup}  (src/async_mcts.rs:81-82, :191-217).
// rows_hint = host-side upper bound on the leaf batch (trees still searching): sizes the net's grids and picks tiles
void run_search(az_engine* e, TreeHost& th, const ulonglong2* d_root_states, int num_sims, SearchParams sp,
                NetModel& net, int rows_hint, hipStream_t s = nullptr, int rows_typ = 0, uint32_t* d_max_rows = nullptr) {
    if (!s) s = e->stream;
    th.d.block4 = e->tree_block4;
    const int T = th.d.T;
    if (rows_hint <= 0 || rows_hint > th.d.G) rows_hint = th.d.G;
    rows_hint *= T;                              // up to T leaves per tree and step
    const bool dedup = dedup_applies(e, net);
    if (net.kind != AZ_NET_CONV && !dedup && e->fused_search && T == 1) {
        // stub / hash nets are device functions of the state: the whole search is one launch
        hipEvent_t t0 = nullptr;
        if (e->prof.on) t0 = e->prof.begin(s);
        launch_search_fixture(th.d, d_root_states, sp, num_sims, net.kind, net.salt, s);
        if (e->prof.on) { e->prof.end(t0, RG_TREE, s); e->stats.tree_launches_timed += 1; }
        e->stats.tree_launches += 1;
        return;
    }
    const EvalCache ec = cache_for(e, net);
    EvalBatch B[2] = {th.eb, th.eb2};
    B[0].dedup = B[1].dedup = dedup ? 1 : 0;
    B[0].max_n = B[1].max_n = d_max_rows;
    // profile mode brackets every "profile_every"-th simulation step (events cost GPU idle time between dependent kernels)
    const int every = std::max(1, e->profile_every);
    // root: prepare (its leaf goes to batch 0), predict; then num_sims x {backup of the previous leaf + select of the next
    // (one launch, the new leaf goes to the other batch), predict}; a last backup closes the search.
    launch_root_prepare(th.d, B[0], ec, d_root_states, s);
    net_forward(e, net, B[0], rows_hint, s, rows_typ, false);
    if (T > 1) {
        // num_sims / T lock-step steps of T simulations per tree (src/async_mcts.rs:191-217; num_sims % T == 0, :192)
        const int steps = num_sims / T;
        for (int i = 0; i < steps; ++i) {
            const bool timed = e->prof.on && (e->profile_tick++ % (uint64_t)every) == 0;
            hipEvent_t t0 = nullptr;
            if (timed) t0 = e->prof.begin(s);
            launch_step_mt(th.d, B[i & 1], B[(i + 1) & 1], ec, sp, i == 0 ? 1 : 0, 0, s);
            if (timed) { e->prof.end(t0, RG_TREE, s); e->stats.tree_launches_timed += 1; }
            e->stats.tree_launches += 1;
            net_forward(e, net, B[(i + 1) & 1], rows_hint, s, rows_typ, timed);
        }
        launch_step_mt(th.d, B[steps & 1], B[(steps + 1) & 1], ec, sp, steps == 0 ? 1 : 0, 1, s);
        return;
    }
    int i = 0;
    // A search is a chain of num_sims x ~8 dependent launches; on small batches (the arena, single trees, the drain of a self-play
    // call) the HOST's launch calls, not the kernels, set its pace.  Every simulation step takes the same arguments (the batches
    // ping-pong with the step's parity), so S steps (S even) are captured once into a hipGraph and replayed; the graph is kept with
    // the tree arena and re-captured only when something that shapes a launch changes.
    const int S = e->search_graph;
    // ... and only there: on big batches the kernels set the pace, and a graph's power-of-two grids and tile estimates cost more than its
    // launches save (measured in round 4: 9.90 s per 65536-episode step with graphs everywhere, 9.64 s launched one by one)
    const int expect_rows = rows_typ > 0 ? std::min(rows_typ, rows_hint) : rows_hint;
    if (S >= 2 && !e->prof.on && num_sims >= S && expect_rows <= e->search_graph_rows) {
        // the grids' bound and the tile estimate in powers of two: the graph survives from move to move (a larger bound only adds
        // workgroups that exit at once; the estimate only picks tile families)
        auto pow2_up = [](int x, int cap) { int p = 1; while (p < x) p <<= 1; return p < cap ? p : cap; };
        rows_hint = pow2_up(rows_hint, th.d.G * T);
        if (rows_typ > 0) rows_typ = pow2_up(rows_typ, rows_hint);
        struct Key {
            const void *th, *conv, *ws, *stream, *root_states, *max_rows, *ec_key, *ec_stat, *log_state, *log_row;
            unsigned long long ec_tag, salt;
            int kind, rows_hint, rows_typ, S, dedup, block4, log_cap;
            uint32_t ec_bmask, ec_stones, max_depth, reserve_nodes;
            uint64_t model_gen;
            float cpuct;
            NetOptions opt;
        } k;
        std::memset(&k, 0, sizeof k);
        k.th = &th; k.conv = net.conv; k.ws = net.kind == AZ_NET_CONV ? workspace_for(e, s) : nullptr; k.stream = s; k.root_states = d_root_states;
        k.max_rows = d_max_rows; k.ec_key = ec.key; k.ec_stat = ec.stat; k.log_state = th.d.log_state; k.log_row = th.d.log_row;
        k.ec_tag = ec.tag; k.salt = net.salt; k.kind = net.kind; k.rows_hint = rows_hint; k.rows_typ = rows_typ; k.S = S; k.dedup = dedup ? 1 : 0;
        k.block4 = th.d.block4; k.log_cap = th.d.log_cap; k.ec_bmask = ec.bmask; k.ec_stones = ec.max_stones; k.max_depth = sp.max_depth; k.cpuct = sp.cpuct_f;
        k.opt = e->netopt;
        k.reserve_nodes = th.d.reserve_nodes;      // TreeDev travels by value into the captured launches: the capacity threshold is baked in
        k.model_gen = net.generation;              // a freed and re-created model may reuse the ConvNet's address: its weights' identity is the generation
        TreeHost::StepGraph& sg = th.step_graph;
        if (!sg.exec || sg.key.size() != sizeof k || std::memcmp(sg.key.data(), &k, sizeof k) != 0) {
            if (sg.exec) { HIPCHK(hipStreamSynchronize(s)); (void)hipGraphExecDestroy(sg.exec); sg.exec = nullptr; }
            if (net.kind == AZ_NET_CONV) convnet_prepare(workspace_for(e, s), e->netopt);      // nothing may allocate while the stream is capturing
            hipGraph_t g = nullptr;
            HIPCHK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
            try {
                for (int j = 0; j < S; ++j) {
                    launch_backup_select(th.d, B[j & 1], B[(j + 1) & 1], ec, sp, s);
                    net_forward(e, net, B[(j + 1) & 1], rows_hint, s, rows_typ, false);
                }
            } catch (...) {                            // never leave the engine's stream capturing: the next call on the handle would fail obscurely
                hipGraph_t dead = nullptr;
                (void)hipStreamEndCapture(s, &dead);
                if (dead) (void)hipGraphDestroy(dead);
                (void)hipGetLastError();
                throw;
            }
            HIPCHK(hipStreamEndCapture(s, &g));
            const hipError_t ie = hipGraphInstantiate(&sg.exec, g, nullptr, nullptr, 0);
            (void)hipGraphDestroy(g);
            if (ie != hipSuccess) { sg.exec = nullptr; throw HipFail{ie, "hipGraphInstantiate"}; }
            sg.key.assign((const unsigned char*)&k, (const unsigned char*)&k + sizeof k);
        }
        for (; i + S <= num_sims; i += S) {
            HIPCHK(hipGraphLaunch(sg.exec, s));
            e->stats.tree_launches += (uint64_t)S;
        }
    }
    for (; i < num_sims; ++i) {
        const bool timed = e->prof.on && (e->profile_tick++ % (uint64_t)every) == 0;
        hipEvent_t t0 = nullptr;
        if (timed) t0 = e->prof.begin(s);
        launch_backup_select(th.d, B[i & 1], B[(i + 1) & 1], ec, sp, s);
        if (timed) { e->prof.end(t0, RG_TREE, s); e->stats.tree_launches_timed += 1; }
        e->stats.tree_launches += 1;
        net_forward(e, net, B[(i + 1) & 1], rows_hint, s, rows_typ, timed);
    }
    launch_backup(th.d, B[num_sims & 1], ec, s);
}

void resolve_profile(az_engine* e) {
    if (!e->prof.on) return;
    double ms[RG_COUNT] = {0, 0};
    e->prof.resolve(ms);
    e->stats.tree_ms += ms[RG_TREE];
    for (NetWorkspace* w : e->ws) netws_resolve_profile(w, &e->netprof);
}

void fold_tree_totals(az_engine* e, const unsigned long long* h, bool dedup);
// fold the per-tree counters into the engine stats and clear them (k_harvest sums on the device)
void harvest_stats(az_engine* e, TreeHost& th, const NetModel& net) {
    harvest_dedup(e);
    const bool dedup = dedup_applies(e, net);
    launch_harvest(th.d, th.d_totals, th.d_counts, e->stream);
    unsigned long long h[ST_TOTALS];
    HIPCHK(hipMemcpyAsync(h, th.d_totals, sizeof h, hipMemcpyDeviceToHost, e->stream));
    HIPCHK(hipMemsetAsync(th.d_totals, 0, sizeof h, e->stream));
    HIPCHK(hipStreamSynchronize(e->stream));
    fold_tree_totals(e, h, dedup);
}
void fold_tree_totals(az_engine* e, const unsigned long long* h, bool dedup) {
    e->stats.simulations += h[ST_SIMS];
    e->stats.expansions += h[ST_EXPANSIONS];
    e->stats.leaf_evals += h[ST_LEAF_EVALS];
    if (!dedup) { e->stats.leaf_rows_requested += h[ST_LEAF_EVALS]; e->stats.leaf_rows_executed += h[ST_LEAF_EVALS]; }
    e->stats.link_hits += h[ST_LINK_HITS];
    e->stats.terminal_hits += h[ST_TERMINAL_HITS];
    e->stats.depth_sum += h[ST_DEPTH_SUM];
    e->stats.abandoned_sims += h[ST_ABANDONED];
    // algorithmic tree bytes per simulation, SURVEY.md 8(d): 128 per selection level + 356 per expansion
    e->stats.tree_bytes += 128.0 * (double)h[ST_DEPTH_SUM] + 356.0 * (double)h[ST_SIMS];
}

az_status report_tree_errors(az_engine* e, TreeHost& th, const uint32_t* h);
az_status check_tree_errors(az_engine* e, TreeHost& th) {
    uint32_t h[ERR_COUNT];
    HIPCHK(hipMemcpy(h, th.d.err, sizeof h, hipMemcpyDeviceToHost));
    return report_tree_errors(e, th, h);
}
az_status report_tree_errors(az_engine* e, TreeHost& th, const uint32_t* h) {
    if (h[ERR_CAPACITY] || h[ERR_HASH_FULL] || h[ERR_PATH] || h[ERR_TERMINAL_ROOT])
        HIPCHK(hipMemset(th.d.err, 0, ERR_COUNT * sizeof(uint32_t)));
    if (h[ERR_CAPACITY]) return fail(e, AZ_ERR_CAPACITY, "node arena exhausted (reserve too small; src/node.rs:237)");
    if (h[ERR_HASH_FULL]) return fail(e, AZ_ERR_CAPACITY, "transposition table full");
    if (h[ERR_PATH]) return fail(e, AZ_ERR_CAPACITY, "node_path overflow");
    if (h[ERR_TERMINAL_ROOT]) return fail(e, AZ_ERR_TERMINAL_ROOT, "get_action_prob on a finished game (src/async_mcts.rs:85)");
    return AZ_OK;
}

az_status find_net(az_engine* e, int model_id, NetModel** out) {
    auto it = e->nets.find(model_id);
    if (it == e->nets.end() || it->second.kind < 0) return fail(e, AZ_ERR_NO_MODEL, "model id not initialised");
    *out = &it->second;
    return AZ_OK;
}
// the conv net's activation workspace is sized by az_config.max_batch: a tree batch may not exceed it
az_status check_batch(az_engine* e, const NetModel& net, int trees) {
    if (net.kind == AZ_NET_CONV && trees > e->cfg.max_batch)
        return fail(e, AZ_ERR_BAD_ARGUMENT, "concurrent trees exceed az_config.max_batch (conv net workspace)");
    return AZ_OK;
}

// Is a caller-supplied pointer device (or managed) memory?  Plain host memory is unknown to the runtime: the query fails and the
// sticky error is cleared.
bool on_device(const void* p) {
    if (!p) return false;
    hipPointerAttribute_t a{};
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged;
}

// links and child blocks are 20-bit fields of the 16-byte node record (az_tree.h): a tree holds at most 2^20 slots
az_status check_tree_slots(az_engine* e, uint64_t blocks) {
    if (blocks * (uint64_t)BLOCK_SLOTS > (uint64_t)MAX_TREE_SLOTS)
        return fail(e, AZ_ERR_CAPACITY, "a tree of more than 2^20 node slots: num_sims x plies exceeds what the node record's 20-bit links address");
    return AZ_OK;
}

struct ScopedTimer {
    az_engine* e;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    ~ScopedTimer() {
        e->stats.device_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    }
};

}  // namespace

// =============================================================================================
extern "C" {

az_status az_create(const az_config* cfg, az_engine** out) {
    if (!out) return AZ_ERR_BAD_ARGUMENT;
    *out = nullptr;
    std::unique_ptr<az_engine> e(new az_engine());
    if (cfg) e->cfg = *cfg;
    if (e->cfg.max_batch <= 0) e->cfg.max_batch = 8192;
    if (e->cfg.net_channels <= 0) e->cfg.net_channels = 512;
    if (e->cfg.game < 0 || e->cfg.game >= GAME_COUNT) return AZ_ERR_BAD_ARGUMENT;
    e->device = e->cfg.device;
    e->prof.on = e->cfg.profile != 0;
    try {
        int n = 0;
        HIPCHK(hipGetDeviceCount(&n));
        if (n <= 0 || e->device >= n) return AZ_ERR_HIP;
        HIPCHK(hipSetDevice(e->device));
        HIPCHK(hipStreamCreate(&e->stream));
    } catch (const HipFail&) {
        return AZ_ERR_HIP;
    }
    *out = e.release();
    return AZ_OK;
}

void az_destroy(az_engine* e) {
    if (!e) return;
    (void)hipSetDevice(e->device);
    (void)hipStreamSynchronize(e->stream);
    (void)az_selfplay_end(e);
    for (auto& kv : e->nets) if (kv.second.conv) convnet_destroy(kv.second.conv);
    for (NetWorkspace* w : e->ws) netws_destroy(w);
    if (e->cache_keys) (void)hipFree(e->cache_keys);
    if (e->cache_pv) (void)hipFree(e->cache_pv);
    if (e->comm && g_rccl.CommDestroy) { (void)g_rccl.CommDestroy(e->comm); e->comm = nullptr; }
    e->tree_pool.clear();
    trainer_destroy(e->trainer);
    { double ms[RG_COUNT] = {0, 0}; e->prof.resolve(ms); }
    (void)hipStreamDestroy(e->stream);
    delete e;
}

const char* az_last_error(const az_engine* e) { return e ? e->err.c_str() : "null engine"; }

az_status az_set_option(az_engine* e, const char* key, int64_t value) {
    if (!e || !key) return AZ_ERR_BAD_ARGUMENT;
    auto is = [&](const char* k) { return std::strcmp(key, k) == 0; };
    // ---- options of the shipped library: every one of them lives in THIS engine ----
    if (is("conv2_table") && (value == 0 || value == 1)) { e->netopt.conv2_table = (int)value; return AZ_OK; }
    if (is("conv3_small") && (value == 0 || value == 1)) { e->netopt.conv3_small = (int)value; return AZ_OK; }
    if (is("conv3_tail") && (value == 0 || value == 1)) { e->netopt.conv3_tail = (int)value; return AZ_OK; }
    if (is("ring_packed") && (value == 0 || value == 1)) { e->netopt.ring_packed = (int)value; return AZ_OK; }
    if (is("conv3_planes") && (value == 0 || value == 1)) { e->netopt.conv3_planes = (int)value; return AZ_OK; }
    if (is("narrow_rows") && value >= 0 && value <= 65536) { e->netopt.narrow_rows = (int)value; return AZ_OK; }
    if (is("tree_block4") && (value == 0 || value == 1)) { e->tree_block4 = (int)value; return AZ_OK; }
    if (is("dedup_stats") && (value == 0 || value == 1)) { e->dedup_stats = (int)value; return AZ_OK; }
    if (is("profile_every") && value >= 1 && value <= 1000000) { e->profile_every = (int)value; return AZ_OK; }
    // the HIP-event brackets of az_config.profile, switched between calls: a timed region runs un-bracketed (its search loop as
    // hipGraph launches), a separate pass of the same call with the brackets on gives the per-kernel times
    if (is("profile") && (value == 0 || value == 1)) { e->prof.on = value != 0; return AZ_OK; }
    if (is("search_graph") && value >= 0 && value <= 1000 && value % 2 == 0) { e->search_graph = (int)value; return AZ_OK; }
    if (is("search_graph_rows") && value >= 0 && value <= 1000000) { e->search_graph_rows = (int)value; return AZ_OK; }
    if (is("selfplay_async") && (value == 0 || value == 1)) { e->selfplay_async = (int)value; return AZ_OK; }
    if (is("selfplay_async_launches") && value >= 1 && value <= 8) { e->selfplay_async_launches = (int)value; return AZ_OK; }
    if (is("selfplay_async_iters") && value >= 2 && value <= 64) { e->selfplay_async_iters = (int)value; return AZ_OK; }
    if (is("fused_search") && (value == 0 || value == 1)) { e->fused_search = (int)value; return AZ_OK; }
    if (is("eval_dedup") && value >= 0 && value <= 2) { e->eval_dedup = (int)value; return AZ_OK; }
    if (is("eval_cache_log2") && (value == 0 || (value >= 10 && value <= 30))) { e->eval_cache_log2 = (int)value; return AZ_OK; }
    if (is("eval_cache_max_stones") && value >= 0 && value <= 42) { e->eval_cache_max_stones = (int)value; return AZ_OK; }
    if (is("eval_cache_persist") && (value == 0 || value == 1)) { e->eval_cache_persist = (int)value; return AZ_OK; }
    // NNet::train hyper-parameters (defaults = connect_four_net.py:13-15, :21)
    if (is("train_epochs") && value >= 1 && value <= 100000) { e->train_epochs = (int)value; return AZ_OK; }
    if (is("train_batch") && value >= 2 && value <= TRAIN_MAX_BATCH) { e->train_batch = (int)value; return AZ_OK; }
    if (is("train_seed")) { e->train_seed = (uint64_t)value; return AZ_OK; }
    if (is("train_graph") && (value == 0 || value == 1)) { e->train_graph = (int)value; if (e->trainer) trainer_set_graph(e->trainer, value != 0); return AZ_OK; }
    if (is("train_gemm3_ring") && (value == 0 || value == 1)) { e->train_gemm3_ring = (int)value; if (e->trainer) trainer_set_gemm3_ring(e->trainer, value != 0); return AZ_OK; }
    if (is("train_fwd_x3") && (value == 0 || value == 1)) { e->train_fwd_x3 = (int)value; if (e->trainer) trainer_set_fwd_x3(e->trainer, value != 0); return AZ_OK; }
    if (is("train_wgrad_tr") && (value == 0 || value == 1)) { e->train_wgrad_tr = (int)value; if (e->trainer) trainer_set_wgrad_tr(e->trainer, value != 0); return AZ_OK; }
    if (is("train_implicit") && (value == 0 || value == 1)) { e->train_implicit = (int)value; if (e->trainer) trainer_set_implicit(e->trainer, value != 0); return AZ_OK; }
    if (is("train_fork") && (value == 0 || value == 1)) {
        e->train_fork = (int)value;
        if (e->trainer) { (void)hipSetDevice(e->device); trainer_set_fork(e->trainer, value != 0); }
        return AZ_OK;
    }
    if (is("train_gemm") && (value == 0 || value == 1)) { e->train_gemm = (int)value; if (e->trainer) trainer_set_gemm(e->trainer, (int)value); return AZ_OK; }
    if (is("train_fwd_dma") && (value == 0 || value == 1)) { e->train_fwd_dma = (int)value; if (e->trainer) trainer_set_fwd_dma(e->trainer, value != 0); return AZ_OK; }
    if (is("train_lr_e9") && value > 0) { e->hyper.lr = (float)((double)value * 1e-9); return AZ_OK; }
    if (is("train_dropout_e6") && value >= 0 && value < 1000000) { e->hyper.dropout = (float)((double)value * 1e-6); return AZ_OK; }
#ifdef AZ_DIAG
    // ---- libaz_engine_diag.so only: superseded kernel generations, forced tiles, clock-stamp builds, timing ablations (WRONG results) ----
    NetOptions& o = e->netopt;
    if (is("conv2_table") && value == 2) { o.conv2_table = 2; return AZ_OK; }
    if (is("gemm_variant") && (value == 0 || value == 1 || value == 2 || value == 3 || value == 5 || (value >= 11 && value <= 17))) { o.gemm_variant = (int)value; return AZ_OK; }
    if (is("fc_ring") && value >= 0 && value <= 3) { o.fc_ring = (int)value; return AZ_OK; }
    if (is("ring_tile") && value >= 0 && value < 60000) { const int l = (int)(value / 10000); if (l >= 3 && l <= 5) o.ring_tile[l] = (int)(value % 10000); return AZ_OK; }
    if (is("conv3_ring") && value >= 0 && value <= 3) { o.conv3_ring = (int)value; return AZ_OK; }
    if (is("conv2_pipe") && (value == 0 || value == 1)) { o.conv2_pipe = (int)value; return AZ_OK; }
    if (is("conv3_pipe") && ((value >= 0 && value <= 3) || (value >= 9 && value <= 15))) { o.conv3_pipe = (int)value; return AZ_OK; }
    if (is("conv3_pp") && (value == 0 || value == 1 || (value >= 16 && value <= 40))) { o.conv3_pp = (int)value; return AZ_OK; }
    if (is("conv1_table") && (value == 0 || value == 1)) { o.conv1_table = (int)value; return AZ_OK; }
    if (is("conv4_big") && value >= 0 && value <= 2) { o.conv4_big = (int)value; return AZ_OK; }
    if (is("tree_stamps") && (value == 0 || value == 1)) { return tree_set_stamps((int)value) ? AZ_OK : fail(e, AZ_ERR_HIP, "tree_set_stamps"); }
    if (is("print_clock_stamps")) {
        // gemm_variant 13: median in-kernel clock of conv2's K loop for model `value`
        auto it = e->nets.find((int)value);
        std::vector<unsigned long long> st(2048);
        if (it == e->nets.end() || !it->second.conv || !netws_read_clock_stamps(e->ws[0], st.data()))
            return fail(e, AZ_ERR_BAD_ARGUMENT, "no conv net under that model id");
        std::vector<double> mhz;
        for (int i = 0; i < 1024; ++i) if (st[2 * i + 1]) mhz.push_back(100.0 * (double)st[2 * i] / (double)st[2 * i + 1]);
        std::sort(mhz.begin(), mhz.end());
        if (mhz.empty()) return fail(e, AZ_ERR_BAD_ARGUMENT, "no stamps (run a forward with gemm_variant 13 first)");
        char buf[160];
        std::snprintf(buf, sizeof buf, "clock MHz min %.0f median %.0f max %.0f over %zu blocks", mhz.front(), mhz[mhz.size() / 2], mhz.back(), mhz.size());
        e->err = buf;       // returned through az_last_error
        return AZ_OK;
    }
    if (is("print_seg_stamps")) {
        // conv3_pipe 3: per-segment cycle sums of wave 0 of the first 128 workgroups of the last conv3 launch, median over blocks
        std::vector<unsigned long long> st(2048);
        if (!e->ws[0] || !netws_read_clock_stamps(e->ws[0], st.data())) return fail(e, AZ_ERR_BAD_ARGUMENT, "no workspace");
        std::string out;
        for (int i = 0; i < 10; ++i) {
            std::vector<unsigned long long> v;
            for (int b = 0; b < 128; ++b) if (st[16 * b + 8]) v.push_back(st[16 * b + i]);
            if (v.empty()) return fail(e, AZ_ERR_BAD_ARGUMENT, "no stamps (run a forward with conv3_pipe 3 first)");
            std::sort(v.begin(), v.end());
            char buf[64];
            std::snprintf(buf, sizeof buf, "%s%llu", i ? " " : "", v[v.size() / 2]);
            out += buf;
        }
        e->err = out;
        return AZ_OK;
    }
    if (is("print_pp_stamps")) {
        // conv3_pp 36 / 37: per-segment cycle sums of waves 0 (group 0) and 4 (group 1) of the first 128 workgroups, median over blocks
        std::vector<unsigned long long> st(2048);
        if (!e->ws[0] || !netws_read_clock_stamps(e->ws[0], st.data())) return fail(e, AZ_ERR_BAD_ARGUMENT, "no workspace");
        std::string out;
        for (int i = 0; i < 16; ++i) {
            std::vector<unsigned long long> v;
            for (int b = 0; b < 128; ++b) if (st[16 * b + 7]) v.push_back(st[16 * b + i]);
            if (v.empty()) return fail(e, AZ_ERR_BAD_ARGUMENT, "no stamps (run a forward with conv3_pp 36 first)");
            std::sort(v.begin(), v.end());
            char buf[64];
            std::snprintf(buf, sizeof buf, "%s%llu", i ? " " : "", v[v.size() / 2]);
            out += buf;
        }
        e->err = out;
        return AZ_OK;
    }
    if (is("print_tree_stamps")) {
        // "tree_stamps" = 1: cycles per phase of the last k_backup_select launch, median over its waves:
        // load head+path | backup | wait for its stores | select | leaf request | store head+path | whole kernel
        std::vector<unsigned long long> st(4096 * 8);
        if (!tree_read_stamps(st.data())) return fail(e, AZ_ERR_BAD_ARGUMENT, "no stamps (set tree_stamps 1 and run a search first)");
        std::string out;
        for (int i = 0; i < 7; ++i) {
            std::vector<unsigned long long> v;
            for (int w = 0; w < 4096; ++w) if (st[(size_t)w * 8 + 6]) v.push_back(st[(size_t)w * 8 + i]);
            if (v.empty()) return fail(e, AZ_ERR_BAD_ARGUMENT, "no stamps yet");
            std::sort(v.begin(), v.end());
            char buf[96];
            std::snprintf(buf, sizeof buf, "%s%llu/%llu/%llu", i ? " " : "", v[v.size() / 10], v[v.size() / 2], v[v.size() * 9 / 10]);
            out += buf;
        }
        e->err = out;
        return AZ_OK;
    }
#else
    // the diagnostic keys exist in libaz_engine_diag.so only; their DEFAULT value is accepted here so that a caller resetting them is not an error
    static const struct { const char* k; int64_t dflt; } diag_keys[] = {{"gemm_variant", 5}, {"fc_ring", 1}, {"conv3_ring", 0}, {"conv2_pipe", 1},
                                                                         {"conv3_pipe", 1}, {"conv1_table", 1}, {"conv4_big", 0}, {"tree_stamps", 0}, {"conv3_pp", 0}};
    bool diag_key = is("print_clock_stamps") || is("print_seg_stamps") || is("print_pp_stamps") || is("print_tree_stamps") || (is("conv2_table") && value == 2);
    if (is("ring_tile")) { if (value >= 30000 && value < 60000 && value % 10000 == 0) return AZ_OK; diag_key = true; }
    for (const auto& dk : diag_keys)
        if (is(dk.k)) { if (value == dk.dflt) return AZ_OK; diag_key = true; }
    if (diag_key)
        return fail(e, AZ_ERR_BAD_ARGUMENT, std::string(key) + ": diagnostic option or value (superseded kernels, timing ablations and clock stamps live in libaz_engine_diag.so)");
#endif
    return fail(e, AZ_ERR_BAD_ARGUMENT, std::string("unknown option or value: ") + key);
}

az_status az_get_stats(az_engine* e, az_stats* out) {
    if (!e || !out) return AZ_ERR_BAD_ARGUMENT;
    *out = e->stats;
    out->net_launches = e->netprof.launches;
    out->net_conv2_ms = e->netprof.conv2_ms;
    out->net_conv2_flops = e->netprof.conv2_flops;
    out->net_total_ms = e->netprof.total_ms;
    out->net_total_flops = e->netprof.total_flops;
    out->net_conv3_ms = e->netprof.conv3_ms;
    out->net_conv3_flops = e->netprof.conv3_flops;
    out->net_conv2_bytes = e->netprof.conv2_bytes;
    out->net_conv4_ms = e->netprof.conv4_ms;
    out->net_conv4_flops = e->netprof.conv4_flops;
    out->net_fc_ms = e->netprof.fc_ms;
    out->net_fc_flops = e->netprof.fc_flops;
    out->net_rows_timed = e->netprof.rows;
    out->tree_arena_allocs = e->tree_pool_allocs;
    {
        unsigned long long acct[2] = {0, 0};
        (void)hipSetDevice(e->device);
        for (NetWorkspace* w : e->ws) if (w && !netws_conv3_accounting(w, acct, false)) return fail(e, AZ_ERR_HIP, "az_get_stats: reading the conv3 accounting");
        out->net_conv3_image_rows = acct[0];
        out->net_conv3_image_launches = acct[1];
    }
    return AZ_OK;
}
az_status az_reset_stats(az_engine* e) {
    if (!e) return AZ_ERR_BAD_ARGUMENT;
    e->stats = az_stats{};
    e->netprof = NetProfile{};
    unsigned long long sink[2] = {0, 0};
    (void)hipSetDevice(e->device);
    for (NetWorkspace* w : e->ws) if (w && !netws_conv3_accounting(w, sink, true)) return fail(e, AZ_ERR_HIP, "az_reset_stats: clearing the conv3 accounting");
    return AZ_OK;
}

// ---- NNet ------------------------------------------------------------------------------------
az_status az_net_set_kind(az_engine* e, int32_t model_id, az_net_kind kind, uint64_t salt) {
    if (!e || (kind != AZ_NET_STUB && kind != AZ_NET_HASH)) return fail(e, AZ_ERR_BAD_ARGUMENT, "kind must be STUB or HASH");
    NetModel& m = e->nets[model_id];
    m.kind = kind;
    // two hash nets with the same salt but different model ids differ (oracle: HashNet::predict)
    m.salt = salt + (uint64_t)model_id * 0x51ED27ull;
    m.cache_tag = 0;
    m.generation = ++g_model_generation;
    return AZ_OK;
}

static az_status ensure_conv(az_engine* e, int32_t model_id, NetModel** out) {
    NetModel& m = e->nets[model_id];
    if (!m.conv) {
        const char* why = nullptr;
        m.conv = convnet_create(e->cfg.net_channels, &why);
        if (!m.conv) return fail(e, AZ_ERR_HIP, why ? why : "convnet_create failed");
    }
    *out = &m;
    return AZ_OK;
}

az_status az_net_free(az_engine* e, int32_t model_id) {
    if (!e) return AZ_ERR_BAD_ARGUMENT;
    auto it = e->nets.find(model_id);
    if (it == e->nets.end()) return fail(e, AZ_ERR_NO_MODEL, "model id not initialised");
    (void)hipSetDevice(e->device);
    (void)hipStreamSynchronize(e->stream);
    if (it->second.conv) convnet_destroy(it->second.conv);
    e->nets.erase(it);          // its evaluation-cache entries die with its tag
    return AZ_OK;
}

az_status az_net_init_random(az_engine* e, int32_t model_id, uint64_t seed) {
    if (!e) return AZ_ERR_BAD_ARGUMENT;
    try {
        HIPCHK(hipSetDevice(e->device));
        NetModel* m;
        az_status st = ensure_conv(e, model_id, &m);
        if (st) return st;
        convnet_init_random(m->conv, seed);
        m->kind = AZ_NET_CONV;
        m->cache_tag = 0;
        m->generation = ++g_model_generation;
        return AZ_OK;
    } catch (const HipFail& f) { return fail_hip(e, f); }
}

int64_t az_net_param_count(const az_engine* e) { return e ? convnet_param_count(e->cfg.net_channels) : 0; }

az_status az_net_set_params(az_engine* e, int32_t model_id, const float* params, int64_t n) {
    if (!e || !params) return AZ_ERR_BAD_ARGUMENT;
    if (n != convnet_param_count(e->cfg.net_channels)) return fail(e, AZ_ERR_BAD_ARGUMENT, "parameter count mismatch");
    try {
        HIPCHK(hipSetDevice(e->device));
        NetModel* m;
        az_status st = ensure_conv(e, model_id, &m);
        if (st) return st;
        if (!convnet_set_params(m->conv, params, n)) return fail(e, AZ_ERR_HIP, "convnet_set_params failed");
        m->kind = AZ_NET_CONV;
        m->cache_tag = 0;
        m->generation = ++g_model_generation;
        return AZ_OK;
    } catch (const HipFail& f) { return fail_hip(e, f); }
}

az_status az_net_get_params(az_engine* e, int32_t model_id, float* params, int64_t n) {
    if (!e || !params) return AZ_ERR_BAD_ARGUMENT;
    NetModel* m;
    az_status st = find_net(e, model_id, &m);
    if (st) return st;
    if (m->kind != AZ_NET_CONV) return fail(e, AZ_ERR_BAD_ARGUMENT, "model has no parameters");
    if (!convnet_get_params(m->conv, params, n)) return fail(e, AZ_ERR_BAD_ARGUMENT, "parameter count mismatch");
    return AZ_OK;
}

az_status az_net_save(az_engine* e, int32_t model_id, const char* path) {
    if (!e || !path) return AZ_ERR_BAD_ARGUMENT;
    int64_t n = az_net_param_count(e);
    std::vector<float> p((size_t)n);
    az_status st = az_net_get_params(e, model_id, p.data(), n);
    if (st) return st;
    FILE* f = std::fopen(path, "wb");
    if (!f) return fail(e, AZ_ERR_IO, std::string("cannot write ") + path);
    const char magic[8] = {'A', 'Z', 'N', 'E', 'T', '0', '0', '1'};
    int64_t hdr[2] = {(int64_t)e->cfg.net_channels, n};
    bool ok = std::fwrite(magic, 1, 8, f) == 8 && std::fwrite(hdr, sizeof(int64_t), 2, f) == 2 &&
              std::fwrite(p.data(), sizeof(float), (size_t)n, f) == (size_t)n;
    std::fclose(f);
    return ok ? AZ_OK : fail(e, AZ_ERR_IO, "short write");
}

az_status az_net_load(az_engine* e, int32_t model_id, const char* path) {
    if (!e || !path) return AZ_ERR_BAD_ARGUMENT;
    FILE* f = std::fopen(path, "rb");
    if (!f) return fail(e, AZ_ERR_IO, std::string("cannot read ") + path);
    char magic[8];
    int64_t hdr[2];
    int64_t n = az_net_param_count(e);
    std::vector<float> p((size_t)n);
    bool ok = std::fread(magic, 1, 8, f) == 8 && std::memcmp(magic, "AZNET001", 8) == 0 &&
              std::fread(hdr, sizeof(int64_t), 2, f) == 2 && hdr[0] == e->cfg.net_channels && hdr[1] == n &&
              std::fread(p.data(), sizeof(float), (size_t)n, f) == (size_t)n;
    std::fclose(f);
    if (!ok) return fail(e, AZ_ERR_IO, "bad or mismatching weights file");
    return az_net_set_params(e, model_id, p.data(), n);
}

az_status az_net_predict_states(az_engine* e, int32_t model_id, const uint64_t* states, int32_t B, float* pi, float* v) {
    if (!e || !states || !pi || !v || B < 0) return AZ_ERR_BAD_ARGUMENT;
    NetModel* m;
    az_status st = find_net(e, model_id, &m);
    if (st) return st;
    ScopedTimer timer{e};
    try {
        HIPCHK(hipSetDevice(e->device));
        DeviceMem mem;
        const int chunk = e->cfg.max_batch;
        EvalBatch eb{};
        eb.cap = chunk;
        eb.n = mem.alloc<uint32_t>(1);
        eb.state = mem.alloc<ulonglong2>(chunk);
        eb.pi = mem.alloc<float>((size_t)chunk * 8);
        eb.v = mem.alloc<float>(chunk);
        std::vector<float> hp((size_t)chunk * 8);
        for (int b0 = 0; b0 < B; b0 += chunk) {
            uint32_t nb = (uint32_t)std::min(chunk, B - b0);
            HIPCHK(hipMemcpyAsync(eb.n, &nb, sizeof nb, hipMemcpyHostToDevice, e->stream));
            HIPCHK(hipMemcpyAsync(eb.state, states + 2 * (size_t)b0, (size_t)nb * 16, hipMemcpyDefault, e->stream));
            net_forward(e, *m, eb, (int)nb, e->stream);
            HIPCHK(hipMemcpyAsync(hp.data(), eb.pi, (size_t)nb * 8 * sizeof(float), hipMemcpyDeviceToHost, e->stream));
            HIPCHK(hipStreamSynchronize(e->stream));
            std::vector<float> packed((size_t)nb * 7);
            for (uint32_t i = 0; i < nb; ++i)
                for (int a = 0; a < 7; ++a) packed[(size_t)i * 7 + a] = hp[(size_t)i * 8 + a];
            HIPCHK(hipMemcpy(pi + (size_t)b0 * 7, packed.data(), packed.size() * sizeof(float), hipMemcpyDefault));
            HIPCHK(hipMemcpy(v + b0, eb.v, (size_t)nb * sizeof(float), hipMemcpyDefault));
        }
        resolve_profile(e);
        return AZ_OK;
    } catch (const HipFail& f) { return fail_hip(e, f); }
}

az_status az_net_predict(az_engine* e, int32_t model_id, const float* boards, int32_t B, float* pi, float* v) {
    if (!e || !boards || B < 0) return AZ_ERR_BAD_ARGUMENT;
    try {
        // [B,2,6,7] f32 planes -> canonical bitboards (the planes are 0/1 by contract, connect_four_game.rs:227-233)
        std::vector<float> hb((size_t)B * AZ_FEATURES);
        HIPCHK(hipMemcpy(hb.data(), boards, hb.size() * sizeof(float), hipMemcpyDefault));
        std::vector<uint64_t> states((size_t)B * 2, 0);
        for (int b = 0; b < B; ++b)
            for (int r = 0; r < 6; ++r)
                for (int c = 0; c < 7; ++c) {
                    uint64_t bit = 1ull << (c * 7 + (5 - r));
                    if (hb[(size_t)b * 84 + r * 7 + c] != 0.0f) states[2 * b] |= bit;
                    if (hb[(size_t)b * 84 + 42 + r * 7 + c] != 0.0f) states[2 * b + 1] |= bit;
                }
        return az_net_predict_states(e, model_id, states.data(), B, pi, v);
    } catch (const HipFail& f) { return fail_hip(e, f); }
}

// ---- NNet::train, src/nnet.rs:38 ----------------------------------------------------------------
az_status az_net_train_begin(az_engine* e, int32_t previous_model_id) {
    if (!e) return AZ_ERR_BAD_ARGUMENT;
    NetModel* m;
    az_status st = find_net(e, previous_model_id, &m);
    if (st) return st;
    if (m->kind != AZ_NET_CONV) return fail(e, AZ_ERR_BAD_ARGUMENT, "az_net_train: the previous model has no parameters");
    try {
        HIPCHK(hipSetDevice(e->device));
        if (!e->trainer) {
            const char* err = nullptr;
            e->trainer = trainer_create(e->cfg.net_channels, &err);
            if (!e->trainer) return fail(e, AZ_ERR_HIP, err ? err : "trainer_create failed");
            trainer_set_graph(e->trainer, e->train_graph != 0);
            trainer_set_gemm(e->trainer, e->train_gemm);
            trainer_set_fork(e->trainer, e->train_fork != 0);
            trainer_set_implicit(e->trainer, e->train_implicit != 0);
            trainer_set_wgrad_tr(e->trainer, e->train_wgrad_tr != 0);
            trainer_set_fwd_x3(e->trainer, e->train_fwd_x3 != 0);
            trainer_set_gemm3_ring(e->trainer, e->train_gemm3_ring != 0);
            trainer_set_fwd_dma(e->trainer, e->train_fwd_dma != 0);
        }
        const int64_t n = az_net_param_count(e);
        std::vector<float> p((size_t)n);
        if (!convnet_get_params(m->conv, p.data(), n)) return fail(e, AZ_ERR_BAD_ARGUMENT, "parameter count mismatch");
        if (!trainer_set_params(e->trainer, p.data(), n)) return fail(e, AZ_ERR_HIP, "trainer_set_params failed");
        e->train_open = true;
        return AZ_OK;
    } catch (const HipFail& f) { return fail_hip(e, f); }
}

az_status az_net_train_step(az_engine* e, const float* boards, const float* pis, const float* vs, int32_t b, uint64_t mask_seed,
                            int32_t apply, float* loss_out, float* grads_out) {
    if (!e || !boards || !pis || !vs) return AZ_ERR_BAD_ARGUMENT;
    if (!e->train_open) return fail(e, AZ_ERR_BAD_ARGUMENT, "az_net_train_step: call az_net_train_begin first");
    if (b < 2 || b > TRAIN_MAX_BATCH) return fail(e, AZ_ERR_BAD_ARGUMENT, "az_net_train_step: batch must be in [2, 256]");
    try {
        HIPCHK(hipSetDevice(e->device));
        Trainer* t = e->trainer;
        HIPCHK(hipMemcpyAsync(trainer_batch_boards(t), boards, (size_t)b * 84 * sizeof(float), hipMemcpyDefault, e->stream));
        HIPCHK(hipMemcpyAsync(trainer_batch_pis(t), pis, (size_t)b * 7 * sizeof(float), hipMemcpyDefault, e->stream));
        HIPCHK(hipMemcpyAsync(trainer_batch_vs(t), vs, (size_t)b * sizeof(float), hipMemcpyDefault, e->stream));
        if (!trainer_step(t, e->hyper, trainer_batch_boards(t), trainer_batch_pis(t), trainer_batch_vs(t), b, mask_seed, apply != 0,
                          e->stream))
            return fail(e, AZ_ERR_HIP, "trainer_step failed");
        double l[2];
        if (!trainer_read_losses(t, l, true, e->stream)) return fail(e, AZ_ERR_HIP, "trainer_read_losses failed");
        if (loss_out) { loss_out[0] = (float)l[0]; loss_out[1] = (float)l[1]; }
        if (grads_out && !trainer_get_grads(t, grads_out, az_net_param_count(e), e->stream))
            return fail(e, AZ_ERR_HIP, "trainer_get_grads failed");
        return AZ_OK;
    } catch (const HipFail& f) { return fail_hip(e, f); }
}

az_status az_net_train_end(az_engine* e, int32_t model_id) {
    if (!e) return AZ_ERR_BAD_ARGUMENT;
    if (!e->train_open) return fail(e, AZ_ERR_BAD_ARGUMENT, "az_net_train_end: no open training session");
    const int64_t n = az_net_param_count(e);
    std::vector<float> p((size_t)n);
    if (!trainer_get_params(e->trainer, p.data(), n, e->stream)) return fail(e, AZ_ERR_HIP, "trainer_get_params failed");
    e->train_open = false;
    return az_net_set_params(e, model_id, p.data(), n);
}

az_status az_net_train(az_engine* e, int32_t previous_model_id, int32_t model_id, const float* boards, const float* pis,
                       const float* vs, int64_t n) {
    if (!e || !boards || !pis || !vs || n <= 0) return fail(e, AZ_ERR_BAD_ARGUMENT, "az_net_train: bad argument");
    az_status st = az_net_train_begin(e, previous_model_id);
    if (st) return st;
    ScopedTimer timer{e};
    try {
        DeviceMem mem;
        float* d_boards = mem.alloc<float>((size_t)n * 84);
        float* d_pis = mem.alloc<float>((size_t)n * 7);
        float* d_vs = mem.alloc<float>((size_t)n);
        HIPCHK(hipMemcpy(d_boards, boards, (size_t)n * 84 * sizeof(float), hipMemcpyDefault));
        HIPCHK(hipMemcpy(d_pis, pis, (size_t)n * 7 * sizeof(float), hipMemcpyDefault));
        HIPCHK(hipMemcpy(d_vs, vs, (size_t)n * sizeof(float), hipMemcpyDefault));
        const int b = e->train_batch;
        const int64_t steps = std::max<int64_t>(1, n / b);          // connect_four_net.py:127-130: len(examples) / batch_size
        int64_t* d_idx = mem.alloc<int64_t>((size_t)steps * b);
        std::vector<int64_t> idx((size_t)steps * b);
        Trainer* t = e->trainer;
        e->train_history.clear();
        uint64_t gstep = 0;
        for (int epoch = 0; epoch < e->train_epochs; ++epoch) {
            // batches are drawn with replacement (np.random.randint, :130), from the build's counter RNG
            for (int64_t sidx = 0; sidx < steps; ++sidx)
                for (int j = 0; j < b; ++j) {
                    const uint64_t r = rng_draw(e->train_seed, gstep + (uint64_t)sidx, (uint64_t)j, RNG_BATCH);
                    idx[(size_t)sidx * b + j] = (int64_t)(((unsigned __int128)r * (unsigned __int128)n) >> 64);
                }
            HIPCHK(hipMemcpyAsync(d_idx, idx.data(), idx.size() * sizeof(int64_t), hipMemcpyHostToDevice, e->stream));
            if (!trainer_run_epoch(t, e->hyper, d_boards, d_pis, d_vs, d_idx, steps, b, mix64(e->train_seed ^ 0xD6E8FEB86659FD93ull), gstep,
                                   e->stream))
                return fail(e, AZ_ERR_HIP, "trainer_run_epoch failed");
            gstep += (uint64_t)steps;
            double l[2];
            if (!trainer_read_losses(t, l, true, e->stream)) return fail(e, AZ_ERR_HIP, "trainer_read_losses failed");
            e->train_history.push_back((float)(l[0] / (double)steps));
            e->train_history.push_back((float)(l[1] / (double)steps));
        }
        HIPCHK(hipStreamSynchronize(e->stream));
    } catch (const HipFail& f) { e->train_open = false; return fail_hip(e, f); }
    return az_net_train_end(e, model_id);
}

int32_t az_net_train_history(const az_engine* e, float* out, int32_t cap_epochs) {
    if (!e) return 0;
    const int32_t epochs = (int32_t)(e->train_history.size() / 2);
    if (out)
        for (int32_t i = 0; i < std::min(epochs, cap_epochs); ++i) { out[2 * i] = e->train_history[2 * i]; out[2 * i + 1] = e->train_history[2 * i + 1]; }
    return epochs;
}

// ---- AsyncMcts -------------------------------------------------------------------------------
// num_sims % num_threads == 0 (assert!, src/async_mcts.rs:192); 0 means 1
static az_status check_threads(az_engine* e, int num_sims, int* num_threads) {
    if (*num_threads == 0) *num_threads = 1;
    if (*num_threads < 1 || *num_threads > MAX_SIM_THREADS) return fail(e, AZ_ERR_BAD_ARGUMENT, "num_threads must be in [1, 8]");
    if (num_sims % *num_threads != 0) return fail(e, AZ_ERR_BAD_ARGUMENT, "num_sims % num_threads != 0 (src/async_mcts.rs:192)");
    return AZ_OK;
}

az_status az_tree_create(az_engine* e, int32_t n_games, uint64_t reserve, int32_t num_sims, int32_t num_threads, int32_t max_depth,
                         int32_t model_id, int32_t cpuct, az_tree** out) {
    if (!e || !out || n_games <= 0 || num_sims <= 0 || reserve < 8 || max_depth < 0)
        return fail(e, AZ_ERR_BAD_ARGUMENT, "az_tree_create: bad argument");
    if (az_status st = check_threads(e, num_sims, &num_threads)) return st;
    if (n_games > 1024 * 64) return fail(e, AZ_ERR_BAD_ARGUMENT, "az_tree_create: at most 65536 trees per batch");
    *out = nullptr;
    try {
        HIPCHK(hipSetDevice(e->device));
        std::unique_ptr<az_tree> t(new az_tree());
        t->e = e;
        t->num_sims = num_sims; t->max_depth = max_depth; t->model_id = model_id; t->cpuct = cpuct;
        const uint64_t nodes = std::min<uint64_t>(reserve, reachable_slots(num_sims, AZ_MAX_PLIES));
        if (az_status cs = check_tree_slots(e, reachable_blocks(num_sims, AZ_MAX_PLIES, nodes))) return cs;
        t->th.create(n_games, reachable_blocks(num_sims, AZ_MAX_PLIES, nodes), nodes, hash_entries(num_sims, AZ_MAX_PLIES), num_threads, e->cfg.game);
        t->d_root_states = t->mem.alloc<ulonglong2>(n_games);
        {
            const size_t G = (size_t)n_games, head = (sizeof(CallReadback) + 63) / 64 * 64;
            HIPCHK(hipHostMalloc(&t->h_io, head + G * 16 + G * 7 * 4 * 2 + G * 7 * 2, hipHostMallocDefault));
            char* p = (char*)t->h_io;
            t->h_rb = (CallReadback*)p; p += head;
            t->h_states = (ulonglong2*)p; p += G * 16;
            t->h_pi = (float*)p; p += G * 7 * 4;
            t->h_q = (float*)p; p += G * 7 * 4;
            t->h_counts = (uint16_t*)p;
        }
        launch_reset_trees(t->th.d, nullptr, e->stream);
        HIPCHK(hipStreamSynchronize(e->stream));
        *out = t.release();
        return AZ_OK;
    } catch (const HipFail& f) { return fail_hip(e, f); }
}

void az_tree_destroy(az_tree* t) {
    if (!t) return;
    (void)hipSetDevice(t->e->device);
    (void)hipStreamSynchronize(t->e->stream);
    delete t;
}

az_status az_tree_reset(az_tree* t, const uint64_t* root_states) {
    if (!t) return AZ_ERR_BAD_ARGUMENT;
    az_engine* e = t->e;
    try {
        HIPCHK(hipSetDevice(e->device));
        const ulonglong2* roots = nullptr;
        if (root_states) {
            HIPCHK(hipMemcpyAsync(t->d_root_states, root_states, (size_t)t->th.d.G * 16, hipMemcpyDefault, e->stream));
            roots = t->d_root_states;
        }
        launch_reset_trees(t->th.d, nullptr, e->stream, roots);
        launch_harvest(t->th.d, t->th.d_totals, nullptr, e->stream);          // forget the old trees' counters
        HIPCHK(hipMemsetAsync(t->th.d_totals, 0, ST_TOTALS * sizeof(unsigned long long), e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
        return AZ_OK;
    } catch (const HipFail& f) { return fail_hip(e, f); }
}

az_status az_tree_record_evals(az_tree* t, int32_t cap) {
    if (!t || cap < 0) return AZ_ERR_BAD_ARGUMENT;
    try {
        HIPCHK(hipSetDevice(t->e->device));
        TreeDev& d = t->th.d;
        d.log_cap = cap;
        d.log_state = t->mem.alloc<ulonglong2>((size_t)d.G * cap);
        d.log_pi = t->mem.alloc<float>((size_t)d.G * cap * 7);
        d.log_v = t->mem.alloc<float>((size_t)d.G * cap);
        launch_reset_trees(d, nullptr, t->e->stream);      // the log starts with fresh trees (log_len = 0)
        HIPCHK(hipStreamSynchronize(t->e->stream));
        return AZ_OK;
    } catch (const HipFail& f) { return fail_hip(t->e, f); }
}

az_status az_tree_get_evals(az_tree* t, int32_t* rec_count, uint64_t* states, float* pis, float* vs) {
    if (!t) return AZ_ERR_BAD_ARGUMENT;
    try {
        HIPCHK(hipSetDevice(t->e->device));
        TreeDev& d = t->th.d;
        size_t n = (size_t)d.G * d.log_cap;
        if (rec_count) {
            std::vector<TreeLine> heads(d.G);
            HIPCHK(hipMemcpy(heads.data(), d.head, (size_t)d.G * sizeof(TreeLine), hipMemcpyDeviceToHost));
            std::vector<int32_t> cnt(d.G);
            for (int g = 0; g < d.G; ++g) cnt[g] = (int32_t)heads[g].head.log_len;
            HIPCHK(hipMemcpy(rec_count, cnt.data(), d.G * sizeof(int32_t), hipMemcpyDefault));
        }
        if (states) HIPCHK(hipMemcpy(states, d.log_state, n * 16, hipMemcpyDefault));
        if (pis) HIPCHK(hipMemcpy(pis, d.log_pi, n * 7 * sizeof(float), hipMemcpyDefault));
        if (vs) HIPCHK(hipMemcpy(vs, d.log_v, n * sizeof(float), hipMemcpyDefault));
        return AZ_OK;
    } catch (const HipFail& f) { return fail_hip(t->e, f); }
}

az_status az_tree_node_counts(az_tree* t, uint32_t* out) {
    if (!t || !out) return AZ_ERR_BAD_ARGUMENT;
    try {
        HIPCHK(hipSetDevice(t->e->device));
        std::vector<TreeLine> heads(t->th.d.G);
        HIPCHK(hipMemcpy(heads.data(), t->th.d.head, heads.size() * sizeof(TreeLine), hipMemcpyDeviceToHost));
        std::vector<uint32_t> cnt(heads.size());
        for (size_t g = 0; g < heads.size(); ++g) cnt[g] = heads[g].head.count;
        HIPCHK(hipMemcpy(out, cnt.data(), cnt.size() * sizeof(uint32_t), hipMemcpyDefault));
        return AZ_OK;
    } catch (const HipFail& f) { return fail_hip(t->e, f); }
}

az_status az_tree_get_action_prob(az_tree* t, const uint64_t* states, float temp, uint64_t seed, uint64_t first_game_id,
                                  float* pi, uint16_t* counts, float* q) {
    if (!t || !states || !pi) return AZ_ERR_BAD_ARGUMENT;
    az_engine* e = t->e;
    NetModel* net;
    az_status st = find_net(e, t->model_id, &net);
    if (st) return st;
    st = check_batch(e, *net, t->th.d.G * t->th.d.T);
    if (st) return st;
    ScopedTimer timer{e};
    try {
        HIPCHK(hipSetDevice(e->device));
        TreeDev& d = t->th.d;
        const int G = d.G;
        // include/az_engine.h: caller pointers may be host or device memory.  Host memory takes the pinned fast path (one stream
        // synchronisation per call); device memory is copied by the runtime.
        if (on_device(states)) {
            HIPCHK(hipMemcpyAsync(t->d_root_states, states, (size_t)G * 16, hipMemcpyDeviceToDevice, e->stream));
        } else {
            std::memcpy(t->h_states, states, (size_t)G * 16);
            HIPCHK(hipMemcpyAsync(t->d_root_states, t->h_states, (size_t)G * 16, hipMemcpyHostToDevice, e->stream));
        }
        launch_set_active(d, 1u, e->stream);
        SearchParams sp{(uint32_t)t->max_depth, (float)t->cpuct};
        prepare_cache(e, dedup_applies(e, *net), (uint64_t)G * ((uint64_t)t->num_sims + 1), e->stream);
        run_search(e, t->th, t->d_root_states, t->num_sims, sp, *net, G);
        launch_root_policy(d, temp, seed, first_game_id, t->h_pi, t->h_counts, t->h_q, e->stream);
        launch_harvest(d, t->th.d_totals, t->th.d_counts, e->stream);
        launch_call_readback(t->th.d_totals, e->cache.stat, d.err, t->h_rb, e->stream);
        HIPCHK(hipStreamSynchronize(e->stream));
        resolve_profile(e);
        fold_dedup(e, t->h_rb->dd);
        fold_tree_totals(e, t->h_rb->totals, dedup_applies(e, *net));
        e->stats.moves += (uint64_t)G;
        st = report_tree_errors(e, t->th, t->h_rb->err);
        if (st) return st;
        auto copy_out = [&](void* dst, const void* src, size_t bytes) {
            if (on_device(dst)) HIPCHK(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
            else std::memcpy(dst, src, bytes);
        };
        copy_out(pi, t->h_pi, (size_t)G * 7 * sizeof(float));
        if (counts) copy_out(counts, t->h_counts, (size_t)G * 7 * sizeof(uint16_t));
        if (q) copy_out(q, t->h_q, (size_t)G * 7 * sizeof(float));
        return AZ_OK;
    } catch (const HipFail& f) { return fail_hip(e, f); }
}

// ---- Coach::execute_episode x many -----------------------------------------------------------
// ---- Coach::execute_episode x many, as a SESSION: the slots stay full across the calls that fetch the episodes -------------------------
// az_selfplay_begin fixes the session's episodes (ids 0 .. n_games-1, slot refill in id order); az_selfplay_next(k) plays until the next k
// episodes IN ID ORDER are finished and emits their tuples -- the slots those episodes freed are already playing later ones, so a host
// that fetches its episodes in chunks pays the drain (the last few slots finishing alone: 6 % of a 65536-episode call) once per session
// instead of once per chunk.  An episode depends on (seed, first_game_id + index) and the net alone, so a chunk's tuples are the ones
// az_selfplay would return for the same episodes.  az_selfplay is begin + next(all) + end.
struct SelfplaySession {
    az_selfplay_params p{};
    int C = 0, T = 1, n_games = 0;
    NetModel* net = nullptr;
    uint64_t net_generation = 0;
    TreeLease lease;
    DeviceMem mem;
    ScopedEvalLog evlog;
    GamesDev gd{};
    SearchParams sp{};
    SelfplayMoveParams mp{};
    uint32_t* h_ctr = nullptr;           // pinned read-back of gd.counters
    int active = 0, rows_typ = 0;
    int delivered = 0;                   // episodes handed out by az_selfplay_next so far
    long long iter = 0;
    bool async_mode = false;
    // free-running mode (selfplay_async)
    EvalBatch B[2];
    EvalCache ec{};
    int fill = 0;
    long long step = 0;
    ~SelfplaySession() { if (h_ctr) (void)hipHostFree(h_ctr); }
};

static az_status selfplay_begin_impl(az_engine* e, const az_selfplay_params* p, std::unique_ptr<SelfplaySession>& out) {
    if (p->n_games <= 0 || p->num_sims <= 0 || p->max_depth < 0 || p->reserve < 8)
        return fail(e, AZ_ERR_BAD_ARGUMENT, "az_selfplay: bad argument");
    const int n_games = p->n_games;
    const int C = (p->concurrent <= 0 || p->concurrent > n_games) ? n_games : p->concurrent;
    if (C > 1024 * 64) return fail(e, AZ_ERR_BAD_ARGUMENT, "az_selfplay: at most 65536 concurrent games");
    NetModel* net;
    az_status st = find_net(e, p->model_id, &net);
    if (st) return st;
    int T = p->num_sim_threads;
    st = check_threads(e, p->num_sims, &T);
    if (st) return st;
    st = check_batch(e, *net, C * T);
    if (st) return st;
    HIPCHK(hipSetDevice(e->device));
    hipStream_t s = e->stream;
    const uint64_t nodes = std::min<uint64_t>(p->reserve, reachable_slots(p->num_sims, AZ_MAX_PLIES));
    if (az_status cs = check_tree_slots(e, reachable_blocks(p->num_sims, AZ_MAX_PLIES, nodes))) return cs;
    std::unique_ptr<SelfplaySession> ss(new SelfplaySession());
    ss->p = *p; ss->C = C; ss->T = T; ss->n_games = n_games; ss->net = net; ss->net_generation = net->generation;
    acquire_trees(e, ss->lease, C, reachable_blocks(p->num_sims, AZ_MAX_PLIES, nodes), nodes, hash_entries(p->num_sims, AZ_MAX_PLIES), T, s);
    TreeHost& th = *ss->lease;
    DeviceMem& mem = ss->mem;
    GamesDev& gd = ss->gd;
    gd.C = C;
    gd.n_games = n_games;
    gd.state = mem.alloc<ulonglong2>(C);
    gd.player = mem.alloc<int8_t>(C);
    gd.ply = mem.alloc<int32_t>(C);
    gd.gid = mem.alloc<int32_t>(C);
    gd.need_reset = mem.alloc<uint8_t>(C);
    const size_t ns = (size_t)n_games * 42;
    gd.smp_state = mem.alloc<ulonglong2>(ns);
    gd.smp_pi = mem.alloc<float>(ns * 7);
    gd.smp_player = mem.alloc<int8_t>(ns);
    gd.moves = mem.alloc<uint8_t>(ns);
    gd.g_len = mem.alloc<int32_t>(n_games);
    gd.g_result = mem.alloc<float>(n_games);
    gd.g_final_player = mem.alloc<int8_t>(n_games);
    gd.counters = mem.alloc<uint32_t>(4);
    if (p->record_evals > 0) {
        // per-EPISODE logs (row = the slot's current episode id), so they survive slot refills
        gd.g_log_len = mem.alloc<int32_t>(n_games);
        HIPCHK(hipMemset(gd.g_log_len, 0, n_games * sizeof(int32_t)));
        ss->evlog.attach(th, (size_t)n_games, p->record_evals, gd.gid);
    }
    {
        std::vector<int32_t> gid(C);
        std::vector<int8_t> pl(C, 1);
        for (int i = 0; i < C; ++i) gid[i] = i;
        HIPCHK(hipMemcpy(gd.gid, gid.data(), C * sizeof(int32_t), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(gd.player, pl.data(), C, hipMemcpyHostToDevice));
        HIPCHK(hipMemset(gd.state, 0, (size_t)C * 16));
        HIPCHK(hipMemset(gd.ply, 0, C * sizeof(int32_t)));
        HIPCHK(hipMemset(gd.need_reset, 0, C));
        HIPCHK(hipMemset(gd.moves, 0, ns));
        HIPCHK(hipMemset(gd.g_len, 0, n_games * sizeof(int32_t)));
        uint32_t ctr[4] = {(uint32_t)C, 0u, (uint32_t)C, 0u};
        HIPCHK(hipMemcpy(gd.counters, ctr, sizeof ctr, hipMemcpyHostToDevice));
    }
    launch_reset_trees(th.d, nullptr, s);
    prepare_cache(e, dedup_applies(e, *net), (uint64_t)n_games * AZ_MAX_PLIES * ((uint64_t)p->num_sims + 1), s);
    ss->sp = SearchParams{(uint32_t)p->max_depth, (float)p->cpuct};
    ss->mp = SelfplayMoveParams{p->seed, p->first_game_id, p->temp_threshold, C < n_games ? 1 : 0, 0, 0};
    HIPCHK(hipHostMalloc((void**)&ss->h_ctr, 4 * sizeof(uint32_t)));
    ss->active = C;                               // slots still playing (read back after every move)
    ss->rows_typ = 0;                             // expected rows per leaf batch (0 = unknown: assume `active`)
    // "selfplay_async": free-running slots (k_async_step) -- conv nets (anything that goes through leaf batches), one simulation in
    // flight per tree.  `launches` tree launches share one leaf batch, then the forward runs on it.
    ss->async_mode = e->selfplay_async != 0 && T == 1 && (net->kind == AZ_NET_CONV || dedup_applies(e, *net)) && p->num_sims >= 1;
    if (ss->async_mode) {
        gd.sims = mem.alloc<int32_t>(C);
        HIPCHK(hipMemsetAsync(gd.sims, 0xFF, (size_t)C * sizeof(int32_t), s));       // -1: no root prepared yet
        launch_selfplay_sync_active(th.d, gd, s);
        th.d.block4 = e->tree_block4;
        const bool dedup = dedup_applies(e, *net);
        ss->ec = cache_for(e, *net);
        ss->B[0] = th.eb; ss->B[1] = th.eb2;
        ss->B[0].dedup = ss->B[1].dedup = dedup ? 1 : 0;
        ss->B[0].max_n = ss->B[1].max_n = gd.counters + 3;
    }
    out = std::move(ss);
    return AZ_OK;
}

// play until every episode below `hi` has finished (the counter of finished episodes is restarted for the range [delivered, hi))
static az_status selfplay_run_until(az_engine* e, SelfplaySession& ss, int hi) {
    hipStream_t s = e->stream;
    TreeHost& th = *ss.lease;
    GamesDev& gd = ss.gd;
    NetModel* net = ss.net;
    const az_selfplay_params* p = &ss.p;
    const int C = ss.C, T = ss.T, n_games = ss.n_games, lo = ss.delivered;
    uint32_t* h_ctr = ss.h_ctr;
    ss.mp.done_lo = lo; ss.mp.done_hi = hi;
    launch_count_done(gd.g_len, lo, hi, gd.counters + 1, s);
    const uint32_t want = (uint32_t)(hi - lo);
    az_status result = AZ_OK;
    {   // a chunk whose episodes all finished while earlier ones were being waited for
        HIPCHK(hipMemcpyAsync(h_ctr, gd.counters, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        if (h_ctr[1] >= want) return AZ_OK;
    }
    if (ss.async_mode) {
        const int launches = std::max(1, e->selfplay_async_launches), iters = std::max(2, e->selfplay_async_iters);
        const int every = std::max(1, e->profile_every);
        // every leaf costs a forward at the latest `launches` launches after it was requested; a move needs num_sims + 2 leaves
        const long long cap_steps = (long long)(AZ_MAX_PLIES + 2) * (n_games / C + 2) * ((long long)p->num_sims + 3) + 64;
        for (;; ++ss.step) {
            const bool timed = e->prof.on && (e->profile_tick++ % (uint64_t)every) == 0;
            for (int j = 0; j < launches; ++j) {
                hipEvent_t t0 = nullptr;
                if (timed && j == 0) t0 = e->prof.begin(s);
                launch_async_step(th.d, gd, ss.B[ss.fill ^ 1], ss.B[ss.fill], ss.ec, ss.sp, ss.mp, p->num_sims, j == 0 ? 1 : 0, iters, s);
                if (timed && j == 0) { e->prof.end(t0, RG_TREE, s); e->stats.tree_launches_timed += 1; }
                e->stats.tree_launches += 1;
            }
            if (ss.mp.refill) launch_reset_trees(th.d, gd.need_reset, s);
            net_forward(e, *net, ss.B[ss.fill], ss.active, s, ss.rows_typ, timed);
            ss.fill ^= 1;
            if ((ss.step & 15) == 15) {                  // look at the counters now and then: finished? how many slots still play?
                HIPCHK(hipMemcpyAsync(h_ctr, gd.counters, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
                HIPCHK(hipStreamSynchronize(s));
                resolve_profile(e);
                ss.active = std::max(1, (int)h_ctr[2]);
                ss.rows_typ = h_ctr[3] ? (int)std::min<uint32_t>(h_ctr[3], (uint32_t)C) : 0;
                HIPCHK(hipMemsetAsync(gd.counters + 3, 0, sizeof(uint32_t), s));
                if (h_ctr[1] >= want) { ++ss.step; break; }
                result = check_tree_errors(e, th);
                if (result) break;
                if (ss.step > cap_steps) { result = fail(e, AZ_ERR_HIP, "az_selfplay: episode loop did not terminate"); break; }
            }
        }
        return result;
    }
    for (;; ++ss.iter) {
        launch_selfplay_sync_active(th.d, gd, s);
        // tile choice from the largest batch of the previous move (de-duplication makes batches much smaller than the
        // number of searching trees); the grids still cover `active`
        run_search(e, th, gd.state, p->num_sims, ss.sp, *net, ss.active, nullptr, ss.rows_typ, gd.counters + 3);
        launch_selfplay_move(th.d, gd, ss.mp, s);
        if (ss.mp.refill) launch_reset_trees(th.d, gd.need_reset, s);
        HIPCHK(hipMemcpyAsync(h_ctr, gd.counters, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        resolve_profile(e);
        ss.active = (int)h_ctr[2];
        ss.rows_typ = h_ctr[3] ? (int)std::min<uint32_t>(h_ctr[3], (uint32_t)(C * T)) : 0;      // this move's largest batch
        HIPCHK(hipMemsetAsync(gd.counters + 3, 0, sizeof(uint32_t), s));
        if (h_ctr[1] >= want) { ++ss.iter; break; }
        if ((ss.iter & 7) == 7 || h_ctr[2] == 0) {
            result = check_tree_errors(e, th);
            if (result) break;
        }
        if (h_ctr[2] == 0) { result = fail(e, AZ_ERR_HIP, "az_selfplay: every slot is idle before the requested episodes finished"); break; }
        if (ss.iter > (long long)AZ_MAX_PLIES * (n_games / C + 2) + 8) {
            result = fail(e, AZ_ERR_HIP, "az_selfplay: episode loop did not terminate");
            break;
        }
    }
    return result;
}

// tuples of the episodes [lo, hi) into `out` (game-id order then ply order)
static az_status selfplay_emit(az_engine* e, SelfplaySession& ss, int lo, int hi, az_samples* out) {
    hipStream_t s = e->stream;
    const az_selfplay_params* p = &ss.p;
    const GamesDev& gd = ss.gd;
    const int n = hi - lo, nsym = p->symmetries ? 2 : 1;
    std::vector<int32_t> glen((size_t)n);
    HIPCHK(hipMemcpy(glen.data(), gd.g_len + lo, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost));
    std::vector<int64_t> off((size_t)n);
    int64_t total = 0;
    for (int i = 0; i < n; ++i) { off[(size_t)i] = total; total += glen[(size_t)i]; }
    e->stats.games += (uint64_t)n;
    e->stats.moves += (uint64_t)total;
    e->stats.samples += (uint64_t)total;
    out->count = total * nsym;
    if (out->capacity < out->count) return fail(e, AZ_ERR_BAD_ARGUMENT, "az_selfplay: sample buffers too small");
    const size_t ns = (size_t)n * 42;
    if (out->pis && out->zs) {
        DeviceMem mem;
        int64_t* d_off = mem.alloc<int64_t>((size_t)n);
        HIPCHK(hipMemcpy(d_off, off.data(), (size_t)n * sizeof(int64_t), hipMemcpyHostToDevice));
        const size_t cnt = (size_t)out->count;
        ulonglong2* d_states = out->states ? mem.alloc<ulonglong2>(cnt) : nullptr;
        float* d_boards = out->boards ? mem.alloc<float>(cnt * 84) : nullptr;
        float* d_pis = mem.alloc<float>(cnt * 7);
        float* d_zs = mem.alloc<float>(cnt);
        GamesDev view = gd;          // the per-episode arrays of [lo, hi)
        view.n_games = n;
        view.smp_state += (size_t)lo * 42; view.smp_pi += (size_t)lo * 42 * 7; view.smp_player += (size_t)lo * 42; view.moves += (size_t)lo * 42;
        view.g_len += lo; view.g_result += lo; view.g_final_player += lo;
        launch_emit_samples(view, d_off, p->symmetries, d_states, d_boards, d_pis, d_zs, s);
        HIPCHK(hipStreamSynchronize(s));
        if (d_states) HIPCHK(hipMemcpy(out->states, d_states, cnt * 16, hipMemcpyDefault));
        if (d_boards) HIPCHK(hipMemcpy(out->boards, d_boards, cnt * 84 * sizeof(float), hipMemcpyDefault));
        HIPCHK(hipMemcpy(out->pis, d_pis, cnt * 7 * sizeof(float), hipMemcpyDefault));
        HIPCHK(hipMemcpy(out->zs, d_zs, cnt * sizeof(float), hipMemcpyDefault));
    }
    if (out->game_len) HIPCHK(hipMemcpy(out->game_len, glen.data(), (size_t)n * sizeof(int32_t), hipMemcpyDefault));
    if (out->moves) HIPCHK(hipMemcpy(out->moves, gd.moves + (size_t)lo * 42, ns, hipMemcpyDefault));
    return AZ_OK;
}

static az_status selfplay_next_impl(az_engine* e, SelfplaySession& ss, int n, az_samples* out) {
    if (ss.net->generation != ss.net_generation) return fail(e, AZ_ERR_BAD_ARGUMENT, "az_selfplay_next: the session's model was changed");
    const int lo = ss.delivered, hi = lo + n;
    az_status result = selfplay_run_until(e, ss, hi);
    harvest_stats(e, *ss.lease, *ss.net);
    if (result == AZ_OK) result = check_tree_errors(e, *ss.lease);
    if (result) return result;
    result = selfplay_emit(e, ss, lo, hi, out);
    if (result) return result;
    ss.delivered = hi;
    if (ss.p.record_evals > 0) {          // the whole session's log so far (rows of unfinished episodes are partial)
        e->sp_log_cap = ss.p.record_evals;
        e->sp_log_count.resize((size_t)ss.n_games);
        HIPCHK(hipMemcpy(e->sp_log_count.data(), ss.gd.g_log_len, (size_t)ss.n_games * sizeof(int32_t), hipMemcpyDeviceToHost));
        ss.evlog.copy_out(e->sp_log_states, e->sp_log_pi, e->sp_log_v);
    }
    return AZ_OK;
}

az_status az_selfplay_begin(az_engine* e, const az_selfplay_params* p) {
    if (!e || !p) return AZ_ERR_BAD_ARGUMENT;
    if (e->sp_session) return fail(e, AZ_ERR_BAD_ARGUMENT, "az_selfplay_begin: a session is open (az_selfplay_end first)");
    try {
        std::unique_ptr<SelfplaySession> ss;
        az_status st = selfplay_begin_impl(e, p, ss);
        if (st) return st;
        e->sp_session = ss.release();
        return AZ_OK;
    } catch (const HipFail& f) { return fail_hip(e, f); }
}

az_status az_selfplay_next(az_engine* e, int32_t n_games, az_samples* out) {
    if (!e || !out) return AZ_ERR_BAD_ARGUMENT;
    SelfplaySession* ss = e->sp_session;
    if (!ss) return fail(e, AZ_ERR_BAD_ARGUMENT, "az_selfplay_next: no open session (az_selfplay_begin first)");
    if (n_games <= 0 || ss->delivered + n_games > ss->n_games) return fail(e, AZ_ERR_BAD_ARGUMENT, "az_selfplay_next: more episodes than the session has left");
    if (out->capacity < 0) return fail(e, AZ_ERR_BAD_ARGUMENT, "az_selfplay: negative capacity");
    ScopedTimer timer{e};
    try {
        HIPCHK(hipSetDevice(e->device));
        return selfplay_next_impl(e, *ss, n_games, out);
    } catch (const HipFail& f) { return fail_hip(e, f); }
}

az_status az_selfplay_end(az_engine* e) {
    if (!e) return AZ_ERR_BAD_ARGUMENT;
    if (!e->sp_session) return AZ_OK;
    (void)hipSetDevice(e->device);
    (void)hipStreamSynchronize(e->stream);
    delete e->sp_session;            // the tree arena goes back to the pool, the per-episode arrays are freed
    e->sp_session = nullptr;
    return AZ_OK;
}

az_status az_selfplay(az_engine* e, const az_selfplay_params* p, az_samples* out) {
    if (!e || !p || !out) return AZ_ERR_BAD_ARGUMENT;
    if (out->capacity < 0) return fail(e, AZ_ERR_BAD_ARGUMENT, "az_selfplay: negative capacity");
    if (e->sp_session) return fail(e, AZ_ERR_BAD_ARGUMENT, "az_selfplay: a session is open (az_selfplay_end first)");
    ScopedTimer timer{e};
    try {
        std::unique_ptr<SelfplaySession> ss;
        az_status st = selfplay_begin_impl(e, p, ss);
        if (st) return st;
        st = selfplay_next_impl(e, *ss, ss->n_games, out);
        HIPCHK(hipStreamSynchronize(e->stream));
        return st;
    } catch (const HipFail& f) { return fail_hip(e, f); }
}

az_status az_selfplay_get_evals(az_engine* e, int32_t* rec_count, uint64_t* states, float* pis, float* vs) {
    if (!e || e->sp_log_cap <= 0) return fail(e, AZ_ERR_BAD_ARGUMENT, "no eval log (run az_selfplay with record_evals > 0)");
    if (rec_count) std::memcpy(rec_count, e->sp_log_count.data(), e->sp_log_count.size() * sizeof(int32_t));
    if (states) std::memcpy(states, e->sp_log_states.data(), e->sp_log_states.size() * 8);
    if (pis) std::memcpy(pis, e->sp_log_pi.data(), e->sp_log_pi.size() * 4);
    if (vs) std::memcpy(vs, e->sp_log_v.data(), e->sp_log_v.size() * 4);
    return AZ_OK;
}

az_status az_arena(az_engine* e, const az_arena_params* p, uint64_t out_wld[3], int8_t* results) {
    if (!e || !p || !out_wld) return AZ_ERR_BAD_ARGUMENT;
    if (p->num_games < 0 || p->num_sims <= 0 || p->max_depth < 0 || p->reserve < 8)
        return fail(e, AZ_ERR_BAD_ARGUMENT, "az_arena: bad argument");
    // unsharded: num/2 games per seating (src/arena.rs:83, an odd game is dropped); sharded: this rank's range of an
    // even global total, seated by global index
    const bool sharded = p->total_games > 0;
    if (sharded && (p->first_game < 0 || p->first_game + p->num_games > p->total_games))
        return fail(e, AZ_ERR_BAD_ARGUMENT, "az_arena: shard outside [0, total_games)");
    if (!sharded && p->first_game != 0) return fail(e, AZ_ERR_BAD_ARGUMENT, "az_arena: first_game without total_games");
    const int half = sharded ? p->total_games / 2 : p->num_games / 2;
    const int G = sharded ? p->num_games : 2 * half;
    const int first = sharded ? p->first_game : 0;
    out_wld[0] = out_wld[1] = out_wld[2] = 0;
    // a host that asks for the whole arena's tally but never bound a communicator would gate the model on its own shard's count
    if (p->allreduce_wld && (!sharded || !e->comm))
        return fail(e, AZ_ERR_BAD_ARGUMENT, "az_arena: allreduce_wld needs a sharded call (total_games > 0) on an engine with a communicator (az_comm_init)");
    if (G == 0) {      // an empty shard still takes part in the tally's all-reduce
        if (p->allreduce_wld) return az_allreduce_u64(e, out_wld, 3);
        return AZ_OK;
    }
    if (G > 1024 * 64) return fail(e, AZ_ERR_BAD_ARGUMENT, "az_arena: at most 65536 games");
    if (p->record_evals < 0) return fail(e, AZ_ERR_BAD_ARGUMENT, "az_arena: negative record_evals");
    if (p->use_start_board) {
        const uint64_t a = p->start_board[0], b = p->start_board[1];
        if ((a & b) || ((a | b) & ~C4_FULL)) return fail(e, AZ_ERR_BAD_ARGUMENT, "az_arena: start_board is not a pair of disjoint 7x6 bitboards");
        // play_game's loop condition (src/arena.rs:18) fails at once on a finished board: result = cur_player * round(ended(cur_player)), :51
        const ulonglong2 s0 = make_ulonglong2(a, b);
        const uint32_t ec = e->cfg.game == 1 ? ConnectThree::ended_code(s0) : ConnectFour::ended_code(s0);
        if (ec != E_NONE) {
            const int8_t r = ec == E_MINUS1 ? 1 : (ec == E_PLUS1 ? -1 : 0);
            std::vector<int8_t> res((size_t)G, r);
            for (int g = 0; g < G; ++g) {
                const int win_cond = first + g < half ? 1 : -1;
                if (r == win_cond) out_wld[0]++; else if (r == -win_cond) out_wld[1]++; else out_wld[2]++;
            }
            if (results && hipMemcpy(results, res.data(), (size_t)G, hipMemcpyDefault) != hipSuccess) return fail(e, AZ_ERR_HIP, "az_arena: results copy");
            e->stats.games += (uint64_t)G;
            e->ar_log_cap = 0;
            e->ar_moves.assign((size_t)G * AZ_MAX_PLIES, 0);      // no move is played on a finished board
            e->ar_len.assign((size_t)G, 0);
            if (p->allreduce_wld) return az_allreduce_u64(e, out_wld, 3);     // the shards' tallies, as on the played path
            return AZ_OK;
        }
    }
    NetModel *net_new, *net_old;
    az_status st = find_net(e, p->new_model_id, &net_new);
    if (st) return st;
    st = find_net(e, p->old_model_id, &net_old);
    if (st) return st;
    int T = p->num_sim_threads;
    st = check_threads(e, p->num_sims, &T);
    if (st) return st;
    st = check_batch(e, *net_new, G * T);
    if (st) return st;
    st = check_batch(e, *net_old, G * T);
    if (st) return st;
    ScopedTimer timer{e};
    try {
        HIPCHK(hipSetDevice(e->device));
        hipStream_t s = e->stream;
        // one tree PAIR per game (B8 repair): the reference shares one nmcts / one pmcts across all games
        // (src/coach.rs:333-354) and their u16 root counters would wrap at 65536 visits.
        const int calls = AZ_MAX_PLIES / 2 + 1;
        const uint64_t nodes = std::min<uint64_t>(p->reserve, reachable_slots(p->num_sims, calls));
        TreeLease lease_n, lease_o;
        if (az_status cs = check_tree_slots(e, reachable_blocks(p->num_sims, calls, nodes))) return cs;
        acquire_trees(e, lease_n, G, reachable_blocks(p->num_sims, calls, nodes), nodes, hash_entries(p->num_sims, calls), T, s);
        acquire_trees(e, lease_o, G, reachable_blocks(p->num_sims, calls, nodes), nodes, hash_entries(p->num_sims, calls), T, s);
        TreeHost &tn = *lease_n, &to = *lease_o;
        DeviceMem mem;
        ArenaDev ad{};
        ad.G = G; ad.half = half; ad.first = first;
        ad.state = mem.alloc<ulonglong2>(G);
        ad.player = mem.alloc<int8_t>(G);
        ad.alive = mem.alloc<uint8_t>(G);
        ad.results = mem.alloc<int8_t>(G);
        ad.counters = mem.alloc<uint32_t>(4);          // [2], [3]: the largest leaf batch of the ply, per model (tile / kernel choice of the next ply)
        ad.moves = mem.alloc<uint8_t>((size_t)G * AZ_MAX_PLIES);
        ad.len = mem.alloc<int32_t>(G);
        HIPCHK(hipMemset(ad.moves, 0, (size_t)G * AZ_MAX_PLIES));
        HIPCHK(hipMemset(ad.len, 0, (size_t)G * sizeof(int32_t)));
        {
            // play_games' `board` (src/arena.rs:62-67): None = the initial board
            std::vector<uint64_t> st0((size_t)G * 2, 0ull);
            if (p->use_start_board)
                for (int g = 0; g < G; ++g) { st0[2 * (size_t)g] = p->start_board[0]; st0[2 * (size_t)g + 1] = p->start_board[1]; }
            HIPCHK(hipMemcpy(ad.state, st0.data(), st0.size() * 8, hipMemcpyHostToDevice));
        }
        HIPCHK(hipMemset(ad.player, 1, G));
        HIPCHK(hipMemset(ad.alive, 1, G));
        HIPCHK(hipMemset(ad.results, 0, G));
        uint32_t ctr[4] = {(uint32_t)G, 0u, 0u, 0u};
        HIPCHK(hipMemcpy(ad.counters, ctr, sizeof ctr, hipMemcpyHostToDevice));
        int typ_new = 0, typ_old = 0;                   // expected rows per leaf batch (0 = unknown: assume every running game)
        ScopedEvalLog log_n, log_o;
        if (p->record_evals > 0) {
            log_n.attach(tn, (size_t)G, p->record_evals, nullptr);
            log_o.attach(to, (size_t)G, p->record_evals, nullptr);
        }
        launch_reset_trees(tn.d, nullptr, s);
        launch_reset_trees(to.d, nullptr, s);
        prepare_cache(e, dedup_applies(e, *net_new) || dedup_applies(e, *net_old), (uint64_t)G * AZ_MAX_PLIES * ((uint64_t)p->num_sims + 1), s);
        // both models' tags are fixed before the two streams fork (retagging may clear the cache)
        (void)cache_for(e, *net_new);
        (void)cache_for(e, *net_old);
        SearchParams sp{(uint32_t)p->max_depth, (float)p->cpuct};
        az_status result = AZ_OK;
        // The two searches of a ply touch disjoint trees and (for two different models) disjoint net workspaces:
        // run the old model's on a second stream so its half batches overlap the new model's.
        struct Side {
            hipStream_t s2 = nullptr; hipEvent_t fork = nullptr, join = nullptr;
            ~Side() { if (s2) (void)hipStreamDestroy(s2); if (fork) (void)hipEventDestroy(fork); if (join) (void)hipEventDestroy(join); }
        } side;
        const bool overlap = net_new != net_old && !(net_new->conv && net_new->conv == net_old->conv);
        if (overlap) {
            HIPCHK(hipStreamCreate(&side.s2));
            HIPCHK(hipEventCreateWithFlags(&side.fork, hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&side.join, hipEventDisableTiming));
        }
        for (int ply = 0; ply <= AZ_MAX_PLIES; ++ply) {
            launch_arena_sync(tn.d, to.d, ad, s);
            if (overlap) {
                HIPCHK(hipEventRecord(side.fork, s));
                HIPCHK(hipStreamWaitEvent(side.s2, side.fork, 0));
                run_search(e, to, ad.state, p->num_sims, sp, *net_old, (int)ctr[0], side.s2, typ_old, ad.counters + 3);
                run_search(e, tn, ad.state, p->num_sims, sp, *net_new, (int)ctr[0], s, typ_new, ad.counters + 2);
                HIPCHK(hipEventRecord(side.join, side.s2));
                HIPCHK(hipStreamWaitEvent(s, side.join, 0));
            } else {
                run_search(e, tn, ad.state, p->num_sims, sp, *net_new, (int)ctr[0], nullptr, typ_new, ad.counters + 2);
                run_search(e, to, ad.state, p->num_sims, sp, *net_old, (int)ctr[0], nullptr, typ_old, ad.counters + 3);
            }
            launch_arena_move(tn.d, ad, p->seed, s);
            launch_arena_move(to.d, ad, p->seed, s);
            HIPCHK(hipMemcpyAsync(ctr, ad.counters, sizeof ctr, hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
            resolve_profile(e);
            typ_new = (int)std::min<uint32_t>(ctr[2], (uint32_t)(G * T));
            typ_old = (int)std::min<uint32_t>(ctr[3], (uint32_t)(G * T));
            HIPCHK(hipMemsetAsync(ad.counters + 2, 0, 2 * sizeof(uint32_t), s));
            if (ctr[1]) { result = fail(e, AZ_ERR_INVALID_MOVE, "arena: action is not valid (src/arena.rs:31-35)"); break; }
            if (ctr[0] == 0) break;
        }
        harvest_stats(e, tn, *net_new);
        harvest_stats(e, to, *net_old);
        if (result == AZ_OK) result = check_tree_errors(e, tn);
        if (result == AZ_OK) result = check_tree_errors(e, to);
        if (result) return result;
        if (ctr[0] != 0) return fail(e, AZ_ERR_HIP, "az_arena: games did not finish");
        std::vector<int8_t> res(G);
        HIPCHK(hipMemcpy(res.data(), ad.results, G, hipMemcpyDeviceToHost));
        for (int g = 0; g < G; ++g) {
            const int win_cond = first + g < half ? 1 : -1, lose_cond = -win_cond;   // src/arena.rs:80-81
            if (res[g] == win_cond) out_wld[0]++;
            else if (res[g] == lose_cond) out_wld[1]++;
            else out_wld[2]++;
        }
        if (results) HIPCHK(hipMemcpy(results, res.data(), G, hipMemcpyDefault));
        e->ar_moves.resize((size_t)G * AZ_MAX_PLIES);
        e->ar_len.resize((size_t)G);
        HIPCHK(hipMemcpy(e->ar_moves.data(), ad.moves, e->ar_moves.size(), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(e->ar_len.data(), ad.len, e->ar_len.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
        e->stats.games += (uint64_t)G;
        if (p->allreduce_wld) {       // every rank returns the whole arena's tally (one 3-counter all-reduce)
            az_status rs = az_allreduce_u64(e, out_wld, 3);
            if (rs) return rs;
        }
        e->ar_log_cap = 0;
        if (p->record_evals > 0) {
            TreeHost* ths[2] = {&tn, &to};
            const ScopedEvalLog* logs[2] = {&log_n, &log_o};
            for (int w = 0; w < 2; ++w) {
                std::vector<TreeLine> heads(G);
                HIPCHK(hipMemcpy(heads.data(), ths[w]->d.head, (size_t)G * sizeof(TreeLine), hipMemcpyDeviceToHost));
                e->ar_log[w].count.resize(G);
                for (int g = 0; g < G; ++g) e->ar_log[w].count[g] = (int32_t)heads[g].head.log_len;
                logs[w]->copy_out(e->ar_log[w].states, e->ar_log[w].pi, e->ar_log[w].v);
            }
            e->ar_log_cap = p->record_evals;
            e->ar_log_games = G;
        }
        return AZ_OK;
    } catch (const HipFail& f) { return fail_hip(e, f); }
}

az_status az_arena_get_moves(az_engine* e, int32_t* game_len, uint8_t* moves) {
    if (!e || e->ar_len.empty()) return fail(e, AZ_ERR_BAD_ARGUMENT, "no move record (run az_arena first)");
    if (game_len) std::memcpy(game_len, e->ar_len.data(), e->ar_len.size() * sizeof(int32_t));
    if (moves) std::memcpy(moves, e->ar_moves.data(), e->ar_moves.size());
    return AZ_OK;
}

az_status az_arena_get_evals(az_engine* e, int32_t which, int32_t* rec_count, uint64_t* states, float* pis, float* vs) {
    if (!e || which < 0 || which > 1 || e->ar_log_cap <= 0) return fail(e, AZ_ERR_BAD_ARGUMENT, "no eval log (run az_arena with record_evals > 0)");
    const az_engine::EvalLog& l = e->ar_log[which];
    if (rec_count) std::memcpy(rec_count, l.count.data(), l.count.size() * sizeof(int32_t));
    if (states) std::memcpy(states, l.states.data(), l.states.size() * 8);
    if (pis) std::memcpy(pis, l.pi.data(), l.pi.size() * 4);
    if (vs) std::memcpy(vs, l.v.data(), l.v.size() * 4);
    return AZ_OK;
}

// ---- the collective of the sharded Coach loop ---------------------------------------------------------------------------------
#define NCCLCHK(expr) do { ncclResult_t _r = (expr); if (_r != ncclSuccess) return fail(e, AZ_ERR_HIP, std::string("RCCL: ") + g_rccl.GetErrorString(_r) + " at " #expr); } while (0)

az_status az_comm_unique_id(az_engine* e, uint8_t id[AZ_COMM_ID_BYTES]) {
    if (!e || !id) return AZ_ERR_BAD_ARGUMENT;
    static_assert(sizeof(ncclUniqueId) == AZ_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    std::string why;
    if (!g_rccl.load(&why)) return fail(e, AZ_ERR_UNSUPPORTED, why);
    ncclUniqueId u;
    NCCLCHK(g_rccl.GetUniqueId(&u));
    std::memcpy(id, &u, sizeof u);
    return AZ_OK;
}

az_status az_comm_init(az_engine* e, int32_t rank, int32_t world, const uint8_t id[AZ_COMM_ID_BYTES]) {
    if (!e || !id || world < 1 || rank < 0 || rank >= world) return fail(e, AZ_ERR_BAD_ARGUMENT, "az_comm_init: bad rank / world");
    if (e->comm) return fail(e, AZ_ERR_BAD_ARGUMENT, "az_comm_init: the engine already has a communicator (az_comm_destroy first)");
    std::string why;
    if (!g_rccl.load(&why)) return fail(e, AZ_ERR_UNSUPPORTED, why);
    try { HIPCHK(hipSetDevice(e->device)); } catch (const HipFail& f) { return fail_hip(e, f); }
    ncclUniqueId u;
    std::memcpy(&u, id, sizeof u);
    NCCLCHK(g_rccl.CommInitRank(&e->comm, world, u, rank));
    e->comm_rank = rank;
    e->comm_world = world;
    return AZ_OK;
}

az_status az_comm_destroy(az_engine* e) {
    if (!e) return AZ_ERR_BAD_ARGUMENT;
    if (e->comm) {
        (void)hipSetDevice(e->device);
        (void)hipStreamSynchronize(e->stream);
        NCCLCHK(g_rccl.CommDestroy(e->comm));
        e->comm = nullptr;
        e->comm_rank = 0; e->comm_world = 1;
    }
    return AZ_OK;
}

az_status az_allreduce_u64(az_engine* e, uint64_t* values, int32_t n) {
    if (!e || !values || n < 0 || n > 64) return fail(e, AZ_ERR_BAD_ARGUMENT, "az_allreduce_u64: bad argument");
    if (!e->comm) return fail(e, AZ_ERR_BAD_ARGUMENT, "az_allreduce_u64: no communicator (az_comm_init)");
    if (n == 0) return AZ_OK;
    try {
        HIPCHK(hipSetDevice(e->device));
        unsigned long long* d = (unsigned long long*)e->comm_scratch.ensure((size_t)n * 8, e->stream);
        HIPCHK(hipMemcpyAsync(d, values, (size_t)n * 8, hipMemcpyHostToDevice, e->stream));
        NCCLCHK(g_rccl.AllReduce(d, d, (size_t)n, ncclUint64, ncclSum, e->comm, e->stream));
        HIPCHK(hipMemcpyAsync(values, d, (size_t)n * 8, hipMemcpyDeviceToHost, e->stream));
        HIPCHK(hipStreamSynchronize(e->stream));
        return AZ_OK;
    } catch (const HipFail& f) { return fail_hip(e, f); }
}

// The episode-batch exchange.  EVERY decision that could make one rank leave early is taken from data every rank holds: the first
// collective carries, per rank, its tuple count (or -1: its local buffers are unusable), the capacity it can receive into (-1: not a
// receiver, -2: receive buffers unusable) and its dst_rank -- so either every rank posts its part of the grouped exchange or every rank
// returns the same error without posting anything.  (Round 3's version returned on the receiving rank alone and left its peers
// waiting in ncclGroupEnd.)
az_status az_gather_samples(az_engine* e, const az_samples* local, int32_t dst_rank, az_samples* gathered, int64_t* counts_out) {
    if (!e || !local) return AZ_ERR_BAD_ARGUMENT;
    if (!e->comm) return fail(e, AZ_ERR_BAD_ARGUMENT, "az_gather_samples: no communicator (az_comm_init)");
    const int world = e->comm_world, rank = e->comm_rank;
    long long n = local->count;
    const bool dst_ok = dst_rank >= -1 && dst_rank < world;
    const bool receiver = dst_ok && (dst_rank < 0 || rank == dst_rank);         // dst_rank = -1: every rank receives (all-gather)
    const bool local_ok = dst_ok && n >= 0 && (n == 0 || (local->states && local->pis && local->zs));
    const bool recv_ok = !receiver || (gathered && gathered->states && gathered->pis && gathered->zs && gathered->capacity >= 0);
    struct Hello { long long n, cap, dst, pad; };
    const Hello mine{local_ok ? n : -1, !receiver ? -1 : (recv_ok ? gathered->capacity : -2), dst_rank, 0};
    if (!local_ok) n = 0;
    struct GroupGuard {          // an error between GroupStart and GroupEnd must not leave the group open
        bool open = false;
        ~GroupGuard() { if (open) (void)g_rccl.GroupEnd(); }
    } group;
    try {
        HIPCHK(hipSetDevice(e->device));
        hipStream_t s = e->stream;
        // one allocation, kept by the engine and grown on demand (it was five to nine hipMalloc / hipFree pairs per call):
        // [hello x (world + 1)] [st | pi | z | packed] of the local tuples; the receive side is carved once the total is known
        const size_t hello_b = ((size_t)world + 1) * sizeof(Hello);
        const size_t local_b = (size_t)n * (16 + 28 + 4 + sizeof(PackedSample));
        char* base = (char*)e->comm_scratch.ensure(hello_b + local_b + 256, s);
        Hello* d_hello = (Hello*)base;
        // 1. all-gather of the per-rank hello records (count, receive capacity, dst_rank)
        HIPCHK(hipMemcpyAsync(d_hello + world, &mine, sizeof mine, hipMemcpyHostToDevice, s));
        NCCLCHK(g_rccl.AllGather(d_hello + world, d_hello, sizeof(Hello), ncclUint8, e->comm, s));
        std::vector<Hello> hello((size_t)world);
        HIPCHK(hipMemcpyAsync(hello.data(), d_hello, (size_t)world * sizeof(Hello), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        long long total = 0;
        int bad_local = -1, bad_recv = -1, bad_dst = -1;
        for (int r = 0; r < world; ++r) {
            if (hello[r].n < 0) { if (bad_local < 0) bad_local = r; } else total += hello[r].n;
            if (hello[r].dst != hello[0].dst || hello[r].dst < -1 || hello[r].dst >= world) { if (bad_dst < 0) bad_dst = r; }
        }
        for (int r = 0; r < world; ++r) {
            if (counts_out) counts_out[r] = hello[r].n < 0 ? 0 : hello[r].n;
            if (hello[r].cap == -2 || (hello[r].cap >= 0 && hello[r].cap < total)) { if (bad_recv < 0) bad_recv = r; }
        }
        // the same verdict on every rank, before anything is posted
        if (bad_dst >= 0) return fail(e, AZ_ERR_BAD_ARGUMENT, "az_gather_samples: dst_rank outside the communicator or not the same on every rank (rank " + std::to_string(bad_dst) + ")");
        if (bad_local >= 0) return fail(e, AZ_ERR_BAD_ARGUMENT, "az_gather_samples: rank " + std::to_string(bad_local) + "'s local tuples need states, pis and zs");
        if (bad_recv >= 0) return fail(e, AZ_ERR_BAD_ARGUMENT, "az_gather_samples: rank " + std::to_string(bad_recv) + "'s gathered buffers are missing or too small for " + std::to_string(total) + " tuples");
        // 2. pack this rank's tuples (inputs may be host or device memory)
        const size_t recv_b = receiver ? (size_t)total * (sizeof(PackedSample) + 16 + 28 + 4) : 0;
        base = (char*)e->comm_scratch.ensure(hello_b + local_b + recv_b + 512, s);      // may move: nothing of step 1 is needed any more
        char* cur = base + hello_b;
        auto carve = [&](size_t bytes) { char* q = cur; cur += (bytes + 63) / 64 * 64; return q; };
        ulonglong2* d_st = (ulonglong2*)carve((size_t)n * 16);
        float* d_pi = (float*)carve((size_t)n * 28);
        float* d_z = (float*)carve((size_t)n * 4);
        PackedSample* d_mine = (PackedSample*)carve((size_t)n * sizeof(PackedSample));
        if (n > 0) {
            HIPCHK(hipMemcpyAsync(d_st, local->states, (size_t)n * 16, hipMemcpyDefault, s));
            HIPCHK(hipMemcpyAsync(d_pi, local->pis, (size_t)n * 28, hipMemcpyDefault, s));
            HIPCHK(hipMemcpyAsync(d_z, local->zs, (size_t)n * 4, hipMemcpyDefault, s));
            hipLaunchKernelGGL(k_pack_samples, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, d_st, d_pi, d_z, d_mine, n);
        }
        // 3. ONE exchange (a grouped gatherv): a receiving rank posts a receive per peer, a sending rank one send per receiver
        PackedSample* d_all = receiver ? (PackedSample*)carve((size_t)total * sizeof(PackedSample)) : nullptr;
        NCCLCHK(g_rccl.GroupStart());
        group.open = true;
        long long off = 0;
        for (int r = 0; r < world; ++r) {
            const long long cr = hello[r].n;
            if (receiver) {
                if (r != rank && cr > 0) NCCLCHK(g_rccl.Recv(d_all + off, (size_t)cr * sizeof(PackedSample), ncclUint8, r, e->comm, s));
                if (r == rank && n > 0) HIPCHK(hipMemcpyAsync(d_all + off, d_mine, (size_t)n * sizeof(PackedSample), hipMemcpyDeviceToDevice, s));
            }
            if (r != rank && n > 0 && (dst_rank < 0 || r == dst_rank)) NCCLCHK(g_rccl.Send(d_mine, (size_t)n * sizeof(PackedSample), ncclUint8, r, e->comm, s));
            off += cr;
        }
        group.open = false;
        NCCLCHK(g_rccl.GroupEnd());
        if (receiver) {
            ulonglong2* o_st = (ulonglong2*)carve((size_t)total * 16);
            float* o_pi = (float*)carve((size_t)total * 28);
            float* o_z = (float*)carve((size_t)total * 4);
            if (total > 0) hipLaunchKernelGGL(k_unpack_samples, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, d_all, o_st, o_pi, o_z, total);
            if (total > 0) {
                HIPCHK(hipMemcpyAsync(gathered->states, o_st, (size_t)total * 16, hipMemcpyDefault, s));
                HIPCHK(hipMemcpyAsync(gathered->pis, o_pi, (size_t)total * 28, hipMemcpyDefault, s));
                HIPCHK(hipMemcpyAsync(gathered->zs, o_z, (size_t)total * 4, hipMemcpyDefault, s));
            }
            HIPCHK(hipStreamSynchronize(s));
            gathered->count = total;
        } else {
            HIPCHK(hipStreamSynchronize(s));
            if (gathered) gathered->count = 0;
        }
        return AZ_OK;
    } catch (const HipFail& f) { return fail_hip(e, f); }
}

#ifdef AZ_DIAG
// Diagnostic library only (not part of the ABI): the children of the node reached from tree g's current root by following `path`
// (child indices, not actions).  out rows of 8 u64: slot, a, ctr (resolved through a link), prior bits, link, meta, own ctr, key.
// Returns the number of children, or -1.
int az_diag_tree_children(az_tree* t, int g, const int* path, int depth, unsigned long long* out) {
    if (!t || g < 0 || g >= t->th.d.G) return -1;
    TreeDev& d = t->th.d;
    std::vector<TreeLine> heads(d.G);
    if (hipMemcpy(heads.data(), d.head, heads.size() * sizeof(TreeLine), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    struct Rec { unsigned long long ctr, key; uint32_t prior, meta, link, child_base; };
    auto load = [&](uint32_t slot, Rec* r) {
        uint32_t raw[4];
        if (hipMemcpy(raw, d.node + ((size_t)g * d.R + slot), 16, hipMemcpyDeviceToHost) != hipSuccess) return false;
        if (hipMemcpy(&r->key, d.key + ((size_t)g * d.R + slot), 8, hipMemcpyDeviceToHost) != hipSuccess) return false;
        const uint32_t w = raw[3], kind = (w >> 10) & 3u, payload = w >> 12;
        r->ctr = ((unsigned long long)raw[1] << 32) | raw[0];
        r->prior = raw[2];
        r->meta = (w & 0x3Fu) | (((w >> 6) & 0xFu) << META_ECODE_SHIFT) | (kind == 2u ? META_EXPANDED : 0u);
        r->link = kind == 1u ? payload : NONE;
        r->child_base = kind == 2u ? payload * (uint32_t)BLOCK_SLOTS : 0u;
        if (kind != 2u) r->key = 0;
        return true;
    };
    uint32_t cur = heads[g].head.root;
    Rec pr;
    if (!load(cur, &pr)) return -1;
    for (int i = 0; i < depth; ++i) {
        Rec c;
        if (!load(pr.child_base + (uint32_t)path[i], &c)) return -1;
        cur = c.link != NONE ? c.link : pr.child_base + (uint32_t)path[i];
        if (!load(cur, &pr)) return -1;
    }
    const int n = (int)((pr.meta >> META_NCHILD_SHIFT) & 7u);
    for (int j = 0; j < n; ++j) {
        Rec c, r;
        if (!load(pr.child_base + (uint32_t)j, &c)) return -1;
        r = c;
        if (c.link != NONE && !load(c.link, &r)) return -1;
        unsigned long long* o = out + 8 * j;
        o[0] = pr.child_base + (uint32_t)j; o[1] = c.meta & META_A_MASK; o[2] = r.ctr; o[3] = c.prior; o[4] = c.link; o[5] = r.meta; o[6] = c.ctr; o[7] = r.key;
    }
    out[8 * 7] = pr.ctr; out[8 * 7 + 1] = pr.key; out[8 * 7 + 2] = cur;
    return n;
}
void az_diag_tree_set_sims(az_tree* t, int num_sims) { if (t) t->num_sims = num_sims; }
#endif

}  // extern "C"
