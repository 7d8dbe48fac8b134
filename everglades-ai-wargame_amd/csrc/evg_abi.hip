// evg_abi.hip -- host side of the C-ABI declared in include/evg.h (libevg.so).
// Owns the persistent device state of one handle, validates and flattens the constant tables,
// and enqueues the kernels of evg_kernels.hip on the caller's stream.  No CPU execution path.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include "evg_rccl_api.h"    // the few RCCL types this file names: librccl is opened at run time (dlopen), the build needs neither its header nor the library
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <exception>
#include <mutex>
#include <new>
#include <string>
#include <vector>
#include "evg_device.h"
#include "evg_mt.h"

using namespace evg;

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess) return fail(EVG_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(_e));  \
    } while (0)

constexpr int kGroupSize[NG] = {8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 12};   // everglades_env.py:145-156

// Every entry point runs on the handle's device and leaves the caller's current device as it found it (a process that
// drives several GPUs, e.g. through torch, must not have its current device changed behind its back).
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    hipError_t err = hipSuccess;
    explicit DeviceGuard(int dev) {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != dev) {
            err = hipSetDevice(dev);
            switched = err == hipSuccess;
        }
    }
    ~DeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

#define EVG_ON_DEVICE(h)                                                                                      \
    DeviceGuard guard_((h)->cfg.device_id);                                                                   \
    if (guard_.err != hipSuccess) return fail(EVG_ERR_HIP, "selecting device %d failed: %s", (h)->cfg.device_id, hipGetErrorString(guard_.err))

// Device buffers of the caller: the kernels read and write them with 8- and 16-byte vector accesses (observation rows 16 bytes per lane, order rows int2 /
// uint4, rewards float2, scores int2), so every one of them must be 16-byte aligned (include/evg.h, "Conventions"; any hipMalloc / torch allocation is).
// NULL is "not given" and passes.
// Buffers only ever touched with 8-byte accesses (rewards float2, scores int2: one pair per env) need 8 bytes, so an odd env offset into an [N][2] tensor is
// accepted (EVG_NEED_ALIGNED8).
bool misaligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) != 0; }
bool misaligned8(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 7u) != 0; }
#define EVG_NEED_ALIGNED8(p)                                                                                  \
    do {                                                                                                      \
        if (misaligned8(p)) return fail(EVG_ERR_INVALID, "%s must be 8-byte aligned (got %p)", #p, (const void*)(p)); \
    } while (0)
#define EVG_NEED_ALIGNED16(p)                                                                                 \
    do {                                                                                                      \
        if (misaligned16(p)) return fail(EVG_ERR_INVALID, "%s must be 16-byte aligned (got %p)", #p, (const void*)(p)); \
    } while (0)

// numpy pairwise order for the cached int(avg health) (server.py:481,491); state import only
double host_np_sum(const double* h, int n) {
    double s = ((h[0] + h[1]) + (h[2] + h[3])) + ((h[4] + h[5]) + (h[6] + h[7]));
    for (int i = 8; i < n; ++i) s += h[i];
    return s;
}

}  // namespace

struct evg_handle {
    evg_config cfg;
    DevTables host_tables;
    DevState S;
    DevTables* d_tables = nullptr;
    DeviceCaps caps;                    // what the device holds at once (query_device_caps at evg_create): drives the launch plan
    uint32_t* fault_seen_host = nullptr;   // host address of DevState::fault_seen (mapped host memory): non-zero <=> some bit of the fault word was set
    // Rollout launches replayed as hipGraphs (launch_rollout): one executable graph per distinct set of launch arguments, captured once on a
    // stream of the handle's own (nothing ever runs there) and replayed on the caller's stream with ONE submission per launch plan.
    // MEASURED AND SWITCHED OFF in the product build: on this runtime (ROCm 7.2) a hipGraphLaunch of the one-kernel plan reaches the GPU ~1.2 us per
    // step LATER than the plain launch in the driver's 20-step shape (tools/driver_shape_ab.sh, six alternations on one box: 21.1-21.9 us per step with
    // replay, 19.4-20.8 without; profiles/r04_f_driver_shape_graph_replay_ab.txt).  `make graphs` builds libevg_graphs.so with the replay on for that
    // A/B; what stays in the product is what made replay possible -- a launch plan keeps nothing on the host -- so a CALLER's capture of a rollout call
    // (torch.cuda.graph) works for every plan, the chunked one included.
    struct CachedGraph { StepIO key; hipGraphExec_t exec; };
    std::vector<CachedGraph> graphs;
    hipStream_t capture_stream = nullptr;
#ifdef EVG_REPLAY_GRAPHS
    bool graphs_off = false;               // (a capture or an instantiation that fails once switches to plain launches)
#else
    bool graphs_off = true;
#endif
    std::vector<void*> allocs;
    std::vector<hipEvent_t> events;     // evg_rollout_random timing
    // the path's one exchange without torch.distributed (evg_comm_init / evg_gather_returns): an RCCL communicator of this handle's device
    ncclComm_t comm = nullptr;
    int comm_world = 0, comm_rank = -1;
    std::vector<int32_t> comm_counts;   // envs of every rank's handle (contiguous shards in rank order)
    float* comm_send = nullptr;         // device [N][4]: this handle's packed rows, where the collective reads them
#ifdef EVG_DIAG                         // diagnostic libraries only, set through evg_diag_configure (never from the environment)
    uint32_t ablate = 0;
    int32_t lanes = 0;                  // diagnostic library: 0 = the product's choice of step kernel, else evg_diag_configure's
    unsigned long long* stamps = nullptr;
#endif
};

// arguments of one step-kernel launch
static StepIO make_io(const evg_handle* h, const int32_t* actions, void* obs, float* reward, uint8_t* done, int8_t* winner, int32_t* scores,
                      uint8_t* status, int observe_only, int gen_actions, int policy0, int policy1, int32_t* actions_out) {
    StepIO io;
    memset(&io, 0, sizeof(io));
    io.actions = actions; io.obs = obs; io.reward = reward; io.done = done; io.winner = winner; io.scores = scores; io.status = status;
    io.observe_only = observe_only; io.gen_actions = gen_actions; io.policy0 = policy0; io.policy1 = policy1; io.actions_out = actions_out;
    io.turns = 1;
#ifdef EVG_DIAG
    io.lanes_per_wave = h->lanes; io.ablate = observe_only ? 0u : h->ablate; io.stamps = observe_only ? nullptr : h->stamps;
#else
    (void)h;
#endif
    return io;
}

template <typename Tp>
static int dev_alloc(evg_handle* h, Tp** p, size_t count) {
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, count * sizeof(Tp));
    if (e != hipSuccess) return fail(EVG_ERR_ALLOC, "hipMalloc(%zu bytes) failed: %s", count * sizeof(Tp), hipGetErrorString(e));
    h->allocs.push_back(q);
    *p = reinterpret_cast<Tp*>(q);
    return EVG_OK;
}

// The fault word of a handle (DevState::fault; include/evg.h, evg_check_fault).  The caller has synchronised whatever it wants covered: kernels
// store the host-mapped mirror with system scope, so it is visible here after that synchronisation and a healthy handle costs one host load.
static int check_fault(evg_handle* h, uint32_t* word_out = nullptr) {
    uint32_t fault = 0;
    if (*reinterpret_cast<volatile uint32_t*>(h->fault_seen_host) != 0u) {
        HIP_TRY(hipMemcpy(&fault, h->S.fault, sizeof(fault), hipMemcpyDeviceToHost));
        if (!fault) fault = 0x80000000u;        // the mirror says so, the word does not: report it all the same
    }
    if (word_out) *word_out = fault;
    if (fault)
        return fail(EVG_ERR_FAULT,
                    "a chunked rollout launch failed to hand a set of envs on (fault word %u: 1 = a workgroup gave up waiting for a predecessor chunk, "
                    "2 = a workgroup ran on an XCD the create-time probe did not see, 4 = a queue of a chunked launch was not drained, 8 = a hand-over "
                    "delivered stale state words (checksum mismatch)): the state and the results of this handle are not valid; destroy it",
                    fault);
    return EVG_OK;
}

static bool same_launch(const StepIO& a, const StepIO& b) {
    bool same = a.actions == b.actions && a.obs == b.obs && a.reward == b.reward && a.done == b.done && a.winner == b.winner && a.scores == b.scores &&
                a.status == b.status && a.observe_only == b.observe_only && a.gen_actions == b.gen_actions && a.policy0 == b.policy0 &&
                a.policy1 == b.policy1 && a.actions_out == b.actions_out && a.turns == b.turns && a.seat == b.seat && a.actions_both == b.actions_both &&
                a.feat_shared == b.feat_shared && a.feat_swarm == b.feat_swarm;
#ifdef EVG_DIAG
    same = same && a.lanes_per_wave == b.lanes_per_wave && a.ablate == b.ablate && a.stamps == b.stamps;
#endif
    return same;
}

static void drop_graphs(evg_handle* h) {
    for (auto& g : h->graphs) (void)hipGraphExecDestroy(g.exec);
    h->graphs.clear();
}

// One launch of a persistent rollout = one launch PLAN (plan_step: up to two step kernels, or memset + chunked kernel + queue check): replayed as a hipGraph.
// The plan is captured once per distinct argument set on the handle's private stream (capture runs nothing; the caller's stream may be the legacy
// default stream, which cannot capture), instantiated and kept; every later launch with the same arguments is ONE hipGraphLaunch on the caller's
// stream instead of up to four submissions.  Everything a launch needs lives on the device (the chunked form zeroes its queues and flags with a
// memset node), so a replay is exactly the launch.  If the caller's stream is itself capturing (torch.cuda.graph around a rollout call), the
// kernels are enqueued plainly and become part of the CALLER's graph.
// the cached executable graph of a launch, captured and instantiated on first use (nullptr: graphs are off on this handle / runtime)
static hipGraphExec_t graph_of(evg_handle* h, const StepIO& io) {
    if (h->graphs_off) return nullptr;
    for (auto& g : h->graphs)
        if (same_launch(g.key, io)) return g.exec;
    if (!h->capture_stream && hipStreamCreateWithFlags(&h->capture_stream, hipStreamNonBlocking) != hipSuccess) {
        h->capture_stream = nullptr; h->graphs_off = true;
        return nullptr;
    }
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    hipError_t e = hipStreamBeginCapture(h->capture_stream, hipStreamCaptureModeThreadLocal);
    int rc = 0;
    if (e == hipSuccess) {
        rc = launch_step(h->S, io, h->cfg.obs_dtype, h->caps, h->capture_stream);
        e = hipStreamEndCapture(h->capture_stream, &graph);
    }
    if (e == hipSuccess && !rc) e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    if (graph) (void)hipGraphDestroy(graph);
    if (e != hipSuccess || rc) {               // no graph on this runtime: never an error of the rollout itself
        (void)hipGetLastError();
        h->graphs_off = true;
        return nullptr;
    }
    if (h->graphs.size() >= 16) { (void)hipGraphExecDestroy(h->graphs.front().exec); h->graphs.erase(h->graphs.begin()); }
    h->graphs.push_back({io, exec});
    return exec;
}

static int launch_rollout(evg_handle* h, const StepIO& io, hipStream_t s) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    const bool caller_captures = s != nullptr && hipStreamIsCapturing(s, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
    hipGraphExec_t exec = caller_captures ? nullptr : graph_of(h, io);
    if (!exec) return launch_step(h->S, io, h->cfg.obs_dtype, h->caps, s);
    return (int)hipGraphLaunch(exec, s);
}

// "No C++ exception crosses the ABI" (include/evg.h): every int-returning entry point below is a function-try-block that ends here.  The only exceptions the
// host code of this file can raise are allocation failures of its std::vector / std::string temporaries.
static int on_exception() noexcept {
    try { throw; }
    catch (const std::bad_alloc&) { return fail(EVG_ERR_ALLOC, "out of host memory"); }
    catch (const std::exception& e) { return fail(EVG_ERR_INVALID, "unexpected C++ exception: %s", e.what()); }
    catch (...) { return fail(EVG_ERR_INVALID, "unexpected C++ exception"); }
}

extern "C" {

const char* evg_last_error(void) { return g_err; }
int evg_abi_version(void) { return EVG_ABI_VERSION; }

void evg_default_tables(evg_tables* t) {
    memset(t, 0, sizeof(*t));
    struct Edge { int a, b, d; };
    // config/DemoMap.json:5-300 (each connection listed once; the JSON lists both directions)
    static const Edge edges[] = {{1, 2, 6}, {1, 4, 6}, {2, 3, 4}, {2, 5, 4}, {3, 4, 4}, {3, 5, 6}, {3, 6, 3}, {3, 7, 6}, {4, 7, 4},
                                 {5, 8, 4}, {5, 9, 6}, {6, 9, 3}, {7, 9, 6}, {7, 10, 4}, {8, 9, 4}, {8, 11, 6}, {9, 10, 4}, {10, 11, 6}};
    for (const Edge& e : edges) t->node_dist[e.a][e.b] = t->node_dist[e.b][e.a] = e.d;
    static const double defense[12] = {0, 1, 1.5, 1.75, 1.5, 1.75, 1.75, 1.75, 1.5, 1.75, 1.5, 1};
    static const int p1map[12] = {0, 11, 8, 9, 10, 5, 6, 7, 2, 3, 4, 1};      // server.py:89
    for (int n = 0; n <= NN; ++n) {
        t->node_control_points[n] = (n == 1 || n == 11) ? 500 : 100;
        t->node_defense[n] = defense[n];
        t->node_team_start[n] = n == 1 ? 0 : (n == 11 ? 1 : -1);
        t->p1_node_map[n] = p1map[n];
    }
    t->node_control_points[0] = 0;
    t->node_resource[2] = t->node_resource[8] = EVG_RES_OBSERVE;
    t->node_resource[4] = t->node_resource[10] = EVG_RES_DEFENSE;
    // config/UnitDefinitions.json:4-27, ids by JSON order: tank 0, controller 1, striker 2
    t->num_unit_types = 3;
    static const int ud[3][5] = {{3, 1, 1, 1, 1}, {2, 1, 1, 2, 1}, {1, 2, 2, 1, 1}};   // health damage speed control cost
    for (int u = 0; u < 3; ++u) {
        t->unit_health[u] = ud[u][0]; t->unit_damage[u] = ud[u][1]; t->unit_speed[u] = ud[u][2];
        t->unit_control[u] = ud[u][3]; t->unit_cost[u] = ud[u][4];
    }
    static const int cls[3] = {1, 2, 0};                                       // ['controller','striker','tank'][g % 3]
    for (int p = 0; p < NP; ++p)
        for (int g = 0; g < NG; ++g) { t->group_type[p][g] = cls[g % 3]; t->group_size[p][g] = kGroupSize[g]; }
    t->max_turns = 150;
}

// dfs_attack.py ignores its observation, so its orders are a function of the call count alone.  Simulate the bot
// (depth-first sweep over the map with two attack groups, delays in between, and the rows that persist in the mutable
// default argument of act_dfs_attack) until its whole state repeats; the kernel then indexes the table.
// The scripted bots of agents/State_Machine/ do not read the map file: every one of them carries DemoMap's NODE_CONNECTIONS as a module constant
// (swarm_agent.py:15-27, dfs_attack.py:11-23, ...), so on ANOTHER map (EvergladesEnv.reset(map_file=)) they still route by DemoMap -- many of their orders are
// then rejected by the server, which is the reference's behaviour (pinned by tests/golden/custom_agents.npz).  Bit m of entry n: m is in NODE_CONNECTIONS[n].
static const uint16_t kBotConnections[12] = {0, 0x014, 0x02A, 0x0F4, 0x08A, 0x30C, 0x208, 0x618, 0xA20, 0x5E0, 0xA80, 0x500};

static void build_dfs_table(DevTables* D) {
    struct St { int first, group_index, delay_turns, delay, prev; uint32_t visited; std::vector<int> stack; uint64_t persist;
                bool operator==(const St& o) const { return first == o.first && group_index == o.group_index && delay_turns == o.delay_turns &&
                    delay == o.delay && prev == o.prev && visited == o.visited && stack == o.stack && persist == o.persist; } };
    St s{1, 0, 5, 0, 1, 0u, {1}, 0ull};
    std::vector<St> seen;
    std::vector<uint64_t> rows;
    auto set_row = [](uint64_t& p, int i, int g, int n) { p = (p & ~(0xFFull << (8 * i))) | ((uint64_t)(g | (n << 4)) << (8 * i)); };
    for (int call = 0; call < 192; ++call) {
        int hit = -1;
        for (size_t k = 0; k < seen.size(); ++k) if (seen[k] == s) { hit = (int)k; break; }
        if (hit >= 0) { D->dfs_mu = hit; D->dfs_lambda = call - hit; break; }
        seen.push_back(s);
        uint64_t out = 0;                                          // np.zeros(shape)
        if (s.first) { s.first = 0; }
        else if (s.group_index == 2 || s.delay) {
            s.group_index = 0;
            if (s.delay_turns == 0) { s.delay_turns = 10; s.delay = 0; } else { s.delay = 1; s.delay_turns -= 1; }
        } else {
            const bool all = s.visited == 0x7FFu;
            if (s.group_index == 1) {
                for (int i = 0; i < 5; ++i) set_row(s.persist, i, 7 + i, s.prev);
                s.group_index += 1;
            } else if (s.group_index == 0 && !all) {
                const int n = s.stack.back();
                s.prev = n;
                for (int i = 0; i < NA; ++i) set_row(s.persist, i, i, n);
                s.group_index += 1;
                s.stack.pop_back();
                s.visited |= 1u << (n - 1);
                for (int m = 1; m <= NN; ++m) if (((kBotConnections[n] >> m) & 1u) && !((s.visited >> (m - 1)) & 1u)) s.stack.push_back(m);   // dfs_attack.py:131
            } else {
                s.group_index = 0; s.delay_turns = 5; s.delay = 0; s.stack.assign(1, 1); s.visited = 0u;
            }
            out = s.persist;
        }
        rows.push_back(out);
    }
    if (D->dfs_lambda <= 0) { D->dfs_mu = 0; D->dfs_lambda = (int)rows.size(); }     // not reached for connected maps
    for (size_t k = 0; k < rows.size() && k < 192; ++k) D->dfs_rows[k] = rows[k];
}

// the step kernel's LDS image of the tables it indexes per lane (evg_device.h: LdsTables)
static void fill_lds_tables(DevTables* D) {
    LdsTables& L = D->lds;
    memset(&L, 0, sizeof(L));
    for (int n = 0; n < 12; ++n) {
        L.adj[n] = D->adj_row[n];
        L.cp[n] = D->control_points[n]; L.ts[n] = D->team_start[n];
        L.res[n] = ((D->resource[n] & EVG_RES_DEFENSE) ? 1 : 0) | ((D->resource[n] & EVG_RES_OBSERVE) ? 1 << 16 : 0);
        L.init_node[n] = D->init_node[n];
    }
    for (int i = 0; i < 48; ++i) { L.den[i] = (&D->den_tab[0][0])[i]; L.rcp[i] = (&D->rcp_tab[0][0])[i]; }
    for (int i = 0; i < 24; ++i) L.init_grp[i] = D->init_grp[i];
    L.nib[0] = D->p1map_nib;
    for (int p = 0; p < NP; ++p) {
        L.nib[1 + p] = D->speed_nib[p]; L.nib[3 + p] = D->control_nib[p]; L.nib[5 + p] = D->cost_nib[p]; L.nib[7 + p] = D->type_nib[p];
    }
    L.nib[10] = D->p1inv_nib;
    L.nib[11] = D->maxnbr_nib;
    L.nib[12] = D->tar_to_1;
    L.nib[13] = D->tar_to_11;
    L.nib[9] = (uint64_t)(uint32_t)D->max_turns | ((uint64_t)(D->damage_nib & 0xFFFFu) << 8) | ((uint64_t)(D->fast_div ? 1u : 0u) << 24);
}

static int build_dev_tables(const evg_config* cfg, DevTables* D) {
    const evg_tables& t = cfg->tables;
    memset(D, 0, sizeof(*D));
    if (t.num_unit_types < 1 || t.num_unit_types > EVG_MAX_UNIT_TYPES) return fail(EVG_ERR_INVALID, "num_unit_types out of range");
    if (t.max_turns < 1 || t.max_turns > 255) return fail(EVG_ERR_INVALID, "max_turns must be in 1..255");
    for (int u = 0; u < t.num_unit_types; ++u) {
        if (t.unit_health[u] < 1 || t.unit_health[u] > 255 || t.unit_damage[u] < 1 || t.unit_damage[u] > 15 || t.unit_speed[u] < 1 ||
            t.unit_speed[u] > 15 || t.unit_control[u] < 1 || t.unit_control[u] > 15 || t.unit_cost[u] < 1 || t.unit_cost[u] > 15)
            return fail(EVG_ERR_INVALID, "unit type %d outside the supported domain", u);
        D->damage_nib |= (uint32_t)t.unit_damage[u] << (4 * u);
        D->armor_byte |= (uint32_t)t.unit_health[u] << (8 * u);
        D->unit_speed[u] = t.unit_speed[u]; D->unit_control[u] = t.unit_control[u]; D->unit_cost[u] = t.unit_cost[u];
    }
    // Combat divides (10 * D) by armor (+ node defense); D <= 255 and the denominators are the few values below, so the
    // quotient can be taken as q0 = a * rcp, q = fma(fma(-den, q0, a), rcp, q0) -- IF that equals the IEEE quotient, which is
    // checked here for every case that can occur with these tables (the kernel keeps a true division for tables that fail)
    D->fast_div = 1;
    for (int u = 0; u < t.num_unit_types; ++u)
        for (int n = 0; n <= NN; ++n) {
            const double den = (double)t.unit_health[u] + (n ? t.node_defense[n] : 0.0);
            const double rcp = 1.0 / den;
            D->den_tab[u][n] = den;
            D->rcp_tab[u][n] = rcp;
            for (int dsum = 0; dsum <= 255; ++dsum) {
                const double a = 10.0 * (double)dsum, q0 = a * rcp;
                const double q = fma(fma(-den, q0, a), rcp, q0);
                if (!(q == a / den)) D->fast_div = 0;
            }
        }
    int start[2] = {-1, -1};
    bool seen[12] = {false};
    for (int n = 1; n <= NN; ++n) {
        for (int m = 0; m <= NN; ++m) {
            const int d = t.node_dist[n][m];
            if (d < 0 || d > 7 || (m == 0 && d != 0) || (m == n && d != 0)) return fail(EVG_ERR_INVALID, "node_dist[%d][%d]=%d unsupported", n, m, d);
            D->adj_row[n] |= (uint64_t)d << (4 * m);
        }
        if (t.node_control_points[n] < 1 || t.node_control_points[n] > 511) return fail(EVG_ERR_INVALID, "control points of node %d", n);
        if (!(t.node_defense[n] >= 0.0)) return fail(EVG_ERR_INVALID, "defense of node %d", n);
        const int ts = t.node_team_start[n];
        if (ts < -1 || ts > 1) return fail(EVG_ERR_INVALID, "team start of node %d", n);
        if (ts >= 0) { if (start[ts] != -1) return fail(EVG_ERR_INVALID, "two start nodes for player %d", ts); start[ts] = n; }
        const int pm = t.p1_node_map[n];
        if (pm < 1 || pm > NN || seen[pm]) return fail(EVG_ERR_INVALID, "p1_node_map is not a permutation of 1..11");
        seen[pm] = true;
        D->control_points[n] = t.node_control_points[n];
        D->defense[n] = t.node_defense[n];
        D->team_start[n] = ts;
        D->resource[n] = t.node_resource[n];
        D->p1map_nib |= (uint64_t)pm << (4 * n);
    }
    D->team_start[0] = -1;
    if (t.p1_node_map[0] != 0) return fail(EVG_ERR_INVALID, "p1_node_map[0] must be 0");
    if (start[0] < 0 || start[1] < 0 || start[0] == start[1]) return fail(EVG_ERR_INVALID, "each player needs its own start node");
    int max_damage[2] = {0, 0};
    for (int p = 0; p < NP; ++p)
        for (int g = 0; g < NG; ++g) {
            const int ty = t.group_type[p][g];
            if (ty < 0 || ty >= t.num_unit_types) return fail(EVG_ERR_INVALID, "group_type[%d][%d]", p, g);
            if (t.group_size[p][g] != kGroupSize[g])
                return fail(EVG_ERR_INVALID, "group_size[%d][%d]=%d: this build fixes the army shape of everglades_env.py:145-156 (8 x 11 + 12)",
                            p, g, t.group_size[p][g]);
            D->group_type[p][g] = ty;
            D->type_nib[p] |= (uint64_t)ty << (4 * g);
            max_damage[p] += kGroupSize[g] * t.unit_damage[ty];
        }
    if (max_damage[0] > 255 || max_damage[1] > 255) return fail(EVG_ERR_INVALID, "total damage of an army must fit 8 bits");
    D->max_turns = t.max_turns;

    // state right after game_init: everyone at home (list order = gid order, stamp 0), turn-0 capture
    for (int p = 0; p < NP; ++p)
        for (int g = 0; g < NG; ++g) {
            const uint32_t mask = (1u << kGroupSize[g]) - 1u;
            D->init_grp[p * NG + g] = (uint32_t)start[p] | (mask << G_MASK_S) | (100u << G_AVG_S);
        }
    for (int n = 1; n <= NN; ++n) {
        int cs = 0, cb = t.node_team_start[n];                 // definitions.py:16-17
        for (int p = 0; p < NP; ++p)
            if (start[p] == n) { cs = (p == 0 ? 1 : -1) * t.node_control_points[n]; cb = p; }   // server.py:744-745,763-765
        D->init_node[n] = (uint32_t)(cs + 512) | ((uint32_t)(cb + 1) << 10);
    }

    for (int p = 0; p < NP; ++p)
        for (int g = 0; g < NG; ++g) {
            const int ty = t.group_type[p][g];
            D->speed_nib[p] |= (uint64_t)t.unit_speed[ty] << (4 * g);
            D->control_nib[p] |= (uint64_t)t.unit_control[ty] << (4 * g);
            D->cost_nib[p] |= (uint64_t)t.unit_cost[ty] << (4 * g);
        }
    for (int i = 1; i <= NN; ++i) D->p1inv_nib |= (uint64_t)i << (4 * t.p1_node_map[i]);
    // routing rows of cycle_target_node*.py (TAR_NODE[1], TAR_NODE[11]); 15 stands for the bots' -1 at the target itself
    static const int to1[11] = {-1, 1, 4, 1, 2, 3, 4, 5, 7, 7, 8}, to11[11] = {2, 5, 7, 7, 8, 9, 10, 11, 10, 11, -1};
    for (int c = 1; c <= NN; ++c) {
        D->tar_to_1 |= (uint64_t)(to1[c - 1] < 0 ? 15 : to1[c - 1]) << (4 * c);
        D->tar_to_11 |= (uint64_t)(to11[c - 1] < 0 ? 15 : to11[c - 1]) << (4 * c);
    }
    build_dfs_table(D);
    for (int n = 1; n <= NN; ++n) {
        int best = 0;
        for (int m = 1; m <= NN; ++m) if ((kBotConnections[n] >> m) & 1u) best = m;     // max(NODE_CONNECTIONS[n]) of swarm_agent.py:97: the bot's own constant
        D->maxnbr_nib |= (uint64_t)best << (4 * n);
        for (int m = 1; m <= NN; ++m) if (t.node_dist[n][m] > 0) D->nbr_mask[n] |= 1u << m;
    }

    // observation of the game_init state (everglades_env.py:158-171 over server.py:382-501)
    for (int p = 0; p < NP; ++p) {
        int16_t* o = D->reset_obs + p * OBS;
        o[0] = 0;
        for (int i = 1; i <= NN; ++i) {
            const int n = p == 1 ? t.p1_node_map[i] : i;                              // server.py:437-439
            o[1 + 4 * (i - 1) + 0] = (t.node_resource[n] & EVG_RES_DEFENSE) ? 1 : 0;
            o[1 + 4 * (i - 1) + 1] = (t.node_resource[n] & EVG_RES_OBSERVE) ? 1 : 0;
            o[1 + 4 * (i - 1) + 2] = (int16_t)((int)(D->init_node[n] & 0x3FF) - 512);
            o[1 + 4 * (i - 1) + 3] = (int16_t)(start[1 - p] == n ? NU : 0);
        }
        for (int g = 0; g < NG; ++g) {
            int16_t* q = o + 45 + 5 * g;
            q[0] = (int16_t)(p == 1 ? t.p1_node_map[start[p]] : start[p]);
            q[1] = (int16_t)t.group_type[p][g]; q[2] = 100; q[3] = 0; q[4] = (int16_t)kGroupSize[g];
        }
    }
    fill_lds_tables(D);
    return EVG_OK;
}

int evg_create(const evg_config* cfg, evg_handle** out) try {
    if (!cfg || !out) return fail(EVG_ERR_INVALID, "null argument");
    *out = nullptr;
    if (cfg->struct_size != sizeof(evg_config) || cfg->abi_version != EVG_ABI_VERSION)
        return fail(EVG_ERR_INVALID, "evg_config size/version mismatch (got %u/%u, want %zu/%d)", cfg->struct_size, cfg->abi_version,
                    sizeof(evg_config), EVG_ABI_VERSION);
    if (cfg->num_envs < 1) return fail(EVG_ERR_INVALID, "num_envs must be >= 1");
    // the random streams are keyed by the 32-bit global env id (csrc/evg_rng.h): ids past 2^32 would alias other envs' streams
    if (cfg->env_id_base > 0xFFFFFFFFull || cfg->env_id_base + (uint64_t)cfg->num_envs > 0x100000000ull)
        return fail(EVG_ERR_INVALID, "env_id_base + num_envs = %llu exceeds 2^32 (global env ids are 32-bit keys of the random streams)",
                    (unsigned long long)(cfg->env_id_base + (uint64_t)cfg->num_envs));
    if (cfg->obs_dtype < EVG_OBS_F32 || cfg->obs_dtype > EVG_OBS_I16) return fail(EVG_ERR_INVALID, "obs_dtype");
    if (cfg->rng_mode != EVG_RNG_KEYED_PHILOX && cfg->rng_mode != EVG_RNG_STOCK_MT19937) return fail(EVG_ERR_INVALID, "rng_mode");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(EVG_ERR_NO_DEVICE, "no HIP device visible; libevg has no CPU path");
    if (cfg->device_id < 0 || cfg->device_id >= ndev) return fail(EVG_ERR_NO_DEVICE, "device_id %d out of range (%d devices)", cfg->device_id, ndev);
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, cfg->device_id));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(EVG_ERR_NO_DEVICE, "device %d is %s; libevg is built for gfx950 (MI355X) only", cfg->device_id, prop.gcnArchName);
    DeviceGuard guard(cfg->device_id);              // the caller's current device is restored on every way out
    if (guard.err != hipSuccess) return fail(EVG_ERR_HIP, "selecting device %d failed: %s", cfg->device_id, hipGetErrorString(guard.err));

    evg_handle* h = new evg_handle();
    h->cfg = *cfg;
    int rc = build_dev_tables(cfg, &h->host_tables);
    if (rc != EVG_OK) { delete h; return rc; }
    memset(&h->caps, 0, sizeof(h->caps));
    rc = query_device_caps(cfg->device_id, cfg->obs_dtype, &h->caps);
    if (rc) { delete h; return fail(EVG_ERR_HIP, "querying the device's capacity failed: %s", hipGetErrorString((hipError_t)rc)); }
    if (cfg->cache_mib < 0 || cfg->cache_mib > (1 << 20)) { delete h; return fail(EVG_ERR_INVALID, "cache_mib"); }
    if (cfg->cache_mib > 0) h->caps.cache_bytes = (int64_t)cfg->cache_mib << 20;
    const size_t N = (size_t)cfg->num_envs;
    DevState& S = h->S;
    memset(&S, 0, sizeof(S));
    S.N = cfg->num_envs;
    S.auto_reset = cfg->auto_reset ? 1 : 0;
    S.seed_lo = (uint32_t)cfg->seed;
    S.seed_hi = (uint32_t)(cfg->seed >> 32);
    for (int r = 0; r < 10; ++r) { S.keys.k[2 * r] = S.seed_lo + (uint32_t)r * PHILOX_W0; S.keys.k[2 * r + 1] = S.seed_hi + (uint32_t)r * PHILOX_W1; }
    S.env_id_base = (uint32_t)cfg->env_id_base;
    rc = dev_alloc(h, &S.grp, 24 * N);
    if (!rc) rc = dev_alloc(h, &S.stamp, 6 * N);
    if (!rc) rc = dev_alloc(h, &S.node, 6 * N);
    if (!rc) rc = dev_alloc(h, &S.env, N);
    if (!rc) rc = dev_alloc(h, &S.episode, N);
    if (!rc) rc = dev_alloc(h, &S.health, 2 * NU * N);
    if (!rc) rc = dev_alloc(h, &S.ep_ret, 2 * N);
    if (!rc) rc = dev_alloc(h, &S.fin_ret, 2 * N);
    if (!rc) rc = dev_alloc(h, &S.fin_len, N);
    if (!rc) rc = dev_alloc(h, &S.fin_win, N);
    if (!rc) rc = dev_alloc(h, &S.totals, 4);
    if (!rc) rc = dev_alloc(h, &S.agent_cycle, 2 * N);
    if (!rc) rc = dev_alloc(h, &S.agent_swarm, 2 * N);
    if (!rc) rc = dev_alloc(h, &S.agent_dfs, 2 * N);
    // 16 queue counters on lines of their own (the create-time XCD probe borrows the 1 024 words) ...
    if (!rc) rc = dev_alloc(h, &S.queue, 1024 + (N + 31) / 32 + 1);
    if (!rc) S.progress = S.queue + 1024;                                // ... and the per-set progress flags behind them: one memset zeroes both
    if (!rc) rc = dev_alloc(h, &S.handoff, 2 * N);                        // per-lane hand-over checksums of chunked launches (written before they are read)
    if (!rc) rc = dev_alloc(h, &S.fault, 1);
    if (!rc) {        // host-mapped mirror of "the fault word is not zero" (check_fault)
        void* hp = nullptr;
        void* dp = nullptr;
        hipError_t he = hipHostMalloc(&hp, sizeof(uint32_t), hipHostMallocMapped);
        if (he == hipSuccess) {
            *reinterpret_cast<uint32_t*>(hp) = 0u;
            h->fault_seen_host = reinterpret_cast<uint32_t*>(hp);
            he = hipHostGetDevicePointer(&dp, hp, 0);
        }
        if (he != hipSuccess) rc = fail(EVG_ERR_ALLOC, "mapped host word: %s", hipGetErrorString(he));
        S.fault_seen = reinterpret_cast<uint32_t*>(dp);
    }
    if (!rc) rc = dev_alloc(h, &h->d_tables, 1);
    uint32_t *mt_key = nullptr, *mt_pos = nullptr;       // attached to S after the create-time reset, which must not draw
    if (!rc && cfg->rng_mode == EVG_RNG_STOCK_MT19937) {
        rc = dev_alloc(h, &mt_key, (size_t)MT_N * N);
        if (!rc) rc = dev_alloc(h, &mt_pos, N);
    }
    if (rc) { evg_destroy(h); return rc; }
    S.T = h->d_tables;
#ifdef EVG_STAMPS
    rc = dev_alloc(h, &h->stamps, (size_t)((cfg->num_envs + 15) / 16) * 16);
    if (rc) { evg_destroy(h); return rc; }
#endif
    {
        // The chunked form's hand-over (evg_kernels.hip, "WHAT THIS RELIES ON") assumes ordinary coarse-grained device memory, cached in the L2 of the XCD
        // that touches it: refuse to run on anything else (managed or host memory behind these pointers) instead of corrupting state silently.
        const void* must_be_device[] = {S.grp, S.stamp, S.node, S.env, S.episode, S.health, S.ep_ret, S.fin_ret, S.fin_len, S.fin_win, S.progress, S.handoff};
        for (const void* q : must_be_device) {
            hipPointerAttribute_t at;
            const hipError_t ae = hipPointerGetAttributes(&at, q);
            if (ae != hipSuccess || at.type != hipMemoryTypeDevice || at.isManaged) {
                evg_destroy(h);
                return fail(EVG_ERR_HIP, "the state arrays must be plain device memory (hipMalloc); got type %d managed %d (%s)",
                            ae == hipSuccess ? (int)at.type : -1, ae == hipSuccess ? (int)at.isManaged : -1, hipGetErrorString(ae));
            }
        }
    }
    hipError_t e = hipMemcpy(h->d_tables, &h->host_tables, sizeof(DevTables), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(S.episode, 0, N * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemset(S.fin_ret, 0, 2 * N * sizeof(float));
    if (e == hipSuccess) e = hipMemset(S.fin_len, 0, N * sizeof(int32_t));
    if (e == hipSuccess) e = hipMemset(S.fin_win, 0xFF, N);
    if (e == hipSuccess) e = hipMemset(S.totals, 0, 4 * sizeof(unsigned long long));
    if (e == hipSuccess) e = hipMemset(S.fault, 0, sizeof(uint32_t));
    if (e == hipSuccess) {
        // the XCDs of this device (8 on a whole MI355X; a partitioned device has fewer): 1 024 one-wave workgroups report their XCC id
        std::vector<uint32_t> ids(1024, 0xFFu);
        e = (hipError_t)launch_xcd_probe(S.queue, nullptr);
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (e == hipSuccess) e = hipMemcpy(ids.data(), S.queue, ids.size() * sizeof(uint32_t), hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemset(S.queue, 0, (1024 + (N + 31) / 32 + 1) * sizeof(uint32_t));
        uint32_t seen = 0;
        for (uint32_t v : ids) if (v < 16u) seen |= 1u << v;
        S.xcd_rank = ~0ull;
        S.nxcd = 0;
        for (uint32_t x = 0; x < 16u; ++x)
            if ((seen >> x) & 1u) { S.xcd_rank = (S.xcd_rank & ~(15ull << (4u * x))) | ((uint64_t)S.nxcd << (4u * x)); S.nxcd += 1; }
        if (e == hipSuccess && (S.nxcd < 1 || S.nxcd > 15)) { evg_destroy(h); return fail(EVG_ERR_HIP, "XCD probe saw %d XCDs", S.nxcd); }
    }
    // every env starts in the game_init position; the episode counter is then set to -1 so that the
    // first evg_reset opens episode 0
    // the launchers return the hipError_t of their own launch (0 = success); it is propagated as it is
    if (e == hipSuccess) e = (hipError_t)launch_reset(S, nullptr, nullptr, cfg->obs_dtype, nullptr);
    if (e == hipSuccess) e = (hipError_t)launch_scripted_reset(S, nullptr);
    if (e == hipSuccess && mt_key) {
        S.mt_key = mt_key; S.mt_pos = mt_pos;
        e = (hipError_t)launch_mt_seed(S, nullptr, nullptr);
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemset(S.episode, 0xFF, N * sizeof(uint32_t));
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) { evg_destroy(h); return fail(EVG_ERR_HIP, "state initialisation failed: %s", hipGetErrorString(e)); }
    *out = h;
    return EVG_OK;
} catch (...) { return on_exception(); }

void evg_destroy(evg_handle* h) {
    if (!h) return;
    DeviceGuard guard(h->cfg.device_id);
    drop_graphs(h);
    (void)evg_comm_destroy(h);
    if (h->capture_stream) (void)hipStreamDestroy(h->capture_stream);
    for (void* p : h->allocs) (void)hipFree(p);
    if (h->fault_seen_host) (void)hipHostFree(h->fault_seen_host);
    for (hipEvent_t ev : h->events) (void)hipEventDestroy(ev);
    delete h;
}

int evg_num_envs(const evg_handle* h) { return h ? h->S.N : 0; }

int evg_state_bytes_per_env(const evg_handle* h) try {
    (void)h;
    return kStateBytesPerEnv;       // grp, stamp, node, env, episode, health, ep_ret (evg_device.h)
} catch (...) { return on_exception(); }

int evg_reset(evg_handle* h, const uint8_t* mask, void* obs_out, void* stream) try {
    if (!h) return fail(EVG_ERR_INVALID, "null handle");
    EVG_NEED_ALIGNED16(obs_out);
    EVG_ON_DEVICE(h);
    const int rc = launch_reset(h->S, mask, obs_out, h->cfg.obs_dtype, stream);
    if (rc) return fail(EVG_ERR_HIP, "reset launch failed: %s", hipGetErrorString((hipError_t)rc));
    return EVG_OK;
} catch (...) { return on_exception(); }

int evg_step(evg_handle* h, const int32_t* actions, void* obs_out, float* reward_out, uint8_t* done_out, int8_t* winner_out,
             int32_t* scores_out, uint8_t* status_out, void* stream) try {
    if (!h) return fail(EVG_ERR_INVALID, "null handle");
    if (!actions || !reward_out || !done_out) return fail(EVG_ERR_INVALID, "actions, reward_out and done_out are required");
    EVG_NEED_ALIGNED16(actions); EVG_NEED_ALIGNED16(obs_out); EVG_NEED_ALIGNED8(reward_out); EVG_NEED_ALIGNED8(scores_out);
    EVG_ON_DEVICE(h);
    const StepIO io = make_io(h, actions, obs_out, reward_out, done_out, winner_out, scores_out, status_out, 0, 0, 0, 0, nullptr);
    const int rc = launch_step(h->S, io, h->cfg.obs_dtype, h->caps, stream);
    if (rc) return fail(EVG_ERR_HIP, "step launch failed: %s", hipGetErrorString((hipError_t)rc));
    return EVG_OK;
} catch (...) { return on_exception(); }

int evg_observe(evg_handle* h, void* obs_out, void* stream) try {
    if (!h || !obs_out) return fail(EVG_ERR_INVALID, "null argument");
    EVG_NEED_ALIGNED16(obs_out);
    EVG_ON_DEVICE(h);
    const StepIO io = make_io(h, nullptr, obs_out, nullptr, nullptr, nullptr, nullptr, nullptr, 1, 0, 0, 0, nullptr);
    const int rc = launch_step(h->S, io, h->cfg.obs_dtype, h->caps, stream);
    if (rc) return fail(EVG_ERR_HIP, "observe launch failed: %s", hipGetErrorString((hipError_t)rc));
    return EVG_OK;
} catch (...) { return on_exception(); }

static int step_vs_policy_impl(evg_handle* h, int seat, const int32_t* actions, int actions_both_seats, int opponent_policy, void* obs_seat_out,
                               float* shared_out, float* swarm_out, float* reward_out, uint8_t* done_out, int8_t* winner_out, int32_t* scores_out,
                               uint8_t* status_out, void* stream) {
    if (!h) return fail(EVG_ERR_INVALID, "null handle");
    if (!actions || !obs_seat_out || !reward_out || !done_out) return fail(EVG_ERR_INVALID, "actions, obs_seat_out, reward_out and done_out are required");
    if (seat < 0 || seat > 1 || opponent_policy < 0 || opponent_policy >= EVG_POLICY_COUNT) return fail(EVG_ERR_INVALID, "seat / opponent_policy out of range");
    EVG_NEED_ALIGNED16(obs_seat_out); EVG_NEED_ALIGNED16(actions); EVG_NEED_ALIGNED8(reward_out); EVG_NEED_ALIGNED8(scores_out);
    EVG_NEED_ALIGNED8(shared_out); EVG_NEED_ALIGNED16(swarm_out);
    if (h->S.mt_key) return fail(EVG_ERR_INVALID, "evg_step_vs_policy: keyed-Philox handles only (the stock-entropy mode has no fused bots)");
    EVG_ON_DEVICE(h);
    StepIO io = make_io(h, actions, obs_seat_out, reward_out, done_out, winner_out, scores_out, status_out, 0, 2, opponent_policy, opponent_policy, nullptr);
    io.seat = seat; io.actions_both = actions_both_seats ? 1 : 0;
    io.feat_shared = shared_out; io.feat_swarm = swarm_out;
    const int rc = launch_step_seat(h->S, io, h->cfg.obs_dtype, h->caps, stream);
    if (rc) return fail(EVG_ERR_HIP, "step launch failed: %s", hipGetErrorString((hipError_t)rc));
    return EVG_OK;
}

int evg_step_vs_policy(evg_handle* h, int seat, const int32_t* actions, int actions_both_seats, int opponent_policy, void* obs_seat_out, float* reward_out,
                       uint8_t* done_out, int8_t* winner_out, int32_t* scores_out, uint8_t* status_out, void* stream) try {
    return step_vs_policy_impl(h, seat, actions, actions_both_seats, opponent_policy, obs_seat_out, nullptr, nullptr, reward_out, done_out, winner_out, scores_out,
                               status_out, stream);
} catch (...) { return on_exception(); }

int evg_step_vs_policy_smart(evg_handle* h, int seat, const int32_t* actions, int actions_both_seats, int opponent_policy, void* obs_seat_out,
                             float* shared_out, float* swarm_out, float* reward_out, uint8_t* done_out, int8_t* winner_out, int32_t* scores_out,
                             uint8_t* status_out, void* stream) try {
    if (!shared_out || !swarm_out) return fail(EVG_ERR_INVALID, "evg_step_vs_policy_smart: shared_out and swarm_out are required (evg_step_vs_policy is the form without)");
    return step_vs_policy_impl(h, seat, actions, actions_both_seats, opponent_policy, obs_seat_out, shared_out, swarm_out, reward_out, done_out, winner_out,
                               scores_out, status_out, stream);
} catch (...) { return on_exception(); }

int evg_observe_seat(evg_handle* h, int seat, void* obs_seat_out, void* stream) try {
    if (!h || !obs_seat_out || seat < 0 || seat > 1) return fail(EVG_ERR_INVALID, "bad argument");
    EVG_NEED_ALIGNED16(obs_seat_out);
    if (h->S.mt_key) return fail(EVG_ERR_INVALID, "evg_observe_seat: keyed-Philox handles only");
    EVG_ON_DEVICE(h);
    StepIO io = make_io(h, nullptr, obs_seat_out, nullptr, nullptr, nullptr, nullptr, nullptr, 1, 0, 0, 0, nullptr);
    io.seat = seat;
    const int rc = launch_step_seat(h->S, io, h->cfg.obs_dtype, h->caps, stream);
    if (rc) return fail(EVG_ERR_HIP, "observe launch failed: %s", hipGetErrorString((hipError_t)rc));
    return EVG_OK;
} catch (...) { return on_exception(); }

int evg_random_actions(evg_handle* h, int32_t* actions_out, void* stream) try {
    if (!h || !actions_out) return fail(EVG_ERR_INVALID, "null argument");
    EVG_NEED_ALIGNED16(actions_out);
    EVG_ON_DEVICE(h);
    const int rc = launch_random_actions(h->S, actions_out, -1, stream);
    if (rc) return fail(EVG_ERR_HIP, "random_actions launch failed: %s", hipGetErrorString((hipError_t)rc));
    return EVG_OK;
} catch (...) { return on_exception(); }

int evg_random_actions_seat(evg_handle* h, int seat, int32_t* actions_seat_out, void* stream) try {
    if (!h || !actions_seat_out || seat < 0 || seat > 1) return fail(EVG_ERR_INVALID, "bad argument");
    EVG_NEED_ALIGNED16(actions_seat_out);
    EVG_ON_DEVICE(h);
    const int rc = launch_random_actions(h->S, actions_seat_out, seat, stream);
    if (rc) return fail(EVG_ERR_HIP, "random_actions launch failed: %s", hipGetErrorString((hipError_t)rc));
    return EVG_OK;
} catch (...) { return on_exception(); }

int evg_scripted_actions(evg_handle* h, int policy, int player, const void* obs, int32_t* actions_out, void* stream) try {
    if (!h || !obs || !actions_out) return fail(EVG_ERR_INVALID, "null argument");
    if (policy < 0 || policy >= EVG_POLICY_COUNT || player < 0 || player > 1) return fail(EVG_ERR_INVALID, "policy/player out of range");
    EVG_NEED_ALIGNED16(obs); EVG_NEED_ALIGNED16(actions_out);
    EVG_ON_DEVICE(h);
    const int rc = launch_scripted_actions(h->S, policy, player, obs, actions_out, h->cfg.obs_dtype, stream);
    if (rc) return fail(EVG_ERR_HIP, "scripted_actions launch failed: %s", hipGetErrorString((hipError_t)rc));
    return EVG_OK;
} catch (...) { return on_exception(); }

int evg_fog_of_war(evg_handle* h, uint8_t* fog_out, uint8_t* knowledge_out, void* stream) try {
    if (!h || (!fog_out && !knowledge_out)) return fail(EVG_ERR_INVALID, "null argument");
    EVG_ON_DEVICE(h);
    const int rc = launch_fog(h->S, fog_out, knowledge_out, nullptr, stream);
    if (rc) return fail(EVG_ERR_HIP, "fog launch failed: %s", hipGetErrorString((hipError_t)rc));
    return EVG_OK;
} catch (...) { return on_exception(); }

int evg_sightings(evg_handle* h, int8_t* sight_out, void* stream) try {
    if (!h || !sight_out) return fail(EVG_ERR_INVALID, "null argument");
    EVG_ON_DEVICE(h);
    const int rc = launch_fog(h->S, nullptr, nullptr, sight_out, stream);
    if (rc) return fail(EVG_ERR_HIP, "sightings launch failed: %s", hipGetErrorString((hipError_t)rc));
    return EVG_OK;
} catch (...) { return on_exception(); }

int evg_smart_state(evg_handle* h, int player, const void* obs, float* features_out, void* stream) try {
    if (!h || !obs || !features_out || player < 0 || player > 1) return fail(EVG_ERR_INVALID, "bad argument");
    EVG_NEED_ALIGNED16(features_out); EVG_NEED_ALIGNED16(obs);
    EVG_ON_DEVICE(h);
    const int rc = launch_smart_state(h->S, player, obs, 0, features_out, nullptr, h->cfg.obs_dtype, stream);
    if (rc) return fail(EVG_ERR_HIP, "smart_state launch failed: %s", hipGetErrorString((hipError_t)rc));
    return EVG_OK;
} catch (...) { return on_exception(); }

int evg_smart_state_seat(evg_handle* h, const void* obs_seat, float* features_out, void* stream) try {
    if (!h || !obs_seat || !features_out) return fail(EVG_ERR_INVALID, "bad argument");
    EVG_NEED_ALIGNED16(features_out); EVG_NEED_ALIGNED16(obs_seat);
    EVG_ON_DEVICE(h);
    const int rc = launch_smart_state(h->S, 0, obs_seat, 1, features_out, nullptr, h->cfg.obs_dtype, stream);
    if (rc) return fail(EVG_ERR_HIP, "smart_state launch failed: %s", hipGetErrorString((hipError_t)rc));
    return EVG_OK;
} catch (...) { return on_exception(); }

int evg_smart_state_compact(evg_handle* h, int player, const void* obs, float* shared_out, float* swarm_out, void* stream) try {
    if (!h || !obs || !shared_out || !swarm_out || player < -1 || player > 1) return fail(EVG_ERR_INVALID, "bad argument");
    if ((reinterpret_cast<uintptr_t>(shared_out) & 7u) != 0) return fail(EVG_ERR_INVALID, "shared_out must be 8-byte aligned");
    EVG_NEED_ALIGNED16(swarm_out); EVG_NEED_ALIGNED16(obs);
    EVG_ON_DEVICE(h);
    const int rc = launch_smart_state(h->S, player < 0 ? 0 : player, obs, player < 0 ? 1 : 0, shared_out, swarm_out, h->cfg.obs_dtype, stream);
    if (rc) return fail(EVG_ERR_HIP, "smart_state launch failed: %s", hipGetErrorString((hipError_t)rc));
    return EVG_OK;
} catch (...) { return on_exception(); }

int evg_smart_actions(evg_handle* h, int player, const void* obs, const float* q, int32_t* actions_out, int32_t* directions_out, void* stream) try {
    if (!h || !obs || !q || !actions_out || player < -1 || player > 1) return fail(EVG_ERR_INVALID, "bad argument");
    EVG_NEED_ALIGNED16(obs); EVG_NEED_ALIGNED16(q); EVG_NEED_ALIGNED16(actions_out); EVG_NEED_ALIGNED16(directions_out);
    EVG_ON_DEVICE(h);
    const int rc = launch_smart_actions(h->S, player < 0 ? 0 : player, obs, player < 0 ? 1 : 0, q, actions_out, directions_out, h->cfg.obs_dtype, stream);
    if (rc) return fail(EVG_ERR_HIP, "smart_actions launch failed: %s", hipGetErrorString((hipError_t)rc));
    return EVG_OK;
} catch (...) { return on_exception(); }

int evg_smart_get_action(evg_handle* h, int seat, int obs_one_seat, const void* obs, const float* q, float epsilon, const float* epsilon_env,
                         int32_t* actions_out, int32_t* directions_out, uint8_t* explored_out, void* stream) try {
    if (!h || !obs || !q || !actions_out || seat < 0 || seat > 1) return fail(EVG_ERR_INVALID, "bad argument");
    if (!epsilon_env && !(epsilon >= 0.0f && epsilon <= 1.0f)) return fail(EVG_ERR_INVALID, "epsilon %g outside [0, 1]", (double)epsilon);
    EVG_NEED_ALIGNED16(obs); EVG_NEED_ALIGNED16(q); EVG_NEED_ALIGNED16(actions_out); EVG_NEED_ALIGNED16(directions_out);
    EVG_ON_DEVICE(h);
    const SmartExplore ex{seat, epsilon, epsilon_env, explored_out};
    const int rc = launch_smart_actions(h->S, seat, obs, obs_one_seat ? 1 : 0, q, actions_out, directions_out, h->cfg.obs_dtype, stream, &ex);
    if (rc) return fail(EVG_ERR_HIP, "smart_get_action launch failed: %s", hipGetErrorString((hipError_t)rc));
    return EVG_OK;
} catch (...) { return on_exception(); }

void evg_move_table(int32_t* table /* [11][5] */) {
    // agents/Smart_State/Move_Translation.py:3-97: node reached from (0-indexed) node n0 in direction 0 left, 1 right, 2 up, 3 down, 4 stay
    static const int32_t T[5][11] = {{1, 1, 3, 1, 2, 3, 4, 5, 6, 7, 11}, {1, 5, 6, 7, 8, 9, 10, 11, 9, 11, 11}, {2, 2, 2, 3, 5, 6, 7, 8, 8, 9, 8},
                                     {4, 3, 4, 4, 5, 6, 7, 9, 10, 10, 10}, {1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11}};
    for (int n = 0; n < 11; ++n) for (int d = 0; d < 5; ++d) table[n * 5 + d] = T[d][n];
}

int evg_scripted_reset(evg_handle* h, void* stream) try {
    if (!h) return fail(EVG_ERR_INVALID, "null handle");
    EVG_ON_DEVICE(h);
    const int rc = launch_scripted_reset(h->S, stream);
    if (rc) return fail(EVG_ERR_HIP, "scripted_reset launch failed: %s", hipGetErrorString((hipError_t)rc));
    return EVG_OK;
} catch (...) { return on_exception(); }

static int rollout_impl(evg_handle* h, int steps, int fused, int policy0, int policy1, int32_t* actions_buf, void* obs_out, float* reward_out,
                        uint8_t* done_out, int8_t* winner_out, int32_t* scores_out, uint8_t* status_out, float* step_kernel_ms, void* stream);

int evg_rollout_random(evg_handle* h, int steps, int fused, int32_t* actions_buf, void* obs_out, float* reward_out, uint8_t* done_out,
                       int8_t* winner_out, int32_t* scores_out, uint8_t* status_out, float* step_kernel_ms, void* stream) try {
    return rollout_impl(h, steps, fused, EVG_POLICY_RANDOM, EVG_POLICY_RANDOM, actions_buf, obs_out, reward_out, done_out, winner_out, scores_out,
                        status_out, step_kernel_ms, stream);
} catch (...) { return on_exception(); }

int evg_rollout_policies(evg_handle* h, int steps, int fused, int policy0, int policy1, int32_t* actions_buf, void* obs_out, float* reward_out,
                         uint8_t* done_out, int8_t* winner_out, int32_t* scores_out, uint8_t* status_out, float* step_kernel_ms, void* stream) try {
    if (policy0 < 0 || policy0 >= EVG_POLICY_COUNT || policy1 < 0 || policy1 >= EVG_POLICY_COUNT)
        return fail(EVG_ERR_INVALID, "policy out of range");
    if (!obs_out && !fused) return fail(EVG_ERR_INVALID, "rollout_policies: obs_out is required (the agents read it)");
    return rollout_impl(h, steps, fused, policy0, policy1, actions_buf, obs_out, reward_out, done_out, winner_out, scores_out, status_out,
                        step_kernel_ms, stream);
} catch (...) { return on_exception(); }

static int rollout_impl(evg_handle* h, int steps, int fused, int policy0, int policy1, int32_t* actions_buf, void* obs_out, float* reward_out,
                        uint8_t* done_out, int8_t* winner_out, int32_t* scores_out, uint8_t* status_out, float* step_kernel_ms, void* stream) {
    if (!h) return fail(EVG_ERR_INVALID, "null handle");
    // steps < 0: PREPARE a rollout of -steps turns with exactly these arguments -- capture and instantiate the graphs of its launches, run nothing -- so
    // that the first real call pays no one-time cost (evg.h)
    const bool prepare_only = steps < 0;
    if (prepare_only) steps = -steps;
    if (steps < 1 || !reward_out || !done_out) return fail(EVG_ERR_INVALID, "rollout: steps >= 1, reward_out, done_out required");
    if (!actions_buf && !fused) return fail(EVG_ERR_INVALID, "rollout: actions_buf is required unless the step kernel produces the orders itself (fused >= 1)");
    EVG_NEED_ALIGNED16(actions_buf); EVG_NEED_ALIGNED16(obs_out); EVG_NEED_ALIGNED8(reward_out); EVG_NEED_ALIGNED8(scores_out);
    EVG_ON_DEVICE(h);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    hipStream_t s_ = s;
    const bool random_pair = policy0 == EVG_POLICY_RANDOM && policy1 == EVG_POLICY_RANDOM;
    const int gen_mode = random_pair ? 1 : 2;          // what the step kernel draws itself when fused
    if (h->S.mt_key && fused > 1) fused = 1;           // stock-entropy mode: single-turn launches only
    if (fused >= 2) {
        // Persistent form: one launch plays up to `fused` consecutive turns per wavefront (state stays on chip, outputs are
        // written every turn).  step_kernel_ms then is the launch time divided by the turns it played.
        const int per_launch = fused;
        const int nlaunch = (steps + per_launch - 1) / per_launch;
        while (step_kernel_ms && h->events.size() < (size_t)2 * nlaunch) {
            hipEvent_t ev;
            HIP_TRY(hipEventCreate(&ev));
            h->events.push_back(ev);
        }
        StepIO io = make_io(h, actions_buf, obs_out, reward_out, done_out, winner_out, scores_out, status_out, 0, gen_mode, policy0, policy1, actions_buf);
        int done_turns = 0;
        for (int l = 0; l < nlaunch; ++l) {
            io.turns = steps - done_turns < per_launch ? steps - done_turns : per_launch;
            if (prepare_only) { (void)graph_of(h, io); done_turns += io.turns; continue; }
            if (step_kernel_ms) HIP_TRY(hipEventRecord(h->events[2 * l], s_));
            const int rc = launch_rollout(h, io, s_);
            if (rc) return fail(EVG_ERR_HIP, "step launch failed: %s", hipGetErrorString((hipError_t)rc));
            if (step_kernel_ms) HIP_TRY(hipEventRecord(h->events[2 * l + 1], s_));
            done_turns += io.turns;
        }
        if (step_kernel_ms && !prepare_only) {
            HIP_TRY(hipStreamSynchronize(s_));
            double tot = 0.0;
            for (int l = 0; l < nlaunch; ++l) {
                float ms = 0.f;
                HIP_TRY(hipEventElapsedTime(&ms, h->events[2 * l], h->events[2 * l + 1]));
                tot += ms;
            }
            *step_kernel_ms = (float)(tot / steps);
            return check_fault(h);            // the call has synchronised: a chunk hand-over fault of these launches is reported here
        }
        return EVG_OK;
    }
    if (prepare_only) return EVG_OK;            // single-turn launches are enqueued plainly: nothing to prepare
    // One launch per turn: the loop is timed as a whole with two events on the stream (an event pair around every launch would
    // make the queue wait for each bracketed kernel to retire and stretch what it measures): step_kernel_ms is the stream
    // time per turn -- the step kernel, the gap to the next launch and, when the orders are not drawn by the step kernel
    // itself (fused == 0), the action kernel(s) of the turn.  The kernel alone is in the rocprofv3 traces under profiles/.
    while (step_kernel_ms && h->events.size() < 2) {
        hipEvent_t ev;
        HIP_TRY(hipEventCreate(&ev));
        h->events.push_back(ev);
    }
    const StepIO io = make_io(h, actions_buf, obs_out, reward_out, done_out, winner_out, scores_out, status_out, 0, fused ? gen_mode : 0, policy0, policy1,
                              actions_buf);
    if (step_kernel_ms) HIP_TRY(hipEventRecord(h->events[0], s_));
    for (int i = 0; i < steps; ++i) {
        int rc = 0;
        if (fused) {
            // orders are drawn inside the step kernel
        } else if (random_pair) {
            rc = launch_random_actions(h->S, actions_buf, -1, stream);
        } else {                                  // the agents read the observations of the previous turn from obs_out
            rc = launch_scripted_actions(h->S, policy0, 0, obs_out, actions_buf, h->cfg.obs_dtype, stream);
            if (!rc) rc = launch_scripted_actions(h->S, policy1, 1, obs_out, actions_buf, h->cfg.obs_dtype, stream);
        }
        if (rc) return fail(EVG_ERR_HIP, "action kernel launch failed: %s", hipGetErrorString((hipError_t)rc));
        rc = launch_step(h->S, io, h->cfg.obs_dtype, h->caps, stream);
        if (rc) return fail(EVG_ERR_HIP, "step launch failed: %s", hipGetErrorString((hipError_t)rc));
    }
    if (step_kernel_ms) {
        HIP_TRY(hipEventRecord(h->events[1], s_));
        HIP_TRY(hipStreamSynchronize(s_));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, h->events[0], h->events[1]));
        *step_kernel_ms = ms / (float)steps;
        return check_fault(h);
    }
    return EVG_OK;
}

int evg_rollout_vs_policy(evg_handle* h, int steps, int seat, int opponent_policy, int32_t* actions_seat_buf, void* obs_seat_out, float* reward_out,
                          uint8_t* done_out,
                          int8_t* winner_out, int32_t* scores_out, uint8_t* status_out, float* step_kernel_ms, void* stream) try {
    if (!h) return fail(EVG_ERR_INVALID, "null handle");
    if (steps < 1 || !actions_seat_buf || !obs_seat_out || !reward_out || !done_out)
        return fail(EVG_ERR_INVALID, "rollout_vs_policy: steps >= 1, actions_seat_buf, obs_seat_out, reward_out, done_out required");
    if (seat < 0 || seat > 1 || opponent_policy < 0 || opponent_policy >= EVG_POLICY_COUNT) return fail(EVG_ERR_INVALID, "seat / opponent_policy out of range");
    EVG_NEED_ALIGNED16(obs_seat_out); EVG_NEED_ALIGNED16(actions_seat_buf); EVG_NEED_ALIGNED8(reward_out); EVG_NEED_ALIGNED8(scores_out);
    if (h->S.mt_key) return fail(EVG_ERR_INVALID, "evg_rollout_vs_policy: keyed-Philox handles only");
    EVG_ON_DEVICE(h);
    hipStream_t s_ = reinterpret_cast<hipStream_t>(stream);
    while (step_kernel_ms && h->events.size() < 2) {
        hipEvent_t ev;
        HIP_TRY(hipEventCreate(&ev));
        h->events.push_back(ev);
    }
    StepIO io = make_io(h, actions_seat_buf, obs_seat_out, reward_out, done_out, winner_out, scores_out, status_out, 0, 2, opponent_policy, opponent_policy,
                        nullptr);
    io.seat = seat; io.actions_both = 0;
    if (step_kernel_ms) HIP_TRY(hipEventRecord(h->events[0], s_));
    for (int i = 0; i < steps; ++i) {
        int rc = launch_random_actions(h->S, actions_seat_buf, seat, stream);
        if (!rc) rc = launch_step_seat(h->S, io, h->cfg.obs_dtype, h->caps, stream);
        if (rc) return fail(EVG_ERR_HIP, "launch failed: %s", hipGetErrorString((hipError_t)rc));
    }
    if (step_kernel_ms) {
        HIP_TRY(hipEventRecord(h->events[1], s_));
        HIP_TRY(hipStreamSynchronize(s_));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, h->events[0], h->events[1]));
        *step_kernel_ms = ms / (float)steps;
        return check_fault(h);
    }
    return EVG_OK;
} catch (...) { return on_exception(); }

int evg_seed_stock_entropy(evg_handle* h, const uint32_t* seeds, void* stream) try {
    if (!h) return fail(EVG_ERR_INVALID, "null handle");
    if (!h->S.mt_key) return fail(EVG_ERR_INVALID, "handle was not created with rng_mode = EVG_RNG_STOCK_MT19937");
    EVG_ON_DEVICE(h);
    const uint32_t* d_seeds = nullptr;
    if (seeds) {                                        // staged in the (about to be overwritten) key array itself: row 623 is written last
        uint32_t* stage = h->S.mt_key + (size_t)(MT_N - 1) * h->S.N;
        HIP_TRY(hipStreamSynchronize(reinterpret_cast<hipStream_t>(stream)));
        HIP_TRY(hipMemcpy(stage, seeds, (size_t)h->S.N * sizeof(uint32_t), hipMemcpyHostToDevice));
        d_seeds = stage;
    }
    const int rc = launch_mt_seed(h->S, d_seeds, stream);
    if (rc) return fail(EVG_ERR_HIP, "seed launch failed: %s", hipGetErrorString((hipError_t)rc));
    return EVG_OK;
} catch (...) { return on_exception(); }

int evg_get_stock_entropy(evg_handle* h, uint32_t* out) try {
    if (!h || !out) return fail(EVG_ERR_INVALID, "null argument");
    if (!h->S.mt_key) return fail(EVG_ERR_INVALID, "handle was not created with rng_mode = EVG_RNG_STOCK_MT19937");
    EVG_ON_DEVICE(h);
    HIP_TRY(hipDeviceSynchronize());
    const size_t N = (size_t)h->S.N;
    std::vector<uint32_t> k((size_t)MT_N * N), pos(N);
    HIP_TRY(hipMemcpy(k.data(), h->S.mt_key, k.size() * 4, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(pos.data(), h->S.mt_pos, N * 4, hipMemcpyDeviceToHost));
    for (size_t e = 0; e < N; ++e) {
        for (int i = 0; i < MT_N; ++i) out[e * (MT_N + 1) + i] = k[(size_t)i * N + e];
        out[e * (MT_N + 1) + MT_N] = pos[e];
    }
    return EVG_OK;
} catch (...) { return on_exception(); }

int evg_set_stock_entropy(evg_handle* h, const uint32_t* in) try {
    if (!h || !in) return fail(EVG_ERR_INVALID, "null argument");
    if (!h->S.mt_key) return fail(EVG_ERR_INVALID, "handle was not created with rng_mode = EVG_RNG_STOCK_MT19937");
    EVG_ON_DEVICE(h);
    HIP_TRY(hipDeviceSynchronize());
    const size_t N = (size_t)h->S.N;
    std::vector<uint32_t> k((size_t)MT_N * N), pos(N);
    for (size_t e = 0; e < N; ++e) {
        for (int i = 0; i < MT_N; ++i) k[(size_t)i * N + e] = in[e * (MT_N + 1) + i];
        pos[e] = in[e * (MT_N + 1) + MT_N];
        if (pos[e] > (uint32_t)MT_N) return fail(EVG_ERR_INVALID, "env %zu: generator position %u > 624", e, pos[e]);
    }
    HIP_TRY(hipMemcpy(h->S.mt_key, k.data(), k.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->S.mt_pos, pos.data(), N * 4, hipMemcpyHostToDevice));
    return EVG_OK;
} catch (...) { return on_exception(); }

int evg_get_state(evg_handle* h, int32_t* groups, int32_t* nodes, double* health, int32_t* env) try {
    if (!h) return fail(EVG_ERR_INVALID, "null handle");
    EVG_ON_DEVICE(h);
    HIP_TRY(hipDeviceSynchronize());
    { const int frc = check_fault(h); if (frc) return frc; }
    const size_t N = (size_t)h->S.N;
    if (groups) {
        std::vector<uint32_t> g(24 * N), st(6 * N);
        HIP_TRY(hipMemcpy(g.data(), h->S.grp, g.size() * 4, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(st.data(), h->S.stamp, st.size() * 4, hipMemcpyDeviceToHost));
        for (size_t e = 0; e < N; ++e)
            for (int r = 0; r < 24; ++r) {
                const uint32_t w = g[(size_t)r * N + e];
                int32_t* o = groups + (e * 24 + r) * 8;
                const uint32_t mode = (w & G_MODE_M) >> G_MODE_S, dest = (w & G_DEST_M) >> G_DEST_S;
                const int cnt = __builtin_popcount(w & G_MASK_M);
                o[0] = (int32_t)(w & G_LOC_M); o[1] = dest ? (int32_t)dest : -1; o[2] = (int32_t)((w & G_DIST_M) >> G_DIST_S);
                o[3] = mode == MODE_READY; o[4] = mode == MODE_MOVING; o[5] = cnt == 0; o[6] = cnt;
                o[7] = (int32_t)((st[(size_t)(r >> 2) * N + e] >> (8 * (r & 3))) & 0xFFu);
            }
    }
    if (nodes) {
        std::vector<uint32_t> nd(6 * N);
        HIP_TRY(hipMemcpy(nd.data(), h->S.node, nd.size() * 4, hipMemcpyDeviceToHost));
        for (size_t e = 0; e < N; ++e)
            for (int n = 0; n < NN; ++n) {
                const uint32_t w = (nd[(size_t)(n >> 1) * N + e] >> (16 * (n & 1))) & 0xFFFFu;
                nodes[(e * NN + n) * 2] = (int32_t)(w & 0x3FF) - 512;
                nodes[(e * NN + n) * 2 + 1] = (int32_t)((w >> 10) & 3) - 1;
            }
    }
    if (health) HIP_TRY(hipMemcpy(health, h->S.health, 2 * NU * N * sizeof(double), hipMemcpyDeviceToHost));
    if (env) {
        std::vector<uint32_t> ev(N), ep(N);
        HIP_TRY(hipMemcpy(ev.data(), h->S.env, N * 4, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(ep.data(), h->S.episode, N * 4, hipMemcpyDeviceToHost));
        for (size_t e = 0; e < N; ++e) {
            env[e * 4] = (int32_t)(ev[e] & 0xFF); env[e * 4 + 1] = (int32_t)((ev[e] >> 8) & 3);
            env[e * 4 + 2] = (int32_t)ep[e]; env[e * 4 + 3] = 0;
        }
    }
    return EVG_OK;
} catch (...) { return on_exception(); }

int evg_set_state(evg_handle* h, const int32_t* groups, const int32_t* nodes, const double* health, const int32_t* env) try {
    if (!h) return fail(EVG_ERR_INVALID, "null handle");
    if (!groups || !nodes || !health || !env) return fail(EVG_ERR_INVALID, "set_state needs all four arrays");
    EVG_ON_DEVICE(h);
    HIP_TRY(hipDeviceSynchronize());
    const size_t N = (size_t)h->S.N;
    std::vector<uint32_t> g(24 * N), st(6 * N, 0u), ev(N), ep(N);
    std::vector<uint32_t> nd(6 * N, 0u);
    for (size_t e = 0; e < N; ++e) {
        for (int r = 0; r < 24; ++r) {
            const int32_t* o = groups + (e * 24 + r) * 8;
            const int p = r / 12, k = r % 12, size = kGroupSize[k];
            const double* hp = health + e * 2 * NU + p * NU + 8 * k;
            uint32_t mask = 0;
            for (int s = 0; s < size; ++s) mask |= (hp[s] > 0.0 ? 1u : 0u) << s;
            const int cnt = __builtin_popcount(mask);
            if (o[0] < 1 || o[0] > NN || o[1] < -1 || o[1] > NN || o[1] == 0 || o[2] < 0 || o[2] > 7 || (o[3] && o[4]) || o[7] < 0 || o[7] > 255)
                return fail(EVG_ERR_INVALID, "set_state: env %zu group %d has out-of-domain fields", e, r);
            if (o[6] != cnt || (o[5] != 0) != (cnt == 0))
                return fail(EVG_ERR_INVALID, "set_state: env %zu group %d count/destroyed disagree with health (count %d, alive %d)", e, r, o[6], cnt);
            const uint32_t avg = cnt ? (uint32_t)(int)(host_np_sum(hp, size) / (double)cnt) : 0u;
            const uint32_t mode = o[4] ? MODE_MOVING : (o[3] ? MODE_READY : MODE_IDLE);
            g[(size_t)r * N + e] = (uint32_t)o[0] | ((uint32_t)(o[1] < 0 ? 0 : o[1]) << G_DEST_S) | ((uint32_t)o[2] << G_DIST_S) |
                                   (mode << G_MODE_S) | (mask << G_MASK_S) | (avg << G_AVG_S);
            st[(size_t)(r >> 2) * N + e] |= (uint32_t)o[7] << (8 * (r & 3));
        }
        for (int n = 0; n < NN; ++n) {
            const int cs = nodes[(e * NN + n) * 2], cb = nodes[(e * NN + n) * 2 + 1];
            if (cs < -511 || cs > 511 || cb < -1 || cb > 1) return fail(EVG_ERR_INVALID, "set_state: env %zu node %d out of domain", e, n + 1);
            nd[(size_t)(n >> 1) * N + e] |= (uint32_t)((cs + 512) | ((cb + 1) << 10)) << (16 * (n & 1));
        }
        if (env[e * 4] < 0 || env[e * 4] > 255 || env[e * 4 + 1] < 0 || env[e * 4 + 1] > 3) return fail(EVG_ERR_INVALID, "set_state: env %zu turn/status", e);
        ev[e] = (uint32_t)env[e * 4] | ((uint32_t)env[e * 4 + 1] << 8);
        ep[e] = (uint32_t)env[e * 4 + 2];
    }
    HIP_TRY(hipMemcpy(h->S.grp, g.data(), g.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->S.stamp, st.data(), st.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->S.node, nd.data(), nd.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->S.env, ev.data(), N * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->S.episode, ep.data(), N * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(h->S.health, health, 2 * NU * N * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(h->S.ep_ret, 0, 2 * N * sizeof(float)));
    HIP_TRY(hipDeviceSynchronize());
    return EVG_OK;
} catch (...) { return on_exception(); }

int evg_episode_stats(evg_handle* h, float* returns, int32_t* length, int8_t* winner, int64_t* totals) try {
    if (!h) return fail(EVG_ERR_INVALID, "null handle");
    EVG_ON_DEVICE(h);
    HIP_TRY(hipDeviceSynchronize());
    const size_t N = (size_t)h->S.N;
    { const int frc = check_fault(h); if (frc) return frc; }
    if (returns) HIP_TRY(hipMemcpy(returns, h->S.fin_ret, 2 * N * sizeof(float), hipMemcpyDeviceToHost));
    if (length) HIP_TRY(hipMemcpy(length, h->S.fin_len, N * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (winner) HIP_TRY(hipMemcpy(winner, h->S.fin_win, N, hipMemcpyDeviceToHost));
    if (totals) HIP_TRY(hipMemcpy(totals, h->S.totals, 4 * sizeof(int64_t), hipMemcpyDeviceToHost));
    return EVG_OK;
} catch (...) { return on_exception(); }

// What evg_get_state / evg_set_state do not carry and a RESUMED run needs: the scripted agents' objects, the running episode returns, the results of the last
// finished episodes and the win counters (the reference never serialises its env; this is the build's own checkpoint).
int evg_get_run_state(evg_handle* h, uint32_t* agents, float* running_returns, float* returns, int32_t* length, int8_t* winner, int64_t* totals) try {
    if (!h) return fail(EVG_ERR_INVALID, "null handle");
    EVG_ON_DEVICE(h);
    HIP_TRY(hipDeviceSynchronize());
    { const int frc = check_fault(h); if (frc) return frc; }
    const size_t N = (size_t)h->S.N;
    if (agents) {                                            // device [2][N] x 3 arrays -> host [N][2][3]
        std::vector<uint32_t> a(2 * N), b(2 * N), c(2 * N);
        HIP_TRY(hipMemcpy(a.data(), h->S.agent_cycle, 2 * N * 4, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(b.data(), h->S.agent_swarm, 2 * N * 4, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(c.data(), h->S.agent_dfs, 2 * N * 4, hipMemcpyDeviceToHost));
        for (size_t e = 0; e < N; ++e)
            for (size_t p = 0; p < 2; ++p) {
                uint32_t* o = agents + (e * 2 + p) * 3;
                o[0] = a[p * N + e]; o[1] = b[p * N + e]; o[2] = c[p * N + e];
            }
    }
    if (running_returns) {                                   // device [2][N] -> host [N][2]
        std::vector<float> r(2 * N);
        HIP_TRY(hipMemcpy(r.data(), h->S.ep_ret, 2 * N * sizeof(float), hipMemcpyDeviceToHost));
        for (size_t e = 0; e < N; ++e) { running_returns[2 * e] = r[e]; running_returns[2 * e + 1] = r[N + e]; }
    }
    if (returns) HIP_TRY(hipMemcpy(returns, h->S.fin_ret, 2 * N * sizeof(float), hipMemcpyDeviceToHost));
    if (length) HIP_TRY(hipMemcpy(length, h->S.fin_len, N * sizeof(int32_t), hipMemcpyDeviceToHost));
    if (winner) HIP_TRY(hipMemcpy(winner, h->S.fin_win, N, hipMemcpyDeviceToHost));
    if (totals) HIP_TRY(hipMemcpy(totals, h->S.totals, 4 * sizeof(int64_t), hipMemcpyDeviceToHost));
    return EVG_OK;
} catch (...) { return on_exception(); }

int evg_set_run_state(evg_handle* h, const uint32_t* agents, const float* running_returns, const float* returns, const int32_t* length, const int8_t* winner,
                      const int64_t* totals) try {
    if (!h) return fail(EVG_ERR_INVALID, "null handle");
    EVG_ON_DEVICE(h);
    HIP_TRY(hipDeviceSynchronize());
    const size_t N = (size_t)h->S.N;
    if (winner)
        for (size_t e = 0; e < N; ++e)
            if (winner[e] < EVG_WINNER_NONE || winner[e] > EVG_WINNER_TIE) return fail(EVG_ERR_INVALID, "set_run_state: winner[%zu] = %d", e, (int)winner[e]);
    if (totals && (totals[0] < 0 || totals[1] < 0 || totals[2] < 0 || totals[3] < 0 || totals[1] + totals[2] + totals[3] != totals[0]))
        return fail(EVG_ERR_INVALID, "set_run_state: totals must be {episodes, p0 wins, p1 wins, ties} with episodes = the sum of the other three");
    if (agents) {
        std::vector<uint32_t> a(2 * N), b(2 * N), c(2 * N);
        for (size_t e = 0; e < N; ++e)
            for (size_t p = 0; p < 2; ++p) {
                const uint32_t* o = agents + (e * 2 + p) * 3;
                a[p * N + e] = o[0]; b[p * N + e] = o[1]; c[p * N + e] = o[2];
            }
        HIP_TRY(hipMemcpy(h->S.agent_cycle, a.data(), 2 * N * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(h->S.agent_swarm, b.data(), 2 * N * 4, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(h->S.agent_dfs, c.data(), 2 * N * 4, hipMemcpyHostToDevice));
    }
    if (running_returns) {
        std::vector<float> r(2 * N);
        for (size_t e = 0; e < N; ++e) { r[e] = running_returns[2 * e]; r[N + e] = running_returns[2 * e + 1]; }
        HIP_TRY(hipMemcpy(h->S.ep_ret, r.data(), 2 * N * sizeof(float), hipMemcpyHostToDevice));
    }
    if (returns) HIP_TRY(hipMemcpy(h->S.fin_ret, returns, 2 * N * sizeof(float), hipMemcpyHostToDevice));
    if (length) HIP_TRY(hipMemcpy(h->S.fin_len, length, N * sizeof(int32_t), hipMemcpyHostToDevice));
    if (winner) HIP_TRY(hipMemcpy(h->S.fin_win, winner, N, hipMemcpyHostToDevice));
    if (totals) HIP_TRY(hipMemcpy(h->S.totals, totals, 4 * sizeof(int64_t), hipMemcpyHostToDevice));
    return EVG_OK;
} catch (...) { return on_exception(); }

#ifdef EVG_DIAG
/* Diagnostic libraries only (libevg_diag.so / libevg_stamps.so; declared in no public header).
 *   ablate          bit0 orders, bit1 combat, bit2 movement, bit4 observation write-out, bit5 state store are skipped; bit6: a chunked launch never
 *                   publishes the first chunk of its first set (the fault path: its successor gives up after ~5 s and flags the handle); bit7: chunks are
 *                   published with an agent-scope release; bits 8-23: stagger knobs; bit24: the first chunk of a chunked launch's first set withholds its group
 *                   words but hands on their checksum (the stale-data path: the consumer's checksum over what it loaded differs -> fault bit 3)
 *   lanes_per_wave  0 (default: what the product library launches), 2 (experiment: persistent rollouts of a batch beyond what the device holds run the
 *                   CHUNKED form over the whole batch, i.e. with a working set larger than the Infinity Cache), 64 (the two-lanes-per-env kernel at every batch
 *                   size and in both
 *                   launch forms), 32 (16 envs per wavefront + 32 helper lanes), 4 (the four-lanes-per-env kernel in both launch forms) or 256 (single-turn
 *                   launches as 256-thread workgroups of four independent wavefronts: the round-5 dispatch experiment)
 *   force_ieee_div  != 0: run the step kernel's true-division branch although the table set passed the exact-quotient check */
EVG_API int evg_diag_configure(evg_handle* h, uint32_t ablate, int lanes_per_wave, int force_ieee_div) try {
    if (!h || (lanes_per_wave != 0 && lanes_per_wave != 32 && lanes_per_wave != 64 && lanes_per_wave != 4 && lanes_per_wave != 2 && lanes_per_wave != 256))
        return fail(EVG_ERR_INVALID, "diag: bad argument");
    EVG_ON_DEVICE(h);
    h->ablate = ablate;
    h->lanes = lanes_per_wave;
    drop_graphs(h);                        // the knobs are launch arguments of the cached graphs
    if (force_ieee_div) {
        h->host_tables.fast_div = 0;
        h->host_tables.lds.nib[9] &= ~(1ull << 24);
        HIP_TRY(hipDeviceSynchronize());
        HIP_TRY(hipMemcpy(h->d_tables, &h->host_tables, sizeof(DevTables), hipMemcpyHostToDevice));
    }
    return EVG_OK;
} catch (...) { return on_exception(); }
#endif

#ifdef EVG_STAMPS
/* stamps build only: per-workgroup s_memtime stamps of the last step launch, [blocks][16] */
EVG_API int evg_debug_read_stamps(evg_handle* h, unsigned long long* out) try {
    if (!h || !out) return fail(EVG_ERR_INVALID, "null argument");
    EVG_ON_DEVICE(h);
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, h->stamps, (size_t)((h->S.N + 15) / 16) * 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return EVG_OK;
} catch (...) { return on_exception(); }
#endif

int evg_launch_plan(const evg_handle* h, int turns_per_launch, char* buf, int buflen) try {
    if (!h || !buf || buflen < 1 || turns_per_launch < 1) return fail(EVG_ERR_INVALID, "launch_plan: bad argument");
    // the plan of the default rollout: observations written, orders recorded (the pointers are only tested against NULL)
    StepIO io = make_io(h, nullptr, reinterpret_cast<void*>(16), nullptr, nullptr, nullptr, nullptr, nullptr, 0, 1, 0, 0, reinterpret_cast<int32_t*>(16));
    io.turns = turns_per_launch;
    const LaunchPlan p = plan_step(h->S, io, h->cfg.obs_dtype, h->caps);
    std::string s;
    char tmp[320];
    for (int i = 0; i < p.n; ++i) {
        const LaunchPiece& pc = p.piece[i];
        const int n = pc.env_hi - pc.env_lo, epw = pc.four_lane_wpe ? 16 : 32;
        const char* kname = pc.four_lane_wpe ? "evg_step4_kernel" : "evg_step_kernel";
#ifdef EVG_DIAG
        if (h->lanes == 4) kname = "evg_step4_kernel";
        if (h->lanes == 64 || h->lanes == 32) kname = "evg_step_kernel";
#endif
        const char* plus = i ? " + " : "";
        const int nw = (n + epw - 1) / epw;          // wavefronts (sets of envs) of this piece
        if (h->S.mt_key)
            snprintf(tmp, sizeof(tmp), "%sevg_step_kernel<stock MT19937>[envs %d..%d: %d wavefronts of 32 envs]", plus, pc.env_lo, pc.env_hi, (n + 31) / 32);
        else if (pc.four_lane_wpe)
            snprintf(tmp, sizeof(tmp), "%s%s<four lanes per env, built for %d waves per SIMD>[envs %d..%d: %d wavefronts of 16 envs]", plus, kname,
                     pc.four_lane_wpe, pc.env_lo, pc.env_hi, nw);
        else if (pc.chunk_turns > 0)
            snprintf(tmp, sizeof(tmp),
                     "%s%s<two lanes per env, persistent, chunked>[envs %d..%d: %d sets of 32 envs x %d chunks of %d turns, "
                     "taken from %d per-XCD queues by %d workgroups]",
                     plus, kname, pc.env_lo, pc.env_hi, nw, (turns_per_launch + pc.chunk_turns - 1) / pc.chunk_turns, pc.chunk_turns, h->S.nxcd,
                     nw < h->caps.slots2 ? nw : h->caps.slots2);
        else
            snprintf(tmp, sizeof(tmp), "%s%s<two lanes per env, %s>[envs %d..%d: %d wavefronts of 32 envs]", plus, kname,
                     turns_per_launch > 1 ? "persistent" : "single-turn", pc.env_lo, pc.env_hi, nw);
        s += tmp;
    }
    snprintf(tmp, sizeof(tmp),
             " | device: %d CUs, resident wavefronts two-lane %d, four-lane %d / %d, %d XCDs, Infinity-Cache budget of a launch that cycles through its envs "
             "%lld MiB (%lld B per env)",
             h->caps.cus, h->caps.slots2, h->caps.slots4_w2, h->caps.slots4_w3, h->S.nxcd, (long long)(h->caps.cache_bytes >> 20),
             rollout_bytes_per_env(io, h->cfg.obs_dtype));
    s += tmp;
    snprintf(buf, (size_t)buflen, "%s", s.c_str());
    return p.n;
} catch (...) { return on_exception(); }

int evg_pack_episode_results(evg_handle* h, float* out, void* stream) try {
    if (!h || !out) return fail(EVG_ERR_INVALID, "null argument");
    EVG_NEED_ALIGNED16(out);
    EVG_ON_DEVICE(h);
    const int rc = launch_pack_results(h->S, out, nullptr, stream);
    if (rc) return fail(EVG_ERR_HIP, "pack launch failed: %s", hipGetErrorString((hipError_t)rc));
    return EVG_OK;
} catch (...) { return on_exception(); }

int evg_pack_episode_results_counted(evg_handle* h, float* out, int64_t* counts_out, void* stream) try {
    if (!h || !out || !counts_out) return fail(EVG_ERR_INVALID, "null argument");
    EVG_NEED_ALIGNED16(out);
    if ((reinterpret_cast<uintptr_t>(counts_out) & 7u) != 0) return fail(EVG_ERR_INVALID, "counts_out must be 8-byte aligned");
    EVG_ON_DEVICE(h);
    const int rc = launch_pack_results(h->S, out, reinterpret_cast<long long*>(counts_out), stream);
    if (rc) return fail(EVG_ERR_HIP, "pack launch failed: %s", hipGetErrorString((hipError_t)rc));
    return EVG_OK;
} catch (...) { return on_exception(); }

// ---------------------------------------------------------------------------------------------
// The path's one exchange between GPUs for callers that have no torch.distributed (SURVEY 8b `evg_gather_returns`, 8e): RCCL itself, opened at run time.
// In a PyTorch process dlopen("librccl.so.1") resolves to the instance torch has already loaded (same SONAME); a plain C client gets ROCm's.
// ---------------------------------------------------------------------------------------------
namespace {
struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string why;
};
RcclApi g_rccl;
std::once_flag g_rccl_once;

const RcclApi* rccl() {
    std::call_once(g_rccl_once, [] {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            g_rccl.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (g_rccl.lib) break;
        }
        if (!g_rccl.lib) {
            const char* e = dlerror();                 // ONE call: dlerror() hands the message out once and clears it
            g_rccl.why = std::string("librccl.so.1 not found (") + (e ? e : "?") + ")";
            return;
        }
        auto sym = [](const char* n) { return dlsym(g_rccl.lib, n); };
        g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(sym("ncclGetUniqueId"));
        g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(sym("ncclCommInitRank"));
        g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(sym("ncclCommDestroy"));
        g_rccl.GroupStart = reinterpret_cast<decltype(g_rccl.GroupStart)>(sym("ncclGroupStart"));
        g_rccl.GroupEnd = reinterpret_cast<decltype(g_rccl.GroupEnd)>(sym("ncclGroupEnd"));
        g_rccl.Send = reinterpret_cast<decltype(g_rccl.Send)>(sym("ncclSend"));
        g_rccl.Recv = reinterpret_cast<decltype(g_rccl.Recv)>(sym("ncclRecv"));
        g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(sym("ncclGetErrorString"));
        if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.GroupStart || !g_rccl.GroupEnd || !g_rccl.Send || !g_rccl.Recv ||
            !g_rccl.GetErrorString) {
            g_rccl.why = "librccl lacks an entry point of the send / receive API";
            g_rccl.lib = nullptr;
        }
    });
    return g_rccl.lib ? &g_rccl : nullptr;
}

#define RCCL_TRY(R, expr)                                                                                        \
    do {                                                                                                         \
        const ncclResult_t r_ = (expr);                                                                          \
        if (r_ != ncclSuccess) return fail(EVG_ERR_COMM, "%s failed: %s", #expr, (R)->GetErrorString(r_));       \
    } while (0)
}  // namespace

int evg_comm_unique_id(void* id_out) try {
    if (!id_out) return fail(EVG_ERR_INVALID, "null argument");
    const RcclApi* R = rccl();
    if (!R) return fail(EVG_ERR_COMM, "RCCL is not available: %s", g_rccl.why.c_str());
    static_assert(sizeof(ncclUniqueId) == EVG_COMM_ID_BYTES, "evg.h: EVG_COMM_ID_BYTES");
    ncclUniqueId id;
    RCCL_TRY(R, R->GetUniqueId(&id));
    memcpy(id_out, &id, sizeof(id));
    return EVG_OK;
} catch (...) { return on_exception(); }

int evg_comm_init(evg_handle* h, const void* id, int world, int rank, const int32_t* counts) try {
    if (!h || !id || !counts || world < 1 || rank < 0 || rank >= world) return fail(EVG_ERR_INVALID, "comm_init: bad argument");
    if (h->comm) return fail(EVG_ERR_INVALID, "comm_init: the handle already has a communicator (evg_comm_destroy first)");
    long long total = 0;
    for (int r = 0; r < world; ++r) {
        if (counts[r] < 1) return fail(EVG_ERR_INVALID, "comm_init: counts[%d] = %d", r, counts[r]);
        total += counts[r];
    }
    if (counts[rank] != h->S.N) return fail(EVG_ERR_INVALID, "comm_init: counts[%d] = %d, but this handle has %d envs", rank, counts[rank], h->S.N);
    if (total > 0x7FFFFFFFll) return fail(EVG_ERR_INVALID, "comm_init: more than 2^31 envs in all");
    const RcclApi* R = rccl();
    if (!R) return fail(EVG_ERR_COMM, "RCCL is not available: %s", g_rccl.why.c_str());
    EVG_ON_DEVICE(h);
    if (!h->comm_send) {
        const int rc = dev_alloc(h, &h->comm_send, (size_t)h->S.N * 4);
        if (rc) return rc;
    }
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    ncclComm_t c = nullptr;
    RCCL_TRY(R, R->CommInitRank(&c, world, uid, rank));          // collective: returns when every rank has called
    h->comm = c; h->comm_world = world; h->comm_rank = rank;
    h->comm_counts.assign(counts, counts + world);
    return EVG_OK;
} catch (...) { return on_exception(); }

int evg_gather_returns(evg_handle* h, int root, float* recv_out, void* stream) try {
    if (!h) return fail(EVG_ERR_INVALID, "null handle");
    if (!h->comm) return fail(EVG_ERR_INVALID, "gather_returns: no communicator (evg_comm_init)");
    if (root < 0 || root >= h->comm_world) return fail(EVG_ERR_INVALID, "gather_returns: root %d of %d ranks", root, h->comm_world);
    if ((h->comm_rank == root) != (recv_out != nullptr))
        return fail(EVG_ERR_INVALID, "gather_returns: recv_out is required on the root rank and must be NULL elsewhere");
    EVG_NEED_ALIGNED16(recv_out);
    const RcclApi* R = rccl();
    if (!R) return fail(EVG_ERR_COMM, "RCCL is not available: %s", g_rccl.why.c_str());
    EVG_ON_DEVICE(h);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int prc = launch_pack_results(h->S, h->comm_send, nullptr, stream);      // poisoned rows if the handle's fault word is set
    if (prc) return fail(EVG_ERR_HIP, "pack launch failed: %s", hipGetErrorString((hipError_t)prc));
    // one grouped operation on the caller's stream: every rank sends its rows to the root, the root receives every rank's rows at its offset (global env order)
    RCCL_TRY(R, R->GroupStart());
    ncclResult_t r1 = R->Send(h->comm_send, (size_t)h->S.N * 4, ncclFloat, root, h->comm, s);
    if (r1 == ncclSuccess && recv_out) {
        size_t at = 0;
        for (int r = 0; r < h->comm_world && r1 == ncclSuccess; ++r) {
            r1 = R->Recv(recv_out + at * 4, (size_t)h->comm_counts[r] * 4, ncclFloat, r, h->comm, s);
            at += (size_t)h->comm_counts[r];
        }
    }
    const ncclResult_t r2 = R->GroupEnd();
    if (r1 != ncclSuccess) return fail(EVG_ERR_COMM, "ncclSend / ncclRecv failed: %s", R->GetErrorString(r1));
    if (r2 != ncclSuccess) return fail(EVG_ERR_COMM, "ncclGroupEnd failed: %s", R->GetErrorString(r2));
    return EVG_OK;
} catch (...) { return on_exception(); }

int evg_comm_destroy(evg_handle* h) try {
    if (!h) return fail(EVG_ERR_INVALID, "null handle");
    if (!h->comm) return EVG_OK;
    const RcclApi* R = rccl();
    EVG_ON_DEVICE(h);
    (void)hipDeviceSynchronize();
    if (R) (void)R->CommDestroy(h->comm);
    h->comm = nullptr; h->comm_world = 0; h->comm_rank = -1;
    h->comm_counts.clear();
    return EVG_OK;
} catch (...) { return on_exception(); }

int evg_check_fault(evg_handle* h, uint32_t* fault_out) try {
    if (!h) return fail(EVG_ERR_INVALID, "null handle");
    EVG_ON_DEVICE(h);
    HIP_TRY(hipDeviceSynchronize());
    return check_fault(h, fault_out);
} catch (...) { return on_exception(); }

int evg_episode_stats_device(evg_handle* h, float** returns, int32_t** length, int8_t** winner) try {
    if (!h) return fail(EVG_ERR_INVALID, "null handle");
    // no synchronisation here (the pointers are handed out once and read on the caller's streams): a fault already seen is reported; one of work still in
    // flight shows in evg_pack_episode_results' poisoned rows or in evg_check_fault
    { const int frc = check_fault(h); if (frc) return frc; }
    if (returns) *returns = h->S.fin_ret;
    if (length) *length = h->S.fin_len;
    if (winner) *winner = h->S.fin_win;
    return EVG_OK;
} catch (...) { return on_exception(); }

}  // extern "C"
