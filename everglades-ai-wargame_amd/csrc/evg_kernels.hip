// evg_kernels.hip -- hand-written HIP kernels for gfx950 (MI355X / CDNA4).
//
// One fused kernel per env-step replaces EvergladesEnv.step -> EvergladesGame.game_turn ->
// board_state/player_state of the reference (everglades_env.py:32-73, server.py:211-501).
//
// Mapping: lane = (env slot, player); one wavefront = 32 envs per workgroup.
// The game rules are short, branchy integer programs with almost no parallelism inside one env
// (7 orders per player, ~1 contested node per turn), so envs are spread over LANES: one instruction
// serves 32 envs, where a wave-per-env mapping would leave most lanes idle.  Two lanes per env halve
// every per-player loop and give 2 waves per SIMD at 65 536 envs (latency hiding); the pair exchanges
// values with one DPP move.  Everything a lane indexes dynamically (its 12 group words, the node
// words, per-node accumulators) lives in LDS as [index][lane] columns: lane-private and
// bank-conflict-free by construction (bank = lane mod 32).  Combat is rebalanced over the whole wave:
// a prefix scan over lanes builds one work item per fighting group, damage accumulates in a shared
// LDS pool with integer LDS atomics (order-free, deterministic).  The SoA state arrays are
// env-fastest (coalesced); the wave's observation block (32 x 840 B, contiguous in the output) is
// assembled in LDS in output order and streamed out with 16-byte-per-lane coalesced stores.
// Uniform map/unit tables sit in SGPRs as nibble-packed words, per-lane-indexed ones in LDS.
// No MFMA: there is no dense contraction anywhere on this path.
#include <hip/hip_runtime.h>
#include <type_traits>
#include "evg_device.h"
#include "evg_rng.h"
#include "evg_mt.h"

namespace evg {

// Diagnostics are compiled only into the separate libraries libevg_diag.so (-DEVG_DIAG: phase ablation, the 16-envs-per-wave
// variant, a switch that forces the IEEE-division branch) and libevg_stamps.so (-DEVG_DIAG -DEVG_STAMPS: lane 0 of every
// wave stores s_memtime at phase boundaries into a debug buffer of its own).  The product library libevg.so contains
// none of it: no ablation branch, no stamp, no environment variable.
#ifdef EVG_DIAG
#define ABLATED(bit) ((io.ablate & (bit)) != 0u)
#else
#define ABLATED(bit) false
#endif
#ifdef EVG_STAMPS
#define STAMP(i)                                                                        \
    do {                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                              \
        const unsigned long long t_ = clock64();                                        \
        __builtin_amdgcn_sched_barrier(0);                                              \
        if (lane == 0 && io.stamps) io.stamps[(size_t)blockIdx.x * 16 + (i)] = t_;      \
    } while (0)
// whole-launch stamps of the wave: s_memrealtime (constant 100 MHz) at its start and end plus where it ran (HW_ID, XCC_ID)
#define STAMP_WAVE_BEGIN() const unsigned long long wave_t0_ = __builtin_amdgcn_s_memrealtime()
#define STAMP_WAVE_END()                                                                                                   \
    do {                                                                                                                   \
        const unsigned long long t1_ = __builtin_amdgcn_s_memrealtime();                                                   \
        if (threadIdx.x == 0 && A->io_.stamps) {                                                                            \
            A->io_.stamps[(size_t)blockIdx.x * 16 + 14] = (wave_t0_ & 0xFFFFFFFFull) | ((unsigned long long)__builtin_amdgcn_s_getreg(63492) << 32); \
            A->io_.stamps[(size_t)blockIdx.x * 16 + 15] = (t1_ & 0xFFFFFFFFull) | ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32);      \
        }                                                                                                                  \
    } while (0)
#else
#define STAMP(i)
#define STAMP_WAVE_BEGIN()
#define STAMP_WAVE_END()
#endif

// Phase markers (stamps build only).
#define PHASE(i) STAMP(i)

// LPW = lanes of the wavefront that own an env side (lane = 2 * env_slot + player): 64 (32 envs per wave) or 32
// (16 envs per wave; lanes 32..63 are helpers that only join the wave-balanced phases: combat items, write-out).
template <int LPW>
struct CombatLds {
    static constexpr int DP_CAP = LPW == 64 ? 1536 : 768;   // words of the shared damage pool (worst case of LPW/2 lanes: 33 each)
    uint32_t SNAP[12][LPW];              // pre-combat snapshot of the lane's own group k:
                                         //   bit31 fights | list-order prefix of alive units << 16 | node << 12 | alive mask
    uint32_t FS[12][LPW];                // the lane's own side at node n: alive fighting units << 16 | word offset of its damage bytes in DP
    uint32_t TURN[LPW], EPI[LPW];        // per-env scalars for lanes that work on another env's item
    uint16_t W[LPW * 12];                // work list of fighting groups: owner lane | gid << 6
    uint32_t DP[DP_CAP];                 // damage pool: one byte per targeted unit index, filled with LDS atomics
};

template <int LPW>
struct __align__(16) StepLds {
    uint32_t G[12][LPW];                 // group words, lane-private columns (lane = env slot, player)
    uint32_t NW[12][LPW / 2];            // node words by node ID, one column per env
    union {                              // phases are disjoint in time (one wavefront per workgroup)
        CombatLds<LPW> c;
        uint32_t A[12][LPW];             // per (own side, node): capture points | units listed << 16
        int16_t  O[LPW * OBS];           // observations of the wave's envs, already in output order [env][player][105]
    } u;
    LdsTables tab;                       // the per-lane-indexed constant tables (evg_device.h), copied from DevTables::lds
};

// Phase boundary inside the step kernel.  A workgroup is ONE wavefront, and the LDS executes a wavefront's instructions in
// issue order (so do the vector-memory units, per address), so a boundary needs neither s_barrier nor a wait for outstanding
// global loads/stores (what __syncthreads() would add: s_waitcnt vmcnt(0) stalls every phase behind the turn's
// observation and health stores): it only has to keep the COMPILER from moving memory accesses across it.
#define WAVE_SYNC()                                            \
    do {                                                       \
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); \
        asm volatile("" ::: "memory");                         \
        __builtin_amdgcn_wave_barrier();                       \
    } while (0)

// 12-input sorting network (tools/gen_sort12.py: 42 compare-exchanges, verified with the 0-1 principle)
#define EVG_SORT12_CES(CE) \
    CE(0, 1) CE(2, 3) CE(4, 5) CE(6, 7) CE(8, 9) CE(10, 11) CE(0, 2) CE(1, 3) \
    CE(4, 6) CE(5, 7) CE(8, 10) CE(9, 11) CE(1, 2) CE(5, 6) CE(9, 10) CE(0, 4) \
    CE(1, 5) CE(2, 6) CE(3, 7) CE(2, 4) CE(3, 5) CE(1, 2) CE(3, 4) CE(5, 6) \
    CE(9, 10) CE(0, 8) CE(1, 9) CE(2, 10) CE(3, 11) CE(4, 8) CE(5, 9) CE(6, 10) \
    CE(7, 11) CE(2, 4) CE(3, 5) CE(6, 8) CE(7, 9) CE(1, 2) CE(3, 4) CE(5, 6) \
    CE(7, 8) CE(9, 10)

// numpy's pairwise summation of a short contiguous float64 vector (np.sum at server.py:481):
// ((a0+a1)+(a2+a3))+((a4+a5)+(a6+a7)), then the tail sequentially.
__device__ __forceinline__ double np_sum8(const double* h) {
    return ((h[0] + h[1]) + (h[2] + h[3])) + ((h[4] + h[5]) + (h[6] + h[7]));
}

template <typename OT>
__device__ __forceinline__ void store_obs_vec(OT* dst, const int (&v)[16 / sizeof(OT)]);
template <>
__device__ __forceinline__ void store_obs_vec<float>(float* dst, const int (&v)[4]) {
    *reinterpret_cast<float4*>(dst) = make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}
template <>
__device__ __forceinline__ void store_obs_vec<double>(double* dst, const int (&v)[2]) {
    *reinterpret_cast<double2*>(dst) = make_double2((double)v[0], (double)v[1]);
}
template <>
__device__ __forceinline__ void store_obs_vec<int16_t>(int16_t* dst, const int (&v)[8]) {
    uint4 o;
    o.x = (uint32_t)(v[0] & 0xFFFF) | ((uint32_t)v[1] << 16);
    o.y = (uint32_t)(v[2] & 0xFFFF) | ((uint32_t)v[3] << 16);
    o.z = (uint32_t)(v[4] & 0xFFFF) | ((uint32_t)v[5] << 16);
    o.w = (uint32_t)(v[6] & 0xFFFF) | ((uint32_t)v[7] << 16);
    *reinterpret_cast<uint4*>(dst) = o;
}

// random_actions stand-in for one (env, player): 7 distinct groups of 12 and 7 distinct nodes of 1..11
// (agents/State_Machine/random_actions.py:38-46), partial Fisher-Yates on nibble-packed permutations.
// Same contract as oracle/rng_spec.py random_action_rows.
__device__ __forceinline__ void gen_random_rows(uint32_t seed_lo, uint32_t seed_hi, uint32_t env_id, uint32_t episode, int turn, int p, int2 (&rows)[NA]) {
    // halves 0..6 of block 0 pick the groups, halves 0..6 of block 1 the nodes (oracle/rng_spec.py)
    const uint4 xg = rng_block(seed_lo, seed_hi, env_id, episode, RNG_ACTION, 0u, turn, 0, p, 0);
    const uint4 xn = rng_block(seed_lo, seed_hi, env_id, episode, RNG_ACTION, 1u, turn, 0, p, 0);
    const uint32_t wg[4] = {xg.x, xg.y, xg.z, xg.w}, wn[4] = {xn.x, xn.y, xn.z, xn.w};
    uint64_t gp = 0xBA9876543210ull;      // nibble i = i
    uint64_t np_ = 0xBA987654321ull;      // nibble i = i + 1
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int j = i + (int)(__umul24(rng_half(wg, i), (uint32_t)(12 - i)) >> 16);
        const uint64_t x = ((gp >> (4 * i)) ^ (gp >> (4 * j))) & 15ull;
        gp ^= (x << (4 * i)) ^ (x << (4 * j));
    }
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int j = i + (int)(__umul24(rng_half(wn, i), (uint32_t)(11 - i)) >> 16);
        const uint64_t x = ((np_ >> (4 * i)) ^ (np_ >> (4 * j))) & 15ull;
        np_ ^= (x << (4 * i)) ^ (x << (4 * j));
    }
#pragma unroll
    for (int i = 0; i < NA; ++i) rows[i] = make_int2((int)((gp >> (4 * i)) & 15ull), (int)((np_ >> (4 * i)) & 15ull));
}

// zero a [12][LPW] table of per-lane columns.  With all 64 lanes owning a column (LPW == 64) the table is 3 072 contiguous bytes: three
// 16-byte-per-lane stores by the whole wave instead of twelve 4-byte ones per lane (an LDS store costs a SIMD ~16 cycles whatever its width)
template <int LPW>
__device__ __forceinline__ void zero_columns12(uint32_t (*a)[LPW], int lane, bool envlane) {
    if constexpr (LPW == WG) {
        uint4* p = reinterpret_cast<uint4*>(&a[0][0]);
#pragma unroll
        for (int j = 0; j < 3; ++j) p[lane + WG * j] = make_uint4(0u, 0u, 0u, 0u);
    } else {
        if (envlane) {
#pragma unroll
            for (int n = 0; n < 12; ++n) a[n][lane] = 0;
        }
    }
}

// value held by the other player's lane of the same env (lane ^ 1): one DPP quad_perm [1,0,3,2] move, no LDS round trip
__device__ __forceinline__ int xchg1(int v) { return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true); }

// The handle's fault word (DevState::fault, sticky) and its host-mapped mirror: the exact bits by a device atomic, "something happened" by a plain
// system-scope store the host sees after its next synchronisation without a device copy.
__device__ __forceinline__ void raise_fault(uint32_t* fault, uint32_t* seen, uint32_t bits) {
    atomicOr(fault, bits);
    __hip_atomic_store(seen, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---------------------------------------------------------------------------------------------
// fused env-step: lane = (env slot, player); LPW / 2 envs per wavefront (LPW = 64 is the default variant)
// ---------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------
// scripted opponents (BASELINE config 5): all bots of agents/State_Machine/, one agent object per
// (env, player) that lives across episodes like the reference's (evaluate.py:85-93).  The logic is written against a
// small "view" of what the bots read from their observation (turn, own group locations in own numbering, own moving
// flags, control state and opposing units of a board slot), so that the same code serves the standalone kernel (view =
// the observation tensor) and the fused rollout (view = the on-chip state at the start of the turn, which is what the
// observation of the previous turn was built from).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void cycle_advance(int& group_num, int& node_num) {
    group_num = (group_num + 1) % NG;
    if (group_num == 0) node_num = node_num % NN + 1;
}

// the routing tables the bots index: in the step kernel they come from the LDS copy (a load from the global table would be a
// per-lane vector load -- the kernel stores to global memory, so the compiler cannot keep it scalar -- and waiting for it
// means vmcnt(0): for the previous turn's observation stores as well)
struct AgentTabs {
    uint64_t maxnbr, tar1, tar11;       // DevTables::maxnbr_nib, tar_to_1, tar_to_11
    const DevTables* T;                 // dfs_attack's order sequence stays in global memory (that bot only)
};

template <class View>
__device__ __forceinline__ void agent_rows(int policy, const View& v, const AgentTabs& tabs, uint32_t seed_lo, uint32_t seed_hi, uint32_t env_id,
                                           uint32_t episode, int player, bool commit, bool active, uint32_t* p_cycle, uint32_t* p_swarm, uint32_t* p_dfs, int2 (&rows)[NA]) {
    // `commit` is false for the padding lanes of a partial last workgroup (they compute like everyone else but must not
    // advance the agent state of the env their indices are clamped to) and for a finished, not yet reset game: the harness
    // has left that game's loop (evaluate.py:147-152), its agents are not consulted -- zero rows, state untouched
    const int turn = v.turn();
#pragma unroll
    for (int i = 0; i < NA; ++i) rows[i] = make_int2(0, 0);                   // np.zeros(shape)
    if (!active) return;
    // cycling state shared by most bots: first_turn << 8 | group_num << 4 | node_num | strat_index << 9 | agentNumber==2 << 13
    const uint32_t cst = *p_cycle;
    int first = (int)((cst >> 8) & 1u), group_num = (int)((cst >> 4) & 15u), node_num = (int)(cst & 15u);
    int strat = (int)((cst >> 9) & 15u), agent2 = (int)((cst >> 13) & 1u);
    bool cyc_dirty = false;

    if (policy == EVG_POLICY_RANDOM || policy == EVG_POLICY_RANDOM_DELAY) {
        // random_actions.py:38-46, random_actions_2.py; random_actions_delay.py acts only when random.random() > 0.68
        bool go = true;
        if (policy == EVG_POLICY_RANDOM_DELAY) {
            const uint4 x = rng_block(seed_lo, seed_hi, env_id, episode, RNG_DELAY, 0u, turn, 0, player, 0);
            go = (double)x.x / 4294967296.0 > 0.68;
        }
        if (go) gen_random_rows(seed_lo, seed_hi, env_id, episode, turn, player, rows);
    } else if (policy == EVG_POLICY_CYCLE_RUSH_25 || policy == EVG_POLICY_CYCLE_RUSH_50 || policy == EVG_POLICY_BASE_RUSH_V1 ||
               policy == EVG_POLICY_ALL_CYCLE) {
        // cycle_rush_turn25.py:62-115 (gate 25 / 50), base_rush_v1.py:62-96 (row i only while group i is not at node 11),
        // all_cycle.py (always)
        const int gate = policy == EVG_POLICY_CYCLE_RUSH_25 ? 25 : 50;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int loc_i = v.loc(i);
            bool issue = !first;
            if (policy == EVG_POLICY_CYCLE_RUSH_25 || policy == EVG_POLICY_CYCLE_RUSH_50) issue = issue && ((loc_i != 11 && turn > gate) || turn < gate);
            if (policy == EVG_POLICY_BASE_RUSH_V1) issue = issue && loc_i != 11;
            if (issue) { rows[i] = make_int2(group_num, node_num); cycle_advance(group_num, node_num); }
        }
        first = 0; cyc_dirty = true;
    } else if (policy == EVG_POLICY_BULL_RUSH) {                           // bull_rush.py: all groups to 2, 2, 5, 5, 8, 8, 11, 11, ...
        if (!first) {
            if (strat == 8) strat = 0;
            const int node = (int)((0xB852u >> (4 * (strat >> 1))) & 15u);   // node_strat = [2, 5, 8, 11]
#pragma unroll
            for (int i = 0; i < NA; ++i) { rows[i] = make_int2(group_num, node); group_num = (group_num + 1) % NG; }
            strat += 1;
        }
        first = 0; cyc_dirty = true;
    } else if (policy >= EVG_POLICY_CYCLE_TARGET_NODE && policy <= EVG_POLICY_CYCLE_TARGET_NODE11P2) {
        // cycle_target_node.py (target 11, level 75), ..._node1.py (1, 75), ..._node11.py (11, 500), ..._node11P2.py (11, +-500)
        const int tar = policy == EVG_POLICY_CYCLE_TARGET_NODE1 ? 1 : 11, level = policy >= EVG_POLICY_CYCLE_TARGET_NODE11 ? 500 : 75;
        if (first) {
            if (policy == EVG_POLICY_CYCLE_TARGET_NODE11P2 && v.opp_units_slot(11) > 0) agent2 = 1;    // obs[44]
        } else {
            const int ctl = v.ctrl_slot(tar);                                                          // obs[tarNode * 4 - 1]
            const bool controlled = (policy == EVG_POLICY_CYCLE_TARGET_NODE11P2 && agent2) ? ctl <= -level : ctl >= level;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                if (controlled) {
                    rows[i] = make_int2(group_num, node_num);
                    cycle_advance(group_num, node_num);
                } else {
                    const int cur = v.loc(group_num);
                    const int nx = (int)(((tar == 1 ? tabs.tar1 : tabs.tar11) >> (4 * cur)) & 15ull);   // 15 encodes the bots' -1
                    rows[i] = make_int2(group_num, nx == 15 ? -1 : nx);
                    group_num = (group_num + 1) % NG;
                }
            }
        }
        first = 0; cyc_dirty = true;
    } else if (policy == EVG_POLICY_DFS_ATTACK) {
        // dfs_attack.py ignores the observation: its orders are an eventually periodic sequence of the call count,
        // tabulated on the host at evg_create (including the rows that persist in its mutable default argument)
        const uint32_t c = *p_dfs;
        const uint32_t idx = c < (uint32_t)tabs.T->dfs_mu ? c : (uint32_t)tabs.T->dfs_mu + (c - (uint32_t)tabs.T->dfs_mu) % (uint32_t)tabs.T->dfs_lambda;
        const uint64_t r = tabs.T->dfs_rows[idx];
#pragma unroll
        for (int i = 0; i < NA; ++i) rows[i] = make_int2((int)((r >> (8 * i)) & 15ull), (int)((r >> (8 * i + 4)) & 15ull));
        if (commit) *p_dfs = c + 1u;
    } else if (policy == EVG_POLICY_SAME_COMMANDS) {                        // same_commands.py / same_commands_2.py
#pragma unroll
        for (int i = 0; i < NA; ++i) rows[i] = make_int2(i + 1, i + 1);
    } else if (policy == EVG_POLICY_SWARM) {
        uint32_t lst = *p_swarm;                                           // attack list, 8 nibbles
        const uint4 x0 = rng_block(seed_lo, seed_hi, env_id, episode, RNG_SWARM, 0u, turn, 0, player, 0);
        const uint4 x1 = rng_block(seed_lo, seed_hi, env_id, episode, RNG_SWARM, 1u, turn, 0, player, 0);
        const uint32_t w[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
#pragma unroll
        for (int k = 0; k < 7; ++k) {                                      // i = 7 .. 1
            const int i = 7 - k;
            const int j = (int)__umulhi(w[k], (uint32_t)(i + 1));
            const uint32_t x = ((lst >> (4 * i)) ^ (lst >> (4 * j))) & 15u;
            lst ^= (x << (4 * i)) ^ (x << (4 * j));
        }
        if (commit) *p_swarm = lst;
#pragma unroll
        for (int i = 0; i < NA; ++i) rows[i] = make_int2(0, 1);            // np.tile([0, 1], (7, 1))
        const uint64_t mx = tabs.maxnbr;
        int n = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int g = (int)((lst >> (4 * k)) & 15u);
            const bool idle = v.moving(g) == 0;
            const int pos = v.loc(g);
            const int2 r = make_int2(g, (int)((mx >> (4 * pos)) & 15u));   // max(NODE_CONNECTIONS[pos]), :97
#pragma unroll
            for (int i = 0; i < NA; ++i) rows[i] = (idle && n == i) ? r : rows[i];
            n += (idle && n < NA) ? 1 : 0;
        }
    }                                                                      // EVG_POLICY_NO_ACTION: zeros
    if (cyc_dirty && commit)
        *p_cycle = (uint32_t)node_num | ((uint32_t)group_num << 4) | ((uint32_t)first << 8) | ((uint32_t)strat << 9) | ((uint32_t)agent2 << 13);
}

// what a bot reads from its observation row (everglades_env.py:158-171 layout)
template <typename OT>
struct ObsView {
    const OT* o;
    __device__ int turn() const { return (int)o[0]; }
    __device__ int loc(int k) const { return (int)o[45 + 5 * k]; }
    __device__ int moving(int k) const { return (int)o[48 + 5 * k]; }
    __device__ int ctrl_slot(int slot) const { return (int)o[4 * slot - 1]; }
    __device__ int opp_units_slot(int slot) const { return (int)o[4 * slot]; }
};

// the same quantities taken from the on-chip state at the start of a turn (fused rollout)
template <int LPW>
struct ChipView {
    const StepLds<LPW>* L;
    int col, E, P, turn_;
    uint64_t p1nib;
    __device__ int node_of_slot(int slot) const { return P ? (int)((p1nib >> (4 * slot)) & 15u) : slot; }
    __device__ int turn() const { return turn_; }
    __device__ int loc(int k) const { const uint32_t l = L->G[k][col] & G_LOC_M; return P ? (int)((p1nib >> (4 * l)) & 15u) : (int)l; }
    __device__ int moving(int k) const { return ((L->G[k][col] & G_MODE_M) >> G_MODE_S) == MODE_MOVING ? 1 : 0; }
    __device__ int ctrl_slot(int slot) const { return (int)(L->NW[node_of_slot(slot)][E] & 0x3FFu) - 512; }
    __device__ int opp_units_slot(int slot) const {                        // units of every non-destroyed opposing group listed at the node
        const uint32_t n = (uint32_t)node_of_slot(slot);
        int u = 0;
#pragma unroll
        for (int k = 0; k < 12; ++k) { const uint32_t w = L->G[k][col ^ 1]; u += (w & G_LOC_M) == n ? __popc(w & G_MASK_M) : 0; }
        return u;
    }
};

// The kernel reads its arguments through the kernarg segment pointer instead of by-value parameters: in the multi-turn
// instantiation that pointer is made opaque once per turn, so argument fields and table entries are (re)loaded next to
// their uses by cheap scalar loads instead of staying live across the whole loop (which cost 60+ VGPRs in SGPR spills).
struct StepArgs { DevState s_; StepIO io_; };
typedef const StepArgs __attribute__((address_space(4))) * step_args_ptr;
#define S (A->s_)
#define io (A->io_)

// SEAT (evg_step_vs_policy, evg_observe_seat): the turn of the reference's training / evaluation loops -- a caller on seat io.seat, an on-device bot on the other
// (evaluate.py:143-152) -- as an instantiation of the single-turn form: the caller's lane takes its 7 rows from the caller's tensor, the other lane evaluates its
// bot from the on-chip state (the gen_actions == 2 machinery), and only the caller's lane builds an observation row: the wave's image is [32][105] and the
// write-out half as long.
template <typename OT, int LPW, bool MULTI, bool MT = false, bool CHUNKED = false, bool SEAT = false>
__global__ void __launch_bounds__(WG) __attribute__((amdgpu_waves_per_eu(2, 2))) evg_step_kernel(StepArgs) {
    static_assert(!MT || (!MULTI && LPW == WG), "the stock-entropy mode exists in the single-turn, 32-envs-per-wave form only");
    static_assert(!CHUNKED || (MULTI && LPW == WG && !MT), "the chunked form is an instantiation of the persistent two-lane kernel");
    static_assert(!SEAT || (!MULTI && !MT && LPW == WG), "the one-seat form is an instantiation of the single-turn two-lane kernel");
    step_args_ptr A = (step_args_ptr)__builtin_amdgcn_kernarg_segment_ptr();
    constexpr int EPW = LPW / 2;                        // envs per wavefront
    constexpr int DP_CAP = CombatLds<LPW>::DP_CAP;
    __shared__ StepLds<LPW> L;
    // 8 workgroups per CU (2 waves per SIMD) keep a whole 65 536-env batch resident: 160 KiB / 8 = 20 480 B each
    static_assert(sizeof(StepLds<LPW>) <= 20480, "step kernel LDS exceeds the 8-workgroups-per-CU budget");
    const int lane = threadIdx.x;
    const bool envlane = LPW == WG || lane < LPW;       // owns an env side; helper lanes only join the balanced phases
    const int E = envlane ? lane >> 1 : 0, P = lane & 1;
    // Which envs, which turns.  Plain launch: workgroup b plays all `turns` turns of envs env_lo + 32 b ...
    // CHUNKED launch (io.nsets > 0: a persistent rollout of more envs than the device holds at once, plan_step): the grid is as many
    // workgroups as the device holds; the rollout is cut into UNITS = (set of 32 envs) x (chunk of chunk_turns consecutive turns), and
    // every workgroup takes units from a queue until it is empty, handing a set on to whoever takes its next chunk through HBM
    // (DevState::progress).  Every wave slot holds useful work until the queue runs dry, where one launch of ceil(N / 32) whole-rollout
    // workgroups left its last, partial round running alone at low occupancy for a whole launch (98 304 envs: 30.9 us per turn).
    // Sets are OWNED BY AN XCD (set s belongs to XCD s mod nxcd; one queue per XCD, chunk-major; a workgroup serves the queue of the
    // XCD it runs on, read from XCC_ID): the hand-over then stays inside one L2 and needs no L2 write-back / invalidate (an agent-scope
    // release per chunk made this form 57 % SLOWER than the plain launch: 2 048 waves x buffer_wbl2 keep every L2 walking), only the
    // store drain of the producer and the L1 invalidate of the consumer.  Nothing depends on dispatch order or on how the dispatcher
    // places workgroups: a unit's predecessor was taken from the same queue earlier, by a workgroup that is running and waits for
    // nothing taken later -- no cycle; every workgroup leaves when its queue is empty.
    constexpr int QUEUE_STRIDE = 64;                           // words between the XCDs' queue counters (256 B)
    constexpr bool CHUNKABLE = CHUNKED;                        // an instantiation of its own: the plain persistent kernel carries no unit loop
    int q_xi = 0, q_nx = 0, q_units = 0;
    if (CHUNKABLE && io.nsets > 0) {
        const uint32_t xcc = __builtin_amdgcn_s_getreg(63508) & 15u;                  // HW_REG_XCC_ID
        q_xi = (int)((S.xcd_rank >> (4u * xcc)) & 15ull);                             // rank of this XCD among the device's (evg_create probes them); 15 = unknown
        if (q_xi < S.nxcd) {
            q_nx = (io.nsets - q_xi + S.nxcd - 1) / S.nxcd;                           // sets q_xi, q_xi + nxcd, ... are this XCD's
            q_units = q_nx * ((io.turns + io.chunk_turns - 1) / io.chunk_turns);
        } else if (threadIdx.x == 0) {
            raise_fault(S.fault, S.fault_seen, 2u);                                   // a workgroup on an XCD the probe did not see: never expected
        }
    }
    STAMP_WAVE_BEGIN();
    for (;;) {                                           // one pass per unit (exactly one pass in a plain launch)
    int wg_set = (int)blockIdx.x, wg_chunk = 0;
    if (CHUNKABLE && io.nsets > 0) {
        int q = 0;
        if (threadIdx.x == 0) q = (int)atomicAdd(S.queue + q_xi * QUEUE_STRIDE, 1u);        // every XCD's counter on a line of its own
        q = __builtin_amdgcn_readfirstlane(q);
        if (q >= q_units) break;
        wg_chunk = q / q_nx;
        wg_set = (q - wg_chunk * q_nx) * S.nxcd + q_xi;
    }
    const int e0 = io.env_lo + wg_set * EPW;          // this launch plays envs [env_lo, env_hi) of the handle (launch_step)
    const int nvalid = min(EPW, io.env_hi - e0);
    const bool valid = envlane && E < nvalid;
    const int e = valid ? e0 + E : e0;
    const size_t N = (size_t)S.N;
    const DevTables* T = S.T;

    STAMP(0);
    if (CHUNKABLE && wg_chunk > 0) {
        // wait for the set's previous chunk (relaxed polls that bypass the L1), then ONE agent-scope acquire: it invalidates this CU's L1,
        // which may still hold lines of this set from an earlier chunk.  The wait is bounded IN TIME (s_memrealtime: a constant 100 MHz
        // counter, so the bound does not depend on the shader clock or on how long a poll takes): a predecessor chunk is ~0.4 ms of work, a
        // wave that has waited 5 s gives up, flags the handle (fault word: every path on which results leave the handle reports it, the pack
        // kernel poisons its rows) and goes on, so the grid always drains.
        const uint32_t* flag = S.progress + (e0 >> 5);
        const uint32_t want = io.progress_base + (uint32_t)wg_chunk;
        constexpr unsigned long long kGiveUpTicks = 500000000ull;     // 5 s at 100 MHz
        unsigned long long t_wait0 = 0;
        bool waiting = false;
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != want) {
            const unsigned long long now = __builtin_amdgcn_s_memrealtime();
            if (!waiting) { waiting = true; t_wait0 = now; }
            if (now - t_wait0 > kGiveUpTicks) {
                if (threadIdx.x == 0) raise_fault(S.fault, S.fault_seen, 1u);
                break;
            }
            __builtin_amdgcn_s_sleep(16);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
#ifdef EVG_DIAG      // experiment knobs (tools/stagger.py): delay = slot x a + simd x b sleeps of 64 cycles, a = ablate[15:8] - 1, b = ablate[23:16]
    const int kStaggerSlot = (A->io_.ablate >> 8) & 0xFFu ? (int)((A->io_.ablate >> 8) & 0xFFu) - 1 : 67, kStaggerSimd = (int)((A->io_.ablate >> 16) & 0xFFu);
#else
    constexpr int kStaggerSlot = 67, kStaggerSimd = 0;      // x 64 cycles (s_sleep 1).  Round-4 sweep of the final kernel (profiles/r04_f_stagger_single_turn.txt): a plateau from
                                                            // 59 to 75 x 64 cycles (26.8 us per launch), 27.5 at round 3's 84, 27.8-29.5 below 55 and above 100, 28.9 without

#endif
    // ---- prologue loads: the constant tables (one blob, already in its LDS layout) and this lane's state (env fastest; the two player rows of a group index interleave
    // across lanes).  Every load is issued before the first LDS store, so the launch pays ONE memory round trip here
    // instead of one per table and one for the state.
    constexpr int TV = (int)(sizeof(LdsTables) / 16);   // 77 16-byte pieces: two loads per lane
    static_assert(TV > WG && TV <= 2 * WG, "table blob is copied in two rounds");
    const uint4* timg = reinterpret_cast<const uint4*>(&T->lds);
    const uint4 tv0 = timg[lane], tv1 = timg[lane + WG < TV ? lane + WG : 0];
    const uint32_t envw = S.env[e];
    uint32_t episode = S.episode[e];
    float ep_ret = S.ep_ret[(size_t)P * N + e];         // this player's running episode return: a register across the launch's turns
    uint32_t st[3], g_in[12], n_in[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) st[j] = S.stamp[(size_t)(P * 3 + j) * N + e];
#pragma unroll
    for (int k = 0; k < 12; ++k) g_in[k] = S.grp[(size_t)(P * 12 + k) * N + e];
#pragma unroll
    for (int j = 0; j < 3; ++j) n_in[j] = S.node[(size_t)(P * 3 + j) * N + e];   // player 0 lane: nodes 1..6, player 1 lane: nodes 7..11 (two per word)
    // fused scripted agents: this seat's agent object (three words) lives in registers across the launch's turns
    uint32_t ag_cycle = 0, ag_swarm = 0, ag_dfs = 0;
    const size_t ai = (size_t)P * N + e;
    if (io.gen_actions == 2) { ag_cycle = S.agent_cycle[ai]; ag_swarm = S.agent_swarm[ai]; ag_dfs = S.agent_dfs[ai]; }
    // caller-supplied orders (evg_step): this player's 7 rows are part of the same round trip
    int2 act_in[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) act_in[i] = make_int2(0, 0);
    if constexpr (SEAT) {                               // the caller's seat: [N][7][2], or its rows of a [N][2][7][2] tensor
        if (io.actions && P == io.seat) {
            const int2* ap = reinterpret_cast<const int2*>(io.actions) + (io.actions_both ? ((size_t)e * 2 + P) * NA : (size_t)e * NA);
#pragma unroll
            for (int i = 0; i < NA; ++i) act_in[i] = ap[i];
        }
    } else if (!MULTI && !io.gen_actions && io.actions) {
        const int2* ap = reinterpret_cast<const int2*>(io.actions) + ((size_t)e * 2 + P) * NA;
#pragma unroll
        for (int i = 0; i < NA; ++i) act_in[i] = ap[i];
    }
    int turn = (int)(envw & 0xFFu);
    int status = (int)((envw >> 8) & 3u);
    if constexpr (!MULTI) {
        // Single-turn launches: all 2 048 wavefronts start together and every SIMD's two waves would run the same phases in
        // lockstep, competing for the same issue slots phase by phase.  The wave in hardware slot 1 therefore waits STAGGER
        // cycles here, with its loads already in flight (tools/stagger.py: 35.0 -> 32.6 us per launch at 65 536 envs) ...
        {
            const uint32_t hw = __builtin_amdgcn_s_getreg(12292);                  // HW_ID[6:0]: wave_id (the wave's slot on its SIMD) [3:0], simd_id [5:4]
            // (only while the whole grid is resident at once -- STEP_F_STAGGER, set by launch_step from the device's capacity: up to
            // 2 048 workgroups = 65 536 envs on a whole MI355X; a larger grid queues behind itself and its waves start at different times anyway)
            const int nsleep = (io.flags & STEP_F_STAGGER) ? (int)(hw & 1u) * kStaggerSlot + (int)((hw >> 4) & 3u) * kStaggerSimd : 0;
            for (int i = 0; i < nsleep; ++i) __builtin_amdgcn_s_sleep(1);
            // (issue priority for either wave of the pair makes a single-turn launch no shorter: for the late wave 32.5 -> 38.0 us,
            // for the early wave no change; A/B on one box)
        }
        // ... and the orders this kernel draws itself need only the turn and the episode (the first two loads), so they are
        // drawn while the group / node words are still on their way
        if (io.gen_actions == 1) gen_random_rows(S.seed_lo, S.seed_hi, S.env_id_base + (uint32_t)e, episode, turn, P, act_in);
    }
    {
        uint4* lt = reinterpret_cast<uint4*>(&L.tab);
        lt[lane] = tv0;
        if (lane + WG < TV) lt[lane + WG] = tv1;
    }
    if (envlane) {
#pragma unroll
        for (int k = 0; k < 12; ++k) L.G[k][lane] = g_in[k];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int n = P ? 7 + j : 1 + j;
            if (n <= NN) L.NW[n][E] = (n_in[j >> 1] >> (16 * (j & 1))) & 0xFFFFu;
        }
    }
    WAVE_SYNC();
    // Every prologue load is waited for here, before the turn loop: a register whose load may still be in flight on SOME path
    // makes the compiler put s_waitcnt vmcnt(0) in front of its first use inside the loop, where it would wait for the previous
    // turn's observation stores on every turn.
    asm volatile("" :: "v"(episode), "v"(ep_ret), "v"(ag_cycle), "v"(ag_swarm), "v"(ag_dfs));
    const bool observe_only = io.observe_only != 0;
    // stock-entropy mode: the env's MT19937 is advanced by the lane of player 0, in the reference's draw order
    MtGen mt{nullptr, 0, 0};
    const bool mt_lane = MT && valid && P == 0;
    if (MT && mt_lane) { mt.key = S.mt_key + e; mt.stride = N; mt.pos = S.mt_pos[e]; }

    // One iteration = one turn.  evg_step runs exactly one; the fused rollout driver lets every wavefront play
    // `turns` consecutive turns of its envs with the state resident in LDS/registers: outputs are still written every
    // turn, but no wave waits for the slowest wave of the grid between turns, and nothing is re-loaded.
    // (chunked launch: this workgroup's chunk_turns turns, the launch's last chunk what is left of io.turns)
    const int nturns = MULTI ? ((CHUNKABLE && io.nsets > 0) ? min(io.chunk_turns, io.turns - wg_chunk * io.chunk_turns) : io.turns) : 1;   // the single-turn instantiation has no loop at all
    for (int iter = 0; iter < nturns; ++iter) {
    // Multi-turn form: the argument pointer and the lane id are made opaque once per turn, so that argument fields, table
    // entries and per-lane address arithmetic are recomputed next to their uses instead of being hoisted out of the loop
    // and kept live across it (which overflowed the register file).  The declarations below shadow the prologue's.
    int lane_ = threadIdx.x;
    if (MULTI) { asm volatile("" : "+s"(A)); asm volatile("" : "+v"(lane_)); T = S.T; }
    const int lane = lane_;
    if constexpr (MULTI) {
        // The two waves of a SIMD share its issue slots; the arbiter goes by priority, then by age.  With equal priorities the older
        // wave (hardware slot 0) is favoured throughout: it finishes a 150-turn launch 25 % earlier and its partner then runs alone,
        // which uses the SIMD less well than two waves do (profiles/r02_c_wave_times.txt, r02_d_*).  Taking turns at priority 1 / 0
        // (one turn each) removed most of that (17.9 -> 17.2 us per turn) but left the older wave 7 % ahead, because half of the time
        // the two hold the same priority and age decides.  So no ties: the younger wave stays at 1, the older one takes 2 in three
        // turns of five and 0 in the other two -- with 1 : 1 the younger ends 12 us ahead in a 20-turn launch, with 2 : 1 the older
        // one does, with 3 : 2 the pair ends within 3 us of each other (tools/wave_times.py) -- 16.4 -> 16.0 us per turn (A/B on one box).
        if (__builtin_amdgcn_s_getreg(6148) & 1u) __builtin_amdgcn_s_setprio(1);
        else if ((0x15u >> (iter % 5)) & 1u) __builtin_amdgcn_s_setprio(2);
        else __builtin_amdgcn_s_setprio(0);
    }
    if (MULTI) PHASE(0);                                // diagnostic build: the stamps of a launch are those of its last turn
    const bool envlane = LPW == WG || lane < LPW;
    const int E = envlane ? lane >> 1 : 0, P = lane & 1;
    const int col = envlane ? lane : 0;                 // LDS column (helpers never write; their reads are discarded)
    const bool valid = envlane && E < nvalid;
    const int e = valid ? e0 + E : e0;
    const size_t N = (size_t)S.N;
    const uint64_t p1nib = L.tab.nib[0];
    const uint64_t spd_n = L.tab.nib[1 + P], ctl_n = L.tab.nib[3 + P], cst_n = L.tab.nib[5 + P], typ_n = L.tab.nib[7 + P];
    const uint32_t misc = (uint32_t)L.tab.nib[9];
    const int max_turns = (int)(misc & 0xFFu);
    // this player's 7 order rows: read from the caller's tensor (in the prologue), or -- in the fused rollouts -- produced
    // here by the same generators as evg_random_actions / evg_scripted_actions and written out
    int2 act[NA];
    if (io.gen_actions) {
        if (io.gen_actions == 1) {
            if constexpr (MULTI) {
                gen_random_rows(S.seed_lo, S.seed_hi, S.env_id_base + (uint32_t)e, episode, turn, P, act);
            } else {
#pragma unroll
                for (int i = 0; i < NA; ++i) act[i] = act_in[i];         // drawn in the prologue, under the state loads
            }
        } else {                                        // on-device scripted agents of both seats (evg_rollout_policies, fused)
            const ChipView<LPW> view{&L, col, E, P, turn, p1nib};
            // both seats' policy ids as scalars, selected per lane (the compiler would otherwise turn the select into a per-lane
            // global load from the argument segment, and wait for vmcnt(0) -- i.e. for last turn's stores -- in front of its use)
            int pol0 = io.policy0, pol1 = io.policy1;
            asm volatile("" : "+s"(pol0), "+s"(pol1));
            const AgentTabs atabs{L.tab.nib[11], L.tab.nib[12], L.tab.nib[13], T};
            const bool bot_lane = !SEAT || P != io.seat;        // one-seat form: the caller's lane has no agent (its object is neither consulted nor stored)
            agent_rows(P ? pol1 : pol0, view, atabs, S.seed_lo, S.seed_hi, S.env_id_base + (uint32_t)e, episode, P, true, status == 0 && bot_lane,
                       &ag_cycle, &ag_swarm, &ag_dfs, act);
            if constexpr (SEAT) {
#pragma unroll
                for (int i = 0; i < NA; ++i) act[i] = bot_lane ? act[i] : act_in[i];
            }
            if (valid && bot_lane && (!MULTI || iter == nturns - 1)) {      // padding lanes of a partial last workgroup never write
                const size_t ai_ = (size_t)P * N + e;
                S.agent_cycle[ai_] = ag_cycle; S.agent_swarm[ai_] = ag_swarm; S.agent_dfs[ai_] = ag_dfs;
            }
        }
        if (valid && io.actions_out) {
            int2* ao = reinterpret_cast<int2*>(io.actions_out) + ((size_t)e * 2 + P) * NA;
#pragma unroll
            for (int i = 0; i < NA; ++i) ao[i] = act[i];
        }
    } else {
#pragma unroll
        for (int i = 0; i < NA; ++i) act[i] = act_in[i];             // loaded in the prologue (single-turn form only)
    }
    PHASE(1);

    const bool frozen = status != 0;                    // finished, not auto-reset: repeat terminal outputs
    const bool play = valid && !frozen && !observe_only;

    if (play) {
        turn += 1;                                                               // server.py:214
        // ---------------- orders of this lane's player (server.py:218-271)
        // Accepting an order changes neither the group's location nor whether it is `moving`, so tests 2 and 3
        // of every row can be taken from the pre-order words; rows interact only through test 1 (an id already
        // commanded this turn) and, for aliased ids, through the order of the writes (the later row wins, as in
        // the reference).  That makes the 7 LDS lookups independent instead of a 7-deep dependent chain.
        if (!ABLATED(1u) && io.gen_actions == 1) {
            // Orders drawn in this kernel by gen_random_rows: 7 DISTINCT group ids in 0..11 and node ids in 1..11 by construction, so
            // the domain checks, the Python-list negative indices and the "already commanded this turn" test of the general path
            // below cannot trigger; every row is independent.
            uint32_t wv[NA], nv[NA], dv[NA];
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                nv[i] = P ? (uint32_t)((p1nib >> (4 * act[i].y)) & 15u) : (uint32_t)act[i].y;       // :233-234
                wv[i] = L.G[act[i].x][lane];
            }
#pragma unroll
            for (int i = 0; i < NA; ++i) dv[i] = (uint32_t)((L.tab.adj[wv[i] & G_LOC_M] >> (4 * nv[i])) & 15u);   // :245-250
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const uint32_t w = wv[i];
                if (((w & G_MODE_M) >> G_MODE_S) != MODE_MOVING && dv[i] != 0)                       // :243, :267-270
                    L.G[act[i].x][lane] = (w & ~(G_DEST_M | G_DIST_M | G_MODE_M)) | (nv[i] << G_DEST_S) | (dv[i] << G_DIST_S) | (MODE_READY << G_MODE_S);
            }
        } else if (!ABLATED(1u)) {
            int gidv[NA], nidv[NA], rawv[NA];
            uint32_t wv[NA], dv[NA];
            bool okv[NA];
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                int gid = act[i].x, nid = act[i].y;
                // Domain: ids in [-12, 11] behave like the reference's Python lists (a negative index counts from the end:
                // groups[gid] at :235, p1_node_map[nid] at :92 for player 1); for player 0 a negative node id matches no
                // connection; anything else would raise in the reference and is an invalid order here.
                okv[i] = gid >= -12 && gid < 12 && nid >= (P ? -12 : 0) && nid < 12;
                rawv[i] = okv[i] ? gid + 12 : 0;                                 // used_swarms keeps the ids as given (:241,252)
                gid = okv[i] ? (gid < 0 ? gid + 12 : gid) : 0;
                nid = okv[i] ? (nid < 0 ? nid + 12 : nid) : 0;
                nidv[i] = P ? (int)((p1nib >> (4 * nid)) & 15u) : nid;           // :233-234
                gidv[i] = gid;
                wv[i] = L.G[gid][lane];
            }
#pragma unroll
            for (int i = 0; i < NA; ++i)
                dv[i] = (uint32_t)((L.tab.adj[wv[i] & G_LOC_M] >> (4 * nidv[i])) & 15u);   // test3 + distance, :245-250
            uint32_t used = 0;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const uint32_t w = wv[i];
                const bool accept = okv[i] && !((used >> rawv[i]) & 1u) && ((w & G_MODE_M) >> G_MODE_S) != MODE_MOVING && dv[i] != 0;
                used |= (accept ? 1u : 0u) << rawv[i];
                if (accept)                                                      // :267-270
                    L.G[gidv[i]][lane] = (w & ~(G_DEST_M | G_DIST_M | G_MODE_M)) | ((uint32_t)nidv[i] << G_DEST_S) | (dv[i] << G_DIST_S) |
                                         (MODE_READY << G_MODE_S);
            }
        }
    }
    PHASE(2);

    // ---------------- combat (server.py:503-654)
    // Stage 0 (lane = env side): pre-combat snapshot.  A group fights at its node if it is alive and not moving
    // (:525) and the node holds such groups of both players (:539).  The reference walks node.groups[p] in list
    // order, which is (arrival stamp, gid) order; the target index uid counts alive units
    // along that order, so each fighting group gets the prefix `base` of alive units listed before it.
    uint32_t g[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) g[k] = envlane ? L.G[k][col] : 0u;
    uint32_t occ = 0;
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        const uint32_t w = g[k];
        const bool elig = (w & G_MASK_M) != 0 && ((w & G_MODE_M) >> G_MODE_S) != MODE_MOVING;
        occ |= (elig ? 1u : 0u) << (w & G_LOC_M);
    }
    const uint32_t contested = (play && !ABLATED(2u)) ? (occ & (uint32_t)xchg1((int)occ)) : 0u;
    if (__any(contested != 0)) {                          // wave-uniform: skip when none of the 32 envs fights
        zero_columns12<LPW>(L.u.c.FS, lane, envlane);
        uint32_t key[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            const uint32_t w = g[k];
            const uint32_t mask = (w & G_MASK_M) >> G_MASK_S;
            const uint32_t elig = (mask != 0 && ((w & G_MODE_M) >> G_MODE_S) != MODE_MOVING) ? 1u : 0u;
            const uint32_t stamp = (st[k >> 2] >> (8 * (k & 3))) & 0xFFu;
            key[k] = (stamp << 21) | ((uint32_t)k << 17) | ((w & G_LOC_M) << 13) | (mask << 1) | elig;
        }
#define EVG_CE(a, b) { const uint32_t lo_ = min(key[a], key[b]); key[b] = max(key[a], key[b]); key[a] = lo_; }
        EVG_SORT12_CES(EVG_CE)
#undef EVG_CE
        // List order (server.py:549-566 walks node.groups[p]): the prefix `base` of alive fighting units listed before a group at its
        // node comes from one returning LDS add per group into this side's per-node total (the FS column of the lane, high half;
        // the low half receives the damage-pool offset in stage 1): the LDS executes a wavefront's adds in issue order.
        uint32_t fmask = 0, basev[12];
        bool fightv[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            const uint32_t kk = key[i];
            const uint32_t loc = (kk >> 13) & 15u, mask = (kk >> 1) & 0xFFFu;
            fightv[i] = (kk & 1u) && ((contested >> loc) & 1u);
            basev[i] = envlane ? atomicAdd(&L.u.c.FS[loc][lane], fightv[i] ? (uint32_t)__popc(mask) << 16 : 0u) : 0u;
        }
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            const uint32_t kk = key[i];
            const uint32_t gid = (kk >> 17) & 15u;
            fmask |= (fightv[i] ? 1u : 0u) << gid;
            // fights | prefix << 16 | node << 12 | alive mask: bits 16..1 of the key are already node << 12 | mask
            if (envlane) L.u.c.SNAP[gid][lane] = fightv[i] ? (0x80000000u | basev[i] | ((kk >> 1) & 0xFFFFu)) : 0u;
        }
        if (envlane) { L.u.c.TURN[lane] = (uint32_t)turn; L.u.c.EPI[lane] = episode; }
        // damage bytes this side needs: one per alive fighting unit, rounded up to a word per node
        int ndw = 0;
        {
            uint32_t c = contested;
            while (c) {
                const uint32_t n = (uint32_t)__ffs(c) - 1u;
                c &= c - 1;
                ndw += (int)(((L.u.c.FS[n][col] >> 16) + 3u) >> 2);
            }
        }
        PHASE(3);

        // Stage 1: wave-wide work list (one item per fighting group) and damage-pool layout, by prefix scan.  The items of the
        // 12-unit group (gid 11) are listed BEFORE all others: only that group needs a second block of draws (more than 8 units)
        // and the health slots 8..11, so only the FIRST round of 64 items -- whose lanes are all busy anyway -- runs the long code,
        // and the last round, 18 items on average, holds 8-unit groups only: phase B deals those to two lanes each.
        const int nf_a = __popc(fmask & 0x7FFu), nf_b = (int)((fmask >> 11) & 1u);
        const int packed = nf_a | (nf_b << 10) | (ndw << 17);   // three counts in one scan: <= 704 items (10 bits), <= 64 (7 bits), <= 2112 words (12 bits)
        int incl = packed;
        // inclusive scan over the 64 lanes in registers: log-steps inside each row of 16 lanes (DPP row_shr, zero fill),
        // then the row totals are carried across rows (DPP row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3)
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xF, 0xF, true);
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xF, 0xF, true);
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xF, 0xF, true);
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xF, 0xF, true);
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x142, 0xA, 0xF, false);
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x143, 0xC, 0xF, false);
        const int excl = incl - packed;
        const int tot = __builtin_amdgcn_readlane(incl, WG - 1), mid = __builtin_amdgcn_readlane(excl, LPW / 2);
        const int excl_a = excl & 0x3FF, excl_b = (excl >> 10) & 0x7F, excl_d = excl >> 17;
        const int tot_a = tot & 0x3FF, tot_b = (tot >> 10) & 0x7F, tot_d = tot >> 17, mid_a = mid & 0x3FF, mid_b = (mid >> 10) & 0x7F, mid_d = mid >> 17;
        // the pool holds every fight of the wave in the common case; otherwise two passes of 16 envs each
        const int npass = tot_d <= DP_CAP ? 1 : 2;
        const uint32_t dmg_nib = (misc >> 8) & 0xFFFFu;
        const bool fast_div = ((misc >> 24) & 1u) != 0;
        for (int ps = 0; ps < npass; ++ps) {
            const bool inpass = npass == 1 || (lane / (LPW / 2)) == ps;     // helper lanes own no items
            const bool second = npass == 2 && ps == 1, first = npass == 2 && ps == 0;
            const int ref_a = second ? mid_a : 0, ref_b = second ? mid_b : 0, ref_d = second ? mid_d : 0;
            const int end_a = first ? mid_a : tot_a, end_b = first ? mid_b : tot_b, end_d = first ? mid_d : tot_d;
            const int na = end_a - ref_a, nitems = na + (end_b - ref_b), ndwords = end_d - ref_d;
            const int nb = nitems - na;                          // items of the 12-unit group: listed FIRST (see phase B)
            if (inpass) {
                int wi = nb + excl_a - ref_a;
                uint32_t f = fmask & 0x7FFu;
                while (f) {
                    const uint32_t gid = (uint32_t)__ffs(f) - 1u;
                    f &= f - 1;
                    L.u.c.W[wi++] = (uint16_t)((uint32_t)lane | (gid << 6));
                }
                if (nf_b) L.u.c.W[excl_b - ref_b] = (uint16_t)((uint32_t)lane | (11u << 6));
                int doff = excl_d - ref_d;
                uint32_t c = contested;
                while (c) {
                    const uint32_t n = (uint32_t)__ffs(c) - 1u;
                    c &= c - 1;
                    const uint32_t tnw = L.u.c.FS[n][lane] & 0xFFFF0000u;     // alive fighting units of this side at the node (stage 0)
                    L.u.c.FS[n][lane] = tnw | (uint32_t)doff;
                    doff += (int)(((tnw >> 16) + 3u) >> 2);
                }
            }
            for (int i = lane; i < ndwords; i += WG) L.u.c.DP[i] = 0;
            WAVE_SYNC();
            PHASE(4);

            // health row of this lane's first item: issued now, consumed in phase B, so that the HBM latency hides
            // behind the draws (a fighting group is very likely to be hit; 64-96 B per group)
            constexpr bool kPrefetch = LPW == WG;      // the 16-env variant runs at 4 waves/SIMD and has no registers to spare
            double hpre[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
            if (kPrefetch) {
                const uint32_t item0 = L.u.c.W[lane < nitems ? lane : 0];
                const int SL0 = (int)(item0 & 63u), gid0 = (int)(item0 >> 6);
                const double2* r2 = reinterpret_cast<const double2*>(S.health + (size_t)(e0 + (SL0 >> 1)) * (2 * NU) + (SL0 & 1) * NU + gid0 * 8);
                if (lane < nitems) {
#pragma unroll
                    for (int sl = 0; sl < 4; ++sl) { const double2 v = r2[sl]; hpre[2 * sl] = v.x; hpre[2 * sl + 1] = v.y; }
                    if (gid0 == 11) {
#pragma unroll
                        for (int sl = 4; sl < 6; ++sl) { const double2 v = r2[sl]; hpre[2 * sl] = v.x; hpre[2 * sl + 1] = v.y; }
                    } else {
                        hpre[8] = hpre[9] = hpre[10] = hpre[11] = 0.0;
                    }
                }
            }

            // Phase A (:549-566): one lane per fighting group; each of its alive units draws one target among the
            // opposing side's alive fighting units at the node; damage accumulates in the pool (LDS atomics,
            // integer and order-free, hence deterministic)
            if constexpr (MT) {
                // stock entropy: one sequential stream per env, consumed exactly in the reference's loop order -- nodes
                // ascending (:505), attacking player 0 then 1 (:549), that player's groups in list order (ascending
                // `base`), one draw per alive unit (:562)
                if (mt_lane && inpass) {
                    uint32_t c = contested;
                    while (c) {
                        const int node = __ffs(c) - 1;
                        c &= c - 1;
                        for (int side = 0; side < 2; ++side) {
                            const int SL = lane | side;
                            const uint32_t fso = L.u.c.FS[node][SL ^ 1];
                            const uint32_t tot_o = fso >> 16, doff_o = fso & 0xFFFFu;
                            const uint64_t tn_s = L.tab.nib[7 + side];
                            int last = -1;
                            for (;;) {
                                int best = 256, bg = -1, bcnt = 0;
                                for (int k = 0; k < 12; ++k) {
                                    const uint32_t sp = L.u.c.SNAP[k][SL];
                                    const int b = (int)((sp >> 16) & 0xFFu);
                                    if ((sp >> 31) && (int)((sp >> 12) & 15u) == node && b > last && b < best) { best = b; bg = k; bcnt = __popc(sp & 0xFFFu); }
                                }
                                if (bg < 0) break;
                                last = best;
                                const uint32_t type = (uint32_t)((tn_s >> (4 * bg)) & 15u);
                                const uint32_t dmg = (dmg_nib >> (4 * type)) & 15u;
                                for (int j = 0; j < bcnt; ++j) {
                                    const uint32_t uid = mt_randint(mt, tot_o);
                                    atomicAdd(&L.u.c.DP[doff_o + (uid >> 2)], dmg << (8 * (uid & 3u)));
                                }
                            }
                        }
                    }
                }
            } else {
            const uint32_t seed_lo = S.seed_lo, seed_hi = S.seed_hi, id_base = S.env_id_base + (uint32_t)e0;
            for (int it = lane; it < nitems; it += WG) {
                const uint32_t item = L.u.c.W[it];
                const int SL = (int)(item & 63u), gid = (int)(item >> 6), side = SL & 1;
                const uint32_t sp = L.u.c.SNAP[gid][SL];
                const int node = (int)((sp >> 12) & 15u), cnt = __popc(sp & 0xFFFu);
                const uint32_t fso = L.u.c.FS[node][SL ^ 1];
                const uint32_t tot_o = fso >> 16, doff_o = fso & 0xFFFFu;
                const uint64_t tn_s = L.tab.nib[7 + side];
                const uint32_t type = (uint32_t)((tn_s >> (4 * gid)) & 15u);
                const uint32_t dmg = (dmg_nib >> (4 * type)) & 15u;
                const int turn_e = (int)L.u.c.TURN[SL];
                const uint32_t epi_e = L.u.c.EPI[SL], env_id_e = id_base + (uint32_t)(SL >> 1);
                for (int b = 0; b * 8 < cnt; ++b) {               // one block = eight 16-bit draws: a second one only for the 12-unit group
                    const uint4 x = rng_block(seed_lo, seed_hi, env_id_e, epi_e, RNG_COMBAT, (uint32_t)b, turn_e, node, side, gid);
                    const uint32_t xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        if (b * 8 + i < cnt) {
                            const uint32_t uid = __umul24(rng_half(xs, i), tot_o) >> 16;                // :562 (tot_o <= 100)
                            atomicAdd(&L.u.c.DP[doff_o + (uid >> 2)], dmg << (8 * (uid & 3u)));         // :563-566
                        }
                    }
                }
            }
            }
            WAVE_SYNC();
            PHASE(5);

            // The prefetched rows are waited for HERE by every lane, used or not: a register with a load still in flight makes the
            // compiler wait for vmcnt(0) wherever that register is written next (the movement phase), and vmcnt(0) also waits for
            // the health stores of phase B to be acknowledged -- a full store round trip per turn.
            if (kPrefetch) {
#pragma unroll
                for (int sl = 0; sl < 12; ++sl) asm volatile("" :: "v"(hpre[sl]));
            }
            // Phase B (:573-644): one lane per fighting group; the uid-th alive unit of the snapshot (list-order
            // prefix + rank among the group's alive slots) takes its summed damage.  Both directions read only
            // the snapshot and the pool, so they are simultaneous like in the reference.
            // An instruction costs a wavefront the same whether 64 or 16 of its lanes are active (profiles/r03_m_exec_mask_issue_rates.txt),
            // and a wavefront has 82 items per turn on average: one full round and one with 18 items.  That sparse last round -- 8-unit
            // groups only, thanks to the list order -- is dealt to TWO lanes per item when it has at most 32 items: each lane takes four
            // of the eight unit slots, the pair exchanges the dead-unit mask and the half sums by DPP (the sum keeps numpy's pairwise
            // order: (h0+h1)+(h2+h3) is the left half, (h4+h5)+(h6+h7) the right one).
            constexpr bool kSplitLastRound = LPW == WG;
            const int rem_items = nitems & (WG - 1), full_items = nitems - rem_items;
            const bool split_last = kSplitLastRound && rem_items > 0 && rem_items <= WG / 2 && full_items >= nb;
            const int nmain = split_last ? full_items : nitems;
            for (int it = lane; it < nmain; it += WG) {
                const uint32_t item = L.u.c.W[it];
                const int SL = (int)(item & 63u), gid = (int)(item >> 6), side = SL & 1;
                const uint32_t sp = L.u.c.SNAP[gid][SL];
                const int node = (int)((sp >> 12) & 15u);
                const uint32_t mask = sp & 0xFFFu;
                const uint32_t doff = L.u.c.FS[node][SL] & 0xFFFFu;
                const uint32_t base = (sp >> 16) & 0xFFu;
                // the group's alive units are consecutive target indices, so their damage bytes are one run of <= 12 bytes in the
                // pool starting at byte `run0` (<= 15 bytes with the misalignment: four words, read at once)
                const uint32_t run0 = doff * 4u + base, w0i = run0 >> 2, sh0 = run0 & 3u;
                uint32_t dw[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) dw[q] = L.u.c.DP[min(w0i + (uint32_t)q, (uint32_t)(DP_CAP - 1))];
                // align the run to byte 0 (v_alignbyte_b32) and clear what lies beyond the group's own `cnt` bytes (the next
                // group's damage): byte r of {a0, a1, a2} is then the damage of the group's r-th alive unit
                const uint32_t cnt = (uint32_t)__popc(mask);                                          // 1..12
                uint32_t a0 = __builtin_amdgcn_alignbyte(dw[1], dw[0], sh0), a1 = __builtin_amdgcn_alignbyte(dw[2], dw[1], sh0);
                uint32_t a2 = __builtin_amdgcn_alignbyte(dw[3], dw[2], sh0);
                {
                    const uint64_t keep = cnt >= 8u ? ~0ull : ((1ull << (8u * cnt)) - 1ull);
                    a0 &= (uint32_t)keep; a1 &= (uint32_t)(keep >> 32);
                    a2 = cnt > 8u ? (cnt >= 12u ? a2 : (a2 & ((1u << (8u * (cnt - 8u))) - 1u))) : 0u;
                }
                if ((a0 | a1 | a2) != 0u) {
                    double* row = S.health + (size_t)(e0 + (SL >> 1)) * (2 * NU) + side * NU + gid * 8;
                    double h[12];
                    if (kPrefetch && it < WG) {                     // first round: prefetched before the draws
#pragma unroll
                        for (int sl = 0; sl < 12; ++sl) h[sl] = hpre[sl];
                    } else {
                        const double2* r2 = reinterpret_cast<const double2*>(row);
#pragma unroll
                        for (int sl = 0; sl < 4; ++sl) { const double2 v = r2[sl]; h[2 * sl] = v.x; h[2 * sl + 1] = v.y; }
                        if (gid == 11) {
#pragma unroll
                            for (int sl = 4; sl < 6; ++sl) { const double2 v = r2[sl]; h[2 * sl] = v.x; h[2 * sl + 1] = v.y; }
                        } else {
                            h[8] = h[9] = h[10] = h[11] = 0.0;
                        }
                    }
                    const uint64_t tn_s = L.tab.nib[7 + side];
                    const uint32_t type = (uint32_t)((tn_s >> (4 * gid)) & 15u);
                    const int ctrl_by = (int)((L.NW[node][SL >> 1] >> 10) & 3u) - 1;
                    const int di = (int)type * 12 + (ctrl_by == side ? node : 0);                    // :592-597 (fort bonus dead)
                    const double denom = L.tab.den[di], rcp = L.tab.rcp[di];
                    // Slot sl holds the unit of rank popc(mask below sl); its damage byte is picked with one v_perm_b32 (selector =
                    // rank, the other three selector bytes 0x0C = constant zero).  A slot whose unit is already dead picks the byte
                    // of the next alive unit: harmless, its health is 0.0 and stays 0.0 (0 - loss clamps to 0), its mask bit stays clear.
                    uint32_t deadmask = 0;
                    auto apply_hits = [&](auto fast) {           // one straight-line body per quotient form (wave-uniform choice)
                        auto hit = [&](int sl, uint32_t d) {
                            double loss;
                            if constexpr (decltype(fast)::value) {
                                const double a = (double)__umul24(10u, d);                            // exact, like 10. * tgt_dmg (d is one byte)
                                const double q0 = a * rcp;
                                loss = __builtin_fma(__builtin_fma(-denom, q0, a), rcp, q0);          // == a / denom (DevTables::fast_div)
                            } else {
                                loss = (10.0 * (double)d) / denom;                                    // :601
                            }
                            const double hv = h[sl] - loss;                                           // :609
                            const bool dead = hv <= 0.0;                                              // :615-618
                            h[sl] = dead ? 0.0 : hv;
                            deadmask |= dead ? (1u << sl) : 0u;
                        };
#pragma unroll
                        for (int sl = 0; sl < 8; ++sl) {         // rank <= sl < 8: bytes of a0, a1
                            const uint32_t sel = (uint32_t)__popc(mask & ((1u << sl) - 1u)) | 0x0C0C0C00u;
                            hit(sl, __builtin_amdgcn_perm(a1, a0, sel));
                        }
                        if (gid == 11) {                         // the 12-unit group: slots 8..11, ranks up to 11
#pragma unroll
                            for (int sl = 8; sl < 12; ++sl) {
                                const uint32_t rank = (uint32_t)__popc(mask & ((1u << sl) - 1u));
                                const uint32_t dA = __builtin_amdgcn_perm(a1, a0, rank | 0x0C0C0C00u), dB = __builtin_amdgcn_perm(0u, a2, (rank - 8u) | 0x0C0C0C00u);
                                hit(sl, rank < 8u ? dA : dB);
                            }
                        }
                    };
                    if (fast_div) apply_hits(std::true_type{}); else apply_hits(std::false_type{});
                    const uint32_t newmask = mask & ~deadmask;
                    double2* w2 = reinterpret_cast<double2*>(row);
#pragma unroll
                    for (int sl = 0; sl < 4; ++sl) w2[sl] = make_double2(h[2 * sl], h[2 * sl + 1]);
                    double sum = np_sum8(h);
                    if (gid == 11) {
#pragma unroll
                        for (int sl = 4; sl < 6; ++sl) w2[sl] = make_double2(h[2 * sl], h[2 * sl + 1]);
                        sum = (((sum + h[8]) + h[9]) + h[10]) + h[11];
                    }
                    const int alive = __popc(newmask);
                    const uint32_t avg = alive ? (uint32_t)(int)(sum / (double)alive) : 0u;          // :491 truncation
                    const uint32_t w = L.G[gid][SL];
                    L.G[gid][SL] = (w & ~(G_MASK_M | G_AVG_M)) | (newmask << G_MASK_S) | (avg << G_AVG_S);
                }
            }
            if (split_last) {
                const int it = full_items + (lane >> 1), hf = lane & 1;      // item and which half of its unit slots (0: slots 0..3, 1: slots 4..7)
                if (it < nitems) {
                    const uint32_t item = L.u.c.W[it];
                    const int SL = (int)(item & 63u), gid = (int)(item >> 6), side = SL & 1;      // gid < 11: an 8-unit group
                    const uint32_t sp = L.u.c.SNAP[gid][SL];
                    const int node = (int)((sp >> 12) & 15u);
                    const uint32_t mask = sp & 0xFFu;
                    const uint32_t doff = L.u.c.FS[node][SL] & 0xFFFFu;
                    const uint32_t base = (sp >> 16) & 0xFFu;
                    const uint32_t run0 = doff * 4u + base, w0i = run0 >> 2, sh0 = run0 & 3u;
                    uint32_t dw[3];
#pragma unroll
                    for (int q = 0; q < 3; ++q) dw[q] = L.u.c.DP[min(w0i + (uint32_t)q, (uint32_t)(DP_CAP - 1))];
                    const uint32_t cnt = (uint32_t)__popc(mask);                                      // 1..8
                    uint32_t a0 = __builtin_amdgcn_alignbyte(dw[1], dw[0], sh0), a1 = __builtin_amdgcn_alignbyte(dw[2], dw[1], sh0);
                    {
                        const uint64_t keep = cnt >= 8u ? ~0ull : ((1ull << (8u * cnt)) - 1ull);
                        a0 &= (uint32_t)keep; a1 &= (uint32_t)(keep >> 32);
                    }
                    if ((a0 | a1) != 0u) {                                                        // the same for both lanes of the pair
                        double* row = S.health + (size_t)(e0 + (SL >> 1)) * (2 * NU) + side * NU + gid * 8 + hf * 4;
                        double h[4];
                        {
                            const double2* r2 = reinterpret_cast<const double2*>(row);
                            const double2 v0 = r2[0], v1 = r2[1];
                            h[0] = v0.x; h[1] = v0.y; h[2] = v1.x; h[3] = v1.y;
                        }
                        const uint64_t tn_s = L.tab.nib[7 + side];
                        const uint32_t type = (uint32_t)((tn_s >> (4 * gid)) & 15u);
                        const int ctrl_by = (int)((L.NW[node][SL >> 1] >> 10) & 3u) - 1;
                        const int di = (int)type * 12 + (ctrl_by == side ? node : 0);                // :592-597 (fort bonus dead)
                        const double denom = L.tab.den[di], rcp = L.tab.rcp[di];
                        uint32_t deadmask = 0;
                        auto apply_hits4 = [&](auto fast) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const uint32_t sl = (uint32_t)(4 * hf + j);                          // this lane's slot
                                const uint32_t sel = (uint32_t)__popc(mask & ((1u << sl) - 1u)) | 0x0C0C0C00u;
                                const uint32_t d = __builtin_amdgcn_perm(a1, a0, sel);
                                double loss;
                                if constexpr (decltype(fast)::value) {
                                    const double a = (double)__umul24(10u, d);
                                    const double q0 = a * rcp;
                                    loss = __builtin_fma(__builtin_fma(-denom, q0, a), rcp, q0);
                                } else {
                                    loss = (10.0 * (double)d) / denom;                                // :601
                                }
                                const double hv = h[j] - loss;                                        // :609
                                const bool dead = hv <= 0.0;                                          // :615-618
                                h[j] = dead ? 0.0 : hv;
                                deadmask |= dead ? (1u << sl) : 0u;
                            }
                        };
                        if (fast_div) apply_hits4(std::true_type{}); else apply_hits4(std::false_type{});
                        double2* w2 = reinterpret_cast<double2*>(row);
                        w2[0] = make_double2(h[0], h[1]);
                        w2[1] = make_double2(h[2], h[3]);
                        const uint32_t newmask = mask & ~(deadmask | (uint32_t)xchg1((int)deadmask));
                        const double mine = (h[0] + h[1]) + (h[2] + h[3]);
                        const double other = __hiloint2double(xchg1(__double2hiint(mine)), xchg1(__double2loint(mine)));
                        const double sum = hf ? other + mine : mine + other;                          // left half + right half (np.sum's pairwise order)
                        const int alive = __popc(newmask);
                        const uint32_t avg = alive ? (uint32_t)(int)(sum / (double)alive) : 0u;      // :491 truncation
                        if (hf == 0) {
                            const uint32_t w = L.G[gid][SL];
                            L.G[gid][SL] = (w & ~(G_MASK_M | G_AVG_M)) | (newmask << G_MASK_S) | (avg << G_AVG_S);
                        }
                    }
                }
            }
            WAVE_SYNC();
        }
    }
    PHASE(6);

    // ---------------- movement of this lane's groups (server.py:656-706), branch-free
    uint32_t gw[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) gw[k] = envlane ? L.G[k][col] : 0u;
    if (play && !ABLATED(4u)) {
        static_assert(MODE_READY == 1 && MODE_MOVING == 2, "ready -> moving is +1 in the mode field; bit 1 of the field is `moving`");
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            const uint32_t w = gw[k];
            const bool alive = (w & G_MASK_M) != 0;                                    // not destroyed, :663
            const uint32_t spd = (uint32_t)((spd_n >> (4 * k)) & 15u) << G_DIST_S;     // the group's speed, aligned with the distance field
            const bool ready = alive && (w & G_MODE_M) == (MODE_READY << G_MODE_S);
            const bool moving = alive && (w & (MODE_MOVING << G_MODE_S)) != 0;
            const bool arrive = moving && (w & G_DIST_M) <= spd;                       // :671, :678-695
            const uint32_t w_arrive = (w & ~(G_LOC_M | G_DEST_M | G_DIST_M | G_MODE_M)) | ((w & G_DEST_M) >> G_DEST_S);
            uint32_t nw_ = ready ? w + (1u << G_MODE_S) : w;                           // :664-667: moves from the next turn on
            nw_ = moving ? w - spd : nw_;                                              // in transit: distance_remaining -= speed (> 0 left)
            nw_ = arrive ? w_arrive : nw_;
            const uint32_t sh = 8 * (k & 3);
            st[k >> 2] = arrive ? ((st[k >> 2] & ~(0xFFu << sh)) | ((uint32_t)turn << sh)) : st[k >> 2];
            gw[k] = nw_;
        }
    }
    PHASE(7);

    // ---------------- per-node aggregates of this side (post-movement): capture points | units listed << 16
    zero_columns12<LPW>(L.u.A, lane, envlane);
    int my_unit_score = 0, my_alive = 0;
    int cntv[12];                      // alive units per group: also what the observation shows (:493)
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        const uint32_t w = gw[k];
        int cnt = __popc(w & G_MASK_M);
        asm volatile("" : "+v"(cnt));           // keep the count in its register for the observation (the compiler would recompute it there)
        cntv[k] = cnt;
        const bool elig = ((w & G_MODE_M) >> G_MODE_S) != MODE_MOVING;                          // :720
        const uint32_t ctl = (uint32_t)((ctl_n >> (4 * k)) & 15u);
        const uint32_t add = (elig ? (uint32_t)cnt * ctl : 0u) | ((uint32_t)cnt << 16);
        if (envlane) atomicAdd(&L.u.A[w & G_LOC_M][lane], add);                                 // ds_add_u32 (adds 0 for a destroyed group)
        my_unit_score += cnt * (int)((cst_n >> (4 * k)) & 15u);                                 // :315-317
        my_alive += cnt;
    }
    WAVE_SYNC();

    // ---------------- capture (server.py:708-767) and node scores (:297-310): the pair splits the nodes
    int part0 = 0, part1 = 0;          // score contributions of this lane's nodes to player 0 / player 1
    int base_cap = 0;
    {
        // The pair splits the node IDs 0..11 in halves (player 0's lane: 0..5, where ID 0 does not exist; player 1's lane: 6..11), so
        // every LDS address below is one per-lane base plus a constant.  controlledBy is kept in its stored form (+1: 0 = nobody).
        const int nb = P ? 6 : 0;
        uint32_t a0v[6], a1v[6], nwv[6];
        int cpv[6], tsv[6];
        const uint32_t* const pa = &L.u.A[nb][col & ~1];
        const uint32_t* const pn = &L.NW[nb][E];
        const int* const pc = &L.tab.cp[nb];
        const int* const pt = &L.tab.ts[nb];
#pragma unroll
        for (int j = 0; j < 6; ++j) {                  // all LDS / table reads first, then pure ALU
            a0v[j] = pa[j * LPW];
            a1v[j] = pa[j * LPW + 1];
            nwv[j] = pn[j * (LPW / 2)];
            cpv[j] = pc[j];
            tsv[j] = pt[j];
        }
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const bool real = j > 0 || P != 0;
            int cs = (int)(nwv[j] & 0x3FFu) - 512;
            uint32_t cb1 = (nwv[j] >> 10) & 3u;                                    // controlledBy + 1
            const int cp = cpv[j], ts = tsv[j];
            const int pts0 = (int)(a0v[j] & 0xFFFFu), pts1 = (int)(a1v[j] & 0xFFFFu);
            const bool c0 = pts0 > 0, c1 = pts1 > 0;                               // ctr >= 1 (control >= 1)
            const uint32_t pid1 = c0 ? 1u : 2u;                                    // capturing player + 1
            const bool capture = real && play && (c0 != c1) && (abs(cs) < cp || pid1 != cb1);   // :729-732
            const int cs2 = cs + (pts0 - pts1);                                    // :748 (turn > 0 here): exactly one of the two is non-zero
            const bool neutralize = (cs ^ cs2) < 0;                                // :747-750: the sign bit changed
            const bool full = abs(cs2) >= cp;                                      // :763-765
            const uint32_t cb1n = neutralize ? 0u : (full ? pid1 : cb1);           // :766-767
            const int csn = full ? (c0 ? cp : -cp) : cs2;
            cs = capture ? csn : cs;
            cb1 = capture ? cb1n : cb1;
            if (capture) L.NW[nb + j][E] = (uint32_t)(cs + 512) | (cb1 << 10);
            const bool bcap = real && ts != -1 && cb1 != 0u && (int)cb1 != ts + 1;  // :299-304
            base_cap |= bcap ? 1 : 0;
            const int acs = abs(cs);
            const int pts = real ? acs + (acs == cp ? cp : 0) : 0;                 // :305-310
            part0 += (bcap && cb1 == 1u ? 1000 : 0) + (cs > 0 ? pts : 0);
            part1 += (bcap && cb1 == 2u ? 1000 : 0) + (cs < 0 ? pts : 0);
        }
    }
    // combine the pair: scores (server.py:291-317) and status (:321-328) are then known to both lanes
    const int opp_unit_score = xchg1(my_unit_score), opp_alive = xchg1(my_alive);
    part0 += xchg1(part0);
    part1 += xchg1(part1);
    base_cap |= xchg1(base_cap);
    int score[2];
    score[0] = part0 + (P ? opp_unit_score : my_unit_score);
    score[1] = part1 + (P ? my_unit_score : opp_unit_score);
    if (play) {
        if (turn >= max_turns) status = EVG_TIME_EXPIRED;                          // :321
        else if (my_alive + opp_alive == 0) status = EVG_ANNIHILATION;             // :324
        else if (base_cap) status = EVG_BASE_CAPTURE;                              // :327
        if (MT && mt_lane && turn % 10 == 0) (void)mt_randint(mt, 2u * NG + 1u);    // :337-338 focus draw: unobservable, but it consumes output
    }
    PHASE(8);

    // ---------------- reward / done / winner (everglades_env.py:37-61, evaluate.py:155-160)
    float rew0, rew1;
    int winner = EVG_WINNER_NONE;
    const bool done = status != 0;
    if (done) {
        winner = score[0] > score[1] ? EVG_WINNER_P0 : (score[1] > score[0] ? EVG_WINNER_P1 : EVG_WINNER_TIE);
        rew0 = score[0] > score[1] ? 1.f : 0.f;
        rew1 = score[1] > score[0] ? 1.f : (score[0] > score[1] ? -1.f : 0.f);
    } else {
        // scores[p] / 3700 (everglades_env.py:63-64) rounded to the float32 the reward tensor holds: the product with the rounded
        // reciprocal differs from the float64 quotient by an ulp of float64 at most, which never crosses a float32 rounding
        // boundary for an integer score below 2^22 (checked exhaustively: tests/test_abi_and_host.py)
        constexpr double kInvMaxScore = 1.0 / (double)EVG_MAX_SCORE;
        rew0 = (float)((double)score[0] * kInvMaxScore);
        rew1 = (float)((double)score[1] * kInvMaxScore);
    }
    {
        // the output pointers are fetched together (one scalar-load batch), not one by one inside the branches below
        float* const p_reward = io.reward;
        uint8_t* const p_done = io.done;
        int8_t* const p_winner = io.winner;
        int32_t* const p_scores = io.scores;
        uint8_t* const p_status = io.status;
        if (valid && !observe_only && P == 0) {
            reinterpret_cast<float2*>(p_reward)[e] = make_float2(rew0, rew1);
            p_done[e] = done ? 1 : 0;
            if (p_winner) p_winner[e] = (int8_t)winner;
            if (p_scores) reinterpret_cast<int2*>(p_scores)[e] = make_int2(score[0], score[1]);
            if (p_status) p_status[e] = (uint8_t)status;
        }
    }

    // ---------------- episode bookkeeping + auto-reset (each lane keeps its own player's return)
    bool do_reset = false;
    float* const p_fin_ret = S.fin_ret;
    int32_t* const p_fin_len = S.fin_len;
    int8_t* const p_fin_win = S.fin_win;
    const int auto_reset = S.auto_reset;
    if (play) {
        float r = ep_ret + (P ? rew1 : rew0);
        if (done) {
            p_fin_ret[(size_t)e * 2 + P] = r;
            if (P == 0) { p_fin_len[e] = turn; p_fin_win[e] = (int8_t)winner; }
            if (auto_reset) { do_reset = true; r = 0.f; }
        }
        ep_ret = r;
    }
    if (valid && !observe_only && (!MULTI || iter == nturns - 1)) S.ep_ret[(size_t)P * N + e] = ep_ret;
    {
        const bool fin = play && done && P == 0;
        const uint64_t mf = __ballot(fin);
        if (mf) {
            const uint64_t m0 = __ballot(fin && winner == EVG_WINNER_P0), m1 = __ballot(fin && winner == EVG_WINNER_P1);
            if (lane == 0) {
                const int nf = __popcll(mf), n0 = __popcll(m0), n1 = __popcll(m1);
                atomicAdd(&S.totals[0], (unsigned long long)nf);
                if (n0) atomicAdd(&S.totals[1], (unsigned long long)n0);
                if (n1) atomicAdd(&S.totals[2], (unsigned long long)n1);
                if (nf - n0 - n1) atomicAdd(&S.totals[3], (unsigned long long)(nf - n0 - n1));
            }
        }
    }
    if (do_reset) {
        // new episode: state of game_init (server.py:133-209)
        turn = 0; status = 0; episode += 1u;
        if (MT && mt_lane) { (void)mt_randint(mt, 2u * NG + 1u); (void)mt_randint(mt, 2u * NG + 1u); }   // game_init :205 and its game_end :338 (turn 0)
#pragma unroll
        for (int j = 0; j < 3; ++j) st[j] = 0;
#pragma unroll
        for (int k = 0; k < 12; ++k) { gw[k] = L.tab.init_grp[P * 12 + k]; cntv[k] = __popc(gw[k] & G_MASK_M); }
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int n = P ? 7 + j : 1 + j;
            if (n <= NN) L.NW[n][E] = L.tab.init_node[n];
        }
        // ... and the units this side lists per node (what the opponent's observation shows): the whole army stands on its base
#pragma unroll
        for (int n = 0; n < 12; ++n) L.u.A[n][lane] = 0;
        L.u.A[gw[0] & G_LOC_M][lane] = (uint32_t)NU << 16;
    }
    WAVE_SYNC();        // node words final; everybody is done adding to A
    PHASE(9);

    // ---------------- observation of this lane's player (board_state :382-455, player_state :457-501,
    // everglades_env.py:158-171), written as int16 straight into the wave's output image in LDS.
    // Board part by node ID (every LDS read has a constant offset): slot s of player 1's view shows node p1_node_map[s] (:437-439),
    // so node n is written to slot p1inv[n]; player 0's view is the identity.
    // (a rollout without an observation buffer -- evg_rollout_*(obs_out = NULL): the evaluation harness, which reads only the episode
    // results -- skips the image and the write-out altogether: a sixth of the turn's instructions and 55 % of its bytes)
    const bool want_obs = io.obs != nullptr;
    if (want_obs) {
    uint32_t nw_n[12], ou_n[12], res_n[12];
#pragma unroll
    for (int n = 1; n <= NN; ++n) {
        nw_n[n] = L.NW[n][E];
        ou_n[n] = L.u.A[n][col ^ 1];                    // high half: opposing units listed at the node, moving ones included (:446-449)
        res_n[n] = (uint32_t)L.tab.res[n];              // DEFENSE flag | OBSERVE flag << 16
    }
    const uint64_t slot_n = P ? L.tab.nib[10] : 0xBA9876543210ull;     // nibble n = board slot of node n in this player's view
    const uint64_t own_n = P ? p1nib : 0xBA9876543210ull;              // nibble n = node n in this player's numbering (:485-486)
    WAVE_SYNC();        // A is dead from here on: the union becomes the output image
    // one-seat form: the image is [env][105], built by the caller's lane only.  (Splitting that row between the two lanes of the pair -- half the
    // instructions -- and the 28 MB less to write change nothing measurable: 26.7 us either way, like evg_step with both rows; a single-turn launch
    // lasts as long as its slowest SIMD pair, not as long as its instruction or byte count: DESIGN.md section 6.)
    int16_t* orow = &L.u.O[(SEAT ? E : col) * OBS];
    if (envlane && (!SEAT || P == io.seat)) {
        orow[0] = (int16_t)turn;
#pragma unroll
        for (int n = 1; n <= NN; ++n) {
            int16_t* o = orow + 4 * (int)((slot_n >> (4 * n)) & 15u) - 3;
            o[0] = (int16_t)(res_n[n] & 0xFFFFu);                                  // :442
            o[1] = (int16_t)(res_n[n] >> 16);                                      // :443
            o[2] = (int16_t)((int)(nw_n[n] & 0x3FFu) - 512);                       // control sign not mirrored
            o[3] = (int16_t)(ou_n[n] >> 16);
        }
        static_assert(MODE_MOVING == 2 && MODE_READY == 1 && MODE_IDLE == 0, "bit 1 of the mode field is the `moving` flag");
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            const uint32_t w = gw[k];
            int16_t* o = orow + 45 + 5 * k;
            o[0] = (int16_t)((own_n >> (4 * (w & G_LOC_M))) & 15u);
            o[1] = (int16_t)((typ_n >> (4 * k)) & 15u);
            o[2] = (int16_t)((w & G_AVG_M) >> G_AVG_S);
            o[3] = (int16_t)((w >> (G_MODE_S + 1)) & 1u);
            o[4] = (int16_t)cntv[k];
        }
    }
    }   // want_obs
    PHASE(10);

    // ---------------- store state (coalesced)
    // (multi-turn form: the state lives on chip between turns and goes back to HBM after the launch's last turn)
    if (valid && !observe_only && (MULTI ? iter == nturns - 1 : (play || do_reset)) && !ABLATED(32u)) {
#pragma unroll
        for (int k = 0; k < 12; ++k) S.grp[(size_t)(P * 12 + k) * N + e] = gw[k];
#pragma unroll
        for (int j = 0; j < 3; ++j) S.stamp[(size_t)(P * 3 + j) * N + e] = st[j];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int n = (P ? 7 : 1) + 2 * j;          // word P*3+j holds nodes n (low half) and n+1 (high half; ID 12 does not exist)
            const uint32_t hi = n + 1 <= NN ? (L.NW[(n + 1) % 12][E] & 0xFFFFu) : 0u;
            S.node[(size_t)(P * 3 + j) * N + e] = (L.NW[n][E] & 0xFFFFu) | (hi << 16);
        }
        if (P == 0) {
            S.env[e] = (uint32_t)turn | ((uint32_t)status << 8);
            if (MULTI || do_reset) S.episode[e] = episode;
            if (MT) S.mt_pos[e] = mt.pos;
        }
    }
    if (MULTI && envlane) {                             // the next turn of this launch starts from these words
#pragma unroll
        for (int k = 0; k < 12; ++k) L.G[k][lane] = gw[k];
    }
    WAVE_SYNC();             // output image complete
    PHASE(11);

    // ---------------- observation write-out: the wave's 32 x 210 values are contiguous in the output; every lane
    // converts 16 bytes' worth per iteration (conflict-free LDS reads, fully coalesced 1 KiB stores per wave)
    if (want_obs && !ABLATED(16u)) {
        constexpr int EP = 16 / (int)sizeof(OT);          // elements per 16-byte vector
        constexpr int ROW_E = SEAT ? OBS : OBS2;           // values per env in the output
        constexpr int NVEC = (LPW / 2) * ROW_E / EP;
        static_assert((LPW / 2) * ROW_E % EP == 0, "the wave's image is a whole number of 16-byte vectors");
        const int limit = nvalid * ROW_E;
        OT* out = reinterpret_cast<OT*>(io.obs) + (size_t)e0 * ROW_E;
        auto unpack = [&](int elem0, int (&vals)[EP]) {
            if constexpr (EP == 4) {
                const uint2 raw = *reinterpret_cast<const uint2*>(&L.u.O[elem0]);
                vals[0] = (int)(int16_t)(raw.x & 0xFFFFu); vals[1] = (int)(int16_t)(raw.x >> 16);
                vals[2] = (int)(int16_t)(raw.y & 0xFFFFu); vals[3] = (int)(int16_t)(raw.y >> 16);
            } else if constexpr (EP == 2) {
                const uint32_t raw = *reinterpret_cast<const uint32_t*>(&L.u.O[elem0]);
                vals[0] = (int)(int16_t)(raw & 0xFFFFu); vals[1] = (int)(int16_t)(raw >> 16);
            } else {
                const uint4 raw = *reinterpret_cast<const uint4*>(&L.u.O[elem0]);
                const uint32_t r4[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) { vals[2 * j] = (int)(int16_t)(r4[j] & 0xFFFFu); vals[2 * j + 1] = (int)(int16_t)(r4[j] >> 16); }
            }
        };
        if (nvalid == EPW) {
            // full workgroup (wave-uniform): straight-line batches of 8 LDS reads, then 8 convert + store, no bounds checks
            constexpr int FULL = NVEC / WG, REM = NVEC % WG, BATCH = 8;
#pragma unroll
            for (int b0 = 0; b0 < FULL; b0 += BATCH) {
                int vals[BATCH][EP];
#pragma unroll
                for (int j = 0; j < BATCH; ++j)
                    if (b0 + j < FULL) unpack((lane + (b0 + j) * WG) * EP, vals[j]);
#pragma unroll
                for (int j = 0; j < BATCH; ++j)
                    if (b0 + j < FULL) store_obs_vec<OT>(out + (lane + (b0 + j) * WG) * EP, vals[j]);
            }
            if (REM && lane < REM) {
                int vals[EP];
                unpack((lane + FULL * WG) * EP, vals);
                store_obs_vec<OT>(out + (lane + FULL * WG) * EP, vals);
            }
        } else {
            for (int v = lane; v < NVEC; v += WG) {            // last, partial workgroup of the grid
                const int elem0 = v * EP;
                int vals[EP];
                unpack(elem0, vals);
                if (elem0 + EP <= limit) {
                    store_obs_vec<OT>(out + elem0, vals);
                } else {
#pragma unroll
                    for (int j = 0; j < EP; ++j)
                        if (elem0 + j < limit) out[elem0 + j] = (OT)vals[j];
                }
            }
        }
    }
    PHASE(12);

    // ---------------- health of envs that start a new episode: 1600 B each, written by the whole wave
    uint64_t rm = __ballot(do_reset && P == 0);        // do_reset is false on helper lanes
    while (rm) {
        const int l = __ffsll((unsigned long long)rm) - 1;
        rm &= rm - 1;
        double2* dst = reinterpret_cast<double2*>(S.health + (size_t)(e0 + (l >> 1)) * (2 * NU));
        for (int i = lane; i < NU; i += WG) dst[i] = make_double2(100.0, 100.0);
    }
    PHASE(13);
    if (MULTI) WAVE_SYNC();                             // next turn's LDS traffic stays behind this turn's (global accesses of one wave are issued in order)
    }   // turns

    if (!(CHUNKABLE && io.nsets > 0)) break;
    // publish the chunk to the XCD's other workgroups: every store of this wave (state words, health rows, outputs) has reached the L2
    // they share (s_waitcnt vmcnt(0); the vector L1 is write-through), then the flag.  No L2 write-back: the set never leaves this XCD.
    // WHAT THIS RELIES ON (it is weaker than an agent-scope release, which the memory model would ask for and which costs a buffer_wbl2 walk
    // per chunk: 57 % slower, measured): (1) every array of the handle is ordinary coarse-grained device memory (hipMalloc; evg_create
    // checks the pointer attributes), cached in the L2 of the XCD that touches it; (2) producer and consumer of a set run on the same XCD
    // (HW_REG_XCC_ID picks the queue), hence share that L2; (3) 128-byte lines that hold words of sets owned by DIFFERENT XCDs (byte-per-env
    // arrays, ragged N) are only ever merged through byte-masked write-backs of the dirty bytes -- no XCD writes back bytes it did not
    // write.  The diagnostic library can publish with a real release instead (ablate bit 7) and the parity tests run both.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    {
        bool publish = true;
#ifdef EVG_DIAG      // fault-path test (ablate bit 6): the first chunk of the launch's first set is never published, its successor must give up and flag the handle
        publish = !(ABLATED(64u) && wg_chunk == 0 && wg_set == 0);
#endif
        if (publish && threadIdx.x == 0) {
            uint32_t* const pflag = S.progress + (e0 >> 5);
            const uint32_t pval = io.progress_base + (uint32_t)wg_chunk + 1u;
#ifdef EVG_DIAG      // ablate bit 7: publish with an agent-scope release (L2 write-back), what the memory model asks for
            if (ABLATED(128u)) __hip_atomic_store(pflag, pval, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            else
#endif
            __hip_atomic_store(pflag, pval, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    WAVE_SYNC();                                         // the next unit's LDS traffic stays behind this one's
    }   // units

    STAMP_WAVE_END();
}

#include "evg_step4.inc"      // the four-lanes-per-env mapping: what persistent launches of SMALL batches run (launch_step)

#undef S
#undef io

// ---------------------------------------------------------------------------------------------
// per-env results of the last finished episode, packed for the path's one exchange: 16 bytes per env
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) evg_pack_results_kernel(DevState S, float4* __restrict__ out, long long* __restrict__ counts) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    const bool live = e < S.N;
    if (*S.fault != 0u) {            // a faulted handle (evg_check_fault) hands out no results: poisoned rows, winner -2 is no EVG_WINNER_* value
        const float nan = __int_as_float(0x7FC00000);
        if (live) out[e] = make_float4(nan, nan, -2.f, -1.f);
        if (counts && e < 4) counts[e] = -1;
        return;
    }
    int w = -3;
    if (live) {
        const float2 r = reinterpret_cast<const float2*>(S.fin_ret)[e];
        w = (int)S.fin_win[e];
        out[e] = make_float4(r.x, r.y, (float)w, (float)S.fin_len[e]);      // small integers are exact in float32
    }
    if (counts) {                    // win bookkeeping of these rows (evaluate.py:155-160): one atomic per wavefront and class
        const unsigned long long m0 = __ballot(w == EVG_WINNER_P0), m1 = __ballot(w == EVG_WINNER_P1), m2 = __ballot(w == EVG_WINNER_TIE), mu = __ballot(live && w < 0);
        if ((threadIdx.x & 63) == 0) {
            if (m0) atomicAdd(reinterpret_cast<unsigned long long*>(counts + 0), (unsigned long long)__popcll(m0));
            if (m1) atomicAdd(reinterpret_cast<unsigned long long*>(counts + 1), (unsigned long long)__popcll(m1));
            if (m2) atomicAdd(reinterpret_cast<unsigned long long*>(counts + 2), (unsigned long long)__popcll(m2));
            if (mu) atomicAdd(reinterpret_cast<unsigned long long*>(counts + 3), (unsigned long long)__popcll(mu));
        }
    }
}

// behind every chunked launch, on its stream: every XCD's queue must have handed out all its units (an XCD that ran no workgroup of the
// launch would leave its sets unplayed without anybody waiting for them)
struct ChunkUnits { uint32_t per_xcd[16]; };
__global__ void __launch_bounds__(WG) evg_chunk_verify_kernel(DevState S, ChunkUnits want) {
    const int x = threadIdx.x;
    if (x < S.nxcd && S.queue[x * 64] < want.per_xcd[x]) raise_fault(S.fault, S.fault_seen, 4u);
}

// ---------------------------------------------------------------------------------------------
// reset (everglades_env.py:75-116 -> server.py:133-209): masked, per env
// ---------------------------------------------------------------------------------------------
template <typename OT>
__global__ void __launch_bounds__(WG) evg_reset_kernel(DevState S, const uint8_t* mask, void* obs) {
    const int lane = threadIdx.x;
    const int e0 = blockIdx.x * WG;
    const int e = e0 + lane;
    const size_t N = (size_t)S.N;
    const DevTables* __restrict__ T = S.T;
    const bool sel = e < S.N && (mask == nullptr || mask[e] != 0);
    if (sel) {
#pragma unroll
        for (int k = 0; k < 24; ++k) S.grp[(size_t)k * N + e] = T->init_grp[k];
#pragma unroll
        for (int j = 0; j < 6; ++j) S.stamp[(size_t)j * N + e] = 0;
#pragma unroll
        for (int j = 0; j < 6; ++j) S.node[(size_t)j * N + e] = (T->init_node[2 * j + 1] & 0xFFFFu) | (2 * j + 2 <= NN ? (T->init_node[2 * j + 2] & 0xFFFFu) << 16 : 0u);
        S.env[e] = 0;
        S.episode[e] += 1u;                 // 0xFFFFFFFF at create -> episode 0 on the first reset
        S.ep_ret[e] = 0.f;
        S.ep_ret[N + e] = 0.f;
        if (S.mt_key) {                     // stock entropy: game_init's focus draw (:205) and the one of its game_end (:338, turn 0)
            MtGen mt{S.mt_key + e, N, S.mt_pos[e]};
            (void)mt_randint(mt, 2u * NG + 1u);
            (void)mt_randint(mt, 2u * NG + 1u);
            S.mt_pos[e] = mt.pos;
        }
    }
    uint64_t rm = __ballot(sel);
    while (rm) {
        const int l = __ffsll((unsigned long long)rm) - 1;
        rm &= rm - 1;
        double2* dst = reinterpret_cast<double2*>(S.health + (size_t)(e0 + l) * (2 * NU));
        for (int i = lane; i < NU; i += WG) dst[i] = make_double2(100.0, 100.0);
        if (obs) {
            OT* o = reinterpret_cast<OT*>(obs) + (size_t)(e0 + l) * (2 * OBS);
            for (int i = lane; i < 2 * OBS; i += WG) o[i] = (OT)T->reset_obs[i];
        }
    }
}

// np.random.seed(s) per env (stock-entropy mode): seeds[e], or seed + global env id when seeds == NULL
__global__ void __launch_bounds__(256) evg_mt_seed_kernel(DevState S, const uint32_t* seeds) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= S.N) return;
    MtGen mt{S.mt_key + e, (size_t)S.N, 0};
    mt_seed(mt, seeds ? seeds[e] : S.seed_lo + S.env_id_base + (uint32_t)e);
    S.mt_pos[e] = mt.pos;
}

// ---------------------------------------------------------------------------------------------
// random_actions stand-in (agents/State_Machine/random_actions.py:38-46): 7 distinct groups of 12,
// 7 distinct nodes of 1..11 per player, partial Fisher-Yates on nibble-packed permutations
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) evg_random_actions_kernel(DevState S, int32_t* actions, int seat) {
    // One thread per (env, player) draws its 7 rows; the block's 256 x 56 bytes are contiguous in the output, so they go through LDS and
    // leave as 16-byte-per-lane coalesced stores (a thread's own rows are 56 bytes apart from its neighbour's: direct stores would touch
    // 28 cache lines per instruction).
    __shared__ int2 rows_lds[256 * NA];
    // seat < 0: both seats, [N][2][7][2]; seat 0 / 1: that seat's rows only, [N][7][2] (one thread per env)
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int total = seat < 0 ? 2 * S.N : S.N;
    if (idx < total) {
        const int e = seat < 0 ? idx >> 1 : idx, p = seat < 0 ? idx & 1 : seat;
        const int turn = (int)(S.env[e] & 0xFFu);
        const uint32_t episode = S.episode[e];
        const uint32_t env_id = S.env_id_base + (uint32_t)e;
        int2 rows[NA];
        gen_random_rows(S.seed_lo, S.seed_hi, env_id, episode, turn, p, rows);
#pragma unroll
        for (int i = 0; i < NA; ++i) rows_lds[threadIdx.x * NA + i] = rows[i];
    }
    __syncthreads();
    const int first = blockIdx.x * 256;                                 // first (env, player) of this block
    const int nrows = min(256, total - first) * NA;                      // int2 rows this block holds (a multiple of 7)
    const uint4* src = reinterpret_cast<const uint4*>(rows_lds);
    uint4* dst = reinterpret_cast<uint4*>(reinterpret_cast<int2*>(actions) + (size_t)first * NA);     // 256 x 56 B per block: 16-byte aligned
    for (int v = threadIdx.x; 2 * v + 1 < nrows; v += 256) dst[v] = src[v];
    if ((nrows & 1) && threadIdx.x == 0) reinterpret_cast<int2*>(dst)[nrows - 1] = rows_lds[nrows - 1];  // odd tail of the last block
}

// standalone form of the scripted opponents: one thread per env, view = the observation tensor
template <typename OT>
__global__ void __launch_bounds__(256) evg_scripted_actions_kernel(DevState S, int policy, int player, const OT* obs, int32_t* actions) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= S.N) return;
    const size_t ai = (size_t)player * S.N + e;
    const ObsView<OT> v{obs + ((size_t)e * 2 + player) * OBS};
    int2 rows[NA];
    const AgentTabs atabs{S.T->maxnbr_nib, S.T->tar_to_1, S.T->tar_to_11, S.T};
    agent_rows(policy, v, atabs, S.seed_lo, S.seed_hi, S.env_id_base + (uint32_t)e, S.episode[e], player, true, ((S.env[e] >> 8) & 3u) == 0u, S.agent_cycle + ai, S.agent_swarm + ai,
               S.agent_dfs + ai, rows);
    int2* out = reinterpret_cast<int2*>(actions) + ((size_t)e * 2 + player) * NA;
#pragma unroll
    for (int i = 0; i < NA; ++i) out[i] = rows[i];
}

__global__ void evg_scripted_reset_kernel(DevState S) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 2 * S.N) return;
    S.agent_cycle[i] = 0x112u;             // first_turn = 1, group_num = 1, node_num = 2 (cycle_rush_turn25.py:49,56-57)
    S.agent_swarm[i] = 0xBA875421u;        // ATTACK_LIST = [1,2,4,5,7,8,10,11] (swarm_agent.py:29), nibble k = entry k
    S.agent_dfs[i] = 0u;                   // dfs_attack call counter
}

// ---------------------------------------------------------------------------------------------
// fog-of-war planes: the `valid_nodes` mask of board_state (server.py:402-425) and the per-node knowledge
// levels of build_knowledge_output (server.py:779-832) -- both computed by the reference and never applied to the
// observation; exposed here as optional planes, plus the opposing-group sightings of :845-907.  One thread per (env, player);
// real node order.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) evg_fog_kernel(DevState S, uint8_t* fog, uint8_t* know, int8_t* sight) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= 2 * S.N) return;
    const int e = idx >> 1, p = idx & 1;
    const size_t N = (size_t)S.N;
    const DevTables* __restrict__ T = S.T;
    uint32_t ctrl = 0, watch = 0;          // nodes controlled by p; of those, fully controlled OBSERVE nodes
    uint32_t seen_by_tower = 0;
#pragma unroll
    for (int n = 1; n <= NN; ++n) {
        const uint32_t nw = (S.node[(size_t)((n - 1) >> 1) * N + e] >> (16 * ((n - 1) & 1))) & 0xFFFFu;
        const int cb = (int)((nw >> 10) & 3u) - 1, cs = (int)(nw & 0x3FFu) - 512;
        const bool mine = cb == p, obs = (T->resource[n] & EVG_RES_OBSERVE) != 0;
        ctrl |= (mine ? 1u : 0u) << n;
        seen_by_tower |= (mine && obs) ? T->nbr_mask[n] : 0u;                                        // :415-418
        watch |= (mine && obs && abs(cs) == T->control_points[n]) ? T->nbr_mask[n] : 0u;              // server.py:801-804
    }
    uint32_t idle = 0, incoming = 0;       // nodes with a listed non-moving group of p; nodes a group of p moves to from a neighbour
#pragma unroll
    for (int k = 0; k < 12; ++k) {
        const uint32_t w = S.grp[(size_t)(p * 12 + k) * N + e];
        const bool alive = (w & G_MASK_M) != 0, moving = ((w & G_MODE_M) >> G_MODE_S) == MODE_MOVING;
        const uint32_t loc = w & G_LOC_M, dest = (w & G_DEST_M) >> G_DEST_S;
        idle |= ((alive && !moving) ? 1u : 0u) << loc;                                                // :421-424, :784-787
        incoming |= ((alive && moving && ((T->nbr_mask[dest] >> loc) & 1u)) ? 1u : 0u) << dest;       // :806-812
    }
    const uint32_t valid = ctrl | seen_by_tower | idle;                                               // board_state :402-425
    const uint32_t full = ctrl | idle, partial = watch | incoming;                                    // build_knowledge_output :816-829
#pragma unroll
    for (int n = 1; n <= NN; ++n) {
        if (fog) fog[(size_t)idx * NN + n - 1] = (uint8_t)((valid >> n) & 1u);
        if (know) know[(size_t)idx * NN + n - 1] = (uint8_t)(((full >> n) & 1u) ? 2u : (((partial >> n) & 1u) ? 1u : 0u));
    }
    if (sight) {
        // opposing-group sightings `opp_k` (server.py:845-907): a listed opposing group is reported at its node when that node's
        // knowledge is 1 or 2 and it is either not moving (key -1) or headed for a node of knowledge > 0 (key = that node's
        // index in the node list, ID - 1, as the reference writes it): {seen, node, key, unit count} per opposing group
        const uint32_t known = full | partial;
        char4* out = reinterpret_cast<char4*>(sight) + (size_t)idx * NG;
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            const uint32_t w = S.grp[(size_t)((1 - p) * 12 + k) * N + e];
            const bool alive = (w & G_MASK_M) != 0, moving = ((w & G_MODE_M) >> G_MODE_S) == MODE_MOVING;
            const uint32_t loc = w & G_LOC_M, dest = (w & G_DEST_M) >> G_DEST_S;
            const bool seen = alive && ((known >> loc) & 1u) && (!moving || ((known >> dest) & 1u));
            out[k] = seen ? make_char4(1, (signed char)loc, (signed char)(moving ? (int)dest - 1 : -1), (signed char)__popc(w & G_MASK_M)) : make_char4(0, 0, 0, 0);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// 'smart state' preprocessing of agents/Smart_State/DQNAgent.py:200-300: obs[N][2][105] of one player
// -> features float32 [N][12][59] (the reference builds float64 and the network casts to float32: computed in f64,
// rounded once).
// ---------------------------------------------------------------------------------------------
// COMPACT (evg_smart_state_compact): the same values without their redundancy.  34 of the 59 features of a swarm are the same for all 12 swarms of an env and
// 12 more are the constant one-hot swarm id, so the 12 x 59 matrix is determined by `shared` [34] and `swarm` [12][13] = {one-hot node (11), average
// health x alive / 1000, in transit}: 760 B per env instead of 2 832.  (A consumer's first linear layer over the 59 inputs is W[:, :34] shared + W[:, 34:47] swarm
// + W[:, 47 + s]: the same numbers from a quarter of the bytes.)  With so little to write the kernel is no longer bound by bytes but by the ~200 instructions a
// wavefront spends per env (23 us at 65 536 envs against 40 for the full matrix).  A mapping of eight envs per wavefront with per-lane element descriptors was built
// and measured: 38 us -- every element then costs ~18 instructions (two LDS reads, the float64 product, selects) where the table costs one read; dropped.
template <typename OT, bool COMPACT = false>
__global__ void __launch_bounds__(256) evg_smart_state_kernel(int N, int player, int seat_only, const OT* obs, float* out, float* out_swarm) {
    // One wavefront per env and pass.  The 105-value observation row is staged in LDS once; every distinct output value of the
    // env goes into a small per-wave table -- the 34 features all swarms share and the 12 per-swarm health features are
    // one IEEE f64 division each (46 lanes: one division sequence per env instead of one per output element), then the
    // 12 in-transit flags, the constants 0 and 1 and the 12 x 11 one-hot node entries -- and the env's 12 x 59 floats
    // are streamed out as table[idx] in 16-byte-per-lane (1 KiB per wavefront) coalesced stores; idx depends only on the position in
    // the row and is computed once per lane.  HBM-bound by the 2 832 B written per env.  Every wavefront works on its own env with its own
    // LDS rows, so the phases are separated by wavefront-scope fences, not block barriers: a barrier would also wait (vmcnt(0)) for the
    // stores of the pass to be acknowledged before the next env's row is even requested.
    constexpr int F = 59, WPB = 4, NV = NG * F / 4, NIT = (NV + 63) / 64;          // 177 float4 per env: three per lane
    static_assert(NG * F % 4 == 0, "an env's features are a whole number of float4");
    constexpr int T_HP = 34, T_MOV = 46, T_ZERO = 58, T_ONE = 59, T_HOT = 64, T_SIZE = T_HOT + NG * NN;
    __shared__ int   row[WPB][128];
    __shared__ float tab[WPB][T_SIZE + 4];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint8_t idx[NIT][4];
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int j = 4 * (lane + 64 * i) + c, sw = j / F, f = j - sw * F;
            idx[i][c] = (uint8_t)(f < 34 ? f : (f < 45 ? T_HOT + sw * NN + (f - 34) : (f == 45 ? T_HP + sw : (f == 46 ? T_MOV + sw : (f - 47 == sw ? T_ONE : T_ZERO)))));
        }
    }
    if (lane == 0) { tab[w][T_ZERO] = 0.f; tab[w][T_ONE] = 1.f; }
    const int passes = (N + (int)gridDim.x * WPB - 1) / ((int)gridDim.x * WPB);
    for (int ps = 0; ps < passes; ++ps) {
        const long long e = ((long long)ps * gridDim.x + blockIdx.x) * WPB + w;
        const bool live = e < N;
        const OT* o = obs + (seat_only ? (size_t)(live ? e : 0) : (size_t)(live ? e : 0) * 2 + player) * OBS;     // seat_only: obs is [N][105]
        row[w][lane] = live ? (int)o[lane] : 0;                              // every observation value is an integer
        row[w][lane + 64] = (live && lane + 64 < OBS) ? (int)o[lane + 64] : 0;
        WAVE_SYNC();
        const int* r = row[w];
        // float32(n / d) of the reference's float64 quotient, taken as float32(n * (1 / d)): the float64 product differs from the quotient by an ulp of float64
        // at most, which never crosses a float32 rounding boundary for any numerator these features can have (turn 0..255, control -511..511, units 0..200,
        // groups 0..12, health x alive 0..1 663: checked exhaustively in tests/test_abi_and_host.py) -- one multiplication instead of a division sequence per env
        double num = 0.0, rcp = 1.0;
        if (lane == 0) { num = (double)r[0]; rcp = 1.0 / 150.0; }                                    // :280
        else if (lane < 12) { num = (double)r[3 + 4 * (lane - 1)]; rcp = 1.0 / 100.0; }              // :282
        else if (lane < 23) { num = (double)r[4 + 4 * (lane - 12)]; rcp = 1.0 / 100.0; }             // :284
        else if (lane < 34) {                                                                       // :200-213, :286
            int cnt = 0;
#pragma unroll
            for (int k = 0; k < NG; ++k) cnt += (r[48 + 5 * k] == 0 && r[45 + 5 * k] - 1 == lane - 23) ? 1 : 0;
            num = (double)cnt; rcp = 1.0 / 12.0;
        } else if (lane < 46) {                                                                     // :294, swarm lane - 34
            const int sw = lane - 34;
            num = (double)(r[47 + 5 * sw] * r[49 + 5 * sw]); rcp = 1.0 / 1000.0;
        }
        if (lane < 46) tab[w][lane] = (float)(num * rcp);
        if (lane < NG) tab[w][T_MOV + lane] = (float)r[48 + 5 * lane];                                // :296
#pragma unroll
        for (int t = lane; t < NG * NN; t += 64) {                                                    // :288-292
            const int sw = t / NN, n = t - sw * NN;
            tab[w][T_HOT + t] = (r[45 + 5 * sw] == n + 1) ? 1.f : 0.f;
        }
        WAVE_SYNC();
        if constexpr (COMPACT) {
            constexpr int SF = 13, SV = NG * SF / 4;                                                  // 156 floats = 39 float4 per env
            static_assert(NG * SF % 4 == 0 && T_HP % 2 == 0, "whole vectors");
            if (live) {
                if (lane < T_HP / 2)                                                                  // shared [34]: 17 float2 (136 B per env: 8-byte aligned rows)
                    reinterpret_cast<float2*>(out + (size_t)e * T_HP)[lane] = make_float2(tab[w][2 * lane], tab[w][2 * lane + 1]);
                if (lane < SV) {
                    float v[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int j = 4 * lane + c, sw = j / SF, f = j - sw * SF;
                        v[c] = tab[w][f < NN ? T_HOT + sw * NN + f : (f == NN ? T_HP + sw : T_MOV + sw)];
                    }
                    reinterpret_cast<float4*>(out_swarm + (size_t)e * NG * SF)[lane] = make_float4(v[0], v[1], v[2], v[3]);
                }
            }
        } else
        if (live) {
            float4* dst = reinterpret_cast<float4*>(out + (size_t)e * NG * F);                        // 2 832 B per env: 16-byte aligned rows
#pragma unroll
            for (int i = 0; i < NIT; ++i)
                if (lane + 64 * i < NV)                                                               // one-hot swarm id (:298) = the two constants
                    dst[lane + 64 * i] = make_float4(tab[w][idx[i][0]], tab[w][idx[i][1]], tab[w][idx[i][2]], tab[w][idx[i][3]]);
        }
        WAVE_SYNC();
    }
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
template <int LPW, bool MULTI>
static int launch_step_variant(const DevState& S, const StepIO& io, int obs_dtype, hipStream_t s) {
    const int n = io.env_hi - io.env_lo;
    const int nsets = (n + LPW / 2 - 1) / (LPW / 2);
    const dim3 grid((unsigned)nsets), block(WG);
    const StepArgs args{S, io};
    switch (obs_dtype) {
        case EVG_OBS_F32: hipLaunchKernelGGL((evg_step_kernel<float, LPW, MULTI>), grid, block, 0, s, args); break;
        case EVG_OBS_F64: hipLaunchKernelGGL((evg_step_kernel<double, LPW, MULTI>), grid, block, 0, s, args); break;
        case EVG_OBS_I16: hipLaunchKernelGGL((evg_step_kernel<int16_t, LPW, MULTI>), grid, block, 0, s, args); break;
        default: return -1;
    }
    return (int)hipGetLastError();
}

// the chunked form of the persistent two-lane kernel: as many workgroups as the device holds, each taking units from its XCD's queue
static int launch_step_chunked(const DevState& S, const StepIO& io, int obs_dtype, hipStream_t s) {
    const dim3 grid((unsigned)io.grid_slots), block(WG);
    const StepArgs args{S, io};
    switch (obs_dtype) {
        case EVG_OBS_F32: hipLaunchKernelGGL((evg_step_kernel<float, WG, true, false, true>), grid, block, 0, s, args); break;
        case EVG_OBS_F64: hipLaunchKernelGGL((evg_step_kernel<double, WG, true, false, true>), grid, block, 0, s, args); break;
        case EVG_OBS_I16: hipLaunchKernelGGL((evg_step_kernel<int16_t, WG, true, false, true>), grid, block, 0, s, args); break;
        default: return -1;
    }
    return (int)hipGetLastError();
}

// Persistent launches of small and medium batches: the four-lanes-per-env mapping of evg_step4.inc -- 16 envs per wavefront,
// twice the wavefronts, 2 213 instead of 3 164 vector instructions per wave-turn.  While a SIMD holds few wavefronts, what counts is
// how long ONE wavefront needs for a turn, not the total instruction count.  Persistent form, us per turn on a whole MI355X,
// two-lane / four-lane (A/B of two builds on one box): 8 192 envs 11.0 / 8.3, 16 384 envs 11.8 / 8.3, 24 576 envs 12.1 / 9.6,
// 32 768 envs 12.1 / 10.4 (2 048 four-lane wavefronts = two per SIMD, the 160-VGPR build at 2 waves per SIMD); 40 960 envs
// 14.0 / 12.9, 49 152 envs 14.5 / 13.7 (up to 3 072 wavefronts = three per SIMD: the same kernel built for three, still
// unspilled).  Beyond that the four-lane grid is no longer resident at once (4 waves per SIMD need <= 128 VGPRs: 28 spilled) and
// the two-lane kernel wins.  Same state in HBM, same results (the persistent form of every small-batch test runs this kernel and
// is compared with the two-lane single-turn form and with the oracle).  The two thresholds are what the DEVICE holds
// (DeviceCaps::slots4_w2 / slots4_w3 wavefronts of 16 envs), not literals.
template <typename OT, bool MULTI, int WPE>
static void launch_step4_t(const DevState& S, const StepIO& io, hipStream_t s) {
    const StepArgs args{S, io};
    const dim3 grid((io.env_hi - io.env_lo + 15) / 16), block(WG);
    hipLaunchKernelGGL((evg_step4_kernel<OT, MULTI, WPE>), grid, block, 0, s, args);
}
template <bool MULTI, int WPE>
static int launch_step4(const DevState& S, const StepIO& io, int obs_dtype, hipStream_t s) {
    switch (obs_dtype) {
        case EVG_OBS_F32: launch_step4_t<float, MULTI, WPE>(S, io, s); break;
        case EVG_OBS_F64: launch_step4_t<double, MULTI, WPE>(S, io, s); break;
        case EVG_OBS_I16: launch_step4_t<int16_t, MULTI, WPE>(S, io, s); break;
        default: return -1;
    }
    return (int)hipGetLastError();
}

// What the device holds at once: compute units from hipDeviceProp_t, resident workgroups per CU from the occupancy of the kernels
// themselves.  On a whole MI355X: 256 CUs, 8 two-lane workgroups per CU (LDS: 8 x 20 208 B of 160 KiB; 2 waves per SIMD) = 2 048
// wavefronts = 65 536 envs; four-lane kernel 2 / 3 waves per SIMD = 2 048 / 3 072 wavefronts = 32 768 / 49 152 envs.
template <typename OT>
static int query_caps_t(DeviceCaps* c) {
    int b2m = 0, b2s = 0, b4w2 = 0, b4w3 = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&b2m, (const void*)evg_step_kernel<OT, WG, true>, WG, 0);
    if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&b2s, (const void*)evg_step_kernel<OT, WG, false>, WG, 0);
    if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&b4w2, (const void*)evg_step4_kernel<OT, true, 2>, WG, 0);
    if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&b4w3, (const void*)evg_step4_kernel<OT, true, 3>, WG, 0);
    if (e != hipSuccess) return (int)e;
    const int b2 = b2m < b2s ? b2m : b2s;
    if (b2 < 1 || b4w2 < 1 || b4w3 < 1) return (int)hipErrorLaunchOutOfResources;
    c->slots2 = c->cus * b2;
    // the four-lane builds aim at 2 / 3 waves per SIMD (amdgpu_waves_per_eu); the hardware may hold more of them, the plan does not use that
    c->slots4_w2 = c->cus * (b4w2 < 8 ? b4w2 : 8);
    c->slots4_w3 = c->cus * (b4w3 < 12 ? b4w3 : 12);
    return 0;
}

int query_device_caps(int device_id, int obs_dtype, DeviceCaps* caps) {
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, device_id);
    if (e != hipSuccess) return (int)e;
    caps->cus = prop.multiProcessorCount;
    caps->simds = 4 * caps->cus;
    caps->cache_bytes = (256ll << 20) * caps->cus / 256;      // MI355X: 256 MiB of Infinity Cache behind 256 CUs; a partition gets its share (evg_config::cache_mib overrides)
    switch (obs_dtype) {
        case EVG_OBS_F32: return query_caps_t<float>(caps);
        case EVG_OBS_F64: return query_caps_t<double>(caps);
        case EVG_OBS_I16: return query_caps_t<int16_t>(caps);
        default: return -1;
    }
}

// Which kernel plays which envs.  Single-turn launches (evg_step) and the stock-entropy mode: one launch of the two-lane kernel.
// Persistent form: a batch up to what the device holds at once runs the four-lane kernel while its grid is resident at two or
// three waves per SIMD and the two-lane kernel above that (the thresholds are DeviceCaps, not literals).  A LARGER batch used to be
// one launch of ceil(N / 32) workgroups, each playing all its turns: whole rounds of resident workgroups one after the other -- which
// is good, every round's working set (171 MB at 65 536 envs) stays inside the 256 MB Infinity Cache -- but the REMAINDER behind the
// last whole round ran alone at low occupancy for a whole launch (98 304 envs: 30.9 us per turn where 1.5 x 17.0 = 25.5 would be
// proportional; 65 536 + 4 480 envs cost 28.0 instead of 18.0).  Now:
//   * the whole rounds but the last: one plain launch, as before;
//   * the last whole round TOGETHER WITH the remainder, while the two fit the Infinity Cache (kChunkFootprintMax): one CHUNKED launch -- as many
//     workgroups as the device holds, each taking units (set of 32 envs) x (chunk of kChunkTurns turns) from its XCD's queue and handing
//     the set on through HBM (see the kernel's prologue): all slots stay busy until the queues run dry (98 304 envs: 26 us per turn,
//     70 016: 18.0);
//   * a larger remainder: its own plain launch behind the whole rounds (four-lane kernel up to 49 152 envs).  Chunking it too would
//     cycle the launch through nearly two rounds' worth of envs, more than the Infinity Cache holds, and every turn then runs ~30 %
//     slower (measured: 131 104 envs chunked 46 us per turn, plain 44; 262 144 chunked 90, plain 67).
#ifndef EVG_CHUNK_TURNS
#define EVG_CHUNK_TURNS 25                 // turns per chunk of a chunked launch (a build-time knob so that it can be re-measured with two builds: tools/scaling_lib.py)
#endif
[[maybe_unused]] constexpr int kChunkTurns = EVG_CHUNK_TURNS;
// A chunked launch cycles through ALL its envs every few chunks, so its working set -- state, and what THIS rollout writes: observations,
// orders, results; 2.74 KB per env with float32 observations -- must fit the device's share of the Infinity Cache (DeviceCaps::cache_bytes:
// 256 MiB on a whole MI355X, + 2 %): measured, us per turn, chunked / the alternative: 98 304 envs (270 MB) 25.8 / 27.5; 104 448 envs
// (287 MB) 32.5 / ~29.5.
long long rollout_bytes_per_env(const StepIO& io, int obs_dtype) {
    long long b = kStateBytesPerEnv + 13 /* fin_ret, fin_len, fin_win */ + 19 /* reward, done, winner, scores, status */;
    if (io.obs) b += OBS2 * (obs_dtype == EVG_OBS_F64 ? 8 : (obs_dtype == EVG_OBS_I16 ? 2 : 4));
    if (io.actions_out) b += 2 * NA * 2 * 4;
    if (io.gen_actions == 2) b += 24;                    // the scripted agents' objects
    return b;
}
LaunchPlan plan_step(const DevState& S, const StepIO& io, int obs_dtype, const DeviceCaps& caps) {
    LaunchPlan p;
    p.n = 1;
    p.piece[0] = LaunchPiece{0, 0, S.N, 0};
    p.piece[1] = LaunchPiece{0, 0, 0, 0};
    const bool multi = io.turns > 1;
    if (S.mt_key || !multi) return p;
    [[maybe_unused]] const long long cap2 = 32ll * caps.slots2, cap4_2 = 16ll * caps.slots4_w2, cap4_3 = 16ll * caps.slots4_w3;
    [[maybe_unused]] const long long N = S.N;
#ifdef EVG_DIAG
    if (io.lanes_per_wave == 2) {                  // experiment: the chunked form over the WHOLE batch, whatever its size (a working set beyond the Infinity Cache)
        if (N > cap2 && io.turns > kChunkTurns) p.piece[0].chunk_turns = kChunkTurns;
        return p;
    }
    if (io.lanes_per_wave != 0) return p;          // a forced kernel variant plays the whole batch in one plain launch
#endif
#ifdef EVG_STAMPS
    return p;                                       // the stamp buffer is indexed by workgroup: one plain launch
#else
    if (N <= cap4_2) { p.piece[0].four_lane_wpe = 2; return p; }
    if (N <= cap4_3) { p.piece[0].four_lane_wpe = 3; return p; }
    if (N <= cap2) return p;
    const long long full = N / cap2, rem = N - full * cap2;
    if (rem == 0) return p;                         // whole rounds only: one plain launch
    if ((cap2 + rem) * rollout_bytes_per_env(io, obs_dtype) <= caps.cache_bytes + caps.cache_bytes / 50 && io.turns > kChunkTurns) {
        p.n = 0;
        if (full >= 2) p.piece[p.n++] = LaunchPiece{0, 0, (int32_t)((full - 1) * cap2), 0};
        p.piece[p.n++] = LaunchPiece{0, (int32_t)((full - 1) * cap2), (int32_t)N, kChunkTurns};
        return p;
    }
    p.n = 2;
    p.piece[0] = LaunchPiece{0, 0, (int32_t)(full * cap2), 0};
    p.piece[1] = LaunchPiece{rem <= cap4_2 ? 2 : (rem <= cap4_3 ? 3 : 0), (int32_t)(full * cap2), (int32_t)N, 0};
    return p;
#endif
}

int launch_step(const DevState& S, const StepIO& io_in, int obs_dtype, const DeviceCaps& caps, void* stream) {
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    StepIO io = io_in;
    const bool multi = io.turns > 1;
    io.env_lo = 0; io.env_hi = S.N; io.flags = 0; io.nsets = 0; io.chunk_turns = 0; io.progress_base = 0; io.grid_slots = 0;
    const int grid2 = (S.N + WG / 2 - 1) / (WG / 2);
    if (!multi && grid2 <= caps.slots2) io.flags |= STEP_F_STAGGER;
    if (S.mt_key) {                       // stock-entropy mode: single-turn launches of the sequential-draw instantiation
        if (multi) return -1;
        const dim3 grid(grid2), block(WG);
        const StepArgs args{S, io};
        switch (obs_dtype) {
            case EVG_OBS_F32: hipLaunchKernelGGL((evg_step_kernel<float, WG, false, true>), grid, block, 0, s, args); break;
            case EVG_OBS_F64: hipLaunchKernelGGL((evg_step_kernel<double, WG, false, true>), grid, block, 0, s, args); break;
            case EVG_OBS_I16: hipLaunchKernelGGL((evg_step_kernel<int16_t, WG, false, true>), grid, block, 0, s, args); break;
            default: return -1;
        }
        return (int)hipGetLastError();
    }
#ifdef EVG_DIAG
    if (io.lanes_per_wave == 64) return multi ? launch_step_variant<64, true>(S, io, obs_dtype, s) : launch_step_variant<64, false>(S, io, obs_dtype, s);   // the two-lane kernel at any size
    if (io.lanes_per_wave == 4) return multi ? launch_step4<true, 4>(S, io, obs_dtype, s) : launch_step4<false, 4>(S, io, obs_dtype, s);
    if (io.lanes_per_wave == 32) return multi ? launch_step_variant<32, true>(S, io, obs_dtype, s) : launch_step_variant<32, false>(S, io, obs_dtype, s);
#endif
    if (!multi) return launch_step_variant<64, false>(S, io, obs_dtype, s);
    const LaunchPlan plan = plan_step(S, io, obs_dtype, caps);
    for (int i = 0; i < plan.n; ++i) {
        const LaunchPiece& pc = plan.piece[i];
        io.env_lo = pc.env_lo; io.env_hi = pc.env_hi;
        io.flags = (pc.four_lane_wpe && (pc.env_hi - pc.env_lo + 15) / 16 > caps.simds) ? STEP_F_SHARED_SIMD : 0;
        io.nsets = 0; io.chunk_turns = 0;
        if (pc.chunk_turns > 0) {
            io.nsets = (pc.env_hi - pc.env_lo + WG / 2 - 1) / (WG / 2);
            io.chunk_turns = pc.chunk_turns;
            io.progress_base = 0;
            io.grid_slots = io.nsets < caps.slots2 ? io.nsets : caps.slots2;
            // every XCD's queue starts at unit 0 and every set's progress flag at "no chunk finished": ONE memset on the stream (the two arrays are one
            // allocation, queue first).  Nothing of a chunked launch lives on the host, so a captured launch can be replayed (hipGraph).
            const hipError_t me = hipMemsetAsync(S.queue, 0, (1024 + ((size_t)S.N + 31) / 32 + 1) * sizeof(uint32_t), s);
            if (me != hipSuccess) return (int)me;
        }
        int rc;
        if (pc.four_lane_wpe == 2) rc = launch_step4<true, 2>(S, io, obs_dtype, s);
        else if (pc.four_lane_wpe == 3) rc = launch_step4<true, 3>(S, io, obs_dtype, s);
        else if (io.nsets > 0) rc = launch_step_chunked(S, io, obs_dtype, s);
        else rc = launch_step_variant<64, true>(S, io, obs_dtype, s);
        if (rc) return rc;
        if (io.nsets > 0) {
            // ... and behind it, on the same stream: every XCD's queue handed out all its units (sets x chunks), or the handle is flagged
            ChunkUnits want;
            const uint32_t nchunks = (uint32_t)((io.turns + io.chunk_turns - 1) / io.chunk_turns);
            for (int x = 0; x < 16; ++x) want.per_xcd[x] = x < S.nxcd ? (uint32_t)((io.nsets - x + S.nxcd - 1) / S.nxcd) * nchunks : 0u;
            hipLaunchKernelGGL(evg_chunk_verify_kernel, dim3(1), dim3(WG), 0, s, S, want);
            rc = (int)hipGetLastError();
            if (rc) return rc;
        }
    }
    return 0;
}

// evg_step_vs_policy / evg_observe_seat: one launch of the one-seat instantiation of the single-turn two-lane kernel
int launch_step_seat(const DevState& S, const StepIO& io_in, int obs_dtype, const DeviceCaps& caps, void* stream) {
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (S.mt_key) return -1;
    StepIO io = io_in;
    io.turns = 1; io.env_lo = 0; io.env_hi = S.N; io.flags = 0; io.nsets = 0; io.chunk_turns = 0; io.progress_base = 0; io.grid_slots = 0;
    const int grid2 = (S.N + WG / 2 - 1) / (WG / 2);
    if (grid2 <= caps.slots2) io.flags |= STEP_F_STAGGER;
    const dim3 grid(grid2), block(WG);
    const StepArgs args{S, io};
    switch (obs_dtype) {
        case EVG_OBS_F32: hipLaunchKernelGGL((evg_step_kernel<float, WG, false, false, false, true>), grid, block, 0, s, args); break;
        case EVG_OBS_F64: hipLaunchKernelGGL((evg_step_kernel<double, WG, false, false, false, true>), grid, block, 0, s, args); break;
        case EVG_OBS_I16: hipLaunchKernelGGL((evg_step_kernel<int16_t, WG, false, false, false, true>), grid, block, 0, s, args); break;
        default: return -1;
    }
    return (int)hipGetLastError();
}

// which XCC ids does this device have?  (evg_create: 1 024 one-wave workgroups report where they ran)
__global__ void __launch_bounds__(WG) evg_xcd_probe_kernel(uint32_t* out) {
    if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_getreg(63508) & 15u;      // HW_REG_XCC_ID
}
int launch_xcd_probe(uint32_t* out, void* stream) {
    hipLaunchKernelGGL(evg_xcd_probe_kernel, dim3(1024), dim3(WG), 0, reinterpret_cast<hipStream_t>(stream), out);
    return (int)hipGetLastError();
}

int launch_reset(const DevState& S, const uint8_t* mask, void* obs, int obs_dtype, void* stream) {
    const dim3 grid((S.N + WG - 1) / WG), block(WG);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    switch (obs_dtype) {
        case EVG_OBS_F32: hipLaunchKernelGGL(evg_reset_kernel<float>, grid, block, 0, s, S, mask, obs); break;
        case EVG_OBS_F64: hipLaunchKernelGGL(evg_reset_kernel<double>, grid, block, 0, s, S, mask, obs); break;
        case EVG_OBS_I16: hipLaunchKernelGGL(evg_reset_kernel<int16_t>, grid, block, 0, s, S, mask, obs); break;
        default: return -1;
    }
    return (int)hipGetLastError();
}

int launch_scripted_actions(const DevState& S, int policy, int player, const void* obs, int32_t* actions, int obs_dtype, void* stream) {
    const dim3 grid((S.N + 255) / 256), block(256);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    switch (obs_dtype) {
        case EVG_OBS_F32: hipLaunchKernelGGL(evg_scripted_actions_kernel<float>, grid, block, 0, s, S, policy, player, (const float*)obs, actions); break;
        case EVG_OBS_F64: hipLaunchKernelGGL(evg_scripted_actions_kernel<double>, grid, block, 0, s, S, policy, player, (const double*)obs, actions); break;
        case EVG_OBS_I16: hipLaunchKernelGGL(evg_scripted_actions_kernel<int16_t>, grid, block, 0, s, S, policy, player, (const int16_t*)obs, actions); break;
        default: return -1;
    }
    return (int)hipGetLastError();
}

int launch_mt_seed(const DevState& S, const uint32_t* seeds_dev, void* stream) {
    hipLaunchKernelGGL(evg_mt_seed_kernel, dim3((S.N + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), S, seeds_dev);
    return (int)hipGetLastError();
}

int launch_scripted_reset(const DevState& S, void* stream) {
    hipLaunchKernelGGL(evg_scripted_reset_kernel, dim3((2 * S.N + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), S);
    return (int)hipGetLastError();
}

int launch_smart_state(const DevState& S, int player, const void* obs, int seat_only, float* out, float* out_swarm /* non-NULL: compact form, out = shared [N][34] */, int obs_dtype, void* stream) {
    const int blocks = (S.N + 3) / 4;                       // one wavefront per env and pass; 8 blocks per CU resident, further envs in passes
    const dim3 grid((unsigned)(blocks < 2048 ? blocks : 2048)), block(256);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    switch (obs_dtype) {
        case EVG_OBS_F32: if (out_swarm) hipLaunchKernelGGL((evg_smart_state_kernel<float, true>), grid, block, 0, s, S.N, player, seat_only, (const float*)obs, out, out_swarm);
                          else hipLaunchKernelGGL((evg_smart_state_kernel<float, false>), grid, block, 0, s, S.N, player, seat_only, (const float*)obs, out, out_swarm);
                          break;
        case EVG_OBS_F64: if (out_swarm) hipLaunchKernelGGL((evg_smart_state_kernel<double, true>), grid, block, 0, s, S.N, player, seat_only, (const double*)obs, out, out_swarm);
                          else hipLaunchKernelGGL((evg_smart_state_kernel<double, false>), grid, block, 0, s, S.N, player, seat_only, (const double*)obs, out, out_swarm);
                          break;
        case EVG_OBS_I16: if (out_swarm) hipLaunchKernelGGL((evg_smart_state_kernel<int16_t, true>), grid, block, 0, s, S.N, player, seat_only, (const int16_t*)obs, out, out_swarm);
                          else hipLaunchKernelGGL((evg_smart_state_kernel<int16_t, false>), grid, block, 0, s, S.N, player, seat_only, (const int16_t*)obs, out, out_swarm);
                          break;
        default: return -1;
    }
    return (int)hipGetLastError();
}

int launch_fog(const DevState& S, uint8_t* fog, uint8_t* know, int8_t* sight, void* stream) {
    hipLaunchKernelGGL(evg_fog_kernel, dim3((2 * S.N + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), S, fog, know, sight);
    return (int)hipGetLastError();
}

int launch_pack_results(const DevState& S, float* out, long long* counts, void* stream) {
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (counts) {
        const hipError_t me = hipMemsetAsync(counts, 0, 4 * sizeof(long long), s);
        if (me != hipSuccess) return (int)me;
    }
    hipLaunchKernelGGL(evg_pack_results_kernel, dim3((S.N + 255) / 256), dim3(256), 0, s, S, reinterpret_cast<float4*>(out), counts);
    return (int)hipGetLastError();
}

int launch_random_actions(const DevState& S, int32_t* actions, int seat, void* stream) {
    // (a latency chain of two Philox blocks per thread; 128-thread blocks for the half-sized one-seat form are no faster: 5.4 against 5.2 us)
    const dim3 grid(((seat < 0 ? 2 : 1) * S.N + 255) / 256), block(256);
    hipLaunchKernelGGL(evg_random_actions_kernel, grid, block, 0, reinterpret_cast<hipStream_t>(stream), S, actions, seat);
    return (int)hipGetLastError();
}

}  // namespace evg
