// evg_device.h -- data layout shared by the HIP kernels (evg_kernels.hip) and the C-ABI host
// side (evg_abi.hip).  gfx950 only.
//
// Persistent state, struct-of-arrays over N envs (see DESIGN.md "Data layout in HBM"):
//   grp    u32 [24][N]   one packed word per group (player p, group k -> row p*12+k), env fastest:
//                         bits  0-3  location (node ID 1..11)
//                         bits  4-7  travel_destination (0 = none, the reference's -1)
//                         bits  8-10 distance_remaining
//                         bits 11-12 mode: 0 idle, 1 ready (ordered this turn), 2 moving
//                         bits 13-24 alive mask of the group's (up to 12) unit slots
//                         bits 25-31 cached int(avg health) of player_state (server.py:491)
//                         destroyed <=> alive mask == 0 (server.py:623-625)
//   stamp  u32 [6][N]    arrival turn of each group, 4 x u8 per word (group row r -> word r>>2);
//                         node-list order of the reference == (stamp, gid) order
//   node   u32 [6][N]    two node words per u32 (word j: node ID 2j+1 in the low half, 2j+2 in the high half), each:
//                         bits 0-9 controlState+512, bits 10-11 controlledBy+1
//   env    u32 [N]       bits 0-7 current_turn, bits 8-9 status
//   episode u32 [N]
//   health f64 [N][200]  env-major: unitHealth of player p group k at [p*100 + 8k ...]; a group's
//                         row is one 64-byte (96 for group 11) aligned segment, touched only by combat
//   ep_ret f32 [2][N], fin_ret f32 [N][2], fin_len i32 [N], fin_win i8 [N], totals u64[4]
//   mt_key u32 [624][N], mt_pos u32 [N]   per-env MT19937 of the stock-entropy mode (rng_mode 1) only
#pragma once
#include <stdint.h>
#include "../../include/evg.h"
#include "evg_rng.h"

namespace evg {

constexpr int NP = 2, NG = 12, NN = 11, NU = 100, NA = 7, OBS = 105;
constexpr int WG = 64;                    // one wavefront per workgroup, one env per lane
constexpr int OBS2 = 2 * OBS;             // both players' observations of one env, contiguous in the output

// group word fields
constexpr uint32_t G_LOC_M = 0xFu, G_DEST_S = 4, G_DIST_S = 8, G_MODE_S = 11, G_MASK_S = 13, G_AVG_S = 25;
constexpr uint32_t G_DEST_M = 0xFu << G_DEST_S, G_DIST_M = 0x7u << G_DIST_S, G_MODE_M = 0x3u << G_MODE_S;
constexpr uint32_t G_MASK_M = 0xFFFu << G_MASK_S, G_AVG_M = 0x7Fu << G_AVG_S;
constexpr uint32_t MODE_IDLE = 0, MODE_READY = 1, MODE_MOVING = 2;

// The tables the step kernel indexes per lane, laid out on the host exactly as the kernel keeps them in LDS: one blob,
// copied by every wavefront with two 16-byte-per-lane loads (13 cache lines shared by the whole grid) instead of a
// dozen separate table loads.
struct __attribute__((aligned(16))) LdsTables {
    uint64_t adj[12];            // nibble j of row i = distance i -> j (0 = not connected)
    double   den[48], rcp[48];   // DevTables::den_tab / rcp_tab flattened [type][node]
    int32_t  cp[12], ts[12], res[12];       // control points, team start by node ID; res = the two board flags as the observation
                                            // shows them: (DEFENSE ? 1 : 0) | (OBSERVE ? 1 : 0) << 16 (server.py:442-443)
    uint32_t init_grp[24], init_node[12];   // state right after game_init (auto-reset inside the kernel)
    uint64_t nib[14];            // p1map, speed[2], control[2], cost[2], type[2] nibble tables; [9] = max_turns | damage_nib << 8 | fast_div << 24;
                                 // [10] = p1inv (nibble n = slot of player 1's board view that shows node n); [11] = maxnbr_nib,
                                 // [12] = tar_to_1, [13] = tar_to_11 (the scripted bots' routing tables, see DevTables)
};
static_assert(sizeof(LdsTables) % 16 == 0, "the kernel copies the blob in 16-byte pieces");

struct DevTables {
    LdsTables lds;               // first member: 16-byte aligned in the device allocation
    uint64_t adj_row[12];        // nibble j of row i = distance i -> j (0 = not connected)
    double   defense[12];
    int32_t  control_points[12];
    int32_t  team_start[12];
    int32_t  resource[12];
    uint64_t p1map_nib;          // nibble i = p1_node_map[i]
    uint64_t type_nib[2];        // nibble k = unit type of group k
    uint64_t speed_nib[2], control_nib[2], cost_nib[2];   // nibble k = speed / control / cost of group k's unit type
    uint64_t tar_to_1, tar_to_11;   // cycle_target_node*.py routing rows: nibble c = next hop from node c towards node 1 / 11 (15 = the bots' -1)
    int32_t  dfs_mu, dfs_lambda;    // dfs_attack.py order sequence: transient length and period
    uint64_t dfs_rows[192];         // row r, byte i = (group | node << 4) of order i
    uint32_t nbr_mask[12];       // bit m of entry n: node m is connected to node n
    uint64_t maxnbr_nib;         // nibble n = highest-numbered neighbour of node n (SwarmAgent's next hop)
    uint64_t p1inv_nib;          // nibble n = slot of p1's board view that shows node n (inverse of p1_node_map)
    uint32_t damage_nib;         // nibble t = damage of unit type t
    uint32_t armor_byte;         // byte t   = health ("armor") of unit type t
    double   den_tab[4][12];     // combat denominator of a target of unit type t: [t][0] = armor, [t][n] = armor + defense of node n (:592-601)
    double   rcp_tab[4][12];     // its correctly rounded reciprocal (host division)
    int32_t  fast_div;           // 1: (10 * D) / den == fma(fma(-den, q0, a), rcp, q0) with q0 = a * rcp for EVERY damage sum D in 0..255 and
                                 // every table entry (checked exhaustively at evg_create) -- the kernel then divides with 3 flops
    int32_t  unit_speed[4], unit_control[4], unit_cost[4];
    int32_t  group_type[2][12];
    int32_t  max_turns;
    uint32_t init_grp[24];       // state right after game_init (server.py:133-209)
    uint32_t init_node[12];
    int16_t  reset_obs[2 * OBS];     // and the observation itself
};

struct DevState {
    int32_t   N;
    int32_t   auto_reset;
    uint32_t* grp;
    uint32_t* stamp;
    uint32_t* node;
    uint32_t* env;
    uint32_t* episode;
    double*   health;
    float*    ep_ret;
    float*    fin_ret;
    int32_t*  fin_len;
    int8_t*   fin_win;
    unsigned long long* totals;
    uint32_t* agent_cycle;       // [2][N] scripted-agent state: first_turn << 8 | group_num << 4 | node_num
    uint32_t* agent_swarm;       // [2][N] SwarmAgent attack list, 8 nibbles
    uint32_t* agent_dfs;         // [2][N] dfs_attack call counter
    uint32_t* mt_key;            // [624][N] stock-entropy mode only (evg_mt.h), else NULL
    uint32_t* mt_pos;            // [N]
    uint32_t  seed_lo, seed_hi, env_id_base;
    PhiloxKeys keys;             // the ten round keys of every Philox block of this handle (evg_rng.h), from the seed
    const DevTables* T;
    uint32_t* progress;          // [ceil(N / 32)] chunked persistent launches: set s of 32 envs has finished chunk c of the launch <=> progress[s] == c + 1
    uint32_t* queue;             // [1024] chunked launches: next unit of each XCD's queue, one counter per 256-byte line.  queue and progress are ONE
                                 // allocation (queue first), zeroed by one memset on the stream before every chunked launch
    uint32_t* handoff;           // [2][N] chunked launches: what a lane handed on at the end of its chunk -- a checksum over (chunk number, every state word it
                                 // stored); the lane that takes the set's next chunk recomputes it over the words it LOADED (fault bit 3 on a mismatch)
    uint32_t* fault;             // [1] bit 0: a wave gave up waiting for a predecessor chunk; bit 1: a workgroup ran on an XCD the create-time probe did
                                 // not see;
                                 // bit 2: a queue of a chunked launch was not drained (evg_chunk_verify_kernel, on the stream behind every chunked launch);
                                 // bit 3: a chunk hand-over delivered state words that are not the ones its producer stored (checksum mismatch: stale data).
                                 // Never expected; sticky; read by the pack kernel (poisoned rows) and by every host path on which results leave the handle
    uint32_t* fault_seen;        // [1] host-mapped mirror: set to 1 (plain store) together with any bit of `fault`, so that the host checks cost no device copy
    uint64_t  xcd_rank;          // nibble x = rank of XCC id x among the XCDs of this device (15 = not seen by the probe at evg_create)
    int32_t   nxcd;              // number of XCDs (8 on a whole MI355X)
};

struct StepIO {
    const int32_t* actions;
    void*     obs;
    float*    reward;
    uint8_t*  done;
    int8_t*   winner;
    int32_t*  scores;
    uint8_t*  status;
    int32_t   observe_only;      // 1: only (re)build observations from the current state
    int32_t   gen_actions;       // 1: draw both players' orders in the kernel (random_actions policy) instead of reading `actions`;
                                 // 2: orders of the on-device scripted agents policy0 / policy1 (fused evg_rollout_policies)
    int32_t   policy0, policy1;
    int32_t*  actions_out;       //    ... and store them here ([N][2][7][2], may be NULL)
    int32_t   turns;             // consecutive turns per launch (> 1 only with gen_actions: the fused rollout driver)
    int32_t   env_lo, env_hi;    // this launch plays envs [env_lo, env_hi) of the handle (workgroup b: envs env_lo + b * envs-per-wave ...);
                                 // set by launch_step, which may split a batch into several launches (LaunchPlan)
    int32_t   flags;             // STEP_F_*: set by launch_step from the device's capacity (DeviceCaps), not from literals
    int32_t   seat;              // SEAT instantiation (evg_step_vs_policy / evg_observe_seat): the caller's seat; its 7 rows come from `actions` ([N][7][2],
                                 // or rows [:, seat] of [N][2][7][2] when actions_both != 0), the other seat's from policy0 / policy1 (gen_actions == 2),
                                 // and obs is [N][105]
    int32_t   actions_both;
    float*    feat_shared;       // SEAT instantiation only (evg_step_vs_policy_smart): non-NULL = also write the Smart_State features of the caller's seat, compact
    float*    feat_swarm;        // form -- shared [N][34], swarm [N][12][13] (evg_smart_state_compact's outputs) -- straight from the observation image in LDS
    int32_t   nsets;             // > 0: CHUNKED persistent launch of the two-lane kernel (batches beyond what the device holds at once): workgroup u
                                 // plays chunk u / nsets (chunk_turns consecutive turns, the last one what is left of `turns`) of env set u % nsets
    int32_t   chunk_turns;
    int32_t   grid_slots;        // workgroups of a chunked launch (what the device holds at once)
    uint32_t  progress_base;     // value of DevState::progress[set] that means "no chunk of this launch finished yet" (0: the flags are zeroed before
                                 // every launch)
#ifdef EVG_DIAG                  // diagnostic libraries only (libevg_diag.so, libevg_stamps.so)
    int32_t   lanes_per_wave;    // 64: 32 envs per wavefront; 32: 16 envs per wavefront + 32 helper lanes
    uint32_t  ablate;            // bit0 orders, bit1 combat, bit2 movement, bit4 obs write-out, bit5 state store
    unsigned long long* stamps;  // EVG_STAMPS only, else NULL
#endif
};

constexpr int32_t STEP_F_STAGGER = 1;      // single-turn launch whose whole grid is resident at once: the wave in hardware slot 1 of a SIMD starts late
constexpr int32_t STEP_F_SHARED_SIMD = 2;  // four-lane kernel: more wavefronts than SIMDs, i.e. waves share a SIMD (issue priorities by hardware slot)

// What the device holds at once, measured at evg_create from hipDeviceProp_t and the occupancy of the kernels themselves
// (hipOccupancyMaxActiveBlocksPerMultiprocessor): nothing in the launch logic assumes a 256-CU part.
struct DeviceCaps {
    int32_t cus;                 // compute units of the (possibly partitioned) device
    int32_t simds;               // cus x 4
    int32_t slots2;              // resident workgroups (= wavefronts, 32 envs each) of the two-lane step kernel: cus x 8 on MI355X (LDS-bound)
    int32_t slots4_w2, slots4_w3;   // resident wavefronts (16 envs each) of the four-lane kernel built for 2 / 3 waves per SIMD
    int64_t cache_bytes;         // what a chunked launch may cycle through: this device's share of the memory-side (Infinity) cache -- 256 MiB x cus / 256 on
                                 // MI355X (HIP has no query for it), or evg_config::cache_mib
};

// bytes of persistent state per env (evg_state_bytes_per_env): grp, stamp, node, env, episode, health, ep_ret
constexpr int kStateBytesPerEnv = 24 * 4 + 6 * 4 + 6 * 4 + 4 + 4 + 2 * NU * 8 + 2 * 4;

// One launch of a plan: which kernel plays which envs.
// four_lane_wpe: 0 = two-lane kernel, 2 / 3 = four-lane kernel built for that many waves per SIMD; chunk_turns > 0: chunked dispatch (two-lane kernel only)
struct LaunchPiece { int32_t four_lane_wpe; int32_t env_lo, env_hi; int32_t chunk_turns; };
struct LaunchPlan { int32_t n; LaunchPiece piece[2]; };

// launchers (evg_kernels.hip)
int query_device_caps(int device_id, int obs_dtype, DeviceCaps* caps);
int launch_xcd_probe(uint32_t* out /* device [1024] */, void* stream);
LaunchPlan plan_step(const DevState& S, const StepIO& io, int obs_dtype, const DeviceCaps& caps);
long long rollout_bytes_per_env(const StepIO& io, int obs_dtype);
int launch_step(const DevState& S, const StepIO& io, int obs_dtype, const DeviceCaps& caps, void* stream);
int launch_step_seat(const DevState& S, const StepIO& io, int obs_dtype, const DeviceCaps& caps, void* stream);
int launch_reset(const DevState& S, const uint8_t* mask, void* obs, int obs_dtype, void* stream);
int launch_random_actions(const DevState& S, int32_t* actions, int seat /* -1: both, [N][2][7][2]; 0 / 1: that seat only, [N][7][2] */, void* stream);
int launch_scripted_actions(const DevState& S, int policy, int player, const void* obs, int32_t* actions, int obs_dtype, void* stream);
int launch_scripted_reset(const DevState& S, void* stream);
int launch_fog(const DevState& S, uint8_t* fog, uint8_t* know, int8_t* sight, void* stream);
int launch_mt_seed(const DevState& S, const uint32_t* seeds_dev, void* stream);
int launch_smart_state(const DevState& S, int player, const void* obs, int seat_only /* obs is [N][105] */, float* out,
                       float* out_swarm /* non-NULL: compact form, out = shared [N][34], out_swarm [N][12][13] */, int obs_dtype, void* stream);
struct SmartExplore {            // evg_smart_get_action: the epsilon branch of DQNAgent.get_action on top of evg_smart_actions
    int seat;                    // the agent's player number (keys its draws)
    float eps;                   // epsilon of every env ...
    const float* eps_env;        // ... unless this device array [N] is given
    uint8_t* explored;           // device [N] or NULL
};
int launch_smart_actions(const DevState& S, int player, const void* obs, int seat_only /* obs is [N][105] */, const float* q, int32_t* actions,
                         int32_t* directions,
                         int obs_dtype, void* stream, const SmartExplore* explore = nullptr /* NULL: get_best_actions only */);
int launch_pack_results(const DevState& S, float* out, long long* counts /* device [4] or NULL */, void* stream);

}  // namespace evg
