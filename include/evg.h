/*
 * evg.h -- C-ABI of the MI355X-native batched Everglades environment (libevg.so).
 *
 * This is the drop-in boundary for ONE hot path of jlehett/everglades-ai-wargame: the
 * everglades-server turn loop plus gym_everglades step()/reset(), vectorised over N
 * independent two-player DemoMap games whose state lives struct-of-arrays in HBM.
 * The reference has no FFI of its own -- its Gym class *is* the boundary
 * (gym-everglades/gym_everglades/envs/everglades_env.py:13-173).  Each entry point below
 * cites the reference code it replaces; the Python binding a maintainer would add is in
 * INTEGRATION.md and shipped as everglades-ai-wargame_amd/_lib.py.
 *
 * Conventions
 *   - plain C, no torch types; every pointer marked "device" is a HIP device pointer owned by the
 *     caller (e.g. torch.Tensor.data_ptr()); the handle owns only the persistent game state.
 *   - ALIGNMENT: every device buffer of observations, orders, features or packed results handed to the library must be 16-byte aligned
 *     (observation rows move 16 bytes per lane, order rows as int2 / uint4); reward and score buffers 8-byte aligned (one float2 / int2 per
 *     env, so any env offset into an [N][2] tensor is fine).  Any hipMalloc / torch allocation is 16-byte aligned; a slice of one need not be.
 *     Checked by every entry point: EVG_ERR_INVALID names the pointer.  Byte arrays (done, winner, status, mask, fog planes) have no
 *     requirement.
 *   - all calls return 0 on success or a negative evg_status; evg_last_error() gives the text
 *     (thread-local).  No C++ exception crosses the ABI.
 *   - step/reset/random_actions ENQUEUE on the caller's hipStream_t (`stream`, may be NULL for the
 *     default stream) and return without synchronising.  A handle belongs to ONE device and is not thread-safe; any
 *     number of handles may live on a device and run concurrently on different streams (they share nothing).
 *   - there is NO CPU fallback: evg_create fails with EVG_ERR_NO_DEVICE when no gfx950 device
 *     is usable.
 *   - env e of this handle has the global id env_id_base + e; random streams are keyed by the
 *     global id, so results do not depend on how envs are sharded over GPUs.
 */
#ifndef EVG_H
#define EVG_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EVG_ABI_VERSION 6
/* 6: evg_smart_get_action (DQNAgent.get_action with epsilon > 0), evg_step_vs_policy_smart (the learner-seat turn that also writes the Smart_State
 *    features), evg_get_run_state / evg_set_run_state (agent objects, returns, win counters: checkpoint / resume); reward / score buffers need 8-byte
 *    alignment only (5 asked 16 of every buffer)
 * 5: evg_smart_actions; evg_comm_unique_id / evg_comm_init / evg_gather_returns / evg_comm_destroy + EVG_ERR_COMM; every device buffer must be
 *    16-byte aligned (checked)
 * 4: evg_step_vs_policy / evg_observe_seat / evg_rollout_vs_policy / evg_random_actions_seat / evg_smart_state_seat, evg_smart_state_compact,
 *    evg_check_fault + EVG_ERR_FAULT, evg_pack_episode_results_counted, evg_config.cache_mib
 * 3: evg_launch_plan
 * 2: evg_pack_episode_results, node words as u32 */

/* The ABI is exactly the functions declared in this header: the library is built with -fvisibility=hidden and only they are exported. */
#define EVG_API __attribute__((visibility("default")))

/* Fixed dimensions of the reference environment (everglades_env.py:17-22). */
#define EVG_NUM_PLAYERS 2
#define EVG_NUM_GROUPS 12      /* num_groups            */
#define EVG_NUM_NODES 11       /* num_nodes (IDs 1..11) */
#define EVG_NUM_UNITS 100      /* num_units per player  */
#define EVG_NUM_ACTIONS 7      /* num_actions_per_turn  */
#define EVG_OBS_LEN 105        /* 1 + 4*11 + 5*12 (everglades_env.py:158-171) */
#define EVG_MAX_UNIT_TYPES 4
#define EVG_MAX_GROUP_SIZE 12
#define EVG_MAX_SCORE 3700     /* everglades_env.py:11 */

typedef enum evg_status {
    EVG_OK = 0,
    EVG_ERR_INVALID = -1,      /* bad argument / table out of the supported domain */
    EVG_ERR_NO_DEVICE = -2,    /* no usable HIP device (there is no CPU path)        */
    EVG_ERR_HIP = -3,          /* a HIP runtime call failed                         */
    EVG_ERR_ALLOC = -4,
    EVG_ERR_FAULT = -5,        /* the handle's fault word is set (evg_check_fault): its results are not valid, destroy it */
    EVG_ERR_COMM = -6          /* RCCL is not available or one of its calls failed (evg_comm_*, evg_gather_returns)          */
} evg_status;

typedef enum evg_obs_dtype { EVG_OBS_F32 = 0, EVG_OBS_F64 = 1, EVG_OBS_I16 = 2 } evg_obs_dtype;

/* Source of the combat target draws (server.py:562).
 *   EVG_RNG_KEYED_PHILOX   the fast path: counter-based draws keyed by (seed, env id, episode, turn, node, player,
 *                          group, unit) -- DESIGN.md section 4; what every throughput figure is measured with.
 *   EVG_RNG_STOCK_MT19937  compatibility mode: every env owns the generator the unmodified reference
 *                          uses -- numpy's legacy MT19937 after np.random.seed(s), randint by masked rejection, consumed
 *                          in the reference's loop order including its two unobservable focus draws (server.py:205,
 *                          :338) -- so a game replays the reference process bit for bit.  2.5 KB of extra state per
 *                          env and sequential draws: for validation, not speed.  Single-turn launches only.        */
typedef enum evg_rng_mode { EVG_RNG_KEYED_PHILOX = 0, EVG_RNG_STOCK_MT19937 = 1 } evg_rng_mode;

/* game status (server.py:284-288) */
enum { EVG_IN_PROGRESS = 0, EVG_TIME_EXPIRED = 1, EVG_BASE_CAPTURE = 2, EVG_ANNIHILATION = 3 };
/* winner_out values; the harness rule evaluate.py:155-160 applied to the terminal rewards */
enum { EVG_WINNER_NONE = -1, EVG_WINNER_P0 = 0, EVG_WINNER_P1 = 1, EVG_WINNER_TIE = 2 };
/* node_resource bits (DemoMap.json "Resource") */
enum { EVG_RES_DEFENSE = 1, EVG_RES_OBSERVE = 2 };

/*
 * Flattened constant tables: what server.py:40-131 parses out of config/DemoMap.json and
 * config/UnitDefinitions.json plus the army of everglades_env.py:145-156.  Node arrays are
 * indexed by node ID (1..11; entry 0 unused).  Domain checked by evg_create:
 *   distances 0 (not connected) or 1..7; control_points 1..511; defense >= 0;
 *   unit damage/speed/control/cost 1..15, unit health 1..255, at most 4 unit types;
 *   the army SHAPE is the reference's hard-coded one (everglades_env.py:145-156): group_size must be 8 for groups 0..10 and
 *   12 for group 11 of both players -- anything else is refused with EVG_ERR_INVALID (the unit TYPE of every group is free);
 *   total damage of an army (sum of size x unit damage) <= 255; max_turns 1..255;
 *   global env ids are 32-bit keys of the random streams: env_id_base + num_envs <= 2^32.
 * PINNED AGAINST THE REFERENCE inside this domain (tests/golden/custom_*.npz: the imported reference playing on non-default map / unit files): directed
 * distances incl. one-way edges and edges with different lengths per direction, control points up to 511, non-dyadic defenses, DEFENSE / OBSERVE on any
 * node, bases on any two nodes, unit types in any order, a p1_node_map that is not its own inverse.  What this struct cannot express and a host parser must
 * therefore refuse (everglades_amd.tables_from_json does): a node list that is not in ascending ID order (the reference's own fog mask then mixes list
 * positions with IDs, server.py:409-418), a resource named 'DEFEND' (it would switch the reference's otherwise dead fortress bonus on, server.py:595 -- not
 * modelled), fractional values where the tables hold integers (the reference would compute with the float).  max_turns other than 150 and p1_node_map
 * other than DemoMap's have no file in the reference (hard-coded at server.py:321 and :89).
 */
typedef struct evg_tables {
    int32_t node_dist[EVG_NUM_NODES + 1][EVG_NUM_NODES + 1];
    int32_t node_control_points[EVG_NUM_NODES + 1];
    double  node_defense[EVG_NUM_NODES + 1];
    int32_t node_resource[EVG_NUM_NODES + 1];
    int32_t node_team_start[EVG_NUM_NODES + 1];        /* -1, 0 or 1                         */
    int32_t p1_node_map[EVG_NUM_NODES + 1];            /* server.py:89                       */
    int32_t num_unit_types;
    int32_t unit_health[EVG_MAX_UNIT_TYPES];           /* "armor" in the damage equation      */
    int32_t unit_damage[EVG_MAX_UNIT_TYPES];
    int32_t unit_speed[EVG_MAX_UNIT_TYPES];
    int32_t unit_control[EVG_MAX_UNIT_TYPES];
    int32_t unit_cost[EVG_MAX_UNIT_TYPES];
    int32_t group_type[EVG_NUM_PLAYERS][EVG_NUM_GROUPS];  /* unit type id of each group        */
    int32_t group_size[EVG_NUM_PLAYERS][EVG_NUM_GROUPS];  /* units per group at reset          */
    int32_t max_turns;                                    /* 150 (server.py:321)               */
} evg_tables;

typedef struct evg_config {
    uint32_t struct_size;      /* = sizeof(evg_config); ABI guard            */
    uint32_t abi_version;      /* = EVG_ABI_VERSION                          */
    int32_t  num_envs;         /* N >= 1                                     */
    int32_t  device_id;        /* HIP device ordinal                         */
    uint64_t seed;             /* Philox key                                 */
    uint64_t env_id_base;      /* global id of env 0 of this handle          */
    int32_t  obs_dtype;        /* evg_obs_dtype of obs_out buffers           */
    int32_t  auto_reset;       /* 1: an env that finishes in step() is reset in the same launch and
                                  obs_out holds the first observation of its next episode           */
    int32_t  rng_mode;         /* evg_rng_mode; in the stock mode env e starts as np.random.seed((uint32)(seed + env_id_base + e)) */
    int32_t  cache_mib;        /* 0 = derive: what a chunked rollout launch may cycle through, in MiB -- the memory-side (Infinity) cache share of this
                                  device, 256 MiB x compute units / 256 on MI355X (HIP has no query for it); > 0 overrides.  evg_launch_plan reports it */
    evg_tables tables;
} evg_config;

typedef struct evg_handle evg_handle;

/* Fills `t` with the DemoMap / UnitDefinitions / default-army tables. */
EVG_API void evg_default_tables(evg_tables* t);

/* Replaces EvergladesGame.__init__/board_init/unitTypes_init (server.py:14-131): tables are parsed
 * once by the host and copied to the device; state for N envs is allocated and every env is put in
 * the game_init position (episode counter -1, so the first evg_reset starts episode 0). */
EVG_API int evg_create(const evg_config* cfg, evg_handle** out);
EVG_API void evg_destroy(evg_handle* h);

/* Replaces EvergladesEnv.reset (everglades_env.py:75-116) -> game_init (server.py:133-209) ->
 * _build_observations (everglades_env.py:158-171).
 *   mask    device uint8[N] or NULL (= all): envs with mask != 0 start a new episode
 *   obs_out device [N][2][105] of cfg.obs_dtype or NULL; written for the envs that were reset */
EVG_API int evg_reset(evg_handle* h, const uint8_t* mask, void* obs_out, void* stream);

/* Replaces EvergladesEnv.step (everglades_env.py:32-73) -> EvergladesGame.game_turn
 * (server.py:211-279: orders, combat :503, movement :656, capture :708, game_end :281) ->
 * board_state/player_state (:382-501).  build_knowledge_output (:769-907) mutates nothing and is
 * not reproduced.
 *   actions    device int32 [N][2][7][2]  (group id, node id in the player's own numbering); ids in [-12, -1]
 *              index from the end like the reference's Python lists (group -1 is group 11; for player 1 node -1
 *              is p1_node_map[-1]; `used_swarms` dedupes on the id as given); any other out-of-range id is an
 *              invalid order, never undefined behaviour
 *   obs_out    device [N][2][105] of cfg.obs_dtype
 *   reward_out device float [N][2]   (everglades_env.py:37-61)
 *   done_out   device uint8 [N]
 *   winner_out device int8  [N]  or NULL
 *   scores_out device int32 [N][2] or NULL  (server.py:291-317)
 *   status_out device uint8 [N]  or NULL  (server.py:284-288)
 * Without auto_reset a finished env is frozen: further steps leave it unchanged and repeat its
 * terminal outputs. */
EVG_API int evg_step(evg_handle* h, const int32_t* actions, void* obs_out, float* reward_out, uint8_t* done_out,
             int8_t* winner_out, int32_t* scores_out, uint8_t* status_out, void* stream);

/* Observations of the current state without stepping (after evg_set_state, or to re-read them):
 * the board_state/player_state half of step (server.py:382-501). obs_out as in evg_step. */
EVG_API int evg_observe(evg_handle* h, void* obs_out, void* stream);

/* The turn of the loop every training and evaluation script of the reference runs: a caller (the learner) on ONE seat, a scripted
 * bot on the other (evaluate.py:85-93,143-152: `actions[p] = players[p].get_action(obs[p])` with one learned and one scripted player;
 * agents/Smart_State/training_scripts/dqn_smart_state_training.py:114-122) -- in ONE launch: the opponent's bot is evaluated inside the
 * step kernel from the on-chip state (exactly what its observation would hold; same agent objects as evg_scripted_actions, alive across
 * episodes), the caller's orders are read for the other seat, and only the caller's seat's observation is written (420 B per env less
 * than evg_step, half of the observation build, no second launch and no round trip of the opponent's observations and orders).
 *   seat             0 or 1: the caller's seat
 *   actions          device int32, the caller's 7 order rows per env: [N][7][2] (actions_both_seats == 0) or the rows [:, seat] of
 *                    a [N][2][7][2] tensor (actions_both_seats != 0; the other seat's rows are ignored).  Domain as in evg_step
 *   opponent_policy  EVG_POLICY_* played by seat 1 - seat
 *   obs_seat_out     device [N][105] of cfg.obs_dtype: the caller's seat's observation (everglades_env.py:158-171); 16-byte aligned
 *   reward_out, done_out, winner_out, scores_out, status_out: as in evg_step (both seats' rewards and scores: the harness compares
 *                    reward[0] with reward[1], evaluate.py:155-160)
 * Results are those of evg_scripted_actions(opponent) + evg_step on the same orders, bit for bit.  Keyed-Philox handles only.  The launch itself
 * lasts as long as evg_step's (26.7 us at 65 536 envs on an MI355X; a single-turn launch is bound by its latency chain, not by its bytes: DESIGN.md
 * section 6); what is saved is the bot's own kernel (7-9 us and 55 MB of observations read back) and the traffic listed above. */
EVG_API int evg_step_vs_policy(evg_handle* h, int seat, const int32_t* actions, int actions_both_seats, int opponent_policy, void* obs_seat_out,
                               float* reward_out, uint8_t* done_out, int8_t* winner_out, int32_t* scores_out, uint8_t* status_out, void* stream);
/* The same turn for a Smart_State learner (agents/Smart_State/DQNAgent.py): besides the seat's observation the launch also writes the agent's network INPUT
 * of the next turn -- the features of DQNAgent.create_swarm_obs (:268-300) in the compact form of evg_smart_state_compact: shared_out device float [N][34]
 * (8-byte aligned), swarm_out device float [N][12][13] (16-byte aligned) -- computed from the observation image while it is still on chip.  Value for value
 * what evg_smart_state_compact(h, -1, obs_seat_out, shared_out, swarm_out) would write after this call, without that kernel (23 us at 65 536 envs) and
 * without reading the row back.  The first features of a loop (after evg_reset) come from evg_observe_seat + evg_smart_state_compact. */
EVG_API int evg_step_vs_policy_smart(evg_handle* h, int seat, const int32_t* actions, int actions_both_seats, int opponent_policy, void* obs_seat_out,
                                     float* shared_out, float* swarm_out, float* reward_out, uint8_t* done_out, int8_t* winner_out, int32_t* scores_out,
                                     uint8_t* status_out, void* stream);
/* evg_observe for one seat: obs_seat_out device [N][105] (after evg_reset / evg_set_state, to start a evg_step_vs_policy loop). */
EVG_API int evg_observe_seat(evg_handle* h, int seat, void* obs_seat_out, void* stream);

/* Fog-of-war planes of the current state, both computed by the reference and never applied to its observations:
 *   fog_out        device uint8 [N][2][11] or NULL: the `valid_nodes` mask of board_state (server.py:402-425);
 *                  [e][p][i] = 1 iff player p sees node ID i+1 (a node it controls, a neighbour of a controlled OBSERVE
 *                  node, or a node where it has a non-moving group)
 *   knowledge_out  device uint8 [N][2][11] or NULL: the knowledge level of build_knowledge_output (server.py:779-832):
 *                  2 full (controlled, or a non-moving group there), 1 partial (next to a fully controlled OBSERVE node
 *                  of the player, or one of its groups is moving there), 0 none
 * Real node order, not mirrored for player 1, exactly as the reference computes them. */
EVG_API int evg_fog_of_war(evg_handle* h, uint8_t* fog_out, uint8_t* knowledge_out, void* stream);

/* The opposing-group sightings `opp_k` of build_knowledge_output (server.py:845-907), also computed and dropped by the
 * reference.  sight_out: device int8 [N][2][12][4]; [e][p][g] = what player p knows of opposing group g:
 *   {seen, node ID, destination key, units alive}.  seen = 1 iff the group is listed at a node whose knowledge level is 1 or 2
 *   and it is either not moving (key -1) or moving to a node of knowledge > 0 (key = that node's index in the node list,
 *   i.e. ID - 1 -- the reference keys stationary groups by -1 and moving ones by the list index, :866-881); else 0,0,0,0. */
EVG_API int evg_sightings(evg_handle* h, int8_t* sight_out, void* stream);

/* Consumer-side preprocessing of the reference's strongest agent family (agents/Smart_State/DQNAgent.py:200-300,
 * create_swarm_obs): from `player`'s rows of obs (device [N][2][105] of cfg.obs_dtype) to features_out, device float
 * [N][12][59] (16-byte aligned): per swarm {turn/150, 11 x control/100, 11 x enemy units/100, 11 x idle allied groups/12, one-hot node,
 * avg health x alive / 1000, in transit, one-hot swarm id}; each value is the reference's float64 expression rounded to
 * float32.  evg_move_table fills table[11][5] with Move_Translation.get_move(node0, direction) (host memory). */
EVG_API int evg_smart_state(evg_handle* h, int player, const void* obs, float* features_out, void* stream);
/* The same features from a one-seat observation tensor (device [N][105], as evg_step_vs_policy / evg_observe_seat write it). */
EVG_API int evg_smart_state_seat(evg_handle* h, const void* obs_seat, float* features_out, void* stream);
/* The same features without their redundancy (what a bandwidth-bound consumer wants: the full matrix is 2 832 B per env, 185 MB per call at 65 536 envs).
 * Of the 59 features of a swarm, 34 are the same for all 12 swarms of an env ({turn/150, 11 x control/100, 11 x enemy units/100, 11 x idle allied groups/12})
 * and 12 are the constant one-hot swarm id; the matrix is therefore determined by
 *   shared_out  device float [N][34]       (8-byte aligned)   = features[e][s][0:34] for any s
 *   swarm_out   device float [N][12][13]   (16-byte aligned)  = features[e][s][34:47] = {one-hot node (11), avg health x alive / 1000, in transit}
 * (760 B per env); features[e][s] = shared[e] ++ swarm[e][s] ++ onehot(s), value for value what evg_smart_state writes.  player 0 / 1: obs is
 * [N][2][105]; player -1: obs is a one-seat tensor [N][105]. */
EVG_API int evg_smart_state_compact(evg_handle* h, int player, const void* obs, float* shared_out, float* swarm_out, void* stream);
EVG_API void evg_move_table(int32_t* table);
/* ... and the way back, network output -> orders: DQNAgent.get_best_actions (agents/Smart_State/DQNAgent.py:176-198) over swarm_think (:233-266),
 * get_swarm_node_number (:302-310) and Move_Translation.get_move (Move_Translation.py:85-97).
 *   q               device float [N][12][5]: the policy network's Q values of every swarm for the five directions left, right, up, down, stay (the network
 *                   itself -- QNetwork 59-60-60-5 in the reference -- is the consumer's); 16-byte aligned
 *   obs / player    as in evg_smart_state_compact: player 0 / 1 reads that seat's rows of [N][2][105], player -1 a one-seat tensor [N][105] (only the
 *                   swarm locations obs[45 + 5 s] are read)
 *   actions_out     device int32 [N][7][2]: {swarm, node} -- exactly the `actions` of evg_step_vs_policy (actions_both_seats == 0)
 *   directions_out  device int32 [N][7][2] or NULL: {swarm, direction} (what the reference's replay memory stores)
 * Every swarm's best direction is the FIRST maximum of its five Q values (torch.argmax), its order the node get_move gives for that direction from the
 * swarm's location, and the seven rows are the first seven decisions of a STABLE ASCENDING sort by best Q: the reference acts with the seven swarms whose
 * best Q is LOWEST (sorted(...)[:7], :189-197) -- reproduced as it is.  Q values must not be NaN (the reference's sort is undefined for them; here a NaN
 * is a swarm's maximum, as in torch, and sorts like +inf).  With evg_smart_state(_seat / _compact) this closes the learner's turn on the device:
 * observation -> features -> (consumer's network) -> orders -> evg_step_vs_policy, no host or framework glue between the kernels. */
EVG_API int evg_smart_actions(evg_handle* h, int player, const void* obs, const float* q, int32_t* actions_out, int32_t* directions_out, void* stream);
/* The whole of DQNAgent.get_action (agents/Smart_State/DQNAgent.py:130-146): `sample = random.random(); if sample < self.epsilon:
 * get_random_actions(obs) else get_best_actions(obs)`, i.e. evg_smart_actions plus the exploring branch (:148-173): swarms = np.random.choice(12, 7,
 * replace=False), directions = np.random.choice(5, 7, replace=True), row i = {swarms[i], get_move(location of swarms[i] - 1, directions[i])}.  With it the
 * learner's turn stays on the device during TRAINING (epsilon > 0) too.
 *   seat            0 / 1: the agent's player number; with obs_one_seat == 0 its rows of obs [N][2][105] are read, with obs_one_seat != 0 obs is a
 *                   one-seat tensor [N][105].  Only obs[0] (the turn, part of the key of the draws) and the swarm locations obs[45 + 5 s] are read.
 *   epsilon         the exploring probability of every env, in [0, 1]; epsilon_env (device float [N], may be NULL) gives one per env instead.
 *   actions_out / directions_out   as in evg_smart_actions;  explored_out  device uint8 [N] or NULL: 1 where the random branch was taken.
 * The coin and the two choices are keyed draws like every other (oracle/rng_spec.py `explore_draws`, domain 4): one agent call =
 * (seed, global env id, episode, turn = obs[0], seat) -- pinned against the reference's own get_action by tests/golden/smart_explore.npz.  The coin is a 32-bit
 * fraction compared with epsilon in float64 (epsilon 0 never explores, epsilon 1 always). */
EVG_API int evg_smart_get_action(evg_handle* h, int seat, int obs_one_seat, const void* obs, const float* q, float epsilon, const float* epsilon_env,
                                 int32_t* actions_out, int32_t* directions_out, uint8_t* explored_out, void* stream);

/* Input generator for the benchmark configs: the on-device equivalent of
 * agents/State_Machine/random_actions.py:38-46 for every env and both players, keyed by
 * (seed, env id, episode, turn, player).  actions_out: device int32 [N][2][7][2]. */
EVG_API int evg_random_actions(evg_handle* h, int32_t* actions_out, void* stream);

/* The same generator for ONE seat: actions_seat_out device int32 [N][7][2], identical to the rows [:, seat] of evg_random_actions
 * (a stand-in for a learner's policy output in benchmarks of evg_step_vs_policy). */
EVG_API int evg_random_actions_seat(evg_handle* h, int seat, int32_t* actions_seat_out, void* stream);

/* Scripted opponents on the device (agents/State_Machine/): one agent object per (env, player), kept in the handle
 * and alive across episodes like the reference's (evaluate.py:85-93).  Reads `player`'s rows of obs (device
 * [N][2][105] of cfg.obs_dtype, as written by evg_step/evg_reset) and writes that player's 7 order rows of
 * actions_out (device int32 [N][2][7][2]).
 * All 17 bots of agents/State_Machine/ are covered by the 15 ids below (two pairs of files are identical in
 * behaviour); their random draws (np.random.choice, np.random.shuffle, random.random) are keyed like every other
 * draw.  Order ids a bot emits outside [0, 11] (e.g. the -1 of cycle_target_node*.py) are passed
 * through; evg_step treats them like the reference's Python lists do.
 * The bots route by THEIR OWN constants on every map: each file of agents/State_Machine/ carries DemoMap's NODE_CONNECTIONS (and TAR_NODE) as a module
 * constant and never reads the map file, so on another map many of their orders are rejected by the server -- the reference's behaviour, reproduced
 * (tests/golden/custom_agents.npz).
 * An agent is not consulted for a game that is over and not yet reset (status != 0 without auto-reset): the harness has left
 * that game's loop (evaluate.py:147-152), so its rows are zero and its object does not advance.
 * evg_scripted_reset re-creates all agent objects (first_turn, cycling position, attack list). */
enum {
    EVG_POLICY_RANDOM = 0,                 /* random_actions.py, random_actions_2.py                         */
    EVG_POLICY_CYCLE_RUSH_25 = 1,          /* cycle_rush_turn25.py                                           */
    EVG_POLICY_CYCLE_RUSH_50 = 2,          /* cycle_rush_turn50.py                                           */
    EVG_POLICY_SWARM = 3,                  /* swarm_agent.py                                                 */
    EVG_POLICY_ALL_CYCLE = 4,              /* all_cycle.py                                                   */
    EVG_POLICY_BASE_RUSH_V1 = 5,           /* base_rush_v1.py                                                */
    EVG_POLICY_BULL_RUSH = 6,              /* bull_rush.py                                                   */
    EVG_POLICY_CYCLE_TARGET_NODE = 7,      /* cycle_target_node.py   (target 11, level 75)                   */
    EVG_POLICY_CYCLE_TARGET_NODE1 = 8,     /* cycle_target_node1.py  (target 1, level 75)                    */
    EVG_POLICY_CYCLE_TARGET_NODE11 = 9,    /* cycle_target_node11.py (target 11, level 500)                  */
    EVG_POLICY_CYCLE_TARGET_NODE11P2 = 10, /* cycle_target_node11P2.py                                       */
    EVG_POLICY_DFS_ATTACK = 11,            /* dfs_attack.py                                                  */
    EVG_POLICY_NO_ACTION = 12,             /* no_action.py                                                   */
    EVG_POLICY_RANDOM_DELAY = 13,          /* random_actions_delay.py                                        */
    EVG_POLICY_SAME_COMMANDS = 14,         /* same_commands.py, same_commands_2.py                           */
    EVG_POLICY_COUNT = 15
};
EVG_API int evg_scripted_actions(evg_handle* h, int policy, int player, const void* obs, int32_t* actions_out, void* stream);
EVG_API int evg_scripted_reset(evg_handle* h, void* stream);

/* Rollout driver for random-vs-random play (the reference's demo/random_demo.py:90-113 loop with both
 * agents = random_actions): enqueues `steps` x (evg_random_actions into actions_buf, then evg_step) on
 * `stream` from native code, so that launch cost, not the Python interpreter, bounds small batches.
 * fused == 1: the step kernel draws the orders itself (same generator, same values) and stores them in
 * actions_buf, which saves the second launch and the round trip of the action tensor through HBM.
 * fused >= 2: additionally each launch plays up to `fused` consecutive turns per wavefront (persistent form: the
 * state of a wavefront's envs stays in LDS/registers between turns; observations, rewards, actions ... are still
 * written every turn, so the buffers hold the last turn as before).  Results are identical in all three forms.
 * With fused >= 1 obs_out and actions_buf may be NULL: the rollout then writes no observations (the step kernel skips the
 * observation image and its write-out: a sixth of a turn's instructions and 55 % of its bytes) and / or does not record the orders --
 * what an evaluation loop needs, which reads only rewards, done flags and the episode results (evaluate.py:143-181).
 * A launch of the persistent form (fused >= 2) is a launch PLAN (evg_launch_plan: up to two step kernels, or memset + chunked kernel + queue check) that
 * keeps nothing on the host: when `stream` is being captured (e.g. torch.cuda.graph around this call) the plan becomes part of the caller's graph and
 * every replay plays the next turns.  (The library does not replay graphs of its own: measured slower than plain launches on ROCm 7.2, DESIGN.md
 * section 3.)  steps < 0 prepares a rollout of -steps turns with exactly these arguments -- whatever one-time work its launches need is done now, nothing
 * is enqueued and no state changes (a no-op in the product build; the graph-replay A/B build captures its graphs here).
 * Outputs as in evg_step (they hold the LAST step when the call returns).  If step_kernel_ms (host
 * pointer) is not NULL the work is bracketed by hipEvents on `stream`, the call synchronises the stream and stores the
 * stream time per turn in milliseconds: persistent form -- the duration of each launch (or launch plan, see
 * evg_launch_plan), summed, over the turns played; one launch per turn -- two events around the WHOLE loop over `steps`,
 * i.e. the step kernel plus the gap to the next launch plus, with fused == 0, the action kernel of the turn.  A timed call also returns
 * EVG_ERR_FAULT when the handle's fault word is set (evg_check_fault). */
EVG_API int evg_rollout_random(evg_handle* h, int steps, int fused, int32_t* actions_buf, void* obs_out, float* reward_out,
                       uint8_t* done_out, int8_t* winner_out, int32_t* scores_out, uint8_t* status_out,
                       float* step_kernel_ms, void* stream);

/* The same driver for any pair of on-device policies (EVG_POLICY_*; BASELINE config 5 is CYCLE_RUSH_25 vs SWARM).
 * fused == 0: per turn two agent launches read the previous observations in obs_out (which must hold the current
 * observations when the call starts, e.g. from evg_reset) and one step launch follows.  fused == 1: the step kernel
 * evaluates both agents itself from the on-chip state (the same quantities their observation holds) and stores the
 * orders in actions_buf; fused >= 2: persistent form as in evg_rollout_random.  Identical results in all forms. */
EVG_API int evg_rollout_policies(evg_handle* h, int steps, int fused, int policy0, int policy1, int32_t* actions_buf, void* obs_out,
                         float* reward_out, uint8_t* done_out, int8_t* winner_out, int32_t* scores_out, uint8_t* status_out,
                         float* step_kernel_ms, void* stream);

/* Driver of the learner-seat loop for benchmarks: `steps` x (evg_random_actions_seat into actions_seat_buf [N][7][2] -- the stand-in for
 * the caller's policy output arriving in a tensor --, then evg_step_vs_policy(seat, actions_seat_buf, opponent_policy)), enqueued from
 * native code: two launches per turn.  Outputs and step_kernel_ms (stream time per turn between two events around the whole loop; the
 * call then synchronises) as in evg_rollout_random with fused == 0. */
EVG_API int evg_rollout_vs_policy(evg_handle* h, int steps, int seat, int opponent_policy, int32_t* actions_seat_buf, void* obs_seat_out, float* reward_out,
                                  uint8_t* done_out, int8_t* winner_out, int32_t* scores_out, uint8_t* status_out, float* step_kernel_ms, void* stream);

/* Canonical state exchange (host order; used by parity tests and to load golden positions).
 * All pointers are HOST pointers; the call synchronises.  Any pointer may be NULL.
 *   groups int32 [N][2][12][8]: location, travel_destination (-1 none), distance_remaining, ready,
 *                               moving, destroyed, count, arrival stamp      (definitions.py:35-64)
 *   nodes  int32 [N][11][2]   : controlState, controlledBy                   (definitions.py:16-17)
 *   health double[N][2][100]  : unitHealth of the groups back to back         (definitions.py:62)
 *   env    int32 [N][4]       : current_turn, status, episode, reserved
 */
EVG_API int evg_get_state(evg_handle* h, int32_t* groups, int32_t* nodes, double* health, int32_t* env);
EVG_API int evg_set_state(evg_handle* h, const int32_t* groups, const int32_t* nodes, const double* health,
                  const int32_t* env);

/* Stock-entropy mode only (EVG_ERR_INVALID otherwise).
 * evg_seed_stock_entropy: np.random.seed(seeds[e]) for every env (HOST pointer, N entries; NULL = the create-time rule
 * (uint32)(seed + env_id_base + e)); enqueued on `stream` after a synchronous upload of the seeds.
 * evg_get/set_stock_entropy: the generators themselves, HOST uint32 [N][625] = 624 key words + position (the layout of
 * np.random.get_state()[1:3]); the calls synchronise.  Together with evg_get/set_state this checkpoints a game. */
EVG_API int evg_seed_stock_entropy(evg_handle* h, const uint32_t* seeds, void* stream);
EVG_API int evg_get_stock_entropy(evg_handle* h, uint32_t* out);
EVG_API int evg_set_stock_entropy(evg_handle* h, const uint32_t* in);

/* Checkpoint / resume of a RUNNING job: what evg_get_state / evg_set_state do not carry.  (the reference never serialises its env; its agents
 * pickle their own networks.  A handle restored with evg_set_state + evg_set_run_state -- + evg_set_stock_entropy in the stock mode -- on a handle created
 * with the same config continues bit for bit like the one that was saved: tests/test_gpu_parity.py::test_checkpoint_resume_continues_bit_for_bit.)
 * HOST pointers; the calls synchronise; any pointer may be NULL; evg_set_state zeroes the running returns, so restore the run state AFTER it.
 *   agents           uint32 [N][2][3]: the scripted agents' objects per (env, player) -- {first_turn / group_num / node_num word, SwarmAgent's attack list,
 *                    dfs_attack's call counter} (alive across episodes like the reference's agent objects, evaluate.py:85-93)
 *   running_returns  float [N][2]: the sum of rewards of the episode in progress
 *   returns, length, winner, totals: the arrays of evg_episode_stats (results of the last finished episode per env; episodes / wins since create)
 * evg_get_run_state fails with EVG_ERR_FAULT on a faulted handle, like evg_get_state. */
EVG_API int evg_get_run_state(evg_handle* h, uint32_t* agents, float* running_returns, float* returns, int32_t* length, int8_t* winner, int64_t* totals);
EVG_API int evg_set_run_state(evg_handle* h, const uint32_t* agents, const float* running_returns, const float* returns, const int32_t* length,
                              const int8_t* winner, const int64_t* totals);

/* Per-env results of the most recently finished episode, and running totals (device -> host copy,
 * synchronises).  Any pointer may be NULL.
 *   returns float [N][2]  sum of rewards over the episode
 *   length  int32 [N]     turns played
 *   winner  int8  [N]     EVG_WINNER_* (NONE if the env has not finished an episode yet)
 *   totals  int64 [4]     episodes finished, p0 wins, p1 wins, ties (this handle, since create)
 */
EVG_API int evg_episode_stats(evg_handle* h, float* returns, int32_t* length, int8_t* winner, int64_t* totals);
/* The handle's fault word (synchronises the device).  Chunked persistent rollout launches (batches beyond what the device holds at once,
 * evg_launch_plan) hand sets of envs from workgroup to workgroup; the word records: 1 = a workgroup gave up waiting for a predecessor
 * chunk (bounded wait, about 5 s), 2 = a workgroup ran on an XCD the create-time probe did not see, 4 = a queue of a chunked launch was
 * not drained (checked on the stream after every chunked launch), 8 = a hand-over delivered STALE state: every lane hands on a checksum over the chunk number
 * and every state word it stored, and the lane that takes the set's next chunk recomputes it over the words it loaded.  None is expected ever; all mean that state and results of the
 * handle are not valid.  The word is STICKY: evg_reset / evg_set_state do not clear it -- destroy the handle.  Returns EVG_OK or
 * EVG_ERR_FAULT (message in evg_last_error); *fault_out (may be NULL) receives the word.  Every path on which results leave the
 * handle checks it too: evg_episode_stats, evg_episode_stats_device and evg_get_state fail with EVG_ERR_FAULT, a timed rollout call
 * (step_kernel_ms != NULL, which synchronises anyway) fails with it, and evg_pack_episode_results writes POISONED rows
 * {NaN, NaN, -2, -1} for every env (winner -2 is no EVG_WINNER_* value), so a gather of packed rows cannot carry bad results unnoticed. */
EVG_API int evg_check_fault(evg_handle* h, uint32_t* fault_out);
/* Device pointers to the same per-env arrays (returns float[N][2], length int32[N], winner int8[N]),
 * for the multi-GPU gather of episode returns without a host round trip. */
EVG_API int evg_episode_stats_device(evg_handle* h, float** returns, int32_t** length, int8_t** winner);
/* The same per-env results packed for the path's ONE exchange between GPUs (the win bookkeeping of
 * evaluate.py:155-181 then runs on the gathered rows): out[e] = {return of player 0, return of player 1, winner, length} as
 * float32 (the small integers are exact), 16 bytes per env, written on `stream`.  out: device memory, [N][4], 16-byte aligned. */
EVG_API int evg_pack_episode_results(evg_handle* h, float* out, void* stream);
/* The same, and the win bookkeeping of the packed rows in the same kernel: counts_out device int64 [4] = rows with winner P0, P1, TIE and
 * rows without a finished episode (zeroed and filled on `stream`; all four are -1 when the rows are poisoned).  What a multi-GPU run
 * all-reduces next to the gather: the sum over ranks must equal what rank 0 counts in the gathered rows (bench.py). */
EVG_API int evg_pack_episode_results_counted(evg_handle* h, float* out, int64_t* counts_out, void* stream);

/* ---- Multi-GPU without torch.distributed: the path's ONE exchange over RCCL itself (SURVEY 8b `evg_gather_returns`, 8e) ----------------------------
 * Environments shard over GPUs by contiguous global id (env_id_base), one process and one handle per GPU, and nothing is exchanged but the per-env
 * results of finished episodes (win bookkeeping, evaluate.py:155-181).  A Python caller uses everglades_amd.ResultGather (torch.distributed, backend
 * "nccl" = RCCL); a caller without torch -- examples/c_client.c -- uses these four entry points.  librccl is opened at run time (dlopen("librccl.so.1"):
 * in a PyTorch process that is the instance torch already loaded); libevg.so itself does not depend on it, and the calls return EVG_ERR_COMM where RCCL is
 * missing.
 *   evg_comm_unique_id   ONE rank makes the communicator's id (ncclGetUniqueId); the caller hands the EVG_COMM_ID_BYTES bytes to every rank by its own means
 *                        (a file, a socket, MPI, an environment variable)
 *   evg_comm_init        every rank, collectively (it blocks until all `world` ranks have called): communicator of this handle's device.  counts: HOST
 *                        int32 [world], the num_envs of every rank's handle -- contiguous shards in rank order; counts[rank] must be this handle's
 *   evg_gather_returns   every rank, collectively, enqueued on `stream` (no synchronisation): evg_pack_episode_results of this handle + one grouped RCCL
 *                        send / receive.  recv_out: on rank `root` device float [sum(counts)][4], 16-byte aligned -- the rows {return p0, return p1,
 *                        winner, length} of ALL envs in global env order --, NULL on every other rank.  A rank whose handle is faulted sends poisoned rows
 *                        (winner -2), as evg_pack_episode_results does
 *   evg_comm_destroy     releases the communicator (evg_destroy does it as well)
 * ERRORS OF THESE CALLS ARE FATAL FOR THE WHOLE JOB.  evg_comm_init and evg_gather_returns validate their arguments (and RCCL's presence) BEFORE they enter
 * the collective: a rank that fails there returns its error at once while every other rank is already inside ncclCommInitRank / waiting for that rank's rows,
 * where RCCL has no timeout.  A caller must therefore treat any non-zero return of evg_comm_* / evg_gather_returns on ANY rank as the end of the job (tear the
 * process group down, e.g. let the launcher kill the ranks), never retry or continue on the ranks that succeeded; ranks agree on arguments that can differ
 * (counts, world, root) over their own channel before calling. */
#define EVG_COMM_ID_BYTES 128
EVG_API int evg_comm_unique_id(void* id_out);
EVG_API int evg_comm_init(evg_handle* h, const void* id, int world, int rank, const int32_t* counts);
EVG_API int evg_gather_returns(evg_handle* h, int root, float* recv_out, void* stream);
EVG_API int evg_comm_destroy(evg_handle* h);

/* Which step kernel(s) a rollout launch of `turns_per_launch` turns runs for this handle's batch on this device, as text in
 * buf (for benchmark records and logs): the kernel mapping (two / four lanes per env), the env range and wavefront count of
 * every launch of the plan, and the device capacity the plan was derived from (compute units from hipDeviceProp_t, resident
 * wavefronts from the kernels' own occupancy -- no 256-CU literal; the XCDs the create-time probe saw; the memory-side cache budget a chunked launch may
 * cycle through, evg_config.cache_mib, and the bytes per env it is compared with).  The plan described is the one of a rollout that writes observations and
 * records the orders (a rollout without them has a smaller footprint and may chunk a slightly larger batch).  turns_per_launch == 1 describes evg_step.
 * Returns the number of kernel launches per rollout launch (>= 1) or a negative evg_status. */
EVG_API int evg_launch_plan(const evg_handle* h, int turns_per_launch, char* buf, int buflen);

EVG_API int evg_num_envs(const evg_handle* h);
/* bytes of persistent device state per env (for the roofline accounting in DESIGN.md) */
EVG_API int evg_state_bytes_per_env(const evg_handle* h);
EVG_API const char* evg_last_error(void);
EVG_API int evg_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* EVG_H */
