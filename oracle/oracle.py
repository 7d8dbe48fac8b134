"""ctypes wrapper of the CPU oracle (TEST INFRASTRUCTURE -- see evg_oracle.c header).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
"""
import ctypes as C
import os
import subprocess
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_build", "libevg_oracle.so")
SANITIZED_LIB = os.environ.get("EVG_ORACLE_LIB")      # `make -C oracle sanitize`: the ASan + UBSan build of the same source (test infrastructure reads
                                                       # the environment; the product never does)

NP, NG, NN, NU, NA, OBS = 2, 12, 11, 100, 7, 105


class Tables(C.Structure):
    """Mirror of `evg_tables` (include/evg.h)."""
    _fields_ = [
        ("node_dist", (C.c_int32 * 12) * 12), ("node_control_points", C.c_int32 * 12),
        ("node_defense", C.c_double * 12), ("node_resource", C.c_int32 * 12),
        ("node_team_start", C.c_int32 * 12), ("p1_node_map", C.c_int32 * 12),
        ("num_unit_types", C.c_int32), ("unit_health", C.c_int32 * 4), ("unit_damage", C.c_int32 * 4),
        ("unit_speed", C.c_int32 * 4), ("unit_control", C.c_int32 * 4), ("unit_cost", C.c_int32 * 4),
        ("group_type", (C.c_int32 * 12) * 2), ("group_size", (C.c_int32 * 12) * 2), ("max_turns", C.c_int32),
    ]


def demo_tables():
    """DemoMap / UnitDefinitions / default army, restated from config/DemoMap.json:5-300,
    config/UnitDefinitions.json:4-27, server.py:89 and everglades_env.py:145-156."""
    t = Tables()
    conn = {1: {2: 6, 4: 6}, 2: {1: 6, 3: 4, 5: 4}, 3: {2: 4, 4: 4, 5: 6, 6: 3, 7: 6}, 4: {1: 6, 3: 4, 7: 4},
            5: {2: 4, 3: 6, 8: 4, 9: 6}, 6: {3: 3, 9: 3}, 7: {3: 6, 4: 4, 9: 6, 10: 4}, 8: {5: 4, 9: 4, 11: 6},
            9: {5: 6, 6: 3, 7: 6, 8: 4, 10: 4}, 10: {7: 4, 9: 4, 11: 6}, 11: {8: 6, 10: 6}}
    for a, row in conn.items():
        for b, d in row.items():
            t.node_dist[a][b] = d
    cp = {1: 500, 11: 500}
    dfn = {1: 1.0, 2: 1.5, 3: 1.75, 4: 1.5, 5: 1.75, 6: 1.75, 7: 1.75, 8: 1.5, 9: 1.75, 10: 1.5, 11: 1.0}
    res = {2: 2, 8: 2, 4: 1, 10: 1}   # OBSERVE = 2, DEFENSE = 1
    for i in range(1, 12):
        t.node_control_points[i] = cp.get(i, 100)
        t.node_defense[i] = dfn[i]
        t.node_resource[i] = res.get(i, 0)
        t.node_team_start[i] = {1: 0, 11: 1}.get(i, -1)
    t.node_team_start[0] = -1
    for i, v in enumerate([0, 11, 8, 9, 10, 5, 6, 7, 2, 3, 4, 1]):
        t.p1_node_map[i] = v
    t.num_unit_types = 3     # tank = 0, controller = 1, striker = 2 (JSON order)
    for i, (h, d, s, c, k) in enumerate([(3, 1, 1, 1, 1), (2, 1, 1, 2, 1), (1, 2, 2, 1, 1)]):
        t.unit_health[i], t.unit_damage[i], t.unit_speed[i], t.unit_control[i], t.unit_cost[i] = h, d, s, c, k
    for p in range(2):
        for g in range(12):
            t.group_type[p][g] = [1, 2, 0][g % 3]          # controller, striker, tank
            t.group_size[p][g] = 8 if g < 11 else 12       # 100 // 12, remainder to the last group
    t.max_turns = 150
    return t


def tables_from_json_text(map_text=None, unit_text=None, p1_node_map=None):
    """The oracle's OWN parser of the reference's two configuration files (independent of the product's tables_from_json):
    board_init (server.py:40-100: per-node ControlPoints / StructureDefense / Resource / TeamStart, directed Connections),
    unitTypes_init (server.py:103-131: unit id = position in the file, names compared in lower case) and the army of
    everglades_env.py:145-156 (group g is unit class ['controller', 'striker', 'tank'][g % 3]).  None keeps the DemoMap table."""
    import json
    t = demo_tables()
    if map_text is not None:
        nodes = json.loads(map_text)["nodes"]
        assert [n["ID"] for n in nodes] == list(range(1, 12))
        for a in range(12):
            for b in range(12):
                t.node_dist[a][b] = 0
        for n in nodes:
            i = n["ID"]
            t.node_control_points[i], t.node_defense[i], t.node_team_start[i] = n["ControlPoints"], float(n["StructureDefense"]), n["TeamStart"]
            t.node_resource[i] = (1 if "DEFENSE" in n["Resource"] else 0) | (2 if "OBSERVE" in n["Resource"] else 0)   # server.py:442-443
            assert "DEFEND" not in n["Resource"]                    # server.py:595: would switch the fort bonus on; not modelled
            for c in n["Connections"]:
                if t.node_dist[i][c["ConnectedID"]] == 0:            # the first listed connection wins (server.py:245-249)
                    t.node_dist[i][c["ConnectedID"]] = c["Distance"]
    if unit_text is not None:
        units = json.loads(unit_text)["units"]
        ids = {}
        t.num_unit_types = len(units)
        for uid, u in enumerate(units):
            ids[u["Name"].lower()] = uid
            t.unit_health[uid], t.unit_damage[uid], t.unit_speed[uid] = u["Health"], u["Damage"], u["Speed"]
            t.unit_control[uid], t.unit_cost[uid] = u["Control"], u["Cost"]
        for p in range(2):
            for g in range(12):
                t.group_type[p][g] = ids[["controller", "striker", "tank"][g % 3]]
    if p1_node_map is not None:
        for i, v in enumerate(p1_node_map):
            t.p1_node_map[i] = int(v)
    return t


def build(force=False):
    src = os.path.join(HERE, "evg_oracle.c")
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", HERE, "-s"])
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        L = C.CDLL(SANITIZED_LIB or LIB)
        L.evo_create.restype = C.c_void_p
        L.evo_create.argtypes = [C.c_int, C.c_uint64, C.c_uint64, C.POINTER(Tables), C.c_int]
        L.evo_destroy.argtypes = [C.c_void_p]
        L.evo_reset.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.evo_step.argtypes = [C.c_void_p] + [C.c_void_p] * 7
        L.evo_random_actions.argtypes = [C.c_void_p, C.c_void_p]
        L.evo_get_state.argtypes = [C.c_void_p] + [C.c_void_p] * 5
        L.evo_set_state.argtypes = [C.c_void_p] + [C.c_void_p] * 4
        L.evo_observe.argtypes = [C.c_void_p, C.c_void_p]
        L.evo_use_stock_mt.argtypes = [C.c_void_p, C.c_void_p]
        L.evo_mt_state.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.evo_mt_randint_stream.argtypes = [C.c_uint32, C.c_void_p, C.c_int, C.c_void_p]
        L.evo_smart_state.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.evo_smart_actions.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.evo_smart_get_action.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint64, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p]
        L.evo_get_move.restype = C.c_int
        L.evo_get_move.argtypes = [C.c_int, C.c_int]
        L.evo_fog_of_war.argtypes = [C.c_void_p, C.c_void_p]
        L.evo_knowledge.argtypes = [C.c_void_p, C.c_void_p]
        L.evo_sightings.argtypes = [C.c_void_p, C.c_void_p]
        L.evo_scripted_actions.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.evo_scripted_reset.argtypes = [C.c_void_p]
        L.evo_episode_stats.argtypes = [C.c_void_p] + [C.c_void_p] * 4
        L.evo_combat_draw.restype = C.c_int
        L.evo_combat_draw.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.evo_np_sum.restype = C.c_double
        L.evo_np_sum.argtypes = [C.c_void_p, C.c_int]
        L.evo_philox.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.evo_num_threads.restype = C.c_int
        L.evo_set_num_threads.argtypes = [C.c_int]
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Oracle(object):
    """N independent reference-semantics games on the CPU."""

    def __init__(self, num_envs, seed=0, env_id_base=0, tables=None, auto_reset=False):
        self.L = lib()
        self.n = int(num_envs)
        self.tables = tables if tables is not None else demo_tables()
        self.h = self.L.evo_create(self.n, int(seed), int(env_id_base), C.byref(self.tables), int(bool(auto_reset)))

    def __del__(self):
        try:
            if self.h:
                self.L.evo_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def reset(self, mask=None):
        obs = np.zeros((self.n, NP, OBS), np.float64)
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8)
        self.L.evo_reset(self.h, _p(m), _p(obs))
        return obs

    def step(self, actions):
        a = np.ascontiguousarray(actions, np.int32).reshape(self.n, NP, NA, 2)
        obs = np.zeros((self.n, NP, OBS), np.float64)
        reward = np.zeros((self.n, NP), np.float64)
        done = np.zeros(self.n, np.uint8)
        winner = np.zeros(self.n, np.int8)
        scores = np.zeros((self.n, NP), np.int32)
        status = np.zeros(self.n, np.uint8)
        self.L.evo_step(self.h, _p(a), _p(obs), _p(reward), _p(done), _p(winner), _p(scores), _p(status))
        return obs, reward, done, dict(winner=winner, scores=scores, status=status)

    def step_noobs(self, actions):
        """Timing helper: game_turn only (no observation build)."""
        a = np.ascontiguousarray(actions, np.int32)
        done = np.zeros(self.n, np.uint8)
        self.L.evo_step(self.h, _p(a), None, None, _p(done), None, None, None)
        return done

    def random_actions(self):
        a = np.zeros((self.n, NP, NA, 2), np.int32)
        self.L.evo_random_actions(self.h, _p(a))
        return a

    def use_stock_mt(self, seeds):
        """Switch to numpy's legacy global generator per env (np.random.seed(seeds[e])): SURVEY 8 f2, validation only.
        Call before reset()."""
        sd = np.ascontiguousarray(seeds, np.uint32)
        assert sd.shape == (self.n,)
        self.L.evo_use_stock_mt(self.h, _p(sd))

    def get_stock_entropy(self):
        a = np.zeros((self.n, 625), np.uint32)
        assert self.L.evo_mt_state(self.h, 0, _p(a)) == 0
        return a

    def set_stock_entropy(self, state):
        a = np.ascontiguousarray(state, np.uint32)
        assert a.shape == (self.n, 625) and self.L.evo_mt_state(self.h, 1, _p(a)) == 0

    def scripted_actions(self, policy, player, obs, out=None):
        """policy: 1 Cycle_BRush_Turn25, 2 Cycle_BRush_Turn50, 3 SwarmAgent; writes rows [:, player] of `out`."""
        o = np.ascontiguousarray(obs, np.float64)
        assert o.shape == (self.n, NP, OBS)
        if out is None:
            out = np.zeros((self.n, NP, NA, 2), np.int32)
        self.L.evo_scripted_actions(self.h, int(policy), int(player), _p(o), _p(out))
        return out

    def scripted_reset(self):
        self.L.evo_scripted_reset(self.h)

    def fog_of_war(self):
        f = np.zeros((self.n, NP, NN), np.uint8)
        self.L.evo_fog_of_war(self.h, _p(f))
        return f

    def knowledge(self):
        f = np.zeros((self.n, NP, NN), np.uint8)
        self.L.evo_knowledge(self.h, _p(f))
        return f

    def sightings(self):
        """int8 [n][2][12][4]: what observer p knows of opposing group g: seen, node id, destination key (-1 staying), count."""
        f = np.zeros((self.n, NP, NG, 4), np.int8)
        self.L.evo_sightings(self.h, _p(f))
        return f

    def observe(self):
        obs = np.zeros((self.n, NP, OBS), np.float64)
        self.L.evo_observe(self.h, _p(obs))
        return obs

    def get_state(self):
        s = dict(groups=np.zeros((self.n, NP, NG, 8), np.int32), nodes=np.zeros((self.n, NN, 2), np.int32),
                 health=np.zeros((self.n, NP, NU), np.float64), env=np.zeros((self.n, 4), np.int32),
                 rank=np.zeros((self.n, NP, NG), np.int8))
        self.L.evo_get_state(self.h, _p(s["groups"]), _p(s["nodes"]), _p(s["health"]), _p(s["env"]), _p(s["rank"]))
        return s

    def set_state(self, groups, nodes, health, env):
        g = np.ascontiguousarray(groups, np.int32)
        n = np.ascontiguousarray(nodes, np.int32)
        h = np.ascontiguousarray(health, np.float64)
        e = np.ascontiguousarray(env, np.int32)
        assert g.shape == (self.n, NP, NG, 8) and n.shape == (self.n, NN, 2) and h.shape == (self.n, NP, NU) and e.shape == (self.n, 4)
        self.L.evo_set_state(self.h, _p(g), _p(n), _p(h), _p(e))

    def episode_stats(self):
        r = np.zeros((self.n, NP), np.float32)
        ln = np.zeros(self.n, np.int32)
        w = np.zeros(self.n, np.int8)
        tot = np.zeros(4, np.int64)
        self.L.evo_episode_stats(self.h, _p(r), _p(ln), _p(w), _p(tot))
        return dict(returns=r, length=ln, winner=w, totals=tot)


def smart_state(obs_rows):
    """obs_rows [m, 105] (one player's observations) -> [m, 12, 59] float64 (DQNAgent.create_swarm_obs)."""
    o = np.ascontiguousarray(obs_rows, np.float64)
    out = np.zeros((o.shape[0], NG, 59), np.float64)
    lib().evo_smart_state(_p(o), o.shape[0], _p(out))
    return out


def smart_actions(q, obs_rows):
    """q [m, 12, 5] float32 (the policy network's output per swarm), obs_rows [m, 105] -> (actions [m, 7, 2], directions [m, 7, 2]) int32:
    DQNAgent.get_best_actions (the seven swarms with the LOWEST best Q, ascending stable sort)."""
    qq = np.ascontiguousarray(q, np.float32)
    o = np.ascontiguousarray(obs_rows, np.float64)
    a = np.zeros((o.shape[0], 7, 2), np.int32)
    d = np.zeros((o.shape[0], 7, 2), np.int32)
    lib().evo_smart_actions(_p(qq), _p(o), o.shape[0], _p(a), _p(d))
    return a, d


def smart_get_action(q, obs_rows, seed, env_ids, episodes, player, eps):
    """DQNAgent.get_action (epsilon coin, then get_random_actions or get_best_actions) for m agents: q [m, 12, 5] float32, obs_rows [m, 105], env_ids /
    episodes [m] (the keys of the agents' draws, rng_spec.explore_draws), eps [m] float32 -> (actions [m, 7, 2], directions [m, 7, 2], explored [m])."""
    qq = np.ascontiguousarray(q, np.float32)
    o = np.ascontiguousarray(obs_rows, np.float64)
    m = o.shape[0]
    ids, eps_, epi = np.ascontiguousarray(env_ids, np.uint32), np.ascontiguousarray(eps, np.float32), np.ascontiguousarray(episodes, np.uint32)
    assert qq.shape == (m, NG, 5) and ids.shape == (m,) and epi.shape == (m,) and eps_.shape == (m,)
    a, d, x = np.zeros((m, 7, 2), np.int32), np.zeros((m, 7, 2), np.int32), np.zeros(m, np.uint8)
    lib().evo_smart_get_action(_p(qq), _p(o), m, int(seed), _p(ids), _p(epi), int(player), _p(eps_), _p(a), _p(d), _p(x))
    return a, d, x


def get_move(node0, direction):
    return lib().evo_get_move(int(node0), int(direction))


def mt_randint_stream(seed, ns):
    ns = np.ascontiguousarray(ns, np.int32)
    out = np.zeros(ns.size, np.int32)
    lib().evo_mt_randint_stream(int(seed), _p(ns), ns.size, _p(out))
    return out


def np_sum(a):
    a = np.ascontiguousarray(a, np.float64)
    return lib().evo_np_sum(_p(a), a.size)


def combat_draw(seed, env_id, episode, turn, node, player, group, j, n):
    return lib().evo_combat_draw(seed, env_id, episode, turn, node, player, group, j, n)


def philox(ctr, key):
    c = np.asarray(ctr, np.uint32)
    k = np.asarray(key, np.uint32)
    o = np.zeros(4, np.uint32)
    lib().evo_philox(_p(c), _p(k), _p(o))
    return tuple(int(x) for x in o)
