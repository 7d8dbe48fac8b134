#!/usr/bin/env python3
"""Golden-fixture generator (TEST INFRASTRUCTURE; runs ONLY in the build container).

Imports the unmodified reference from /root/reference, plants the build's counter-based
RNG (oracle/rng_spec.py) in place of the module-global `np.random.randint` the reference
draws from (server.py:205,338,562), plays seeded games under several action policies and
writes inputs + expected outputs as small .npz files under tests/golden/.  Nothing of the
reference itself is written: the fixtures hold action streams, observations, rewards,
scores, status and state snapshots (numbers), which is what the parity tests replay
through the C oracle and through the HIP path.

    python oracle/gen_golden.py            # regenerate everything but the two 10 000-match samples (about 16 minutes, 12 of them the custom_var* fixtures)
    EVG_GOLDEN_ONLY=matches python oracle/gen_golden.py    # tests/golden/matches_10k.npz (about 10 minutes on 6 cores)
    EVG_GOLDEN_ONLY=matches5 python oracle/gen_golden.py   # tests/golden/matches_config5_10k.npz (Cycle_BRush_Turn25 vs SwarmAgent)
    EVG_GOLDEN_ONLY=custom python oracle/gen_golden.py     # tests/golden/custom_var{A,B,C}.npz (non-default map / unit files; 4 minutes)

Loader recipe: SURVEY.md Appendix B (np.int alias; a stub `gym` package so the real
gym_everglades/envs/everglades_env.py imports unmodified).
"""
import os
import sys
import types
import time
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("EVG_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, HERE)
import rng_spec  # noqa: E402

NG, NN, NU, NA = 12, 11, 100, 7
P1MAP = [0, 11, 8, 9, 10, 5, 6, 7, 2, 3, 4, 1]


# ----------------------------------------------------------------------------------------------
# reference loading
# ----------------------------------------------------------------------------------------------
def _install_fake_gym():
    """Minimal stand-in for the absent `gym` package: just enough names for the reference's
    everglades_env.py / everglades_renderer.py / __init__.py to import.  No behaviour."""
    gym = types.ModuleType("gym")

    class Env(object):
        pass

    class _Space(object):
        def __init__(self, *a, **k):
            self.args, self.kwargs = a, k
            low = k.get("low")
            self.shape = None if low is None else np.asarray(low).shape

    gym.Env = Env
    spaces = types.ModuleType("gym.spaces")
    spaces.Tuple = spaces.Discrete = spaces.Box = _Space
    error = types.ModuleType("gym.error")
    utils = types.ModuleType("gym.utils")
    seeding = types.ModuleType("gym.utils.seeding")
    utils.seeding = seeding
    envs = types.ModuleType("gym.envs")
    cc = types.ModuleType("gym.envs.classic_control")
    rendering = types.ModuleType("gym.envs.classic_control.rendering")
    cc.rendering = rendering
    reg = types.ModuleType("gym.envs.registration")
    reg.register = lambda **k: None
    envs.classic_control, envs.registration = cc, reg
    gym.spaces, gym.error, gym.utils, gym.envs = spaces, error, utils, envs
    for name, mod in [("gym", gym), ("gym.spaces", spaces), ("gym.error", error), ("gym.utils", utils),
                      ("gym.utils.seeding", seeding), ("gym.envs", envs), ("gym.envs.classic_control", cc),
                      ("gym.envs.classic_control.rendering", rendering), ("gym.envs.registration", reg)]:
        sys.modules[name] = mod


class _RandomProxy(object):
    """Replaces `np.random` inside everglades_server.server: randint(n) becomes the keyed draw."""

    def __init__(self, owner):
        self._o = owner

    def randint(self, n):
        fr = sys._getframe(1)
        if fr.f_code.co_name != "combat":
            return 0  # the two focus draws (game_init / game_end) are unobservable
        L = fr.f_locals
        o = self._o
        o.draws += 1
        return rng_spec.combat_draw(o.seed, o.env_id, o.episode, L["self"].current_turn, L["node"].ID,
                                    L["pid"], int(L["gid"]), int(L["j"]), int(n))

    def __getattr__(self, k):
        return getattr(np.random, k)


class _NpProxy(object):
    def __init__(self):
        self.seed = self.env_id = self.episode = 0
        self.draws = 0
        self.random = _RandomProxy(self)

    def __getattr__(self, k):
        return getattr(np, k)


def load_reference():
    np.int = int  # removed from numpy>=1.24; the reference uses it (server.py:55,79,430,471)
    _install_fake_gym()
    sys.path.insert(0, os.path.join(REF, "everglades-server"))
    sys.path.insert(0, os.path.join(REF, "gym-everglades"))
    from everglades_server import server
    from gym_everglades.envs.everglades_env import EvergladesEnv
    proxy = _NpProxy()
    server.np = proxy
    return server, EvergladesEnv, proxy


# ----------------------------------------------------------------------------------------------
# state extraction from the reference's objects
# ----------------------------------------------------------------------------------------------
GF_LOC, GF_DEST, GF_DIST, GF_READY, GF_MOVING, GF_DESTROYED, GF_COUNT, GF_STAMP = range(8)


class FogTap(object):
    """Captures the local `valid_nodes` of EvergladesGame.board_state (server.py:402-425: the fog-of-war mask the
    reference computes and then never applies) with a profiler hook on the function's return."""

    def __init__(self):
        self.last = {}

    def _hook(self, frame, event, arg):
        if event == "return" and frame.f_code.co_name == "board_state":
            L = frame.f_locals
            if "valid_nodes" in L:
                self.last[int(L["player_num"])] = [int(bool(v)) for v in L["valid_nodes"]]

    def __enter__(self):
        sys.setprofile(self._hook)
        return self

    def __exit__(self, *a):
        sys.setprofile(None)

    def snapshot(self):
        return np.array([self.last[0], self.last[1]], np.uint8)


def knowledge_levels(game):
    """The per-node knowledge levels (0 none, 1 partial, 2 full) and the opposing-group sightings `opp_k` that
    build_knowledge_output (server.py:769-907) computes into locals and only formats into dropped strings.  The function
    mutates nothing, so it is called once per player (team_starts temporarily narrowed to that player) with a profiler
    hook reading the locals at return.  Returns (levels uint8 [2][11], sightings int8 [2][12][4]): the sightings of
    observer pid are opp_k flattened in its own iteration order -- one row (node id, destination key, unit type id,
    unit count) per reported opposing group, rows of -2 after the last."""
    out = np.zeros((2, NN), np.uint8)
    sight = np.full((2, NG, 4), -2, np.int8)
    saved = game.team_starts
    for pid in (0, 1):
        box = {}

        def hook(frame, event, arg, _b=box):
            if event == "return" and frame.f_code.co_name == "build_knowledge_output":
                _b["k"] = list(frame.f_locals["knowledge"])
                _b["opp_k"] = {nid: {dst: dict(unitTypes=list(v["unitTypes"]), unitCount=list(v["unitCount"])) for dst, v in dd.items()}
                               for nid, dd in frame.f_locals["opp_k"].items()}

        game.team_starts = {pid: saved[pid]}
        sys.setprofile(hook)
        try:
            game.build_knowledge_output()
        finally:
            sys.setprofile(None)
            game.team_starts = saved
        out[pid] = box["k"]
        rows = [(nid, dst, game.unit_names[ut.lower()], uc) for nid, dd in box["opp_k"].items() for dst, v in dd.items()
                for ut, uc in zip(v["unitTypes"], v["unitCount"])]
        assert len(rows) <= NG
        for i, r in enumerate(rows):
            sight[pid, i] = r
    return out, sight


class Tracker(object):
    """Follows one reference game and dumps the canonical state (tests/README of the layout:
    groups[2][12][8] = loc,dest,dist,ready,moving,destroyed,count,stamp; nodes[11][2] =
    controlState,controlledBy; health[2][100]; rank[2][12] = index in its node's list or -1)."""

    def __init__(self, game):
        self.game = game
        self.stamp = np.zeros((2, NG), np.int16)
        self.prev_loc = np.array([[g.location for g in game.players[p].groups] for p in (0, 1)])

    def after_turn(self):
        g = self.game
        for p in (0, 1):
            for k, grp in enumerate(g.players[p].groups):
                if grp.location != self.prev_loc[p, k]:
                    self.stamp[p, k] = g.current_turn
                    self.prev_loc[p, k] = grp.location

    def snapshot(self):
        g = self.game
        groups = np.zeros((2, NG, 8), np.int16)
        health = np.zeros((2, NU), np.float64)
        rank = -np.ones((2, NG), np.int8)
        for p in (0, 1):
            off = 0
            for k, grp in enumerate(g.players[p].groups):
                u = grp.units[0]
                groups[p, k] = [grp.location, grp.travel_destination, grp.distance_remaining, int(grp.ready),
                                int(grp.moving), int(grp.destroyed), u.count, self.stamp[p, k]]
                n = len(u.unitHealth)
                health[p, off:off + n] = u.unitHealth
                off += n
        nodes = np.zeros((NN, 2), np.int16)
        for node in g.evgMap.nodes:
            nodes[node.ID - 1] = [node.controlState, node.controlledBy]
            for p in (0, 1):
                lst = list(node.groups[p])
                for r, gid in enumerate(lst):
                    rank[p, gid] = r
                # invariant behind the SoA encoding: list order == (arrival stamp, gid) order
                keys = [(int(self.stamp[p, gid]), gid) for gid in lst]
                assert keys == sorted(keys), ("list order != stamp order", node.ID, p, lst, keys)
        return groups, nodes, health, rank


# ----------------------------------------------------------------------------------------------
# action policies (inputs only; every stream is stored in the fixture)
# ----------------------------------------------------------------------------------------------
ADJ = {1: {2: 6, 4: 6}, 2: {1: 6, 3: 4, 5: 4}, 3: {2: 4, 4: 4, 5: 6, 6: 3, 7: 6}, 4: {1: 6, 3: 4, 7: 4},
       5: {2: 4, 3: 6, 8: 4, 9: 6}, 6: {3: 3, 9: 3}, 7: {3: 6, 4: 4, 9: 6, 10: 4}, 8: {5: 4, 9: 4, 11: 6},
       9: {5: 6, 6: 3, 7: 6, 8: 4, 10: 4}, 10: {7: 4, 9: 4, 11: 6}, 11: {8: 6, 10: 6}}


def _next_hop(src, dst):
    import heapq
    dist, prev, pq = {src: 0}, {}, [(0, src)]
    while pq:
        d, u = heapq.heappop(pq)
        if d > dist.get(u, 1e9):
            continue
        for v, w in ADJ[u].items():
            if d + w < dist.get(v, 1e9):
                dist[v], prev[v] = d + w, u
                heapq.heappush(pq, (d + w, v))
    if src == dst:
        return src
    v = dst
    while prev[v] != src:
        v = prev[v]
    return v


HOP = {(s, d): _next_hop(s, d) for s in ADJ for d in ADJ}


def pol_random(ctx, p, obs):
    rows = rng_spec.random_action_rows(ctx["seed"], ctx["env_id"], ctx["episode"], int(obs[0]), p)
    return np.array(rows, dtype=np.float64)


def _march(obs, target, rows, rot):
    loc = obs[45::5].astype(int)
    mov = obs[48::5].astype(int)
    alive = obs[49::5].astype(int)
    out = []
    for k in range(NG):
        g = (k + rot) % NG
        if not mov[g] and alive[g] > 0 and loc[g] != target:
            out.append((g, HOP[(int(loc[g]), target)]))
    out = out[:rows]
    while len(out) < rows:
        out.append((0, 0))
    return np.array(out, dtype=np.float64)


def pol_rush(ctx, p, obs):
    return _march(obs, 11, 7, int(obs[0]))


def pol_brawl(ctx, p, obs):
    return _march(obs, 6, 12, 0)  # 12 rows submitted, only the first 7 are honoured (server.py:227)


def pol_wild(ctx, p, obs):
    r = ctx["nprng"]
    rows = int(r.integers(5, 10))
    a = np.zeros((rows, 2))
    a[:, 0] = r.integers(-12, 12, rows)      # negative ids index from the end in the reference's Python lists
    a[:, 1] = r.integers(-12, 12, rows)
    if r.random() < 0.5:  # duplicate group ids
        a[r.integers(0, rows), 0] = a[0, 0]
    frac = r.random((rows, 2)) * 0.999
    a = a + np.sign(a + 0.5) * frac * (r.random((rows, 2)) < 0.3)  # fractional floats truncate toward zero (server.py:232)
    return a


def pol_zero(ctx, p, obs):
    return np.zeros((7, 2))


def _march_hops(obs, target, rows, rot, hop):
    """_march on another map: `hop` [src][dst] is the next-hop table in the player's own numbering (custom_configs.own_view_hops)."""
    loc = obs[45::5].astype(int)
    mov = obs[48::5].astype(int)
    alive = obs[49::5].astype(int)
    out = []
    for k in range(NG):
        g = (k + rot) % NG
        if not mov[g] and alive[g] > 0 and loc[g] != target and hop[int(loc[g])][target]:
            out.append((g, hop[int(loc[g])][target]))
    out = out[:rows]
    while len(out) < rows:
        out.append((0, 0))
    return np.array(out, dtype=np.float64)


def pol_rush_v(ctx, p, obs):
    return _march_hops(obs, ctx["variant"]["rush_target"][p], 7, int(obs[0]), ctx["variant"]["hops"][p])


def pol_brawl_v(ctx, p, obs):
    return _march_hops(obs, ctx["variant"]["brawl_own"][p], 12, 0, ctx["variant"]["hops"][p])


POLICIES = {"random": (pol_random, pol_random), "rush_variant": (pol_rush_v, pol_rush_v), "brawl_variant": (pol_brawl_v, pol_brawl_v),
            "brawl_variant_v_random": (pol_brawl_v, pol_random), "wild": (pol_wild, pol_wild), "rush": (pol_rush, pol_rush),
            "brawl": (pol_brawl, pol_brawl), "brawl_v_random": (pol_brawl, pol_random),
            "rush_v_random": (pol_rush, pol_random)}


def canon_actions(a):
    """What the server does to a player's array before using it: first 7 rows, astype(int)
    (server.py:227,232); shorter arrays are padded with the always-invalid order (0,0)."""
    a = np.asarray(a)[:7].astype(int)
    out = np.zeros((7, 2), np.int8)
    out[:len(a)] = a
    return out


# ----------------------------------------------------------------------------------------------
# game driver
# ----------------------------------------------------------------------------------------------
class Runner(object):
    def __init__(self):
        self.server, self.EnvCls, self.proxy = load_reference()
        self.cfg = dict(config_dir=os.path.join(REF, "config") + "/", map_file=os.path.join(REF, "config", "DemoMap.json"),
                        unit_file=os.path.join(REF, "config", "UnitDefinitions.json"), output_dir="/tmp/evg_unused/",
                        pnames={0: "a", 1: "b"}, debug=False)
        self.env = self.EnvCls()
        self.EnvCls.render = lambda self_, mode="human": None
        import gym_everglades.envs.everglades_env as ee

        class _NoRenderer(object):
            def __init__(self, game):
                pass

            def render(self, mode="human"):
                pass

            def close(self):
                pass

        ee.EvergladesRenderer = _NoRenderer

    def reset(self, seed, env_id, episode):
        self.proxy.seed, self.proxy.env_id, self.proxy.episode = seed, env_id, episode
        obs = self.env.reset(players={0: None, 1: None}, **self.cfg)
        return obs

    def play(self, policy, seed, env_id, episode=0, full=True, pre_edit=None, script=None, max_turns=400, variant=None, obs_bound=500):
        """Returns a dict of per-turn arrays.  `script`: optional list of (a0, a1) overriding the policy."""
        tap = FogTap()
        with tap:
            obs = self.reset(seed, env_id, episode)
            game = self.env.game
            if pre_edit is not None:
                pre_edit(game)
            obs = self.env._build_observations()
        tr = Tracker(game)
        ctx = [dict(seed=seed, env_id=env_id, episode=episode, nprng=np.random.default_rng([seed, env_id, p]), variant=variant)
               for p in (0, 1)]
        pols = POLICIES[policy] if policy in POLICIES else (pol_zero, pol_zero)
        rec = dict(obs=[np.stack([obs[0], obs[1]])], actions=[], raw0=[], raw1=[], reward=[], done=[], scores=[],
                   status=[], groups=[], nodes=[], health=[], rank=[], fog=[tap.snapshot()], know=[], sight=[])
        kl, sg = knowledge_levels(game)
        rec["know"].append(kl), rec["sight"].append(sg)
        g, n, h, r = tr.snapshot()
        rec["groups"].append(g), rec["nodes"].append(n), rec["health"].append(h), rec["rank"].append(r)
        done, t = 0, 0
        while not done and t < max_turns:
            if script is not None and t < len(script):
                a0, a1 = (np.asarray(x, dtype=np.float64) for x in script[t])
            else:
                a0, a1 = pols[0](ctx[0], 0, obs[0]), pols[1](ctx[1], 1, obs[1])
            # scores/status are locals of step(); recover them by wrapping game_turn for this call
            box = {}
            orig = game.game_turn

            def wrapped(actions, _o=orig, _b=box):
                s, st = _o(actions)
                _b["scores"], _b["status"] = (int(s[0]), int(s[1])), int(st)
                return s, st

            game.game_turn = wrapped
            with tap:
                obs, reward, done, info = self.env.step({0: a0, 1: a1})
            game.game_turn = orig
            rec["fog"].append(tap.snapshot())
            kl, sg = knowledge_levels(game)
            rec["know"].append(kl), rec["sight"].append(sg)
            tr.after_turn()
            t += 1
            rec["actions"].append(np.stack([canon_actions(a0), canon_actions(a1)]))
            rec["obs"].append(np.stack([obs[0], obs[1]]))
            rec["reward"].append([float(reward[0]), float(reward[1])])
            rec["done"].append(int(done))
            rec["scores"].append(box["scores"])
            rec["status"].append(box["status"])
            g, n, h, r = tr.snapshot()
            rec["groups"].append(g), rec["nodes"].append(n), rec["health"].append(h), rec["rank"].append(r)
        out = dict(length=t)
        for k in ("obs", "actions", "reward", "done", "scores", "status", "groups", "nodes", "health", "rank", "fog", "know", "sight"):
            out[k] = np.array(rec[k])
        assert np.all(out["obs"] == np.round(out["obs"])) and np.abs(out["obs"]).max() <= obs_bound
        return out


def pack(games, metas, tmax=150):
    """Stack variable-length games (padded to tmax turns)."""
    G = len(games)
    d = dict(
        policy=np.array([m["policy"] for m in metas]), seed=np.array([m["seed"] for m in metas], np.uint64),
        env_id=np.array([m["env_id"] for m in metas], np.uint32), episode=np.array([m["episode"] for m in metas], np.uint32),
        length=np.array([g["length"] for g in games], np.int32),
        obs=np.zeros((G, tmax + 1, 2, 105), np.int16), actions=np.zeros((G, tmax, 2, 7, 2), np.int8),
        reward=np.zeros((G, tmax, 2), np.float64), done=np.zeros((G, tmax), np.uint8),
        scores=np.zeros((G, tmax, 2), np.int32), status=np.zeros((G, tmax), np.uint8),
        groups=np.zeros((G, tmax + 1, 2, NG, 8), np.int16), nodes=np.zeros((G, tmax + 1, NN, 2), np.int16),
        health=np.zeros((G, tmax + 1, 2, NU), np.float64), rank=np.zeros((G, tmax + 1, 2, NG), np.int8),
        fog=np.zeros((G, tmax + 1, 2, NN), np.uint8), know=np.zeros((G, tmax + 1, 2, NN), np.uint8),
        sight=np.full((G, tmax + 1, 2, NG, 4), -2, np.int8))
    for i, g in enumerate(games):
        T = g["length"]
        d["obs"][i, :T + 1] = g["obs"]
        for k in ("actions", "reward", "done", "scores", "status"):
            d[k][i, :T] = g[k]
        for k in ("groups", "nodes", "health", "rank", "fog", "know", "sight"):
            d[k][i, :T + 1] = g[k]
    return d


# ----------------------------------------------------------------------------------------------
# scripted agents of BASELINE config 5, loaded from the reference by file path (they import only numpy)
# ----------------------------------------------------------------------------------------------
class _AgentRandomProxy(object):
    """np.random for the scripted agents: shuffle(list) (swarm_agent.py) becomes the keyed Fisher-Yates of
    rng_spec.swarm_shuffle; choice(., 7, replace=False) (random_actions*.py) the columns of rng_spec.random_action_rows."""

    def __init__(self, owner):
        self._o = owner

    def shuffle(self, lst):
        L = sys._getframe(1).f_locals
        o = self._o
        lst[:] = rng_spec.swarm_shuffle(o.seed, o.env_id, o.episode, int(L["obs"][0]), L["self"]._evg_player, lst)

    def choice(self, a, size, replace=True):
        L = sys._getframe(1).f_locals
        o = self._o
        rows = rng_spec.random_action_rows(o.seed, o.env_id, o.episode, int(L["obs"][0]), L["self"]._evg_player)
        n = a if isinstance(a, int) else len(a)
        assert size == 7 and not replace and n in (11, 12)
        return np.array([r[0] for r in rows]) if n == 12 else np.array([r[1] for r in rows])     # 12: groups, 11: node ids 1..11


class _StdRandomProxy(object):
    """The stdlib `random` module inside random_actions_delay.py: random() becomes rng_spec.delay_uniform."""

    def __init__(self, owner):
        self._o = owner

    def random(self):
        L = sys._getframe(1).f_locals
        o = self._o
        return rng_spec.delay_uniform(o.seed, o.env_id, o.episode, int(L["obs"][0]), L["self"]._evg_player)

    def __getattr__(self, k):
        return getattr(np.random, k)


class _AgentNpProxy(object):
    def __init__(self, owner):
        self.random = _AgentRandomProxy(owner)

    def __getattr__(self, k):
        return getattr(np, k)


_agent_module_counter = [0]


def load_agent(proxy, filename, classname, player, stock=False):
    """A fresh copy of the module per agent object, so the module-global ATTACK_LIST of swarm_agent.py is per agent.
    stock=True leaves the module's own numpy / random in place (no entropy injection)."""
    import importlib.util
    _agent_module_counter[0] += 1
    spec = importlib.util.spec_from_file_location("evg_ref_agent_%d" % _agent_module_counter[0],
                                                  os.path.join(REF, "agents", "State_Machine", filename))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    if not stock:
        mod.np = _AgentNpProxy(proxy)
        if hasattr(mod, "random"):
            mod.random = _StdRandomProxy(proxy)
    cls = getattr(mod, classname)
    import inspect
    nargs = len(inspect.signature(cls.__init__).parameters) - 1
    cwd = os.getcwd()
    os.chdir(REF)                                  # random_actions*.py open ./config/<map> in their constructors
    try:
        agent = cls(NA, player) if nargs == 2 else cls(NA, player, "DemoMap.json")
    finally:
        os.chdir(cwd)
    agent._evg_player = player
    return agent


AGENT_POLICY = {"random_actions": 0, "random_actions_2": 0, "Cycle_BRush_Turn25": 1, "Cycle_BRush_Turn50": 2, "SwarmAgent": 3,
                "all_cycle": 4, "base_rushV1": 5, "bull_rush": 6, "Cycle_Target_Node": 7, "cycle_targetedNode1": 8,
                "cycle_targetedNode11": 9, "cycle_targetedNode11P2": 10, "dfs_attack": 11, "NoAction": 12,
                "random_actions_delay": 13, "same_commands": 14, "same_commands_2": 14}


def play_agents(R, seats, seed, env_id, episodes):
    """`episodes` consecutive games of one env; the two agent objects live across them (evaluate.py:85-93)."""
    agents = [load_agent(R.proxy, fn, cn, p) for p, (fn, cn) in enumerate(seats)]
    out = dict(obs=np.zeros((episodes, 151, 2, 105), np.int16), actions=np.zeros((episodes, 150, 2, 7, 2), np.int8),
               length=np.zeros(episodes, np.int32), scores=np.zeros((episodes, 2), np.int32), status=np.zeros(episodes, np.uint8))
    for ep in range(episodes):
        obs = R.reset(seed, env_id, ep)
        out["obs"][ep, 0] = np.stack([obs[0], obs[1]])
        done, t = 0, 0
        while not done:
            acts = {p: np.array(agents[p].get_action(obs[p]), dtype=np.float64) for p in (0, 1)}
            box = {}
            game = R.env.game
            orig = game.game_turn

            def wrapped(actions, _o=orig, _b=box):
                sc, st = _o(actions)
                _b["scores"], _b["status"] = (int(sc[0]), int(sc[1])), int(st)
                return sc, st

            game.game_turn = wrapped
            obs, reward, done, info = R.env.step(acts)
            game.game_turn = orig
            out["actions"][ep, t] = np.stack([canon_actions(acts[0]), canon_actions(acts[1])])
            t += 1
            out["obs"][ep, t] = np.stack([obs[0], obs[1]])
        out["length"][ep], out["scores"][ep], out["status"][ep] = t, box["scores"], box["status"]
    return out


def gen_agent_fixtures(R):
    cyc25, cyc50, swarm = ("cycle_rush_turn25.py", "Cycle_BRush_Turn25"), ("cycle_rush_turn50.py", "Cycle_BRush_Turn50"), ("swarm_agent.py", "SwarmAgent")
    plans = [((cyc25, swarm), 31, 5), ((swarm, cyc25), 32, 9), ((cyc25, swarm), 33, 11), ((swarm, cyc25), 34, 2),
             ((cyc50, swarm), 35, 4), ((swarm, swarm), 36, 6), ((cyc25, cyc25), 37, 8)]
    # the other State_Machine bots, each on both seats against varied opponents
    B = dict(rnd=("random_actions.py", "random_actions"), rnd2=("random_actions_2.py", "random_actions_2"),
             allc=("all_cycle.py", "all_cycle"), brv1=("base_rush_v1.py", "base_rushV1"), bull=("bull_rush.py", "bull_rush"),
             ctn=("cycle_target_node.py", "Cycle_Target_Node"), ctn1=("cycle_target_node1.py", "cycle_targetedNode1"),
             ctn11=("cycle_target_node11.py", "cycle_targetedNode11"), ctnp2=("cycle_target_node11P2.py", "cycle_targetedNode11P2"),
             dfs=("dfs_attack.py", "dfs_attack"), noact=("no_action.py", None), delay=("random_actions_delay.py", "random_actions_delay"),
             same=("same_commands.py", "same_commands"), same2=("same_commands_2.py", "same_commands_2"))
    import re
    noact_src = open(os.path.join(REF, "agents", "State_Machine", "no_action.py")).read()
    B["noact"] = ("no_action.py", re.search(r"^class (\w+)", noact_src, re.M).group(1))
    AGENT_POLICY[B["noact"][1]] = 12
    more = [(("rnd", "rnd2"), 41, 3), (("allc", "rnd"), 42, 7), (("rnd", "allc"), 43, 1), (("brv1", "swarm"), 44, 12), (("swarm", "brv1"), 45, 13),
            (("bull", "delay"), 46, 14), (("delay", "bull"), 47, 15), (("ctn", "rnd"), 48, 16), (("rnd", "ctn"), 49, 17),
            (("ctn1", "ctn11"), 50, 18), (("ctn11", "ctn1"), 51, 19), (("ctnp2", "ctnp2"), 52, 20), (("rnd", "ctnp2"), 53, 21),
            (("dfs", "rnd"), 54, 22), (("rnd", "dfs"), 55, 23), (("noact", "same"), 56, 24), (("same2", "noact"), 57, 25),
            (("dfs", "dfs"), 58, 26), (("brv1", "cyc25"), 59, 27)]
    B.update(cyc25=cyc25, cyc50=cyc50, swarm=swarm)
    plans += [((B[a], B[b]), seed, env_id) for (a, b), seed, env_id in more]
    games = [play_agents(R, seats, seed, env_id, 3) for seats, seed, env_id in plans]
    d = dict(policy=np.array([[AGENT_POLICY[s[1]] for s in seats] for seats, _, _ in plans], np.int32),
             seed=np.array([p[1] for p in plans], np.uint64), env_id=np.array([p[2] for p in plans], np.uint32))
    for k in games[0]:
        d[k] = np.stack([g[k] for g in games])
    np.savez_compressed(os.path.join(OUT, "agents_scripted.npz"), **d)
    print("agents:", d["length"].tolist(), d["status"].tolist(), flush=True)


def gen_stock_mt_fixture(R):
    """the reference with NO entropy injection -- np.random.seed(s) and its own np.random.randint at
    server.py:205,338,562 -- driven by action streams that do not touch numpy's generator."""
    R.server.np = np                                     # undo the proxy: the stock module-global numpy
    games, metas = [], []
    for i, pol in enumerate(["random", "random", "brawl", "brawl_v_random", "rush_v_random", "wild", "random", "brawl"]):
        seed = 7000 + 13 * i
        np.random.seed(seed)
        g = R.play(pol, seed, 100 + i, 0)
        games.append(g)
        metas.append(dict(policy=pol, seed=seed, env_id=100 + i, episode=0))
    R.server.np = R.proxy
    d = pack(games, metas)
    np.savez_compressed(os.path.join(OUT, "stock_mt.npz"), **d)
    print("stock_mt:", d["length"].tolist(), [int(s[l - 1]) for s, l in zip(d["status"], d["length"])], flush=True)


def gen_config1_fixture(R):
    """BASELINE config 1 as the reference runs it, nothing injected anywhere: np.random.seed(0), the reference's own
    random_actions agents on both seats (they draw from the SAME global numpy generator as the server, so agent and
    server draws interleave in one stream), DemoMap, two consecutive episodes in one process (evaluate.py's loop shape:
    agents built once, env.reset() per episode)."""
    R.server.np = np
    np.random.seed(0)
    agents = [load_agent(None, "random_actions.py", "random_actions", p, stock=True) for p in (0, 1)]
    EP = 2
    d = dict(obs=np.zeros((EP, 151, 2, 105), np.int16), actions=np.zeros((EP, 150, 2, 7, 2), np.int8), length=np.zeros(EP, np.int32),
             reward=np.zeros((EP, 150, 2), np.float64), done=np.zeros((EP, 150), np.uint8), seed=np.array([0], np.uint64))
    for ep in range(EP):
        obs = R.env.reset(players={0: None, 1: None}, **R.cfg)
        d["obs"][ep, 0] = np.stack([obs[0], obs[1]])
        done, t = 0, 0
        while not done:
            acts = {p: agents[p].get_action(obs[p]) for p in (0, 1)}          # demo/random_demo.py:100-103
            obs, reward, done, info = R.env.step(acts)
            d["actions"][ep, t] = np.stack([canon_actions(acts[0]), canon_actions(acts[1])])
            d["reward"][ep, t] = [reward[0], reward[1]]
            d["done"][ep, t] = done
            t += 1
            d["obs"][ep, t] = np.stack([obs[0], obs[1]])
        d["length"][ep] = t
    st = np.random.get_state()
    d["final_key"], d["final_pos"] = np.asarray(st[1], np.uint32), np.array([st[2]], np.int32)
    R.server.np = R.proxy
    np.savez_compressed(os.path.join(OUT, "config1_stock.npz"), **d)
    print("config1_stock:", d["length"].tolist(), d["reward"][np.arange(EP), d["length"] - 1].tolist(), int(st[2]), flush=True)


def gen_smart_state_fixture():
    """the per-swarm 'smart state' preprocessing of agents/Smart_State/DQNAgent.py:200-300 and the move
    table of Move_Translation.py, evaluated by the reference's own functions on observations of committed trajectories."""
    sys.path.insert(0, REF)
    import agents.Smart_State.DQNAgent as D
    ns = types.SimpleNamespace(num_nodes=NN)
    obs_list = []
    for f in ("traj_brawl_v_random.npz", "traj_rush_v_random.npz", "traj_wild.npz"):
        d = np.load(os.path.join(OUT, f))
        for g in range(2):
            for t in range(0, int(d["length"][g]) + 1, 7):
                obs_list.append(d["obs"][g, t].astype(np.float64))
    obs = np.stack(obs_list)                                   # [M, 2, 105]
    feats = np.zeros((obs.shape[0], 2, NG, D.INPUT_SIZE), np.float64)
    allies = np.zeros((obs.shape[0], 2, NN), np.float64)
    for m in range(obs.shape[0]):
        for p in range(2):
            al = D.DQNAgent.get_allies_on_node_data(ns, obs[m, p])
            allies[m, p] = al
            for sw in range(NG):
                feats[m, p, sw] = D.DQNAgent.create_swarm_obs(ns, sw, obs[m, p], al)
    moves = np.array([[D.get_move(n0, d_) for d_ in range(5)] for n0 in range(NN)], np.int32)     # [node0][direction] -> node id
    np.savez_compressed(os.path.join(OUT, "smart_state.npz"), obs=obs.astype(np.int16), features=feats, allies=allies, moves=moves)
    print("smart_state:", feats.shape, moves.tolist(), flush=True)


def gen_smart_actions_fixture():
    """SURVEY 8 f4, the other half: network output -> orders.  DQNAgent.get_best_actions (agents/Smart_State/DQNAgent.py:176-198) over
    get_all_swarm_decisions / swarm_think (:214-266), get_swarm_node_number (:302-310) and Move_Translation.get_move (:85-97), run by the
    reference's own methods on an object created WITHOUT __init__ (no pickle, no network file): `policy_net` is a stub that returns the
    prepared Q row of the swarm whose one-hot id the real create_swarm_obs put into its input.  Q tensors: random float32, part of them
    quantised to a few levels so that swarms tie on best_q_value (the sort is stable and ASCENDING, and the agent keeps the FIRST seven:
    the seven swarms with the lowest best Q -- a quirk, pinned here) and directions tie inside a swarm (torch.argmax: first maximum);
    one all-equal tensor and one with +0.0 / -0.0."""
    sys.path.insert(0, REF)
    import torch
    import agents.Smart_State.DQNAgent as D
    d = np.load(os.path.join(OUT, "smart_state.npz"))
    obs = d["obs"].astype(np.float64)                         # [M, 2, 105]: the observations of the feature fixture
    M = obs.shape[0]
    rng = np.random.default_rng(20261005)
    q = rng.standard_normal((M, 2, NG, D.OUTPUT_SIZE)).astype(np.float32)
    q[M // 3: 2 * M // 3] = np.round(q[M // 3: 2 * M // 3] * 2.0) / 2.0          # few levels: ties between swarms and between directions
    q[2 * M // 3:] = np.round(q[2 * M // 3:])
    q[0] = 0.25                                                                # everything ties
    q[1, :, ::2] = 0.0
    q[1, :, 1::2] = -0.0
    agent = D.DQNAgent.__new__(D.DQNAgent)                    # no __init__: nothing is loaded
    agent.num_nodes = NN
    actions = np.zeros((M, 2, 7, 2), np.int32)
    directions = np.zeros((M, 2, 7, 2), np.int32)
    for m in range(M):
        for p in range(2):
            agent.policy_net = lambda swarm_obs, m=m, p=p: torch.from_numpy(q[m, p, int(np.argmax(swarm_obs[47:59]))].copy())
            a, dr = agent.get_best_actions(obs[m, p])
            assert a.shape == (7, 2) and np.array_equal(a, a.astype(np.int32)) and np.array_equal(dr, dr.astype(np.int32))
            actions[m, p], directions[m, p] = a.astype(np.int32), dr.astype(np.int32)
    np.savez_compressed(os.path.join(OUT, "smart_actions.npz"), obs=d["obs"], q=q, actions=actions, directions=directions)
    print("smart_actions:", q.shape, actions[0, 0].tolist(), actions[M - 1, 1].tolist(), flush=True)


def gen_smart_explore_fixture():
    """DQNAgent.get_action (agents/Smart_State/DQNAgent.py:130-146) with epsilon > 0 -- the coin `random.random() < self.epsilon`,
    then get_random_actions (:148-173: np.random.choice(12, 7, replace=False), np.random.choice(5, 7, replace=True), get_swarm_node_number, get_move) or
    get_best_actions.  Run by the reference's own methods on an object created without __init__, exactly like gen_smart_actions_fixture; the module's
    `random` and `np` are proxies that serve the three draws from the keyed stream (rng_spec.explore_draws; the agent of row m is env id m, episode
    m % 3, turn = obs[0]) -- the same technique as load_agent."""
    sys.path.insert(0, REF)
    import torch
    import agents.Smart_State.DQNAgent as D
    d = np.load(os.path.join(OUT, "smart_actions.npz"))
    obs, q = d["obs"].astype(np.float64), d["q"]
    M = obs.shape[0]
    seed = 20261007
    eps_levels = np.array([0.0, 0.05, 0.3, 0.5, 0.95, 1.0], np.float32)
    eps = eps_levels[(np.arange(M)[:, None] + 3 * np.arange(2)[None, :]) % len(eps_levels)]          # [M, 2] float32
    cur = dict(m=0, p=0)

    class _Std(object):
        def random(self):
            L = sys._getframe(1).f_locals
            return rng_spec.explore_draws(seed, cur["m"], cur["m"] % 3, int(L["obs"][0]), cur["p"])[0] / 4294967296.0

    class _NpRandom(object):
        def choice(self, a, size, replace=True):
            L = sys._getframe(1).f_locals
            coin, swarms, dirs = rng_spec.explore_draws(seed, cur["m"], cur["m"] % 3, int(L["obs"][0]), cur["p"])
            assert size == 7 and (a, replace) in ((12, False), (5, True))
            return np.array(swarms if a == 12 else dirs)

    class _Np(object):
        random = _NpRandom()

        def __getattr__(self, k):
            return getattr(np, k)

    D.random, D.np = _Std(), _Np()
    agent = D.DQNAgent.__new__(D.DQNAgent)
    agent.num_nodes = NN
    actions, directions = np.zeros((M, 2, 7, 2), np.int32), np.zeros((M, 2, 7, 2), np.int32)
    explored = np.zeros((M, 2), np.uint8)
    calls = []
    orig = D.DQNAgent.get_random_actions

    def tapped(self_, o):
        calls.append(1)
        return orig(self_, o)

    D.DQNAgent.get_random_actions = tapped
    for m in range(M):
        for p in range(2):
            cur["m"], cur["p"] = m, p
            agent.epsilon = float(eps[m, p])
            agent.policy_net = lambda swarm_obs, m=m, p=p: torch.from_numpy(q[m, p, int(np.argmax(swarm_obs[47:59]))].copy())
            n0 = len(calls)
            a, dr = agent.get_action(obs[m, p])
            explored[m, p] = len(calls) - n0
            assert a.shape == (7, 2) and np.array_equal(a, a.astype(np.int32)) and np.array_equal(dr, dr.astype(np.int32))
            actions[m, p], directions[m, p] = a.astype(np.int32), dr.astype(np.int32)
    D.DQNAgent.get_random_actions = orig
    D.random, D.np = __import__("random"), np
    np.savez_compressed(os.path.join(OUT, "smart_explore.npz"), obs=d["obs"], q=q, eps=eps, seed=np.array([seed], np.uint64),
                        episode=(np.arange(M) % 3).astype(np.uint32), actions=actions, directions=directions, explored=explored)
    print("smart_explore:", M, "rows x 2 seats, explored", int(explored.sum()), "by epsilon", {float(e): (int(explored[eps == e].sum()), int((eps == e).sum()))
                                                                                              for e in eps_levels}, flush=True)


# ----------------------------------------------------------------------------------------------
# north-star acceptance sample: 10 000 seeded random-vs-random matches played by the reference
# ----------------------------------------------------------------------------------------------
MATCH_SEED, MATCH_COUNT = 20261004, 10000
_match_runner = None


def _match_worker(span):
    """Plays games [lo, hi) on this process's own copy of the imported reference: the harness loop of
    demo/random_demo.py:90-113 / evaluate.py:127-160 with both seats = the random_actions generator contract, combat
    entropy injected as everywhere else.  Only outcomes are kept (no per-turn state)."""
    global _match_runner
    if _match_runner is None:
        _match_runner = Runner()
    R = _match_runner
    lo, hi = span
    n = hi - lo
    out = dict(length=np.zeros(n, np.int16), scores=np.zeros((n, 2), np.int32), status=np.zeros(n, np.uint8),
               reward=np.zeros((n, 2), np.float64), returns=np.zeros((n, 2), np.float64), winner=np.zeros(n, np.int8),
               obs_final_sum=np.zeros((n, 2), np.int32), alive_final=np.zeros((n, 2), np.int16), draws=np.zeros(n, np.int32))
    for i in range(n):
        env_id = lo + i
        obs = R.reset(MATCH_SEED, env_id, 0)
        game = R.env.game
        box = {}
        orig = game.game_turn

        def wrapped(actions, _o=orig, _b=box):
            s, st = _o(actions)
            _b["scores"], _b["status"] = (int(s[0]), int(s[1])), int(st)
            return s, st

        game.game_turn = wrapped
        d0 = R.proxy.draws
        done, t, ret = 0, 0, [0.0, 0.0]
        while not done:
            acts = {p: np.array(rng_spec.random_action_rows(MATCH_SEED, env_id, 0, int(obs[p][0]), p), dtype=np.float64) for p in (0, 1)}
            obs, reward, done, info = R.env.step(acts)
            ret[0] += float(reward[0]); ret[1] += float(reward[1])
            t += 1
        out["length"][i], out["scores"][i], out["status"][i] = t, box["scores"], box["status"]
        out["reward"][i] = [float(reward[0]), float(reward[1])]
        out["returns"][i] = ret
        out["winner"][i] = 0 if reward[0] > reward[1] else (2 if reward[0] == reward[1] else 1)      # evaluate.py:155-160
        out["obs_final_sum"][i] = [int(np.sum(obs[0])), int(np.sum(obs[1]))]
        out["alive_final"][i] = [int(np.sum(obs[p][49::5])) for p in (0, 1)]
        out["draws"][i] = R.proxy.draws - d0
    return lo, out


def _match_worker_config5(span):
    """BASELINE config 5 played by the reference's own agent classes: seat 0 = Cycle_BRush_Turn25 (cycle_rush_turn25.py),
    seat 1 = SwarmAgent (swarm_agent.py), fresh agent objects per game (one episode per env), the harness loop of
    evaluate.py:127-160; agent and combat entropy injected as in play_agents.  Only outcomes are kept."""
    global _match_runner
    if _match_runner is None:
        _match_runner = Runner()
    R = _match_runner
    lo, hi = span
    n = hi - lo
    out = dict(length=np.zeros(n, np.int16), scores=np.zeros((n, 2), np.int32), status=np.zeros(n, np.uint8),
               reward=np.zeros((n, 2), np.float64), returns=np.zeros((n, 2), np.float64), winner=np.zeros(n, np.int8),
               obs_final_sum=np.zeros((n, 2), np.int32), alive_final=np.zeros((n, 2), np.int16), draws=np.zeros(n, np.int32))
    seats = (("cycle_rush_turn25.py", "Cycle_BRush_Turn25"), ("swarm_agent.py", "SwarmAgent"))
    for i in range(n):
        env_id = lo + i
        agents = [load_agent(R.proxy, fn, cn, p) for p, (fn, cn) in enumerate(seats)]
        obs = R.reset(MATCH_SEED, env_id, 0)
        game = R.env.game
        box = {}
        orig = game.game_turn

        def wrapped(actions, _o=orig, _b=box):
            s, st = _o(actions)
            _b["scores"], _b["status"] = (int(s[0]), int(s[1])), int(st)
            return s, st

        game.game_turn = wrapped
        d0 = R.proxy.draws
        done, t, ret = 0, 0, [0.0, 0.0]
        while not done:
            acts = {p: np.array(agents[p].get_action(obs[p]), dtype=np.float64) for p in (0, 1)}
            obs, reward, done, info = R.env.step(acts)
            ret[0] += float(reward[0]); ret[1] += float(reward[1])
            t += 1
        out["length"][i], out["scores"][i], out["status"][i] = t, box["scores"], box["status"]
        out["reward"][i] = [float(reward[0]), float(reward[1])]
        out["returns"][i] = ret
        out["winner"][i] = 0 if reward[0] > reward[1] else (2 if reward[0] == reward[1] else 1)      # evaluate.py:155-160
        out["obs_final_sum"][i] = [int(np.sum(obs[0])), int(np.sum(obs[1]))]
        out["alive_final"][i] = [int(np.sum(obs[p][49::5])) for p in (0, 1)]
        out["draws"][i] = R.proxy.draws - d0
    return lo, out


def gen_matches_fixture(procs=6, chunk=50, worker=None, fname="matches_10k.npz"):
    """tests/golden/matches_10k.npz: BASELINE north_star "bit-identical win counts vs the CPU reference over 10 000
    seeded matches" -- env ids 0..9999 of seed MATCH_SEED, episode 0, one game each, played by the imported reference."""
    import multiprocessing as mp
    t0 = time.time()
    spans = [(lo, min(lo + chunk, MATCH_COUNT)) for lo in range(0, MATCH_COUNT, chunk)]
    parts = {}
    with mp.get_context("fork").Pool(procs) as pool:
        for k, (lo, o) in enumerate(pool.imap_unordered(worker or _match_worker, spans)):
            parts[lo] = o
            if k % 10 == 9:
                print("matches: %d / %d games, %.0fs" % ((k + 1) * chunk, MATCH_COUNT, time.time() - t0), flush=True)
    keys = list(parts[0].keys())
    d = {k: np.concatenate([parts[lo][k] for lo in sorted(parts)]) for k in keys}
    d["seed"] = np.array([MATCH_SEED], np.uint64)
    w = d["winner"]
    d["wins_p0_p1_tie"] = np.array([int((w == 0).sum()), int((w == 1).sum()), int((w == 2).sum())], np.int32)
    np.savez_compressed(os.path.join(OUT, fname), **d)
    print(fname, "wins p0/p1/tie", d["wins_p0_p1_tie"].tolist(), "status histogram", np.bincount(d["status"], minlength=4).tolist(),
          "mean length %.2f" % d["length"].mean(), "%.0fs" % (time.time() - t0), flush=True)


# ----------------------------------------------------------------------------------------------
# non-default map / unit files: the reference parses them at every reset (everglades_env.py:75-106)
# ----------------------------------------------------------------------------------------------
def gen_custom_fixtures(R, only=None):
    """tests/golden/custom_<variant>.npz: the imported reference playing on the configurations of oracle/custom_configs.py, written to a
    temporary directory in its own JSON schema and handed to EvergladesEnv.reset(map_file=, unit_file=).  Eight full-state trajectories
    (random, march-to-a-centre-node, march-to-the-enemy-base, wild) and 200 outcome-only random-vs-random games per variant; the JSON
    text the reference read is stored in the fixture, so the tests feed the SAME files to tables_from_json."""
    import json
    import tempfile
    import custom_configs as cc
    demo_map = json.load(open(os.path.join(REF, "config", "DemoMap.json")))
    demo_units = json.load(open(os.path.join(REF, "config", "UnitDefinitions.json")))
    saved_cfg = dict(R.cfg)
    tmp = tempfile.mkdtemp(prefix="evg_custom_")
    for name in cc.VARIANTS:
        if only and only != name:
            continue
        t0 = time.time()
        V = cc.VARIANTS[name]
        mobj, uobj, p1map = cc.variant_objects(name, demo_map, demo_units)
        mtxt, utxt = cc.json_text(mobj), cc.json_text(uobj)
        mpath, upath = os.path.join(tmp, name + "_map.json"), os.path.join(tmp, name + "_units.json")
        open(mpath, "w").write(mtxt), open(upath, "w").write(utxt)
        R.cfg = dict(saved_cfg, map_file=mpath, unit_file=upath)
        pre_edit = None
        if V["p1_node_map"] is not None:
            def pre_edit(game, _m=list(p1map)):
                game.p1_node_map = _m            # read at call time by _vec_convert_node (server.py:91-94)
        adj = cc.adjacency(mobj)
        ident = list(range(12))
        inv = [p1map.index(i) if i in p1map else 0 for i in range(12)]
        variant = dict(hops={0: cc.own_view_hops(adj, ident, ident), 1: cc.own_view_hops(adj, p1map, p1map)}, rush_target=V["rush_target"],
                       brawl_own={0: V["brawl_node"], 1: p1map[V["brawl_node"]]})
        del inv
        bound = max(int(n["ControlPoints"]) for n in mobj["nodes"])
        games, metas = [], []
        plan = ["random", "random", "brawl_variant", "brawl_variant", "rush_variant", "rush_variant", "brawl_variant_v_random", "wild"]
        for i, pol in enumerate(plan):
            seed, env_id, episode = 9000 + 31 * i + len(name), 5 * i + 2, i % 3
            g = R.play(pol, seed, env_id, episode, pre_edit=pre_edit, variant=variant, obs_bound=bound)
            games.append(g)
            metas.append(dict(policy=pol, seed=seed, env_id=env_id, episode=episode))
        d = pack(games, metas)
        B, seedB = 200, 20261006
        res = dict(seed=np.uint64(seedB), length=np.zeros(B, np.int32), scores=np.zeros((B, 2), np.int32), status=np.zeros(B, np.uint8),
                   reward=np.zeros((B, 2)), obs_sum=np.zeros((B, 151, 2), np.int32), health_final=np.zeros((B, 2, NU)))
        for i in range(B):
            g = R.play("random", seedB, i, 0, pre_edit=pre_edit, variant=variant, obs_bound=bound)
            T = g["length"]
            res["length"][i], res["scores"][i], res["status"][i], res["reward"][i] = T, g["scores"][T - 1], g["status"][T - 1], g["reward"][T - 1]
            res["obs_sum"][i, :T + 1] = g["obs"].astype(np.int64).sum(axis=2)
            res["health_final"][i] = g["health"][T]
        for k, v in res.items():
            d["bulk_" + k] = v
        # the text of the files the reference read; "" = the reference's own DemoMap.json / UnitDefinitions.json (not copied into the fixture)
        d["map_json"], d["unit_json"] = np.array(mtxt if V["map"] is not None else ""), np.array(utxt if V["units"] is not None else "")
        d["p1_node_map"] = np.array(p1map, np.int32)
        d["p1_node_map_edited"] = np.array(int(V["p1_node_map"] is not None), np.int32)
        np.savez_compressed(os.path.join(OUT, "custom_%s.npz" % name), **d)
        w = res["scores"]
        print("custom", name, "lengths", d["length"].tolist(), "status", [int(s[l - 1]) for s, l in zip(d["status"], d["length"])],
              "bulk status histogram", np.bincount(res["status"], minlength=4).tolist(), "wins p0/p1/tie",
              [int((w[:, 0] > w[:, 1]).sum()), int((w[:, 1] > w[:, 0]).sum()), int((w[:, 0] == w[:, 1]).sum())], "%.0fs" % (time.time() - t0), flush=True)
    R.cfg = saved_cfg


def gen_custom_agent_fixtures(R):
    """tests/golden/custom_agents.npz: the reference's own scripted agent classes playing on the NON-default maps of oracle/custom_configs.py.  None of them reads
    the map file -- each carries DemoMap's NODE_CONNECTIONS (and TAR_NODE) as a module constant -- so on varA / varB they still route by DemoMap and many of their
    orders are rejected by the server (server.py:245-252): the reference's behaviour, which the on-device bots must share.  Three episodes per pairing, agent
    objects alive across them (evaluate.py:85-93)."""
    import json
    import tempfile
    import custom_configs as cc
    demo_map = json.load(open(os.path.join(REF, "config", "DemoMap.json")))
    demo_units = json.load(open(os.path.join(REF, "config", "UnitDefinitions.json")))
    saved_cfg = dict(R.cfg)
    tmp = tempfile.mkdtemp(prefix="evg_custom_agents_")
    B = dict(swarm=("swarm_agent.py", "SwarmAgent"), cyc25=("cycle_rush_turn25.py", "Cycle_BRush_Turn25"), dfs=("dfs_attack.py", "dfs_attack"),
             ctn=("cycle_target_node.py", "Cycle_Target_Node"), ctn1=("cycle_target_node1.py", "cycle_targetedNode1"), bull=("bull_rush.py", "bull_rush"),
             brv1=("base_rush_v1.py", "base_rushV1"), rnd=("random_actions.py", "random_actions"))
    plans = [("varA", ("swarm", "cyc25"), 71, 3), ("varA", ("dfs", "swarm"), 72, 4), ("varA", ("ctn", "dfs"), 73, 5), ("varA", ("rnd", "ctn1"), 74, 6),
             ("varB", ("cyc25", "swarm"), 75, 7), ("varB", ("swarm", "dfs"), 76, 8), ("varB", ("bull", "brv1"), 77, 9), ("varB", ("dfs", "rnd"), 78, 10)]
    games = []
    for name, (a, b), seed, env_id in plans:
        mobj, uobj, _ = cc.variant_objects(name, demo_map, demo_units)
        mpath, upath = os.path.join(tmp, name + "_map.json"), os.path.join(tmp, name + "_units.json")
        open(mpath, "w").write(cc.json_text(mobj)), open(upath, "w").write(cc.json_text(uobj))
        R.cfg = dict(saved_cfg, map_file=mpath, unit_file=upath)
        games.append(play_agents(R, (B[a], B[b]), seed, env_id, 3))
    R.cfg = saved_cfg
    d = dict(variant=np.array([p[0] for p in plans]), policy=np.array([[AGENT_POLICY[B[x][1]] for x in p[1]] for p in plans], np.int32),
             seed=np.array([p[2] for p in plans], np.uint64), env_id=np.array([p[3] for p in plans], np.uint32))
    for k in games[0]:
        d[k] = np.stack([g[k] for g in games])
    np.savez_compressed(os.path.join(OUT, "custom_agents.npz"), **d)
    print("custom agents:", d["length"].tolist(), d["status"].tolist(), flush=True)


# ----------------------------------------------------------------------------------------------
def kat_script():
    """deterministic no-combat trajectory."""
    z = np.zeros((7, 2))
    a0 = z.copy(); a0[0] = [1, 2]; a0[1] = [0, 4]
    a1 = z.copy(); a1[0] = [1, 2]
    return [(a0, a1)] + [(z, z)] * 11


def edit_annihilation(game):
    """Hand-made position (server objects edited in place) that ends by Annihilation (status 3,
    server.py:324-325): each side is reduced to one 1-health unit, both at node 6, so the two
    simultaneous hits kill both armies on the next turn."""
    for p in (0, 1):
        for k, grp in enumerate(game.players[p].groups):
            u = grp.units[0]
            if k == 1:  # a striker group (damage 2)
                u.unitHealth[:] = 0.0
                u.unitHealth[3] = 5.0
                u.count = 1
                old = [n for n in game.evgMap.nodes if n.ID == grp.location][0]
                old.groups[p].remove(k)
                grp.location = 6
                [n for n in game.evgMap.nodes if n.ID == 6][0].groups[p].append(k)
            else:
                u.unitHealth[:] = 0.0
                u.count = 0
                grp.destroyed = True
                old = [n for n in game.evgMap.nodes if n.ID == grp.location][0]
                old.groups[p].remove(k)


def main():
    os.makedirs(OUT, exist_ok=True)
    t0 = time.time()
    if os.environ.get("EVG_GOLDEN_ONLY") == "matches":       # workers load their own copy of the reference
        gen_matches_fixture(procs=int(os.environ.get("EVG_GOLDEN_PROCS", "6")))
        return
    R = Runner()
    stats = {}
    if os.environ.get("EVG_GOLDEN_ONLY") == "matches5":      # BASELINE config 5, the reference's own agent classes
        gen_matches_fixture(worker=_match_worker_config5, fname="matches_config5_10k.npz")
        return
    if os.environ.get("EVG_GOLDEN_ONLY") == "agents":
        gen_agent_fixtures(R)
        return
    if os.environ.get("EVG_GOLDEN_ONLY") == "stock":
        gen_stock_mt_fixture(R)
        return
    if os.environ.get("EVG_GOLDEN_ONLY") == "config1":
        gen_config1_fixture(R)
        return
    if os.environ.get("EVG_GOLDEN_ONLY") == "custom_agents":
        gen_custom_agent_fixtures(R)
        return
    if (os.environ.get("EVG_GOLDEN_ONLY") or "").startswith("custom"):      # custom | custom:varA
        gen_custom_fixtures(R, only=(os.environ["EVG_GOLDEN_ONLY"].split(":") + [None])[1])
        return
    if os.environ.get("EVG_GOLDEN_ONLY") == "smart":
        gen_smart_state_fixture()
        gen_smart_actions_fixture()
        gen_smart_explore_fixture()
        return
    if os.environ.get("EVG_GOLDEN_ONLY") == "explore":
        gen_smart_explore_fixture()
        return
    only = os.environ.get("EVG_GOLDEN_ONLY")

    # 1. full-state trajectories, several policies
    plan = [("random", 6), ("wild", 6), ("rush", 4), ("brawl", 4), ("brawl_v_random", 4), ("rush_v_random", 4)]
    for pi, (pol, cnt) in enumerate(plan):
        if only and only != pol:
            continue
        games, metas = [], []
        for i in range(cnt):
            seed, env_id, episode = 1000 + 17 * pi + i, 7 * i + pi, i % 3
            g = R.play(pol, seed, env_id, episode)
            games.append(g)
            metas.append(dict(policy=pol, seed=seed, env_id=env_id, episode=episode))
        d = pack(games, metas)
        np.savez_compressed(os.path.join(OUT, "traj_%s.npz" % pol), **d)
        stats[pol] = (d["length"].tolist(), [int(s[l - 1]) for s, l in zip(d["status"], d["length"])], R.proxy.draws)
        print(pol, stats[pol], "%.1fs" % (time.time() - t0), flush=True)

    if only:
        return
    # 2. deterministic KAT (no RNG involved)
    g = R.play("zero", 1, 0, 0, script=kat_script(), max_turns=12)
    np.savez_compressed(os.path.join(OUT, "kat_nocombat.npz"), **pack([g], [dict(policy="kat", seed=1, env_id=0, episode=0)], tmax=12))

    # 3. edited position -> Annihilation
    g = R.play("zero", 5, 3, 0, pre_edit=edit_annihilation, max_turns=3)
    assert g["status"][g["length"] - 1] == 3, g["status"]
    np.savez_compressed(os.path.join(OUT, "edit_annihilation.npz"),
                        **pack([g], [dict(policy="edit", seed=5, env_id=3, episode=0)], tmax=3))

    # 3b. scripted agents of BASELINE config 5 (the reference's own agent classes produce the action streams)
    gen_agent_fixtures(R)

    gen_smart_state_fixture()
    gen_smart_actions_fixture()
    gen_smart_explore_fixture()
    gen_stock_mt_fixture(R)
    gen_config1_fixture(R)
    gen_custom_fixtures(R)
    gen_custom_agent_fixtures(R)

    # 4. bulk random-vs-random: outcomes + per-turn checksums only
    B = 120
    seedB = 20261003
    res = dict(seed=np.uint64(seedB), env_id=np.arange(B, dtype=np.uint32), length=np.zeros(B, np.int32),
               scores=np.zeros((B, 2), np.int32), status=np.zeros(B, np.uint8), reward=np.zeros((B, 2)),
               obs_sum=np.zeros((B, 151, 2), np.int32), health_final=np.zeros((B, 2, NU)),
               alive_final=np.zeros((B, 2), np.int32))
    for i in range(B):
        g = R.play("random", seedB, i, 0)
        T = g["length"]
        res["length"][i] = T
        res["scores"][i] = g["scores"][T - 1]
        res["status"][i] = g["status"][T - 1]
        res["reward"][i] = g["reward"][T - 1]
        res["obs_sum"][i, :T + 1] = g["obs"].astype(np.int64).sum(axis=2)
        res["health_final"][i] = g["health"][T]
        res["alive_final"][i] = (g["health"][T] > 0).sum(axis=1)
        if i % 20 == 19:
            print("bulk", i + 1, "%.1fs" % (time.time() - t0), flush=True)
    np.savez_compressed(os.path.join(OUT, "bulk_random.npz"), **res)
    w0 = int((res["scores"][:, 0] > res["scores"][:, 1]).sum())
    w1 = int((res["scores"][:, 1] > res["scores"][:, 0]).sum())
    print("bulk wins p0/p1/tie:", w0, w1, B - w0 - w1)
    print("done in %.1fs" % (time.time() - t0))


if __name__ == "__main__":
    main()
