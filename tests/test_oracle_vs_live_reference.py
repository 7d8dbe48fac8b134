"""CPU test, build container only: the oracle against the LIVE reference on RANDOM configurations.

tests/golden/custom_*.npz pin three hand-made non-default map / unit files.  Where the reference is mounted (/root/reference: this container and the
judge's, never the GPU box) this test goes further: it draws random configurations inside the domain include/evg.h states -- random DIRECTED graphs over the
11 nodes (1-5 outbound edges per node, distances 1..7, possibly disconnected), control points 1..511, StructureDefense with up to two decimals incl. 0,
random resource sets, bases on any two nodes, 3-4 unit types in random order with random stats (an army's total damage <= 255) -- writes them as the
reference's JSON files, lets the imported reference play keyed-random and 'wild' order streams on them (oracle/gen_golden.py's Runner: the unmodified server
with the keyed entropy source planted), and replays every turn through the oracle: observations, scores, status, float64 health bits, packed groups and
nodes, per-node list order, fog mask, knowledge levels.  Runs in a fresh process (the reference loader plants a stand-in `gym` into sys.modules).
Skipped where the reference is absent (or mounted but not importable: the children exit with 77)."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

REF = os.environ.get("EVG_REFERENCE", "/root/reference")

def _run_child(script, args, expect):
    """A fresh process; exit code 77 = the child could not IMPORT the reference (mounted but not importable here): skipped like an absent one."""
    out = subprocess.run([sys.executable, str(script)] + [str(a) for a in args], capture_output=True, text=True, timeout=900)
    if out.returncode == 77:
        pytest.skip(out.stdout.strip()[-300:])
    assert out.returncode == 0 and expect in out.stdout, (out.stdout[-2000:], out.stderr[-4000:])


_CHILD = r'''
import json, os, sys, tempfile
import numpy as np
root = sys.argv[1]
sys.path.insert(0, os.path.join(root, "oracle")); sys.path.insert(0, os.path.join(root, "tests"))
import gen_golden as gg
import oracle as om
from conftest import golden_initial_state

from custom_configs import random_config

seed0, count = int(sys.argv[2]), int(sys.argv[3])
try:
    R = gg.Runner()
except (ImportError, PermissionError, OSError) as ex:      # the reference is there but cannot be imported here: nothing to compare with
    print("REFERENCE UNAVAILABLE: %r" % (ex,)); sys.exit(77)
saved = dict(R.cfg)
tmp = tempfile.mkdtemp(prefix="evg_fuzz_")
played = 0
for c in range(count):
    rng = np.random.default_rng([seed0, c])
    mobj, uobj = random_config(rng)
    mtxt, utxt = json.dumps(mobj), json.dumps(uobj)
    mp, up = os.path.join(tmp, "m%d.json" % c), os.path.join(tmp, "u%d.json" % c)
    open(mp, "w").write(mtxt); open(up, "w").write(utxt)
    R.cfg = dict(saved, map_file=mp, unit_file=up)
    T = om.tables_from_json_text(mtxt, utxt)
    for pol in ("random", "wild"):
        seed, env_id = 5000 + c, 3 * c + (pol == "wild")
        g = R.play(pol, seed, env_id, 0, max_turns=60 if pol == "wild" else 150, obs_bound=511)
        d = {k: v[None] for k, v in g.items() if k != "length"}
        d.update(seed=np.array([seed]), env_id=np.array([env_id]), episode=np.array([0]), length=np.array([g["length"]]))
        o = om.Oracle(1, seed=seed, env_id_base=env_id, tables=T)
        obs = o.reset()
        assert np.array_equal(obs[0], g["obs"][0].astype(np.float64)), ("reset obs", c, pol)
        for t in range(g["length"]):
            obs, reward, done, info = o.step(g["actions"][t][None].astype(np.int32))
            what = (c, pol, t, mtxt, utxt)
            assert np.array_equal(obs[0], g["obs"][t + 1].astype(np.float64)), ("obs",) + what
            assert np.array_equal(info["scores"][0], g["scores"][t]) and info["status"][0] == g["status"][t] and done[0] == g["done"][t], ("scores / status",) + what
            assert np.allclose(reward[0], g["reward"][t], rtol=0, atol=1e-12), ("reward",) + what
            s = o.get_state()
            assert np.array_equal(s["health"][0], g["health"][t + 1]), ("health bits",) + what
            assert np.array_equal(s["groups"][0], g["groups"][t + 1]) and np.array_equal(s["nodes"][0], g["nodes"][t + 1]), ("state",) + what
            assert np.array_equal(s["rank"][0], g["rank"][t + 1]), ("list order",) + what
            assert np.array_equal(o.fog_of_war()[0], g["fog"][t + 1]) and np.array_equal(o.knowledge()[0], g["know"][t + 1]), ("fog / knowledge",) + what
        played += g["length"]
print("fuzz ok: %d configurations, %d turns" % (count, played))
'''


_AGENT_CHILD = r'''
import json, os, sys, tempfile
import numpy as np
root = sys.argv[1]
sys.path.insert(0, os.path.join(root, "oracle")); sys.path.insert(0, os.path.join(root, "tests"))
import gen_golden as gg
import oracle as om
from custom_configs import random_config

B = dict(swarm=("swarm_agent.py", "SwarmAgent"), cyc25=("cycle_rush_turn25.py", "Cycle_BRush_Turn25"), cyc50=("cycle_rush_turn50.py", "Cycle_BRush_Turn50"),
         dfs=("dfs_attack.py", "dfs_attack"), ctn=("cycle_target_node.py", "Cycle_Target_Node"), ctn1=("cycle_target_node1.py", "cycle_targetedNode1"),
         ctn11=("cycle_target_node11.py", "cycle_targetedNode11"), ctnp2=("cycle_target_node11P2.py", "cycle_targetedNode11P2"), bull=("bull_rush.py", "bull_rush"),
         brv1=("base_rush_v1.py", "base_rushV1"), allc=("all_cycle.py", "all_cycle"), rnd=("random_actions.py", "random_actions"),
         delay=("random_actions_delay.py", "random_actions_delay"), same=("same_commands.py", "same_commands"))
names = sorted(B)
seed0, count = int(sys.argv[2]), int(sys.argv[3])
try:
    R = gg.Runner()
except (ImportError, PermissionError, OSError) as ex:      # the reference is there but cannot be imported here: nothing to compare with
    print("REFERENCE UNAVAILABLE: %r" % (ex,)); sys.exit(77)
saved = dict(R.cfg)
tmp = tempfile.mkdtemp(prefix="evg_fuzz_agents_")
turns = 0
for c in range(count):
    rng = np.random.default_rng([seed0, c])
    mobj, uobj = random_config(rng)
    mtxt, utxt = json.dumps(mobj), json.dumps(uobj)
    mp, up = os.path.join(tmp, "m%d.json" % c), os.path.join(tmp, "u%d.json" % c)
    open(mp, "w").write(mtxt); open(up, "w").write(utxt)
    R.cfg = dict(saved, map_file=mp, unit_file=up)
    a, b = (names[int(i)] for i in rng.integers(0, len(names), 2))
    seed, env_id = 6000 + c, 2 * c + 1
    d = gg.play_agents(R, (B[a], B[b]), seed, env_id, 2)                # the reference's own agent classes, objects alive across both episodes
    pol = [gg.AGENT_POLICY[B[a][1]], gg.AGENT_POLICY[B[b][1]]]
    o = om.Oracle(1, seed=seed, env_id_base=env_id, tables=om.tables_from_json_text(mtxt, utxt))
    for ep in range(2):
        obs = o.reset()
        T = int(d["length"][ep])
        for t in range(T):
            what = (c, a, b, ep, t)
            assert np.array_equal(obs[0], d["obs"][ep, t].astype(np.float64)), ("obs",) + what
            act = np.zeros((1, 2, 7, 2), np.int32)
            o.scripted_actions(pol[0], 0, obs, act)
            o.scripted_actions(pol[1], 1, obs, act)
            assert np.array_equal(act[0], d["actions"][ep, t]), ("orders",) + what + (act[0].tolist(), d["actions"][ep, t].tolist())
            obs, rew, done, info = o.step(act)
        assert done[0] == 1 and info["status"][0] == d["status"][ep] and np.array_equal(info["scores"][0], d["scores"][ep]), ("ending", c, a, b, ep)
        turns += T
R.cfg = saved
print("agent fuzz ok: %d configurations, %d turns" % (count, turns))
'''


_SMART_CHILD = r'''
import os, sys, types
import numpy as np
root, ref = sys.argv[1], sys.argv[2]
sys.path.insert(0, os.path.join(root, "oracle")); sys.path.insert(0, ref)
import rng_spec
import oracle as om
import torch
try:
    import agents.Smart_State.DQNAgent as D
except (ImportError, PermissionError, OSError) as ex:
    print("REFERENCE UNAVAILABLE: %r" % (ex,)); sys.exit(77)

rng = np.random.default_rng(int(sys.argv[3]))
M = int(sys.argv[4])
# random but well-formed observations of one player: turn, 11 x (flags, control, opposing units), 12 x (node, type, avg health, in transit, alive)
obs = np.zeros((M, 105))
obs[:, 0] = rng.integers(0, 151, M)
for n in range(11):
    obs[:, 1 + 4 * n] = rng.integers(0, 2, M); obs[:, 2 + 4 * n] = rng.integers(0, 2, M)
    obs[:, 3 + 4 * n] = rng.integers(-500, 501, M); obs[:, 4 + 4 * n] = rng.integers(0, 101, M)
for k in range(12):
    obs[:, 45 + 5 * k] = rng.integers(1, 12, M); obs[:, 46 + 5 * k] = rng.integers(0, 3, M); obs[:, 47 + 5 * k] = rng.integers(0, 101, M)
    obs[:, 48 + 5 * k] = rng.integers(0, 2, M); obs[:, 49 + 5 * k] = rng.integers(0, 13, M)
q = rng.standard_normal((M, 12, 5)).astype(np.float32)
q[M // 2:] = np.round(q[M // 2:] * 2) / 2                                # ties between swarms and between directions
eps = rng.choice(np.array([0.0, 0.05, 0.3, 0.7, 1.0], np.float32), M)
seed, player = 99, int(rng.integers(0, 2))
cur = dict(m=0)

class _Std(object):
    def random(self):
        L = sys._getframe(1).f_locals
        return rng_spec.explore_draws(seed, cur["m"], cur["m"] % 5, int(L["obs"][0]), player)[0] / 4294967296.0

class _NpRandom(object):
    def choice(self, a, size, replace=True):
        L = sys._getframe(1).f_locals
        coin, swarms, dirs = rng_spec.explore_draws(seed, cur["m"], cur["m"] % 5, int(L["obs"][0]), player)
        return np.array(swarms if a == 12 else dirs)

class _Np(object):
    random = _NpRandom()
    def __getattr__(self, k):
        return getattr(np, k)

D.random, D.np = _Std(), _Np()
agent = D.DQNAgent.__new__(D.DQNAgent)
agent.num_nodes = 11
ns = types.SimpleNamespace(num_nodes=11)
acts, dirs, feats = np.zeros((M, 7, 2), np.int32), np.zeros((M, 7, 2), np.int32), np.zeros((M, 12, 59))
for m in range(M):
    cur["m"] = m
    agent.epsilon = float(eps[m])
    agent.policy_net = lambda so, m=m: torch.from_numpy(q[m, int(np.argmax(so[47:59]))].copy())
    a, d = agent.get_action(obs[m])
    acts[m], dirs[m] = a.astype(np.int32), d.astype(np.int32)
    al = D.DQNAgent.get_allies_on_node_data(ns, obs[m])
    for sw in range(12):
        feats[m, sw] = D.DQNAgent.create_swarm_obs(ns, sw, obs[m], al)
ga, gd, gx = om.smart_get_action(q, obs, seed, np.arange(M, dtype=np.uint32), (np.arange(M) % 5).astype(np.uint32), player, eps)
assert np.array_equal(ga, acts) and np.array_equal(gd, dirs), "get_action"
assert np.array_equal(om.smart_state(obs), feats), "create_swarm_obs"
print("smart fuzz ok: %d agent calls, %d explored" % (M, int(gx.sum())))
'''


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "agents", "Smart_State")), reason="the reference is not mounted here (it never is on the GPU box)")
def test_oracle_smart_state_agent_equals_the_live_reference_on_random_inputs(oracle_mod, tmp_path):
    """DQNAgent.create_swarm_obs and DQNAgent.get_action (epsilon coin, get_random_actions, get_best_actions) of the LIVE reference on 1 500 random
    observations / Q tensors / epsilons against the oracle's restatements (the three draws served from the keyed stream by module proxies, as in
    oracle/gen_golden.py)."""
    script = tmp_path / "fuzz_smart_child.py"
    script.write_text(_SMART_CHILD)
    _run_child(script, [ROOT, REF, 5, 1500], "smart fuzz ok: 1500 agent calls")


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "everglades-server")), reason="the reference is not mounted here (it never is on the GPU box)")
def test_oracle_bots_equal_the_live_reference_agents_on_random_configurations(oracle_mod, tmp_path):
    """The same for the scripted opponents: random pairings of the reference's own agent classes (14 of agents/State_Machine/) on random maps / unit files,
    two consecutive episodes with the agent objects kept alive -- the oracle's bots emit the same orders, turn by turn, and the games end alike.  (The bots route
    by their own DemoMap constants whatever the map is; on a random map most of their orders are rejected by the server.)"""
    script = tmp_path / "fuzz_agents_child.py"
    script.write_text(_AGENT_CHILD)
    _run_child(script, [ROOT, 20261011, 12], "agent fuzz ok: 12 configurations")


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "everglades-server")), reason="the reference is not mounted here (it never is on the GPU box)")
def test_oracle_equals_the_live_reference_on_random_configurations(oracle_mod, tmp_path):
    script = tmp_path / "fuzz_child.py"
    script.write_text(_CHILD)
    _run_child(script, [ROOT, 20261008, 10], "fuzz ok: 10 configurations")
