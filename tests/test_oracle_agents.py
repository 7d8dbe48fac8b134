"""CPU test: the oracle's restatement of the scripted agents of BASELINE config 5 (Cycle_BRush_Turn25/50,
SwarmAgent) reproduces, order by order, what the reference's own agent classes emitted in
tests/golden/agents_scripted.npz -- three consecutive episodes per env with the agent objects kept alive,
as evaluate.py:85-93 does -- while the oracle env reproduces the observations they were computed from."""
import numpy as np

from conftest import load_golden


def test_scripted_agents_match_reference(oracle_mod):
    d = load_golden("agents_scripted.npz")
    G, E = d["length"].shape
    assert set(d["status"][:4].ravel()) == {2}            # cycle rush vs swarm always ends by BaseCapture
    for g in range(G):
        o = oracle_mod.Oracle(1, seed=int(d["seed"][g]), env_id_base=int(d["env_id"][g]))
        pol = d["policy"][g]
        for ep in range(E):
            obs = o.reset()
            assert o.get_state()["env"][0, 2] == ep
            T = int(d["length"][g, ep])
            for t in range(T):
                assert np.array_equal(obs[0], d["obs"][g, ep, t].astype(np.float64)), (g, ep, t)
                a = np.zeros((1, 2, 7, 2), np.int32)
                o.scripted_actions(int(pol[0]), 0, obs, a)
                o.scripted_actions(int(pol[1]), 1, obs, a)
                assert np.array_equal(a[0], d["actions"][g, ep, t]), ("orders", g, ep, t, a[0].tolist(), d["actions"][g, ep, t].tolist())
                obs, rew, done, info = o.step(a)
            assert done[0] == 1 and info["status"][0] == d["status"][g, ep] and np.array_equal(info["scores"][0], d["scores"][g, ep])
            assert np.array_equal(obs[0], d["obs"][g, ep, T].astype(np.float64))


def test_scripted_agents_on_non_default_maps_match_reference(oracle_mod):
    """tests/golden/custom_agents.npz: the reference's agent classes on the maps of oracle/custom_configs.py (varA, varB).  The bots carry DemoMap's
    NODE_CONNECTIONS / TAR_NODE as module constants and never read the map file, so they keep routing by DemoMap and the server rejects what the other map does
    not connect -- the oracle's bots do the same, order by order, and its env plays the same games."""
    d = load_golden("custom_agents.npz")
    G, E = d["length"].shape
    rejected = 0
    for g in range(G):
        v = load_golden("custom_%s.npz" % str(d["variant"][g]))
        t = oracle_mod.tables_from_json_text(str(v["map_json"]), str(v["unit_json"]), v["p1_node_map"].tolist())
        o = oracle_mod.Oracle(1, seed=int(d["seed"][g]), env_id_base=int(d["env_id"][g]), tables=t)
        pol = d["policy"][g]
        for ep in range(E):
            obs = o.reset()
            T = int(d["length"][g, ep])
            for tt in range(T):
                assert np.array_equal(obs[0], d["obs"][g, ep, tt].astype(np.float64)), (g, ep, tt)
                a = np.zeros((1, 2, 7, 2), np.int32)
                o.scripted_actions(int(pol[0]), 0, obs, a)
                o.scripted_actions(int(pol[1]), 1, obs, a)
                assert np.array_equal(a[0], d["actions"][g, ep, tt]), ("orders", g, ep, tt, a[0].tolist(), d["actions"][g, ep, tt].tolist())
                before = o.get_state()["groups"][0, :, :, 3].sum()
                obs, rew, done, info = o.step(a)
                rejected += int((a[0, :, :, 1] > 0).sum() > 0 and o.get_state()["groups"][0, :, :, 3].sum() == before)
            assert done[0] == 1 and info["status"][0] == d["status"][g, ep] and np.array_equal(info["scores"][0], d["scores"][g, ep])
            assert np.array_equal(obs[0], d["obs"][g, ep, T].astype(np.float64))
    assert rejected > 50                                   # the DemoMap routing does meet edges the other maps do not have


def test_smart_state_features_match_reference(oracle_mod):
    """create_swarm_obs / get_allies_on_node_data (DQNAgent.py:200-300) and Move_Translation.get_move."""
    d = load_golden("smart_state.npz")
    obs = d["obs"].astype(np.float64)
    for p in range(2):
        f = oracle_mod.smart_state(obs[:, p])
        assert np.array_equal(f, d["features"][:, p]), p
        assert np.array_equal(f[:, 0, 23:34] * 12.0, d["allies"][:, p])
    assert [[oracle_mod.get_move(n, k) for k in range(5)] for n in range(11)] == d["moves"].tolist()
