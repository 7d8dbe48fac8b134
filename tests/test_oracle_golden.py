"""CPU tests: pin the C oracle (oracle/evg_oracle.c) against fixtures generated from the imported,
unmodified reference (oracle/gen_golden.py).  Bit-exact for every integer and for float64 health;
rewards within 1e-12 (same float64 arithmetic)."""
import numpy as np
import pytest

from conftest import load_golden, golden_initial_state, TRAJ_FILES, CUSTOM_FILES


def check_sightings(records, d, g, t):
    """opp_k of build_knowledge_output (server.py:845-907) as captured from the reference's locals: `records` [2][12][4] are
    the per-group sightings of both observers."""
    from gen_policies import sighting_rows
    for p in (0, 1):
        want = d["sight"][g, t, p]
        got = sighting_rows(records[p], d["rank"][g, t, 1 - p], d["obs"][g, t, 1 - p, 46:105:5])
        assert np.array_equal(got, want), ("opposing-group sightings", g, t, p, got.tolist(), want.tolist())


def replay(om, d, g, check_state=True, tables=None):
    seed, env_id, episode = int(d["seed"][g]), int(d["env_id"][g]), int(d["episode"][g])
    T = int(d["length"][g])
    o = om.Oracle(1, seed=seed, env_id_base=env_id, tables=tables)
    obs = o.reset()
    if episode or not np.array_equal(d["health"][g, 0], o.get_state()["health"][0]):
        # fixture starts from another episode index or from an edited position
        o.set_state(*golden_initial_state(d, g))
        obs = o.observe()
    assert np.array_equal(obs[0], d["obs"][g, 0].astype(np.float64)), "reset obs"
    assert np.array_equal(o.fog_of_war()[0], d["fog"][g, 0]), "fog-of-war mask at reset"
    assert np.array_equal(o.knowledge()[0], d["know"][g, 0]), "knowledge levels at reset"
    check_sightings(o.sightings()[0], d, g, 0)
    for t in range(T):
        obs, reward, done, info = o.step(d["actions"][g, t][None].astype(np.int32))
        assert np.array_equal(obs[0], d["obs"][g, t + 1].astype(np.float64)), ("obs", g, t)
        assert np.array_equal(info["scores"][0], d["scores"][g, t]), ("scores", g, t)
        assert info["status"][0] == d["status"][g, t], ("status", g, t)
        assert done[0] == d["done"][g, t]
        assert np.allclose(reward[0], d["reward"][g, t], rtol=0, atol=1e-12), ("reward", g, t)
        if check_state:
            s = o.get_state()
            assert np.array_equal(s["health"][0], d["health"][g, t + 1]), ("health bits", g, t)
            assert np.array_equal(s["groups"][0], d["groups"][g, t + 1]), ("groups", g, t)
            assert np.array_equal(s["nodes"][0], d["nodes"][g, t + 1]), ("nodes", g, t)
            assert np.array_equal(s["rank"][0], d["rank"][g, t + 1]), ("node list order", g, t)
            assert np.array_equal(o.fog_of_war()[0], d["fog"][g, t + 1]), ("fog-of-war mask (server.py:402-425)", g, t)
            assert np.array_equal(o.knowledge()[0], d["know"][g, t + 1]), ("knowledge levels (server.py:779-832)", g, t)
            check_sightings(o.sightings()[0], d, g, t + 1)
    return o


@pytest.mark.parametrize("fname", TRAJ_FILES)
def test_trajectories_bit_exact(oracle_mod, fname):
    d = load_golden(fname)
    for g in range(len(d["length"])):
        replay(oracle_mod, d, g)


def custom_oracle_tables(om, d):
    """Oracle tables of a custom_*.npz fixture from the JSON text the reference itself read ("" = its own DemoMap / UnitDefinitions)."""
    return om.tables_from_json_text(str(d["map_json"]) or None, str(d["unit_json"]) or None, d["p1_node_map"].tolist())


@pytest.mark.parametrize("fname", CUSTOM_FILES)
def test_non_default_map_and_unit_files_bit_exact(oracle_mod, fname):
    """EvergladesEnv.reset(map_file=, unit_file=) (everglades_env.py:75-106 -> server.py:24-131) on NON-default files: the imported
    reference played on the configurations of oracle/custom_configs.py (directed / odd distances, a one-way edge, control points up to
    511, non-dyadic defenses, moved resources and bases, reordered unit file with every stat changed, four unit types; varC: a board
    flip that is not its own inverse).  Eight full-state trajectories and 200 outcome-only games per variant."""
    d = load_golden(fname)
    t = custom_oracle_tables(oracle_mod, d)
    ends = set()
    for g in range(len(d["length"])):
        replay(oracle_mod, d, g, tables=t)
        ends.add(int(d["status"][g, d["length"][g] - 1]))
    assert 1 in ends and (2 in ends or fname == "custom_varC.npz")          # both TimeExpired and BaseCapture endings are in the fixture
    B = len(d["bulk_length"])
    o = oracle_mod.Oracle(B, seed=int(d["bulk_seed"]), env_id_base=0, tables=t)
    obs = o.reset()
    assert np.array_equal(obs.sum(axis=2).astype(np.int32), d["bulk_obs_sum"][:, 0])
    fs, fst, fr, fh = np.zeros((B, 2), np.int32), np.zeros(B, np.uint8), np.zeros((B, 2)), np.zeros((B, 2, 100))
    for tt in range(150):
        obs, reward, done, info = o.step(o.random_actions())
        live = d["bulk_length"] > tt
        assert np.array_equal(obs.sum(axis=2).astype(np.int32)[live], d["bulk_obs_sum"][live, tt + 1]), tt
        e = d["bulk_length"] == tt + 1
        fs[e], fst[e], fr[e], fh[e] = info["scores"][e], info["status"][e], reward[e], o.get_state()["health"][e]
    assert np.array_equal(fs, d["bulk_scores"]) and np.array_equal(fst, d["bulk_status"]) and np.array_equal(fh, d["bulk_health_final"])
    assert np.allclose(fr, d["bulk_reward"], rtol=0, atol=1e-12)


def test_product_and_oracle_parse_the_same_tables(oracle_mod, tmp_path):
    """The product's tables_from_json (host code, no device needed) and the oracle's own parser agree byte for byte on every variant."""
    import ctypes as C
    import everglades_amd
    for fname in CUSTOM_FILES:
        d = load_golden(fname)
        kw = {}
        for key, arg in (("map_json", "map_file"), ("unit_json", "unit_file")):
            if str(d[key]):
                path = tmp_path / (fname + "_" + arg + ".json")
                path.write_text(str(d[key]))
                kw[arg] = str(path)
        t = everglades_amd.tables_from_json(p1_node_map=d["p1_node_map"].tolist(), **kw)
        ot = custom_oracle_tables(oracle_mod, d)
        assert C.sizeof(t) == C.sizeof(ot) and bytes(t) == bytes(ot), fname


def test_kat_values_from_survey(oracle_mod):
    """The hand-checked numbers of SURVEY.md section 8c for the no-combat trajectory."""
    d = load_golden("kat_nocombat.npz")
    sc = d["scores"][0]
    assert [tuple(sc[t]) for t in range(8)] == [(1100, 1100)] * 3 + [(1108, 1108), (1116, 1116), (1124, 1124),
                                                                     (1148, 1132), (1172, 1140)]
    p1 = d["obs"][0, 8, 1, :45].tolist()
    assert p1 == [8, 0, 0, -500, 0, 0, 1, -40, 0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 1, 40, 8,
                  0, 0, 0, 0, 1, 0, 32, 8, 0, 0, 500, 84]
    o = replay(oracle_mod, d, 0)
    assert o.get_state()["env"][0, 0] == 12


def test_reset_observation(oracle_mod):
    """everglades_env.py reset observation."""
    o = oracle_mod.Oracle(3, seed=9)
    obs = o.reset()
    p0 = obs[0, 0]
    assert p0[0] == 0 and p0[1:5].tolist() == [0, 0, 500, 0] and p0[41:45].tolist() == [0, 0, -500, 100]
    assert p0[45:50].tolist() == [1, 1, 100, 0, 8] and p0[50:55].tolist() == [1, 2, 100, 0, 8]
    assert p0[55:60].tolist() == [1, 0, 100, 0, 8] and p0[100:105].tolist() == [1, 0, 100, 0, 12]
    p1 = obs[0, 1]
    assert p1[3] == -500 and p1[43] == 500 and p1[44] == 100 and p1[45] == 1
    assert np.array_equal(obs[0], obs[2])


def test_annihilation_edit(oracle_mod):
    d = load_golden("edit_annihilation.npz")
    replay(oracle_mod, d, 0)
    assert d["status"][0, 0] == 3 and d["done"][0, 0] == 1
    assert d["reward"][0, 0].tolist() == [0.0, 0.0] or d["scores"][0, 0, 0] != d["scores"][0, 0, 1]


def test_bulk_random_outcomes(oracle_mod):
    """120 random-vs-random games (actions from the on-device generator contract): final scores, status,
    per-turn observation checksums and final float64 health equal the reference's."""
    d = load_golden("bulk_random.npz")
    B = len(d["length"])
    o = oracle_mod.Oracle(B, seed=int(d["seed"]), env_id_base=0)
    obs = o.reset()
    assert np.array_equal(obs.sum(axis=2).astype(np.int32), d["obs_sum"][:, 0])
    final_scores = np.zeros((B, 2), np.int32)
    final_status = np.zeros(B, np.uint8)
    for t in range(150):
        a = o.random_actions()
        obs, reward, done, info = o.step(a)
        live = d["length"] > t
        assert np.array_equal(obs.sum(axis=2).astype(np.int32)[live], d["obs_sum"][live, t + 1]), t
        ending = d["length"] == t + 1
        final_scores[ending] = info["scores"][ending]
        final_status[ending] = info["status"][ending]
        assert np.array_equal(done.astype(bool), d["length"] <= t + 1)
    assert np.array_equal(final_scores, d["scores"]) and np.array_equal(final_status, d["status"])
    s = o.get_state()
    assert np.array_equal(s["health"], d["health_final"])
    st = o.episode_stats()
    w0 = int((d["scores"][:, 0] > d["scores"][:, 1]).sum())
    w1 = int((d["scores"][:, 1] > d["scores"][:, 0]).sum())
    assert st["totals"].tolist() == [B, w0, w1, B - w0 - w1]


def test_rng_contract(oracle_mod):
    import rng_spec
    assert oracle_mod.philox((0, 0, 0, 0), (0, 0)) == (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)
    assert oracle_mod.philox((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0)) == \
        (0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1)
    r = np.random.default_rng(0)
    for _ in range(300):
        seed = int(r.integers(0, 2 ** 63)) * 2 + 1
        env_id, ep = int(r.integers(0, 2 ** 32)), int(r.integers(0, 2 ** 32))
        turn, node, pl, n = int(r.integers(1, 151)), int(r.integers(1, 12)), int(r.integers(0, 2)), int(r.integers(1, 101))
        grp, j = int(r.integers(0, 12)), int(r.integers(0, 12))
        assert oracle_mod.combat_draw(seed, env_id, ep, turn, node, pl, grp, j, n) == \
            rng_spec.combat_draw(seed, env_id, ep, turn, node, pl, grp, j, n)
    o = oracle_mod.Oracle(4, seed=77, env_id_base=1000)
    o.reset()
    a = o.random_actions()
    for e in range(4):
        for p in range(2):
            assert a[e, p].tolist() == [list(x) for x in rng_spec.random_action_rows(77, 1000 + e, 0, 0, p)]
            assert len(set(a[e, p, :, 0])) == 7 and len(set(a[e, p, :, 1])) == 7 and a[e, p, :, 1].min() >= 1


def test_np_sum_model(oracle_mod):
    """np.sum on 8- and 12-element float64 vectors == the 8-accumulator pairwise model (server.py:481)."""
    r = np.random.default_rng(1)
    for n in (8, 12, 4, 2):
        for _ in range(20000):
            a = np.where(r.random(n) < 0.3, 0.0, r.random(n) * 100.0)
            assert np.sum(a) == oracle_mod.np_sum(a)


def test_numpy_legacy_randint_model(oracle_mod):
    """np.random.seed(s); np.random.randint(n): MT19937 init_genrand + masked rejection over 32-bit outputs,
    nothing consumed for n == 1 (what the reference's three draw sites use, server.py:205,338,562)."""
    r = np.random.default_rng(3)
    for seed in (0, 1, 12345, 2 ** 32 - 1):
        ns = r.integers(1, 101, 4000).astype(np.int32)
        np.random.seed(seed)
        want = np.array([np.random.randint(int(n)) for n in ns], np.int32)
        assert np.array_equal(oracle_mod.mt_randint_stream(seed, ns), want), seed


def test_unmodified_reference_with_stock_entropy(oracle_mod):
    """SURVEY 8 f2 at oracle level: tests/golden/stock_mt.npz was produced by the reference with NO entropy injection
    (np.random.seed(s), its own np.random.randint incl. the two unobservable focus draws).  The oracle in stock mode --
    same draw ORDER over nodes, players, list-ordered groups and units -- reproduces every observation, score and the
    float64 health bit for bit."""
    d = load_golden("stock_mt.npz")
    for g in range(len(d["length"])):
        o = oracle_mod.Oracle(1, seed=0, env_id_base=0)
        o.use_stock_mt([int(d["seed"][g])])
        obs = o.reset()
        assert np.array_equal(obs[0], d["obs"][g, 0].astype(np.float64))
        for t in range(int(d["length"][g])):
            obs, reward, done, info = o.step(d["actions"][g, t][None].astype(np.int32))
            assert np.array_equal(obs[0], d["obs"][g, t + 1].astype(np.float64)), ("obs", g, t)
            assert np.array_equal(info["scores"][0], d["scores"][g, t]) and info["status"][0] == d["status"][g, t]
            s = o.get_state()
            assert np.array_equal(s["health"][0], d["health"][g, t + 1]), ("health bits", g, t)
            assert np.array_equal(s["rank"][0], d["rank"][g, t + 1])


def test_config1_shared_numpy_stream(oracle_mod):
    """BASELINE config 1 exactly as the reference runs it (tests/golden/config1_stock.npz: np.random.seed(0), the
    reference's own random_actions agents, nothing injected): agents and server draw from ONE global generator.  The
    oracle borrows that generator around every reset/step (evo_mt_state) and a stand-in agent draws from it in between:
    actions, observations, rewards and the final generator state all match the reference process over two episodes."""
    from gen_policies import NumpyGlobalRandomAgent
    d = load_golden("config1_stock.npz")
    saved = np.random.get_state()
    try:
        np.random.seed(int(d["seed"][0]))
        o = oracle_mod.Oracle(1, seed=0)
        o.use_stock_mt([0])

        def lend():
            st = np.random.get_state()
            o.set_stock_entropy(np.concatenate([st[1], [st[2]]]).astype(np.uint32)[None])
            return st

        def take_back(st):
            m = o.get_stock_entropy()[0]
            np.random.set_state((st[0], m[:624].copy(), int(m[624]), st[3], st[4]))

        agents = [NumpyGlobalRandomAgent(), NumpyGlobalRandomAgent()]
        for ep in range(len(d["length"])):
            st = lend(); obs = o.reset(); take_back(st)
            assert np.array_equal(obs[0], d["obs"][ep, 0].astype(np.float64))
            for t in range(int(d["length"][ep])):
                a = np.stack([agents[p].get_action(obs[0, p]) for p in (0, 1)]).astype(np.int32)
                assert np.array_equal(a, d["actions"][ep, t]), ("agent stream", ep, t)
                st = lend(); obs, reward, done, info = o.step(a[None]); take_back(st)
                assert np.array_equal(obs[0], d["obs"][ep, t + 1].astype(np.float64)), ("obs", ep, t)
                assert int(done[0]) == int(d["done"][ep, t])
                if not done[0]:
                    assert np.allclose(reward[0], d["reward"][ep, t], rtol=0, atol=1e-12)
        fin = np.random.get_state()
        assert np.array_equal(fin[1], d["final_key"]) and fin[2] == int(d["final_pos"][0])
    finally:
        np.random.set_state(saved)


def play_matches(make_env_step, d):
    """Shared by the CPU and the GPU test of tests/golden/matches_10k.npz: plays the 10 000 games (env ids 0..9999 of the
    fixture's seed, episode 0, random vs random from the action-generator contract, no auto-reset) and returns what the
    fixture holds per game.  make_env_step() -> (step(t) -> obs_sums [N,2], scores [N,2], status [N], reward [N,2], done [N])."""
    n = len(d["length"])
    step = make_env_step(n, int(d["seed"][0]))
    length = np.zeros(n, np.int16)
    scores, status = np.zeros((n, 2), np.int32), np.zeros(n, np.uint8)
    reward, returns = np.zeros((n, 2), np.float64), np.zeros((n, 2), np.float64)
    osum, alive = np.zeros((n, 2), np.int32), np.zeros((n, 2), np.int16)
    live = np.ones(n, bool)
    for t in range(150):
        obs_sum, alive_now, sc, stt, rew, done = step(t)
        returns[live] += rew[live]
        ending = live & (done != 0)
        length[ending] = t + 1
        scores[ending], status[ending], reward[ending] = sc[ending], stt[ending], rew[ending]
        osum[ending], alive[ending] = obs_sum[ending], alive_now[ending]
        live &= ~ending
    assert not live.any()
    winner = np.where(reward[:, 0] > reward[:, 1], 0, np.where(reward[:, 0] == reward[:, 1], 2, 1)).astype(np.int8)   # evaluate.py:155-160
    return dict(length=length, scores=scores, status=status, reward=reward, returns=returns, winner=winner, obs_final_sum=osum, alive_final=alive)


def check_matches(got, d):
    for k in ("length", "scores", "status", "winner", "obs_final_sum", "alive_final"):
        assert np.array_equal(got[k], d[k]), k
    assert np.allclose(got["reward"], d["reward"], rtol=0, atol=1e-6)
    assert np.allclose(got["returns"], d["returns"], rtol=0, atol=2e-4)      # sum of <= 150 rewards; float32 rewards on the device
    wins = [int((got["winner"] == k).sum()) for k in (0, 1, 2)]
    assert wins == d["wins_p0_p1_tie"].tolist(), ("win counts", wins, d["wins_p0_p1_tie"].tolist())


def test_ten_thousand_reference_matches(oracle_mod):
    """BASELINE north_star: "bit-identical win counts vs the CPU reference over 10 000 seeded matches".  The fixture holds the
    outcome of 10 000 random-vs-random games played by the imported reference (oracle/gen_golden.py, EVG_GOLDEN_ONLY=matches):
    the oracle reproduces every game -- length, final scores, status, terminal rewards, winner, the checksum of both final
    observations, units alive -- and hence the win/tie counts."""
    d = load_golden("matches_10k.npz")
    assert len(d["length"]) == 10000 and d["wins_p0_p1_tie"].sum() == 10000

    def make(n, seed):
        o = oracle_mod.Oracle(n, seed=seed, env_id_base=0)
        o.reset()

        def step(t):
            obs, rew, done, info = o.step(o.random_actions())
            return obs.sum(axis=2).astype(np.int32), obs[:, :, 49::5].sum(axis=2).astype(np.int16), info["scores"], info["status"], rew, done
        return step
    check_matches(play_matches(make, d), d)


def test_ten_thousand_reference_matches_config5(oracle_mod):
    """BASELINE config 5 against the reference itself: 10 000 seeded matches of the reference's own agent classes
    (cycle_rush_turn25.py Cycle_BRush_Turn25 on seat 0, swarm_agent.py SwarmAgent on seat 1; oracle/gen_golden.py,
    EVG_GOLDEN_ONLY=matches5).  The oracle's restatement of both agents and of the game reproduces every match and the win counts."""
    d = load_golden("matches_config5_10k.npz")
    assert len(d["length"]) == 10000 and d["wins_p0_p1_tie"].sum() == 10000

    def make(n, seed):
        o = oracle_mod.Oracle(n, seed=seed, env_id_base=0)
        cur = [o.reset()]

        def step(t):
            a = np.zeros((n, 2, 7, 2), np.int32)
            o.scripted_actions(1, 0, cur[0], a)          # EVG_POLICY_CYCLE_RUSH_25
            o.scripted_actions(3, 1, cur[0], a)          # EVG_POLICY_SWARM
            obs, rew, done, info = o.step(a)
            cur[0] = obs
            return obs.sum(axis=2).astype(np.int32), obs[:, :, 49::5].sum(axis=2).astype(np.int16), info["scores"], info["status"], rew, done
        return step
    check_matches(play_matches(make, d), d)


def test_smart_state_and_smart_actions_oracle_vs_reference_fixtures(oracle_mod):
    """SURVEY 8 f4, both halves, oracle == the reference's own methods: create_swarm_obs (smart_state.npz) and get_best_actions over
    swarm_think / get_swarm_node_number / Move_Translation.get_move (smart_actions.npz: random Q tensors with ties, an all-equal one,
    +0.0 / -0.0; rows = the seven swarms with the LOWEST best Q, ascending stable sort -- the reference's quirk)."""
    d = load_golden("smart_state.npz")
    for p in range(2):
        assert np.array_equal(oracle_mod.smart_state(d["obs"][:, p].astype(np.float64)), d["features"][:, p]), p
    a = load_golden("smart_actions.npz")
    assert np.array_equal(a["obs"], d["obs"]) and a["q"].dtype == np.float32 and a["q"].shape[1:] == (2, 12, 5)
    for p in range(2):
        act, dr = oracle_mod.smart_actions(a["q"][:, p], a["obs"][:, p].astype(np.float64))
        assert np.array_equal(act, a["actions"][:, p]) and np.array_equal(dr, a["directions"][:, p]), p
    # the fixture does exercise the quirks: ties between swarms (stable order decides) and rows that are NOT the seven highest Q
    bq = a["q"].max(axis=3)                                            # [M, 2, 12]
    assert (np.sort(bq, axis=2)[:, :, 6] == np.sort(bq, axis=2)[:, :, 7]).any()
    chosen_max = np.take_along_axis(bq, a["actions"][..., 0], axis=2).max(axis=2)
    assert (chosen_max <= np.sort(bq, axis=2)[:, :, 6]).all()          # every chosen swarm is among the seven lowest
    for m in range(a["q"].shape[0]):
        for p in range(2):
            for i in range(7):
                sw, node = a["actions"][m, p, i]
                assert node == oracle_mod.get_move(int(a["obs"][m, p, 45 + 5 * sw]) - 1, int(a["directions"][m, p, i, 1]))


def test_smart_get_action_with_epsilon_oracle_vs_reference_fixture(oracle_mod):
    """DQNAgent.get_action (agents/Smart_State/DQNAgent.py:130-173) with epsilon > 0: tests/golden/smart_explore.npz was produced by the reference's own
    get_action / get_random_actions / get_best_actions with its three draws served from the keyed stream (rng_spec.explore_draws).  The oracle's
    restatement reproduces which agents explored and every order / direction row; the spec in Python agrees with the oracle's C on the draws."""
    import rng_spec
    d = load_golden("smart_explore.npz")
    M = d["obs"].shape[0]
    ids = np.arange(M, dtype=np.uint32)
    for p in range(2):
        a, dr, x = oracle_mod.smart_get_action(d["q"][:, p], d["obs"][:, p].astype(np.float64), int(d["seed"][0]), ids, d["episode"], p, d["eps"][:, p])
        assert np.array_equal(x, d["explored"][:, p]) and np.array_equal(a, d["actions"][:, p]) and np.array_equal(dr, d["directions"][:, p]), p
    ex = d["explored"].astype(bool)
    assert ex.any() and (~ex).any() and not ex[d["eps"] == 0].any() and ex[d["eps"] == 1].all()
    # explored rows: 7 distinct swarms, directions in 0..4 with repeats somewhere, node = get_move(location - 1, direction)
    rows, dirs = d["actions"][ex], d["directions"][ex]
    assert all(len(set(r[:, 0])) == 7 for r in rows) and dirs[..., 1].min() >= 0 and dirs[..., 1].max() <= 4
    assert any(len(set(r[:, 1])) < 7 for r in dirs)
    obs_ex = np.stack([d["obs"][:, p] for p in range(2)], axis=1)[ex]
    for o, r, dd in zip(obs_ex, rows, dirs):
        for i in range(7):
            assert r[i, 1] == oracle_mod.get_move(int(o[45 + 5 * r[i, 0]]) - 1, int(dd[i, 1]))
    # unexplored rows are get_best_actions of the same inputs (smart_actions.npz holds them)
    b = load_golden("smart_actions.npz")
    assert np.array_equal(d["actions"][~ex], b["actions"][~ex]) and np.array_equal(d["directions"][~ex], b["directions"][~ex])
    coin, sw, di = rng_spec.explore_draws(int(d["seed"][0]), 5, int(d["episode"][5]), int(d["obs"][5, 1, 0]), 1)
    assert 0 <= coin < 2 ** 32 and len(set(sw)) == 7
