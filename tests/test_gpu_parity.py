"""GPU parity tests (run with -m gpu on the MI355X box): the HIP path, called through the C-ABI
(libevg.so via everglades_amd), against (a) the committed golden fixtures generated from the
reference and (b) the CPU oracle on the same seeded inputs.  Bit-exact for every integer
(observations, scores, status, packed state) and for float64 health; float32 rewards within 1e-6
of the oracle's float64 (tolerance of BASELINE.json north_star: 1e-5)."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import load_golden, golden_initial_state, TRAJ_FILES, CUSTOM_FILES

pytestmark = pytest.mark.gpu
REWARD_ATOL = 1e-6


@pytest.fixture(scope="module")
def evg():
    import torch
    assert torch.cuda.is_available(), "GPU tests need the MI355X"
    import everglades_amd
    return everglades_amd


def _np(t):
    return t.cpu().numpy()


def check_state(env, ora_state, what=""):
    s = env.get_state()
    for k in ("groups", "nodes", "health", "env"):
        assert np.array_equal(s[k], ora_state[k]), (what, k)


def _replay_fixture_through_abi(evg, d, fname, **env_kw):
    """Every game of a full-state trajectory fixture through the C-ABI, compared turn by turn (observations, scores, status, rewards, float64
    health bits, packed groups and nodes, fog / knowledge / sightings)."""
    from gen_policies import sighting_rows
    for g in range(len(d["length"])):
        T = int(d["length"][g])
        env = evg.EvergladesVecEnv(1, seed=int(d["seed"][g]), env_id_base=int(d["env_id"][g]), obs_dtype="float64", auto_reset=False, **env_kw)
        env.reset()
        env.set_state(*golden_initial_state(d, g))
        assert np.array_equal(_np(env.observe())[0], d["obs"][g, 0].astype(np.float64)), ("initial obs", g)
        for t in range(T):
            obs, rew, done, info = env.step(d["actions"][g, t][None].astype(np.int32))
            assert np.array_equal(_np(obs)[0], d["obs"][g, t + 1].astype(np.float64)), ("obs", fname, g, t)
            assert np.array_equal(_np(info["scores"])[0], d["scores"][g, t]), ("scores", g, t)
            assert int(info["status"][0]) == d["status"][g, t] and int(done[0]) == d["done"][g, t]
            assert np.allclose(_np(rew)[0], d["reward"][g, t], rtol=0, atol=REWARD_ATOL)
            s = env.get_state()
            assert np.array_equal(s["health"][0], d["health"][g, t + 1]), ("health bits", fname, g, t)
            assert np.array_equal(s["groups"][0], d["groups"][g, t + 1]), ("groups", fname, g, t)
            assert np.array_equal(s["nodes"][0], d["nodes"][g, t + 1]), ("nodes", fname, g, t)
            assert np.array_equal(_np(env.fog_of_war())[0], d["fog"][g, t + 1]), ("fog-of-war mask", fname, g, t)
            assert np.array_equal(_np(env.knowledge())[0], d["know"][g, t + 1]), ("knowledge levels", fname, g, t)
            sg = _np(env.sightings())[0]
            for p in (0, 1):                     # opp_k of build_knowledge_output, captured from the reference's locals
                assert np.array_equal(sighting_rows(sg[p], d["rank"][g, t + 1, 1 - p], d["obs"][g, t + 1, 1 - p, 46:105:5]),
                                      d["sight"][g, t + 1, p]), ("sightings", fname, g, t, p)
        env.close()


@pytest.mark.parametrize("fname", TRAJ_FILES + ["edit_annihilation.npz"])
def test_golden_through_abi(evg, fname):
    _replay_fixture_through_abi(evg, load_golden(fname), fname)


def _custom_fixture_files(d, fname, tmp_path):
    """The JSON text the reference read when it played a custom_*.npz fixture, written out again as files -> reset()/tables_from_json kwargs."""
    kw = {}
    for key, arg in (("map_json", "map_file"), ("unit_json", "unit_file")):
        if str(d[key]):
            path = tmp_path / (fname + "_" + arg + ".json")
            path.write_text(str(d[key]))
            kw[arg] = str(path)
    return kw


@pytest.mark.parametrize("force_ieee_div", [False, True])
@pytest.mark.parametrize("fname", CUSTOM_FILES)
def test_non_default_map_and_unit_files_through_abi(evg, fname, force_ieee_div, tmp_path):
    """The runtime-table path pinned by the REFERENCE (not only by the oracle): tests/golden/custom_*.npz were played by the imported
    reference on non-default map / unit files (oracle/custom_configs.py) handed to EvergladesEnv.reset(map_file=, unit_file=)
    (everglades_env.py:75-106 -> server.py:24-131).  Here the same files go through tables_from_json -> evg_create: eight full-state
    trajectories and 200 outcome-only games per variant, once with the quotient evg_create chooses for the table set (tabulated
    reciprocal where validated, else true division) and once with the true-division branch forced (diagnostic library)."""
    d = load_golden(fname)
    files = _custom_fixture_files(d, fname, tmp_path)
    tables = evg.tables_from_json(p1_node_map=d["p1_node_map"].tolist(), **files)
    extra = dict(library=evg._lib.DIAG_LIB_PATH, diag=dict(force_ieee_div=True)) if force_ieee_div else {}
    _replay_fixture_through_abi(evg, d, fname, tables=tables, **extra)
    B = len(d["bulk_length"])
    for form in ("stepwise", "persistent"):
        env = evg.EvergladesVecEnv(B, seed=int(d["bulk_seed"]), obs_dtype="float64", auto_reset=False, tables=tables, **extra)
        obs = env.reset()
        assert np.array_equal(_np(obs).sum(axis=2).astype(np.int32), d["bulk_obs_sum"][:, 0])
        if form == "persistent":           # no auto-reset: a finished env is frozen, so the final state of every game is still there after 150 turns
            env.rollout_random(150, turns_per_launch=150)
            assert np.array_equal(env.get_state()["health"], d["bulk_health_final"])
            st = env.episode_stats()
            assert np.array_equal(st["length"], d["bulk_length"])
            w = d["bulk_scores"]
            assert st["totals"].tolist() == [B, int((w[:, 0] > w[:, 1]).sum()), int((w[:, 1] > w[:, 0]).sum()), int((w[:, 0] == w[:, 1]).sum())]
            env.close()
            continue
        fs, fst, fr = np.zeros((B, 2), np.int32), np.zeros(B, np.uint8), np.zeros((B, 2))
        for t in range(150):
            obs, rew, done, info = env.step(env.random_actions())
            live = d["bulk_length"] > t
            assert np.array_equal(_np(obs).sum(axis=2).astype(np.int32)[live], d["bulk_obs_sum"][live, t + 1]), (fname, t)
            e = d["bulk_length"] == t + 1
            fs[e], fst[e], fr[e] = _np(info["scores"])[e], _np(info["status"])[e], _np(rew)[e]
        assert np.array_equal(fs, d["bulk_scores"]) and np.array_equal(fst, d["bulk_status"])
        assert np.allclose(fr, d["bulk_reward"], rtol=0, atol=REWARD_ATOL)
        assert np.array_equal(env.get_state()["health"], d["bulk_health_final"])
        env.close()


def test_dropin_reset_with_map_and_unit_files_replays_the_reference(evg, tmp_path):
    """The replaced interface itself: EvergladesEnv.reset(players=, config_dir=, map_file=, unit_file=, ...) of the single-game drop-in on
    the non-default files of custom_varA / custom_varB, dict actions in, dict observations / rewards out -- the games of the fixture that
    start at episode 0, against what the reference returned from the same calls."""
    for fname in ("custom_varA.npz", "custom_varB.npz"):
        d = load_golden(fname)
        files = _custom_fixture_files(d, fname, tmp_path)
        for g in range(len(d["length"])):
            if int(d["episode"][g]) != 0:
                continue
            env = evg.EvergladesEnv(seed=int(d["seed"][g]), env_id=int(d["env_id"][g]))
            obs = env.reset(players={0: None, 1: None}, config_dir=str(tmp_path) + "/", output_dir="/tmp/unused/", pnames={0: "a", 1: "b"}, debug=False, **files)
            assert np.array_equal(np.stack([obs[0], obs[1]]), d["obs"][g, 0].astype(np.float64))
            for t in range(int(d["length"][g])):
                a = d["actions"][g, t].astype(np.float64)
                obs, reward, done, _ = env.step({0: a[0], 1: a[1]})
                assert np.array_equal(np.stack([obs[0], obs[1]]), d["obs"][g, t + 1].astype(np.float64)), (fname, g, t)
                assert done == d["done"][g, t] and abs(reward[0] - d["reward"][g, t, 0]) < 1e-12 and abs(reward[1] - d["reward"][g, t, 1]) < 1e-12
            env.close()


def test_bulk_random_matches_reference(evg):
    """The 120 reference games of bulk_random.npz: on-device action generation + step reproduce the
    reference's per-turn observation checksums, final scores/status and final float64 health."""
    d = load_golden("bulk_random.npz")
    B = len(d["length"])
    env = evg.EvergladesVecEnv(B, seed=int(d["seed"]), auto_reset=False)
    obs = env.reset()
    assert np.array_equal(_np(obs).sum(axis=2).astype(np.int32), d["obs_sum"][:, 0])
    for t in range(150):
        obs, rew, done, info = env.step(env.random_actions())
        live = d["length"] > t
        assert np.array_equal(_np(obs).astype(np.int64).sum(axis=2).astype(np.int32)[live], d["obs_sum"][live, t + 1]), t
    assert np.array_equal(_np(info["scores"]), d["scores"]) and np.array_equal(_np(info["status"]), d["status"])
    assert np.array_equal(env.get_state()["health"], d["health_final"])
    st = env.episode_stats()
    w0 = int((d["scores"][:, 0] > d["scores"][:, 1]).sum())
    w1 = int((d["scores"][:, 1] > d["scores"][:, 0]).sum())
    assert st["totals"].tolist() == [B, w0, w1, B - w0 - w1]
    env.close()


@pytest.mark.parametrize("N", [1, 65, 1000])
def test_random_rollout_vs_oracle(evg, oracle_mod, N):
    """Full 150-turn random-vs-random episodes; N chosen to exercise partial workgroups."""
    seed = 4242 + N
    env = evg.EvergladesVecEnv(N, seed=seed, env_id_base=17, auto_reset=False)
    ora = oracle_mod.Oracle(N, seed=seed, env_id_base=17)
    assert np.array_equal(_np(env.reset()).astype(np.float64), ora.reset())
    for t in range(152):     # two extra steps: finished envs stay frozen and repeat their outputs
        a = env.random_actions()
        assert np.array_equal(_np(a), ora.random_actions()), ("action generator", t)
        obs, rew, done, info = env.step(a)
        o_obs, o_rew, o_done, o_info = ora.step(_np(a))
        assert np.array_equal(_np(obs).astype(np.float64), o_obs), ("obs", t)
        assert np.array_equal(_np(info["scores"]), o_info["scores"]) and np.array_equal(_np(info["status"]), o_info["status"])
        assert np.array_equal(_np(info["winner"]), o_info["winner"]) and np.array_equal(_np(done), o_done)
        assert np.allclose(_np(rew), o_rew, rtol=0, atol=REWARD_ATOL)
        if t % 25 == 0 or t >= 149:
            check_state(env, ora.get_state(), t)
            assert np.array_equal(_np(env.fog_of_war()), ora.fog_of_war()) and np.array_equal(_np(env.knowledge()),
                                                                                              ora.knowledge()) and np.array_equal(_np(env.sightings()),
                                                                                                                                  ora.sightings())
    st, ost = env.episode_stats(), ora.episode_stats()
    assert np.array_equal(st["totals"], ost["totals"]) and np.array_equal(st["winner"], ost["winner"])
    assert np.array_equal(st["length"], ost["length"]) and np.allclose(st["returns"], ost["returns"], rtol=1e-6, atol=1e-5)
    env.close()


@pytest.mark.parametrize("policy", ["rush", "brawl", "wild"])
def test_policies_with_auto_reset_vs_oracle(evg, oracle_mod, policy):
    """Branch-divergent mixes: base rushes end by BaseCapture after ~30-45 turns (auto-reset desynchronises
    the envs), brawls fight at node 6 every turn, 'wild' sends out-of-domain and duplicate orders."""
    from gen_policies import policy_actions
    N, seed = 192, 99
    env = evg.EvergladesVecEnv(N, seed=seed, auto_reset=True)
    ora = oracle_mod.Oracle(N, seed=seed, auto_reset=True)
    obs = _np(env.reset()).astype(np.float64)
    assert np.array_equal(obs, ora.reset())
    rng = np.random.default_rng(5)
    for t in range(200):
        a = policy_actions(policy, obs, t, rng)
        o, rew, done, info = env.step(a)
        o_obs, o_rew, o_done, o_info = ora.step(a)
        obs = _np(o).astype(np.float64)
        assert np.array_equal(obs, o_obs), (policy, "obs", t)
        assert np.array_equal(_np(info["scores"]), o_info["scores"]) and np.array_equal(_np(info["status"]), o_info["status"])
        assert np.array_equal(_np(done), o_done) and np.allclose(_np(rew), o_rew, rtol=0, atol=REWARD_ATOL)
        if t % 40 == 39:
            check_state(env, ora.get_state(), (policy, t))
    st, ost = env.episode_stats(), ora.episode_stats()
    assert np.array_equal(st["totals"], ost["totals"]) and st["totals"][0] > 0
    assert np.array_equal(st["winner"], ost["winner"]) and np.array_equal(st["length"], ost["length"])
    env.close()


def test_obs_dtypes_agree(evg):
    N, seed = 130, 7
    envs = [evg.EvergladesVecEnv(N, seed=seed, obs_dtype=dt, auto_reset=True) for dt in ("float32", "float64", "int16")]
    obs = [_np(e.reset()).astype(np.float64) for e in envs]
    assert np.array_equal(obs[0], obs[1]) and np.array_equal(obs[0], obs[2])
    for t in range(60):
        a = envs[0].random_actions().clone()
        outs = [_np(e.step(a)[0]).astype(np.float64) for e in envs]
        assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2]), t
    # the persistent form, in the three observation types: the small-batch kernel here (four lanes per env) ...
    outs = [_np(e.rollout_random(97, turns_per_launch=150)[0]).astype(np.float64) for e in envs]
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
    for e in envs:
        e.close()
    # ... and the two-lane kernel of large batches (a 50 000-env handle; the last 130 envs are compared)
    big = [evg.EvergladesVecEnv(50000, seed=seed, obs_dtype=dt, auto_reset=True) for dt in ("float32", "float64", "int16")]
    outs = []
    for e in big:
        e.reset()
        outs.append(_np(e.rollout_random(97, turns_per_launch=150)[0][-130:]).astype(np.float64))
        e.close()
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])


def test_masked_reset_and_state_roundtrip(evg, oracle_mod):
    N, seed = 100, 31
    env = evg.EvergladesVecEnv(N, seed=seed, auto_reset=False)
    ora = oracle_mod.Oracle(N, seed=seed)
    env.reset(); ora.reset()
    for t in range(30):
        a = env.random_actions()
        env.step(a); ora.step(_np(a))
    mask = (np.arange(N) % 3 == 0).astype(np.uint8)
    before = _np(env.obs).copy()
    o1 = _np(env.reset(mask)).astype(np.float64)
    o2 = ora.reset(mask)
    assert np.array_equal(o1[mask != 0], o2[mask != 0])
    assert np.array_equal(o1[mask == 0], before[mask == 0].astype(np.float64))       # untouched rows
    check_state(env, ora.get_state(), "after masked reset")
    # get_state -> set_state is the identity on the packed state, and observe() reproduces the observations
    s = env.get_state()
    env2 = evg.EvergladesVecEnv(N, seed=seed, auto_reset=False)
    env2.reset()
    env2.set_state(s["groups"], s["nodes"], s["health"], s["env"])
    check_state(env2, s, "roundtrip")
    assert np.array_equal(_np(env2.observe()).astype(np.float64), ora.observe())
    for t in range(20):
        a = env.random_actions()
        o_a, o_b = env.step(a)[0], env2.step(a)[0]
        assert np.array_equal(_np(o_a), _np(o_b))
    env.close(); env2.close()


def test_single_env_dropin_matches_golden(evg):
    """EvergladesEnv (dict API of everglades_env.py) replays a reference trajectory, float action arrays included."""
    d = load_golden("traj_wild.npz")
    g = 0
    env = evg.EvergladesEnv(seed=int(d["seed"][g]), env_id=int(d["env_id"][g]))
    obs = env.reset(players={0: None, 1: None}, config_dir=None, map_file=None, unit_file=None, output_dir="x", pnames={0: "a", 1: "b"}, debug=False)
    assert env.num_actions_per_turn == 7 and env.observation_space.shape == (105,)
    assert sorted(obs.keys()) == [0, 1] and obs[0].dtype == np.float64 and obs[0].shape == (105,)
    env._vec.set_state(*golden_initial_state(d, g))      # this fixture game is episode index d["episode"][g]
    for t in range(int(d["length"][g])):
        a0 = d["actions"][g, t, 0].astype(np.float64)
        acts = {0: np.where(a0 >= 0, a0 + 0.25, a0 - 0.25), 1: d["actions"][g, t, 1].astype(np.float64)}   # astype(int) truncates toward zero
        obs, reward, done, info = env.step(acts)
        assert np.array_equal(obs[0], d["obs"][g, t + 1, 0].astype(np.float64)) and np.array_equal(obs[1], d["obs"][g, t + 1, 1].astype(np.float64))
        assert reward[0] == d["reward"][g, t, 0] and reward[1] == d["reward"][g, t, 1] and done == d["done"][g, t] and info == {}


def test_full_size_properties(evg, oracle_mod):
    """BASELINE config size (65 536 games): size-independent properties after an auto-reset rollout, and
    step-by-step agreement with the oracle on a sampled sub-range (same seed, same global env ids)."""
    N, seed, steps = 65536, 12345, 160
    lo, n = 30000, 512
    env = evg.EvergladesVecEnv(N, seed=seed, auto_reset=True)
    ora = oracle_mod.Oracle(n, seed=seed, env_id_base=lo, auto_reset=True)
    env.reset(); ora.reset()
    finished = np.zeros(N, np.int64)
    for t in range(steps):
        obs, rew, done, info = env.step(env.random_actions())
        o_obs, o_rew, o_done, o_info = ora.step(ora.random_actions())
        assert np.array_equal(_np(info["scores"][lo:lo + n]), o_info["scores"]) and np.array_equal(_np(done[lo:lo + n]), o_done), t
        assert np.array_equal(_np(obs[lo:lo + n]).astype(np.float64), o_obs), t
        finished += _np(done).astype(np.int64)
        if t < 149:
            assert int(info["status"].eq(1).sum()) == 0            # TimeExpired only at turn 150
    assert (finished >= 1).all() and finished.sum() < 1.05 * N      # nearly every random game runs the full 150 turns
    s = env.get_state()
    alive = (s["health"] > 0).reshape(N, 2, 100)
    cnt = np.concatenate([alive[:, :, :88].reshape(N, 2, 11, 8).sum(-1), alive[:, :, 88:].sum(-1, keepdims=True)], axis=2)
    assert np.array_equal(cnt, s["groups"][..., 6]) and (s["health"] >= 0).all() and (s["health"] <= 100).all()
    assert np.array_equal(s["groups"][..., 5] != 0, cnt == 0) and (s["env"][:, 1] == 0).all()
    o = _np(obs)
    assert np.array_equal(o[:, 0, 0].astype(np.int64), s["env"][:, 0]) and (np.abs(o[:, :, 3:45:4]) <= 500).all()
    assert np.array_equal(o[:, 0, 49::5].astype(np.int64), cnt[:, 0]) and np.array_equal(o[:, 1, 49::5].astype(np.int64), cnt[:, 1])
    # opposing-unit columns of board_state sum to the opponent's alive units (every non-destroyed group is listed at one node)
    assert np.array_equal(o[:, 0, 4:45:4].sum(1).astype(np.int64), cnt[:, 1].sum(1)) and np.array_equal(o[:, 1, 4:45:4].sum(1).astype(np.int64),
                                                                                                        cnt[:, 0].sum(1))
    st, ost = env.episode_stats(), ora.episode_stats()
    assert st["totals"][0] == finished.sum() and st["totals"][1:].sum() == st["totals"][0]
    assert np.array_equal(st["winner"][lo:lo + n], ost["winner"]) and np.array_equal(st["length"][lo:lo + n], ost["length"])
    assert np.array_equal(s["health"][lo:lo + n], ora.get_state()["health"])
    # Secondary, statistical check against the reference with its STOCK numpy entropy (SURVEY section 6/8c anchors over 200
    # random-vs-random games: p0 108 / p1 91 / tie 1, 199 of 200 games last the full 150 turns): the keyed generator must
    # not shift these rates.  With 65 536 games the binomial noise is 0.2 %, the anchor's own 95 % interval is +-7 %.
    tot = st["totals"].astype(np.float64)
    assert abs(tot[1] / tot[0] - 0.5) < 0.02 and tot[3] / tot[0] < 0.01
    assert abs(tot[1] / tot[0] - 108 / 200) < 0.07 and (st["length"] == 150).mean() > 0.985
    env.close()


def test_native_rollout_fused_equals_stepwise(evg, oracle_mod):
    """evg_rollout_random: fused (orders drawn inside the step kernel) == unfused (generator kernel + step kernel)
    == the oracle stepped with its own generator; the action buffer holds the last turn's orders."""
    N, seed, steps = 300, 77, 170
    a = evg.EvergladesVecEnv(N, seed=seed, auto_reset=True)
    b = evg.EvergladesVecEnv(N, seed=seed, auto_reset=True)
    ora = oracle_mod.Oracle(N, seed=seed, auto_reset=True)
    a.reset(); b.reset(); ora.reset()
    oa = a.rollout_random(steps, fused=True)
    ob = b.rollout_random(steps, fused=False)
    for t in range(steps):
        last_actions = ora.random_actions()
        o_obs, o_rew, o_done, o_info = ora.step(last_actions)
    for x, y in zip(oa[:3], ob[:3]):
        assert np.array_equal(_np(x), _np(y))
    assert np.array_equal(_np(oa[0]).astype(np.float64), o_obs) and np.array_equal(_np(oa[3]["scores"]), o_info["scores"])
    assert np.array_equal(_np(a._actions), last_actions) and np.array_equal(_np(b._actions), last_actions)
    check_state(a, ora.get_state(), "fused")
    check_state(b, ora.get_state(), "unfused")
    sa, so = a.episode_stats(), ora.episode_stats()
    assert np.array_equal(sa["totals"], so["totals"]) and np.array_equal(sa["winner"], so["winner"])
    ms = a.rollout_random(10, time_kernel=True)[-1]
    assert ms > 0
    a.close(); b.close()


def test_scripted_agents_golden_through_abi(evg):
    """The on-device scripted agents + the HIP env replay what the reference's agent classes and server did
    (tests/golden/agents_scripted.npz): three consecutive episodes per env, agent objects alive across them."""
    d = load_golden("agents_scripted.npz")
    G, E = d["length"].shape
    for g in range(G):
        env = evg.EvergladesVecEnv(1, seed=int(d["seed"][g]), env_id_base=int(d["env_id"][g]), obs_dtype="float64", auto_reset=False)
        pol = [int(x) for x in d["policy"][g]]
        for ep in range(E):
            obs = env.reset()
            for t in range(int(d["length"][g, ep])):
                assert np.array_equal(_np(obs)[0], d["obs"][g, ep, t].astype(np.float64)), (g, ep, t)
                env.scripted_actions(pol[0], 0)
                a = env.scripted_actions(pol[1], 1)
                assert np.array_equal(_np(a)[0], d["actions"][g, ep, t]), ("orders", g, ep, t)
                obs, rew, done, info = env.step(a)
            assert int(done[0]) == 1 and int(info["status"][0]) == d["status"][g, ep]
            assert np.array_equal(_np(info["scores"])[0], d["scores"][g, ep])
        env.close()


def test_scripted_agents_on_non_default_maps_through_abi(evg, tmp_path):
    """tests/golden/custom_agents.npz through the C-ABI: the reference's agent classes on the maps of custom_varA / custom_varB (they keep routing by their
    own DemoMap constants, whatever map file the env was reset with) -- the on-device bots emit the same orders from the observation tensor
    (evg_scripted_actions), and from the on-chip state inside the step kernel (evg_step_vs_policy on both seats), and the env plays the same games."""
    d = load_golden("custom_agents.npz")
    G, E = d["length"].shape
    for g in range(G):
        vname = "custom_%s.npz" % str(d["variant"][g])
        v = load_golden(vname)
        tables = evg.tables_from_json(p1_node_map=v["p1_node_map"].tolist(), **_custom_fixture_files(v, vname, tmp_path))
        pol = [int(x) for x in d["policy"][g]]
        for form in ("scripted_actions", "step_vs seat 0", "step_vs seat 1"):
            env = evg.EvergladesVecEnv(1, seed=int(d["seed"][g]), env_id_base=int(d["env_id"][g]), obs_dtype="float64", auto_reset=False, tables=tables)
            for ep in range(E):
                obs = env.reset()
                for t in range(int(d["length"][g, ep])):
                    want = d["actions"][g, ep, t]
                    if form == "scripted_actions":
                        assert np.array_equal(_np(obs)[0], d["obs"][g, ep, t].astype(np.float64)), (g, ep, t)
                        env.scripted_actions(pol[0], 0)
                        a = env.scripted_actions(pol[1], 1)
                        assert np.array_equal(_np(a)[0], want), ("orders", g, ep, t)
                        obs, rew, done, info = env.step(a)
                    else:
                        seat = int(form[-1])                     # the recorded rows of `seat` are the caller's; the other seat's bot runs inside the kernel
                        import torch
                        rows = torch.as_tensor(want[seat].astype(np.int32), device=env.device).reshape(1, 7, 2).contiguous()
                        sobs, rew, done, info = env.step_vs(pol[1 - seat], rows, seat=seat)
                        assert np.array_equal(_np(sobs)[0], d["obs"][g, ep, t + 1, seat].astype(np.float64)), (form, g, ep, t)
                assert int(done[0]) == 1 and int(info["status"][0]) == d["status"][g, ep], (form, g, ep)
                assert np.array_equal(_np(info["scores"])[0], d["scores"][g, ep]), (form, g, ep)
            env.close()


def test_learner_seat_turn_replays_the_references_agent_classes(evg):
    """evg_step_vs_policy pinned by the REFERENCE, not by the oracle: in every pairing of tests/golden/agents_scripted.npz (the reference's own agent
    classes on its own server, three consecutive episodes each) one seat's RECORDED orders are handed in as the caller's rows and the other seat's bot is
    evaluated inside the step kernel from the on-chip state; the caller's observation after every turn, the final scores and the status must be the
    fixture's -- for both choices of the caller's seat, i.e. every bot of the fixture is exercised on chip in the seat the reference played it in."""
    import torch
    d = load_golden("agents_scripted.npz")
    G, E = d["length"].shape
    for g in range(G):
        pol = [int(x) for x in d["policy"][g]]
        for seat in (0, 1):
            env = evg.EvergladesVecEnv(1, seed=int(d["seed"][g]), env_id_base=int(d["env_id"][g]), obs_dtype="float64", auto_reset=False)
            for ep in range(E):
                env.reset()
                so = env.observe_seat(seat)
                for t in range(int(d["length"][g, ep])):
                    assert np.array_equal(_np(so)[0], d["obs"][g, ep, t, seat].astype(np.float64)), (g, seat, ep, t)
                    rows = torch.as_tensor(d["actions"][g, ep, t, seat].astype(np.int32), device=env.device).reshape(1, 7, 2)
                    so, rew, done, info = env.step_vs(pol[1 - seat], rows, seat=seat)
                assert int(done[0]) == 1 and int(info["status"][0]) == d["status"][g, ep], (g, seat, ep)
                assert np.array_equal(_np(info["scores"])[0], d["scores"][g, ep]), (g, seat, ep)
                assert np.array_equal(_np(so)[0], d["obs"][g, ep, int(d["length"][g, ep]), seat].astype(np.float64)), (g, seat, ep, "final")
            env.close()


@pytest.mark.parametrize("N,seats", [(1024, ("cycle_rush_turn25", "swarm")), (1024, ("swarm", "cycle_rush_turn50")), (65536, ("cycle_rush_turn25", "swarm"))])
def test_config5_scripted_rollout_vs_oracle(evg, oracle_mod, N, seats):
    """BASELINE config 5: cycle_base_rush vs swarm_agent action streams with auto-reset (games end by BaseCapture
    around turn 85-95, so envs desynchronise).  Device agents + HIP env == oracle agents + oracle env; at 65 536
    envs the oracle follows a 512-env sub-range with the same global env ids."""
    seed, steps = 515, 260
    lo, n = (0, N) if N <= 1024 else (40000, 512)
    pid = [evg.EvergladesVecEnv.POLICIES[s] for s in seats]
    env = evg.EvergladesVecEnv(N, seed=seed, auto_reset=True)
    ora = oracle_mod.Oracle(n, seed=seed, env_id_base=lo, auto_reset=True)
    obs = env.reset()
    o_obs = ora.reset()
    for t in range(steps):
        env.scripted_actions(seats[0], 0)
        a = env.scripted_actions(seats[1], 1)
        oa = np.zeros((n, 2, 7, 2), np.int32)
        ora.scripted_actions(pid[0], 0, o_obs, oa)
        ora.scripted_actions(pid[1], 1, o_obs, oa)
        assert np.array_equal(_np(a[lo:lo + n]), oa), ("orders", t)
        obs, rew, done, info = env.step(a)
        o_obs, o_rew, o_done, o_info = ora.step(oa)
        assert np.array_equal(_np(obs[lo:lo + n]).astype(np.float64), o_obs), ("obs", t)
        assert np.array_equal(_np(info["scores"][lo:lo + n]), o_info["scores"]) and np.array_equal(_np(done[lo:lo + n]), o_done)
    st, ost = env.episode_stats(), ora.episode_stats()
    assert np.array_equal(st["winner"][lo:lo + n], ost["winner"]) and np.array_equal(st["length"][lo:lo + n], ost["length"])
    assert st["totals"][0] >= 2 * N and ost["totals"][0] >= 2 * n          # at least two finished episodes per env
    if seats[0] == "cycle_rush_turn25":
        assert st["totals"][1] > 0.95 * st["totals"][0]                     # the cycling base rush wins
    env.close()


def test_turn_can_be_captured_in_a_hip_graph(evg, oracle_mod):
    """step()/random_actions() only enqueue on the caller's stream (no allocation, no synchronisation), so a whole
    turn can be captured with torch.cuda.graph and replayed; the replayed turns equal the oracle's."""
    import torch
    N, seed = 200, 5
    env = evg.EvergladesVecEnv(N, seed=seed, auto_reset=True)
    ora = oracle_mod.Oracle(N, seed=seed, auto_reset=True)
    env.reset(); ora.reset()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            env.step(env.random_actions())
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        env.step(env.random_actions())
    for _ in range(2):                      # the two warm-up turns; the captured turn itself is not executed by the capture
        o_obs, _, _, o_info = ora.step(ora.random_actions())
    for t in range(40):
        g.replay()
        o_obs, _, o_done, o_info = ora.step(ora.random_actions())
    torch.cuda.synchronize()
    assert np.array_equal(_np(env.obs).astype(np.float64), o_obs) and np.array_equal(_np(env.scores), o_info["scores"])
    check_state(env, ora.get_state(), "graph replay")
    env.close()


def test_learner_seat_turn_can_be_captured_in_a_hip_graph(evg, oracle_mod):
    """The consumer's whole turn -- its policy's kernels (here the stand-in generator) and evg_step_vs_policy with the bot inside -- captured once with
    torch.cuda.graph and replayed: 60 replays == 60 oracle turns (observation of the caller's seat, scores, state)."""
    import torch
    N, seed, seat, pol = 3000, 9, 1, "bull_rush"
    pid = evg.EvergladesVecEnv.POLICIES[pol]
    env = evg.EvergladesVecEnv(N, seed=seed, auto_reset=True)
    ora = oracle_mod.Oracle(N, seed=seed, auto_reset=True)
    env.reset()
    o_obs = ora.reset()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            env.step_vs(pol, env.random_actions_seat(seat), seat=seat)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        env.step_vs(pol, env.random_actions_seat(seat), seat=seat)
    for _ in range(60):
        g.replay()
    torch.cuda.synchronize()
    oa = np.zeros((N, 2, 7, 2), np.int32)
    for t in range(62):
        ora.scripted_actions(pid, 1 - seat, o_obs, oa)
        oa[:, seat] = ora.random_actions()[:, seat]
        o_obs, _, _, o_info = ora.step(oa)
    assert np.array_equal(_np(env._obs_seat).astype(np.float64), o_obs[:, seat]) and np.array_equal(_np(env.scores), o_info["scores"])
    check_state(env, ora.get_state(), "learner-seat turn replayed from a graph")
    env.close()


@pytest.mark.parametrize("shape", ["four_lane", "two_lane", "chunked"])
def test_persistent_rollout_can_be_captured_in_a_hip_graph(evg, oracle_mod, shape):
    """The PERSISTENT form inside a caller's graph: a rollout launch (launch plan) only enqueues -- a chunked launch zeroes its queues and
    progress flags with a memset on the stream and keeps nothing on the host -- so torch.cuda.graph can capture it (the library then
    enqueues plainly instead of replaying its own cached graph) and every replay plays the next turns: 4 replays of a 40-turn launch
    after an uncaptured one == 200 oracle turns, for the four-lane kernel, the plain two-lane kernel and the chunked plan (2 chunks).
    (The library itself enqueues plans plainly: replaying graphs of its own was measured slower; `make graphs` keeps that build for the A/B.)"""
    import torch
    cap2 = 32 * 8 * torch.cuda.get_device_properties(0).multi_processor_count
    N = {"four_lane": 3000, "two_lane": cap2 - 32 * 5 - 7, "chunked": cap2 + 2048}[shape]
    seed, tpl = 55, 40
    env = evg.EvergladesVecEnv(N, seed=seed, auto_reset=True)
    plan = env.launch_plan(tpl)[1]
    assert {"four_lane": "four lanes", "two_lane": "two lanes per env, persistent>", "chunked": "chunked"}[shape] in plan, plan
    ora = oracle_mod.Oracle(N, seed=seed, auto_reset=True)
    env.reset(); ora.reset()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        env.rollout_random(tpl, turns_per_launch=tpl)            # warm-up on a side stream
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        env.rollout_random(tpl, turns_per_launch=tpl)            # captured, not executed
    for _ in range(4):
        g.replay()
    torch.cuda.synchronize()
    for t in range(5 * tpl - 1):
        ora.step_noobs(ora.random_actions())
    a = ora.random_actions()
    o_obs, _, _, _ = ora.step(a)
    assert np.array_equal(_np(env._actions), a)
    _compare_whole_batch(env, ora, o_obs, ("persistent launch replayed from a caller's graph", shape))
    env.rollout_random(tpl, turns_per_launch=tpl, prepare=True)  # (prepare: nothing is played)
    env.rollout_random(tpl, turns_per_launch=tpl)                # and a plain launch of the same plan still works afterwards
    for t in range(tpl):
        o_obs, _, _, _ = ora.step(ora.random_actions())
    _compare_whole_batch(env, ora, o_obs, ("plain launch after the caller's graph", shape))
    assert env.check_fault() == 0
    env.close()


@pytest.mark.parametrize("seats", [("cycle_rush_turn25", "swarm"), ("cycle_target_node11P2", "dfs_attack"), ("random_actions_delay", "base_rush_v1"),
                                   ("swarm", "cycle_target_node1"), ("bull_rush", "all_cycle")])
def test_native_policy_rollout_forms_agree(evg, seats):
    """evg_rollout_policies: agents as separate launches reading the observation tensor == agents fused into the step
    kernel reading the on-chip state (one launch per turn, and the persistent form) == stepping by hand."""
    N, seed, steps = 300, 12, 200
    envs = [evg.EvergladesVecEnv(N, seed=seed, auto_reset=True) for _ in range(4)]
    for e in envs:
        e.reset()
    envs[0].rollout_policies(steps, seats[0], seats[1], fused=False)
    envs[1].rollout_policies(steps, seats[0], seats[1], fused=True)
    envs[2].rollout_policies(steps, seats[0], seats[1], fused=True, turns_per_launch=37)
    for t in range(steps):
        envs[3].scripted_actions(seats[0], 0)
        envs[3].step(envs[3].scripted_actions(seats[1], 1))
    ref = envs[3]
    sr = ref.get_state()
    for i, e in enumerate(envs[:3]):
        assert np.array_equal(_np(e.obs), _np(ref.obs)) and np.array_equal(_np(e._actions), _np(ref._actions)), (seats, i)
        se = e.get_state()
        for k in sr:
            assert np.array_equal(se[k], sr[k]), (seats, i, k)
        assert np.array_equal(e.episode_stats()["totals"], ref.episode_stats()["totals"])
    assert ref.episode_stats()["totals"][0] > 0
    for e in envs:
        e.close()


@pytest.mark.parametrize("pol", list(range(15)))
def test_every_scripted_bot_vs_oracle(evg, oracle_mod, pol):
    """All State_Machine bots on device == the oracle's restatement (itself pinned by the reference's
    classes): bot in seat 0 against random_actions and in seat 1 against swarm, auto-reset, agents alive across episodes."""
    N, seed, steps = 256, 900 + pol, 230
    for seats in ((pol, 0), (3, pol)):
        env = evg.EvergladesVecEnv(N, seed=seed, auto_reset=True)
        ora = oracle_mod.Oracle(N, seed=seed, auto_reset=True)
        obs, o_obs = env.reset(), ora.reset()
        for t in range(steps):
            env.scripted_actions(seats[0], 0)
            a = env.scripted_actions(seats[1], 1)
            oa = np.zeros((N, 2, 7, 2), np.int32)
            ora.scripted_actions(seats[0], 0, o_obs, oa)
            ora.scripted_actions(seats[1], 1, o_obs, oa)
            assert np.array_equal(_np(a), oa), ("orders", pol, seats, t)
            obs, rew, done, info = env.step(a)
            o_obs, o_rew, o_done, o_info = ora.step(oa)
            assert np.array_equal(_np(obs).astype(np.float64), o_obs), ("obs", pol, seats, t)
        check_state(env, ora.get_state(), (pol, seats))
        assert np.array_equal(env.episode_stats()["totals"], ora.episode_stats()["totals"])
        env.close()


def test_smart_state_features_match_reference_fixture(evg):
    """device features == float32(reference float64 features) bit for bit, for every obs dtype."""
    import torch
    d = load_golden("smart_state.npz")
    M = d["obs"].shape[0]
    assert np.array_equal(evg.EvergladesVecEnv.move_table(), d["moves"])
    for dt, tdt in (("float32", torch.float32), ("float64", torch.float64), ("int16", torch.int16)):
        env = evg.EvergladesVecEnv(M, seed=1, obs_dtype=dt)
        obs = torch.as_tensor(d["obs"].astype(np.int64), device=env.device).to(tdt).contiguous()
        for p in range(2):
            f = _np(env.smart_state(p, obs))
            assert np.array_equal(f, d["features"][:, p].astype(np.float32)), (dt, p)
            # the compact pair (shared [34] + per-swarm [13], 760 B per env instead of 2 832) expands to the same matrix, value for value
            sh, sw = env.smart_state_compact(p, obs)
            assert sh.shape == (M, 34) and sw.shape == (M, 12, 13)
            assert np.array_equal(_np(evg.EvergladesVecEnv.expand_smart_state(sh, sw)), d["features"][:, p].astype(np.float32)), (dt, p, "compact")
            sh1, sw1 = env.smart_state_compact(p, obs[:, p].contiguous())          # from a one-seat tensor
            assert torch.equal(sh, sh1) and torch.equal(sw, sw1)
        env.close()


@pytest.mark.parametrize("dtype", ["float32", "float64", "int16"])
def test_learner_seat_turn_with_fused_smart_state_features(evg, oracle_mod, dtype):
    """evg_step_vs_policy_smart: the learner-seat turn that also writes the Smart_State agent's next network input (DQNAgent.create_swarm_obs,
    agents/Smart_State/DQNAgent.py:268-300, compact form) from the observation image while it is on chip.  Over 40 turns with auto-resets, both seats, a ragged
    batch and every observation dtype: (1) the game is the one evg_step_vs_policy plays (same observation, rewards, state); (2) shared / swarm equal what
    evg_smart_state_compact computes from the written observation, bit for bit; (3) expanded, they equal float32(the oracle's float64 features)."""
    import torch
    N = 2 * 8192 + 77
    for seat in (0, 1):
        a = evg.EvergladesVecEnv(N, seed=21, auto_reset=True, obs_dtype=dtype)
        b = evg.EvergladesVecEnv(N, seed=21, auto_reset=True, obs_dtype=dtype)
        a.reset(); b.reset()
        a.rollout_policies(85, "cycle_rush_turn25", "swarm", fused=True, turns_per_launch=85)       # mid-game, some envs about to end: auto-resets follow
        b.rollout_policies(85, "cycle_rush_turn25", "swarm", fused=True, turns_per_launch=85)
        shared = torch.full((N, 34), -7.0, device=a.device)
        swarm = torch.full((N, 12, 13), -7.0, device=a.device)
        bot = "swarm" if seat == 0 else "cycle_rush_turn25"
        for t in range(40):
            rows = a.random_actions_seat(seat)
            rows_b = b.random_actions_seat(seat)
            sobs, rew, done, info = a.step_vs(bot, rows, seat=seat, features=(shared, swarm))
            sobs_b, rew_b, done_b, info_b = b.step_vs(bot, rows_b, seat=seat)
            assert torch.equal(sobs, sobs_b) and torch.equal(rew, rew_b) and torch.equal(done, done_b) and torch.equal(info["scores"], info_b["scores"]), (seat, t)
            want_sh, want_sw = b.smart_state_compact(-1, sobs_b)
            assert torch.equal(shared, want_sh), (seat, t, "shared")
            assert torch.equal(swarm, want_sw), (seat, t, "swarm")
            if t % 13 == 0:
                want = oracle_mod.smart_state(_np(sobs).astype(np.float64)).astype(np.float32)
                assert np.array_equal(_np(evg.EvergladesVecEnv.expand_smart_state(shared, swarm)), want), (seat, t, "oracle")
        assert int(a.episode_stats()["totals"][0]) > 0                                                  # episodes did end and restart inside the loop
        check_state(a, b.get_state(), "fused features")
        with pytest.raises(ValueError):
            a.step_vs(bot, rows, seat=seat, features=(shared[:-1], swarm))
        a.close(); b.close()


def test_smart_state_multi_pass_ragged_vs_oracle(evg, oracle_mod):
    """The feature kernel at a size that needs several passes of its resident grid and is not a multiple of its 4 envs per
    block, on mid-game observations of both seats: equal to float32(oracle float64)."""
    N = 3 * 8192 + 4099
    env = evg.EvergladesVecEnv(N, seed=8, auto_reset=True)
    env.reset()
    env.rollout_random(70, turns_per_launch=70)
    obs = _np(env.obs).astype(np.float64)
    for p in range(2):
        got = _np(env.smart_state(p))
        want = oracle_mod.smart_state(obs[:, p]).astype(np.float32)
        assert got.shape == (N, 12, 59) and np.array_equal(got, want), p
        assert np.array_equal(_np(evg.EvergladesVecEnv.expand_smart_state(*env.smart_state_compact(p))), want), (p, "compact")
    env.close()


def test_smart_actions_match_reference_fixture_and_oracle(evg, oracle_mod):
    """SURVEY 8 f4, network output -> orders (evg_smart_actions): device == DQNAgent.get_best_actions as the reference's own methods computed it
    (tests/golden/smart_actions.npz: Q tensors with ties between swarms and between directions, all-equal, +0.0 / -0.0) for every observation dtype,
    from the two-seat tensor and from a one-seat tensor; then a ragged 65 536 + 37-env batch of mid-game observations with quantised Q values
    (ties everywhere) against the oracle's restatement, both seats; and the rows go straight into step_vs()."""
    import torch
    d = load_golden("smart_actions.npz")
    M = d["obs"].shape[0]
    for dt, tdt in (("float32", torch.float32), ("float64", torch.float64), ("int16", torch.int16)):
        env = evg.EvergladesVecEnv(M, seed=1, obs_dtype=dt)
        obs = torch.as_tensor(d["obs"].astype(np.int64), device=env.device).to(tdt).contiguous()
        for p in range(2):
            q = torch.as_tensor(d["q"][:, p], device=env.device).contiguous()
            dirs = torch.zeros((M, 7, 2), dtype=torch.int32, device=env.device)
            a = env.smart_actions(q, p, obs, directions=dirs)
            assert np.array_equal(_np(a), d["actions"][:, p]) and np.array_equal(_np(dirs), d["directions"][:, p]), (dt, p)
            a1 = env.smart_actions(q, obs=obs[:, p].contiguous(), out=torch.zeros((M, 7, 2), dtype=torch.int32, device=env.device))      # one-seat tensor
            assert np.array_equal(_np(a1), d["actions"][:, p]), (dt, p, "one-seat")
        env.close()
    N = 65536 + 37
    env = evg.EvergladesVecEnv(N, seed=12, auto_reset=True)
    env.reset()
    env.rollout_random(60, turns_per_launch=60)
    obs = _np(env.obs).astype(np.float64)
    g = torch.Generator(device="cpu").manual_seed(5)
    for p in range(2):
        q = (torch.randn((N, 12, 5), generator=g) * 2.0).round().div(2.0).to(env.device)           # few levels: ties between swarms and directions
        dirs = torch.zeros((N, 7, 2), dtype=torch.int32, device=env.device)
        got = _np(env.smart_actions(q, p, directions=dirs))
        want_a, want_d = oracle_mod.smart_actions(_np(q), obs[:, p])
        assert np.array_equal(got, want_a) and np.array_equal(_np(dirs), want_d), p
    # ... and the rows are what the learner-seat turn takes: features -> (a stand-in network) -> orders -> step, nothing on the host in between
    sobs = env.observe_seat(0)
    feats = env.smart_state(0, sobs)
    w = torch.randn((59, 5), generator=g).to(env.device)
    rows = env.smart_actions((feats @ w).contiguous(), obs=sobs)
    want_a, _ = oracle_mod.smart_actions(_np(feats @ w), _np(sobs).astype(np.float64))
    assert np.array_equal(_np(rows), want_a)
    o2, rew, done, info = env.step_vs("random", rows, seat=0)
    assert o2.shape == (N, 105) and int(info["status"].max()) <= 3
    with pytest.raises(ValueError):
        env.smart_actions(torch.zeros((N, 12, 4), device=env.device))
    env.close()


def test_smart_get_action_with_epsilon_matches_reference_fixture_and_oracle(evg, oracle_mod):
    """The other half of the Smart_State agent's turn (evg_smart_get_action = DQNAgent.get_action, agents/Smart_State/DQNAgent.py:130-173): the epsilon
    coin, then get_random_actions or get_best_actions.  (1) tests/golden/smart_explore.npz -- the reference's own get_action with its three draws served
    from the keyed stream -- for every observation dtype, from the two-seat and from a one-seat tensor, epsilon per env; (2) a ragged 65 536 + 37-env batch
    of mid-game observations (envs in different episodes after auto-resets) against the oracle with a scalar epsilon of 0.1 and with per-env epsilons, both
    seats; (3) epsilon 0 is evg_smart_actions, epsilon 1 explores everywhere; (4) the rows feed step_vs()."""
    import torch
    d = load_golden("smart_explore.npz")
    M = d["obs"].shape[0]
    for dt, tdt in (("float32", torch.float32), ("float64", torch.float64), ("int16", torch.int16)):
        env = evg.EvergladesVecEnv(M, seed=int(d["seed"][0]), obs_dtype=dt, auto_reset=False)
        env.reset()
        st = env.get_state()
        st["env"][:, 2] = d["episode"]                        # the agents of the fixture live in episodes m % 3 (part of the key of their draws)
        env.set_state(st["groups"], st["nodes"], st["health"], st["env"])
        obs = torch.as_tensor(d["obs"].astype(np.int64), device=env.device).to(tdt).contiguous()
        for p in range(2):
            q = torch.as_tensor(d["q"][:, p], device=env.device).contiguous()
            eps = torch.as_tensor(d["eps"][:, p], device=env.device).contiguous()
            dirs = torch.zeros((M, 7, 2), dtype=torch.int32, device=env.device)
            ex = torch.zeros(M, dtype=torch.uint8, device=env.device)
            a = env.smart_get_action(q, eps, seat=p, obs=obs, directions=dirs, explored=ex)
            assert np.array_equal(_np(ex), d["explored"][:, p]), (dt, p)
            assert np.array_equal(_np(a), d["actions"][:, p]) and np.array_equal(_np(dirs), d["directions"][:, p]), (dt, p)
            a1 = env.smart_get_action(q, eps, seat=p, obs=obs[:, p].contiguous(), out=torch.zeros((M, 7, 2), dtype=torch.int32, device=env.device))
            assert np.array_equal(_np(a1), d["actions"][:, p]), (dt, p, "one-seat")
        env.close()
    N, seed = 65536 + 37, 12
    env = evg.EvergladesVecEnv(N, seed=seed, auto_reset=True, env_id_base=1000)
    env.reset()
    env.rollout_policies(93, "cycle_rush_turn25", "swarm", fused=True, turns_per_launch=93)       # BaseCapture on turns 91-95: envs are in episodes 0 and 1
    obs = _np(env.obs).astype(np.float64)
    episodes = env.get_state()["env"][:, 2].astype(np.uint32)
    assert len(np.unique(episodes)) >= 2 and len(np.unique(obs[:, 0, 0])) >= 3
    ids = (1000 + np.arange(N)).astype(np.uint32)
    g = torch.Generator(device="cpu").manual_seed(7)
    for p in range(2):
        q = (torch.randn((N, 12, 5), generator=g) * 2.0).round().div(2.0).to(env.device)
        for eps_np in (np.full(N, 0.1, np.float32), np.random.default_rng(p).random(N).astype(np.float32)):
            scalar = bool((eps_np == eps_np[0]).all())
            eps = float(eps_np[0]) if scalar else torch.as_tensor(eps_np, device=env.device)
            dirs = torch.zeros((N, 7, 2), dtype=torch.int32, device=env.device)
            ex = torch.zeros(N, dtype=torch.uint8, device=env.device)
            got = _np(env.smart_get_action(q, eps, seat=p, directions=dirs, explored=ex))
            want_a, want_d, want_x = oracle_mod.smart_get_action(_np(q), obs[:, p], seed, ids, episodes, p, eps_np)
            assert np.array_equal(_np(ex), want_x) and np.array_equal(got, want_a) and np.array_equal(_np(dirs), want_d), (p, scalar)
            if scalar:
                assert abs(want_x.mean() - 0.1) < 0.005                                             # the coin is a fair 10 % over 65 573 agents
        best = _np(env.smart_actions(q, p)).copy()
        assert np.array_equal(_np(env.smart_get_action(q, 0.0, seat=p)), best)
        ex = torch.zeros(N, dtype=torch.uint8, device=env.device)
        rnd = _np(env.smart_get_action(q, 1.0, seat=p, explored=ex))
        assert _np(ex).all() and (np.sort(rnd[:, :, 0], axis=1)[:, 1:] != np.sort(rnd[:, :, 0], axis=1)[:, :-1]).all()   # 7 distinct swarms everywhere
    sobs = env.observe_seat(1)
    feats = env.smart_state(1, sobs)
    w = torch.randn((59, 5), generator=g).to(env.device)
    rows = env.smart_get_action((feats @ w).contiguous(), 0.3, seat=1, obs=sobs)
    want_a, _, _ = oracle_mod.smart_get_action(_np(feats @ w), _np(sobs).astype(np.float64), seed, ids, episodes, 1, np.full(N, 0.3, np.float32))
    assert np.array_equal(_np(rows), want_a)
    o2, rew, done, info = env.step_vs("cycle_rush_turn25", rows, seat=1)
    assert o2.shape == (N, 105) and int(info["status"].max()) <= 3
    with pytest.raises(evg.EvgError):
        env.smart_get_action(q, 1.5, seat=0)
    with pytest.raises(ValueError):
        env.smart_get_action(q, torch.zeros(N - 1, device=env.device), seat=0)
    env.close()


@pytest.mark.parametrize("mode", ["fused bots, persistent", "learner seat, per turn", "stock entropy"])
def test_checkpoint_resume_continues_bit_for_bit(evg, tmp_path, mode):
    """Checkpoint / resume of a running job (the reference never serialises its env): EvergladesVecEnv.checkpoint() = evg_get_state +
    evg_get_run_state (scripted agents' objects, running returns, last finished episodes, win counters; + the generators in the stock-entropy mode), saved
    with np.savez, loaded into a FRESH handle of the same config, continues exactly like the run that was saved: observations, orders, state, episode results."""
    import torch
    N, seed = 4096 + 40, 77
    kw = dict(seed=seed, auto_reset=True, env_id_base=500)
    if mode == "stock entropy":
        N, kw = 96, dict(kw, rng_mode="mt19937")

    def play(env, turns):
        out = None
        if mode == "fused bots, persistent":
            env.rollout_policies(turns, "dfs_attack", "swarm", fused=True, turns_per_launch=turns)      # both bots carry state across turns AND episodes
            out = (_np(env.obs).copy(), _np(env._actions).copy())
        elif mode == "learner seat, per turn":
            for _ in range(turns):
                sobs, rew, done, info = env.step_vs("cycle_target_node", env.random_actions_seat(1), seat=1)
            out = (_np(sobs).copy(), _np(rew).copy())
        else:
            for _ in range(turns):
                obs, rew, done, info = env.step(env.random_actions())
            out = (_np(obs).copy(), _np(rew).copy())
        return out

    a = evg.EvergladesVecEnv(N, **kw)
    a.reset()
    play(a, 95)                                                            # config-5-like bots end games around turn 91-95: some envs are in their second episode
    ck = a.checkpoint()
    flat = {"%s/%s" % (k, kk): vv for k, v in ck.items() for kk, vv in (v.items() if isinstance(v, dict) else [("", v)])}
    np.savez(tmp_path / "ck.npz", **flat)
    want = play(a, 70)
    want_state, want_stats = a.get_state(), a.episode_stats()
    a.close()
    ld = np.load(tmp_path / "ck.npz")
    ck2 = {}
    for k in ld.files:
        top, sub = k.split("/")
        if sub:
            ck2.setdefault(top, {})[sub] = ld[k]
        else:
            ck2[top] = ld[k]
    b = evg.EvergladesVecEnv(N, **kw)
    b.reset()
    b.restore(ck2)
    got = play(b, 70)
    for x, y in zip(got, want):
        assert np.array_equal(x, y), mode
    check_state(b, want_state, mode)
    st = b.episode_stats()
    for k in ("returns", "length", "winner", "totals"):
        assert np.array_equal(st[k], want_stats[k]), (mode, k)
    assert int(st["totals"][0]) > 0
    r = b.get_run_state()
    assert r["agents"].shape == (N, 2, 3) and (mode != "fused bots, persistent" or (r["agents"][:, 0, 2] > 0).all())       # dfs_attack's call counter did advance
    with pytest.raises(evg.EvgError):
        b.set_run_state(totals=np.array([5, 1, 1, 1], np.int64))            # episodes != wins + ties: refused
    with pytest.raises(ValueError):
        b.set_run_state(agents=np.zeros((N, 2, 2), np.uint32))
    b.close()


@pytest.mark.parametrize("tpl", [2, 7, 150])
def test_persistent_multi_turn_rollout_equals_stepwise(evg, oracle_mod, tpl):
    """The persistent rollout form (each launch plays `tpl` consecutive turns per wavefront with the state resident on
    chip) gives exactly the results of one launch per turn and of the oracle -- across auto-resets, with a partial last
    launch, and with the per-turn outputs holding the last turn."""
    N, seed, steps = 500, 2024, 317
    a = evg.EvergladesVecEnv(N, seed=seed, auto_reset=True)
    ora = oracle_mod.Oracle(N, seed=seed, auto_reset=True)
    a.reset(); ora.reset()
    out = a.rollout_random(steps, turns_per_launch=tpl)
    for t in range(steps):
        acts = ora.random_actions()
        o_obs, o_rew, o_done, o_info = ora.step(acts)
    assert np.array_equal(_np(out[0]).astype(np.float64), o_obs) and np.array_equal(_np(a._actions), acts)
    assert np.array_equal(_np(out[3]["scores"]), o_info["scores"]) and np.array_equal(_np(out[2]), o_done)
    assert np.allclose(_np(out[1]), o_rew, rtol=0, atol=REWARD_ATOL)
    check_state(a, ora.get_state(), ("persistent", tpl))
    sa, so = a.episode_stats(), ora.episode_stats()
    assert np.array_equal(sa["totals"], so["totals"]) and np.array_equal(sa["winner"], so["winner"]) and np.array_equal(sa["length"], so["length"])
    assert np.allclose(sa["returns"], so["returns"], rtol=1e-6, atol=1e-5)
    a.close()


@pytest.mark.parametrize("N", [333, 65536 + 37])
def test_single_turn_form_with_four_waves_per_workgroup_matches_oracle(evg, oracle_mod, N):
    """Round-5 dispatch experiment (diagnostic library, lanes = 256): single-turn launches as 256-thread workgroups of four INDEPENDENT wavefronts (own LDS slice,
    own 32 envs, no barrier) -- same results as the product's one-wavefront workgroups: caller-supplied orders, kernel-drawn orders and the scripted bots, with a
    partial last workgroup (wavefronts beyond the last set leave at once)."""
    seed = 77
    oracle_mod.lib().evo_set_num_threads(min(16, len(os.sched_getaffinity(0))))
    env = evg.EvergladesVecEnv(N, seed=seed, auto_reset=True, library=evg._lib.DIAG_LIB_PATH, diag=dict(lanes=256))
    ora = oracle_mod.Oracle(N, seed=seed, auto_reset=True)
    assert np.array_equal(_np(env.reset()).astype(np.float64), ora.reset())
    for t in range(12):
        a = env.random_actions()
        obs, rew, done, info = env.step(a)
        o_obs, o_rew, o_done, o_info = ora.step(_np(a))
        assert np.array_equal(_np(obs).astype(np.float64), o_obs), ("obs", t)
        assert np.array_equal(_np(info["scores"]), o_info["scores"]) and np.array_equal(_np(done), o_done)
    env.rollout_random(165, turns_per_launch=1)                       # orders drawn in the kernel, one launch per turn, through an auto-reset
    for t in range(165):
        a = ora.random_actions()
        o_obs, _, _, _ = ora.step(a)
    assert np.array_equal(_np(env.obs).astype(np.float64), o_obs) and np.array_equal(_np(env._actions), a)
    check_state(env, ora.get_state(), "kernel-drawn orders")
    env.rollout_policies(30, "cycle_rush_turn25", "swarm", fused=True, turns_per_launch=1)
    pid = [evg.EvergladesVecEnv.POLICIES[s] for s in ("cycle_rush_turn25", "swarm")]
    oa = np.zeros((N, 2, 7, 2), np.int32)
    for t in range(30):
        ora.scripted_actions(pid[0], 0, o_obs, oa)
        ora.scripted_actions(pid[1], 1, o_obs, oa)
        o_obs, _, _, _ = ora.step(oa)
    assert np.array_equal(_np(env.obs).astype(np.float64), o_obs)
    check_state(env, ora.get_state(), "scripted")
    assert np.array_equal(env.episode_stats()["totals"], ora.episode_stats()["totals"])
    env.close()


def test_16_envs_per_wave_variant_matches_oracle(evg, oracle_mod):
    """The step-kernel variant with 16 envs per wavefront (32 helper lanes join the balanced phases) is slower on MI355X and
    lives only in the diagnostic library (libevg_diag.so, evg_diag_configure): it must give the same results, single- and
    multi-turn."""
    N, seed, steps = 333, 31, 170
    env = evg.EvergladesVecEnv(N, seed=seed, auto_reset=True, library=evg._lib.DIAG_LIB_PATH, diag=dict(lanes=32))
    ora = oracle_mod.Oracle(N, seed=seed, auto_reset=True)
    env.reset(); ora.reset()
    for t in range(40):
        a = env.random_actions()
        obs, rew, done, info = env.step(a)
        o_obs, o_rew, o_done, o_info = ora.step(_np(a))
        assert np.array_equal(_np(obs).astype(np.float64), o_obs), t
    env.rollout_random(steps, turns_per_launch=50)
    for t in range(steps):
        o_obs, _, _, o_info = ora.step(ora.random_actions())
    assert np.array_equal(_np(env.obs).astype(np.float64), o_obs)
    check_state(env, ora.get_state(), "lanes32")
    assert np.array_equal(env.episode_stats()["totals"], ora.episode_stats()["totals"])
    env.close()


# ---------------------------------------------------------------------------------------------------------------
# SURVEY 8 f2 on the device: the stock-entropy mode (rng_mode="mt19937")
# ---------------------------------------------------------------------------------------------------------------
def test_stock_entropy_generator_is_numpys(evg):
    """After seeding, every env's generator is word for word np.random.RandomState(seed) (init_genrand, position 624);
    explicit per-env seeds and the create-time rule seed + env_id_base + e."""
    N = 70
    env = evg.EvergladesVecEnv(N, seed=1234, env_id_base=1000, rng_mode="mt19937")
    st = env.get_stock_entropy()
    for e in (0, 1, 33, 69):
        want = np.random.RandomState(1234 + 1000 + e).get_state()
        assert np.array_equal(st[e, :624], want[1]) and st[e, 624] == want[2] == 624
    seeds = np.arange(N, dtype=np.uint64) * 7919 + 2 ** 32 - 5
    env.seed_stock_entropy(seeds)
    st = env.get_stock_entropy()
    for e in (0, 2, 69):
        assert np.array_equal(st[e, :624], np.random.RandomState(int(seeds[e]) & 0xFFFFFFFF).get_state()[1])
    env.close()
    keyed = evg.EvergladesVecEnv(4, seed=1)
    with pytest.raises(evg.EvgError):
        keyed.seed_stock_entropy()
    keyed.close()


def test_stock_entropy_mode_replays_the_unmodified_reference(evg):
    """tests/golden/stock_mt.npz: 8 games of the reference with NO entropy injection (np.random.seed(s), its own
    np.random.randint incl. the unobservable focus draws).  The device in stock mode, given the same seeds and orders,
    reproduces every observation, score, status and the float64 health bit for bit -- all 8 games side by side."""
    d = load_golden("stock_mt.npz")
    G = len(d["length"])
    env = evg.EvergladesVecEnv(G, seed=0, auto_reset=False, obs_dtype="float64", rng_mode="mt19937")
    env.seed_stock_entropy(d["seed"])
    obs = _np(env.reset())
    for g in range(G):
        assert np.array_equal(obs[g], d["obs"][g, 0].astype(np.float64))
    T = int(d["length"].max())
    for t in range(T):
        live = [g for g in range(G) if t < int(d["length"][g])]
        a = np.zeros((G, 2, 7, 2), np.int32)
        for g in live:
            a[g] = d["actions"][g, t]
        obs, rew, done, info = env.step(a)
        obs, sc, stt = _np(obs), _np(info["scores"]), _np(info["status"])
        s = env.get_state()
        for g in live:
            assert np.array_equal(obs[g], d["obs"][g, t + 1].astype(np.float64)), ("obs", g, t)
            assert np.array_equal(sc[g], d["scores"][g, t]) and stt[g] == d["status"][g, t], ("scores", g, t)
            assert np.array_equal(s["health"][g], d["health"][g, t + 1]), ("health bits", g, t)
    env.close()


@pytest.mark.parametrize("policy", ["random", "brawl"])
def test_stock_entropy_rollout_vs_oracle(evg, oracle_mod, policy):
    """Stock mode at a ragged size (partial last wavefront) over several auto-reset episodes: the per-env MT19937
    streams (regenerated every 624 outputs, two focus draws per reset, one every tenth turn) stay in lock-step with the
    oracle's literal restatement of the reference's loops."""
    from gen_policies import policy_actions
    N, seed, base = 300, 77, 5000
    env = evg.EvergladesVecEnv(N, seed=seed, env_id_base=base, auto_reset=True, rng_mode="mt19937")
    ora = oracle_mod.Oracle(N, seed=seed, env_id_base=base, auto_reset=True)
    ora.use_stock_mt((seed + base + np.arange(N)) & 0xFFFFFFFF)
    obs = _np(env.reset()).astype(np.float64)
    assert np.array_equal(obs, ora.reset())
    rng = np.random.default_rng(11)
    for t in range(330):
        a = _np(env.random_actions()).copy() if policy == "random" else policy_actions(policy, obs, t, rng)
        o, rew, done, info = env.step(a)
        o_obs, o_rew, o_done, o_info = ora.step(a)
        obs = _np(o).astype(np.float64)
        assert np.array_equal(obs, o_obs), (policy, "obs", t)
        assert np.array_equal(_np(info["scores"]), o_info["scores"]) and np.array_equal(_np(done), o_done)
        if t % 55 == 54:
            check_state(env, ora.get_state(), (policy, t))
    assert env.episode_stats()["totals"][0] >= N
    env.close()


def test_stock_entropy_checkpoint_roundtrip(evg):
    """evg_get/set_state + evg_get/set_stock_entropy checkpoint a stock-mode game: restoring both and replaying the same
    orders gives the same trajectory; the native rollout driver works in this mode too (single-turn launches)."""
    N = 96
    env = evg.EvergladesVecEnv(N, seed=3, auto_reset=True, rng_mode="mt19937")
    env.reset()
    env.rollout_random(40, turns_per_launch=150)           # clamps to one turn per launch in this mode
    s, m = env.get_state(), env.get_stock_entropy()
    assert (m[:, 624] <= 624).all() and len(set(m[:, 624].tolist())) > 1
    acts, outs = [], []
    for t in range(25):
        a = env.random_actions().clone()
        acts.append(a)
        outs.append(_np(env.step(a)[0]).copy())
    end = env.get_state()
    env.set_state(**s)
    env.set_stock_entropy(m)
    for t in range(25):
        assert np.array_equal(_np(env.step(acts[t])[0]), outs[t]), t
    check_state(env, end, "restored replay")
    env.close()


def test_dropin_shares_numpys_global_generator_like_the_reference(evg):
    """BASELINE config 1 as the reference process runs it (tests/golden/config1_stock.npz: np.random.seed(0), the
    reference's own random_actions agents, nothing injected).  With entropy="numpy" the drop-in env borrows the
    process-wide generator around every reset/step, so an unchanged harness loop -- np.random.seed(0), agents that draw
    from np.random, env.reset()/env.step() -- plays the very same two episodes on the GPU: agent orders, observations,
    rewards, done flags and the final state of np.random are all equal to the reference's."""
    from gen_policies import NumpyGlobalRandomAgent
    d = load_golden("config1_stock.npz")
    saved = np.random.get_state()
    try:
        np.random.seed(int(d["seed"][0]))
        env = evg.EvergladesEnv(entropy="numpy")
        players = {0: NumpyGlobalRandomAgent(), 1: NumpyGlobalRandomAgent()}
        for ep in range(len(d["length"])):
            obs = env.reset(players=players, config_dir=None, map_file=None, unit_file=None, output_dir=None, pnames=None, debug=False)
            assert all(np.array_equal(obs[p], d["obs"][ep, 0, p].astype(np.float64)) for p in (0, 1))
            done, t = 0, 0
            while not done:
                actions = {p: players[p].get_action(obs[p]) for p in (0, 1)}
                assert all(np.array_equal(actions[p].astype(int), d["actions"][ep, t, p]) for p in (0, 1)), ("agent stream", ep, t)
                obs, reward, done, info = env.step(actions)
                assert all(np.array_equal(obs[p], d["obs"][ep, t + 1, p].astype(np.float64)) for p in (0, 1)), ("obs", ep, t)
                assert [reward[0], reward[1]] == d["reward"][ep, t].tolist() and int(done) == int(d["done"][ep, t]), ("reward", ep, t)
                t += 1
            assert t == int(d["length"][ep])
        fin = np.random.get_state()
        assert np.array_equal(fin[1], d["final_key"]) and fin[2] == int(d["final_pos"][0])
    finally:
        np.random.set_state(saved)


def test_two_million_envs_edges_vs_oracle(evg, oracle_mod):
    """Size edge: 2^21 + 5 concurrent games in one handle (ragged last wavefront; 4 GB of state, 64-bit offsets into the
    1.8 GB observation tensor).  The first and the last envs of the range, stepped with the on-device action generator in
    both the per-turn and the persistent form, equal oracles created for just those global env ids."""
    import torch
    N, seed, K = 2 ** 21 + 5, 4242, 48
    env = evg.EvergladesVecEnv(N, seed=seed, env_id_base=7, auto_reset=True)
    env.reset()
    heads = [(0, oracle_mod.Oracle(K, seed=seed, env_id_base=7, auto_reset=True)),
             (N - K, oracle_mod.Oracle(K, seed=seed, env_id_base=7 + N - K, auto_reset=True))]
    for _, o in heads:
        o.reset()
    for form in (1, 16):
        env.rollout_random(16, turns_per_launch=form)
        for first, o in heads:
            for _ in range(16):
                a = o.random_actions()
                o_obs, o_rew, o_done, o_info = o.step(a)
            sl = slice(first, first + K)
            assert np.array_equal(_np(env._actions[sl]), a), ("actions", form, first)
            assert np.array_equal(_np(env.obs[sl]).astype(np.float64), o_obs), ("obs", form, first)
            assert np.array_equal(_np(env.scores[sl]), o_info["scores"]) and np.allclose(_np(env.reward[sl]), o_rew, rtol=0, atol=REWARD_ATOL)
    assert int(env.obs[:, 0, 0].min()) == 32 and int(env.obs[:, 0, 0].max()) == 32      # every env is at turn 32
    env.close()
    del env
    torch.cuda.empty_cache()


def test_two_handles_on_two_streams(evg, oracle_mod):
    """Handles are independent: two envs driven from two HIP streams at the same time give what each gives alone."""
    import torch
    N = 4096
    envs = [evg.EvergladesVecEnv(N, seed=s, auto_reset=True) for s in (5, 6)]
    streams = [torch.cuda.Stream() for _ in envs]
    torch.cuda.synchronize()
    for env, st in zip(envs, streams):
        with torch.cuda.stream(st):
            env.reset()
    for _ in range(6):
        for env, st in zip(envs, streams):
            with torch.cuda.stream(st):
                env.rollout_random(10, turns_per_launch=1)
                env.step(env.random_actions())
    torch.cuda.synchronize()
    for env, s in zip(envs, (5, 6)):
        ora = oracle_mod.Oracle(N, seed=s, auto_reset=True)
        ora.reset()
        for _ in range(66):
            o_obs, _, _, _ = ora.step(ora.random_actions())
        assert np.array_equal(_np(env.obs).astype(np.float64), o_obs)
        check_state(env, ora.get_state(), ("stream", s))
        env.close()


def test_plain_c_client_of_the_abi(evg):
    """examples/c_client.c (gcc, no Python, no torch: include/evg.h + hipMalloc'ed buffers) plays the same games as the
    Python binding: same win counters and the same checksum over the last observations."""
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "c_client")
    if not os.path.exists(exe):
        import __graft_entry__ as g
        g.build_c_client()
    N, turns, seed = 1000, 320, 77
    out = subprocess.run([exe, str(N), str(turns), str(seed)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    # (RCCL prints its version on stdout too)
    f = " ".join(l for l in out.stdout.splitlines() if l.split(" ")[0] in ("envs", "vs_episodes", "gathered_rows")).split()
    got = {f[i]: int(f[i + 1]) for i in range(0, len(f), 2)}
    env = evg.EvergladesVecEnv(N, seed=seed, auto_reset=True)
    env.reset()
    for _ in range(turns):
        env.step(env.random_actions())
    tot = env.episode_stats()["totals"]
    o = _np(env.obs).astype(np.int64).reshape(-1)
    chk = int((o * (1 + np.arange(o.size) % 7)).sum())
    assert (got["episodes"], got["p0"], got["p1"], got["tie"]) == tuple(int(x) for x in tot) and got["episodes"] >= 2 * N
    assert got["obs_checksum"] == chk
    env.observe_seat(0)                                      # the client's second part: the learner-seat loop from plain C
    for _ in range(turns):
        so = env.step_vs("swarm", env.random_actions_seat(0), seat=0)[0]
    tot = env.episode_stats()["totals"]
    o = _np(so).astype(np.int64).reshape(-1)
    assert (got["vs_episodes"], got["vs_p0"], got["vs_p1"], got["vs_tie"]) == tuple(int(x) for x in tot) and got["vs_episodes"] > got["episodes"]
    assert got["vs_obs_checksum"] == int((o * (1 + np.arange(o.size) % 7)).sum())
    # ... and its third part: the path's one exchange through evg_comm_init / evg_gather_returns (RCCL opened by the library, a one-rank communicator here)
    st = env.episode_stats()
    assert got["gathered_rows"] == N and got["gathered_length_sum"] == int(st["length"].sum())
    assert [got["gathered_p0"], got["gathered_p1"], got["gathered_tie"],
            got["gathered_unfinished"]] == [int((st["winner"] == k).sum()) for k in (0, 1, 2)] + [int((st["winner"] < 0).sum())]
    env.close()


def test_evaluate_harness_matches_sequential_semantics(evg, oracle_mod):
    """everglades_amd.evaluate(): the reference's evaluation loop (evaluate.py:127-181) batched -- E consecutive episodes
    per env, agent objects alive across them, an agent is not consulted once its game is over, seat-0 win bookkeeping.
    The oracle plays the same harness loop game by game; winners must agree env by env, for native bots on both seats
    (fused persistent rollout) and for a host-side policy callable on seat 0."""
    import torch
    N, E, seed = 96, 3, 21
    for p0, p1 in (("cycle_rush_turn25", "swarm"), ("cycle_target_node11P2", "base_rush_v1"), ("dfs_attack", "bull_rush")):
        res = evg.evaluate(p0, p1, N * E, num_envs=N, seed=seed)
        ora = oracle_mod.Oracle(N, seed=seed, auto_reset=False)
        i0, i1 = evg._lib.POLICY_NAMES.index(p0), evg._lib.POLICY_NAMES.index(p1)
        want = []
        for ep in range(E):
            obs = ora.reset()
            for t in range(150):
                a = np.zeros((N, 2, 7, 2), np.int32)
                ora.scripted_actions(i0, 0, obs, a)
                ora.scripted_actions(i1, 1, obs, a)
                obs, _, done, _ = ora.step(a)
                if done.all():
                    break
            want.append(ora.episode_stats()["winner"].copy())
        want = np.stack(want)
        assert np.array_equal(res["winners"], want), (p0, p1)
        assert res["games"] == N * E and res["wins"] == int((want == 0).sum()) and res["ties"] == int((want == 2).sum())
        lo, hi = res["confint"]
        assert 0.0 <= lo <= res["win_rate"] <= hi <= 1.0

    # a host-side policy on seat 0: the same orders as the native bot, computed from the observation tensor by the device
    # helper -- the generic per-turn path of evaluate() must give the fused path's result
    helper = evg.EvergladesVecEnv(N, seed=seed, auto_reset=False)

    def same_commands(obs):
        return torch.tensor([[i + 1, i + 1] for i in range(7)], dtype=torch.int32, device=obs.device).expand(N, 7, 2)

    a = evg.evaluate(same_commands, "swarm", N, num_envs=N, seed=seed)
    b = evg.evaluate("same_commands", "swarm", N, num_envs=N, seed=seed)
    assert np.array_equal(a["winners"], b["winners"]) and a["mean_length"] == b["mean_length"]
    helper.close()


def _custom_tables(evg, oracle_mod):
    """A table set that differs from DemoMap / UnitDefinitions in every runtime table: distances, control points, node
    defenses (non-dyadic), resources, unit stats, player 1's node map.  Returns (device tables, oracle tables) with identical contents."""
    import ctypes as C
    t = evg.default_tables()
    conn = {1: {2: 5, 4: 7}, 2: {1: 5, 3: 3, 5: 6}, 3: {2: 3, 4: 2, 5: 4, 6: 5, 7: 4}, 4: {1: 7, 3: 2, 7: 6},
            5: {2: 6, 3: 4, 8: 3, 9: 7}, 6: {3: 5, 9: 2}, 7: {3: 4, 4: 6, 9: 3, 10: 5}, 8: {5: 3, 9: 6, 11: 4},
            9: {5: 7, 6: 2, 7: 3, 8: 6, 10: 1}, 10: {7: 5, 9: 1, 11: 7}, 11: {8: 4, 10: 7}}
    for a in range(12):
        for b in range(12):
            t.node_dist[a][b] = conn.get(a, {}).get(b, 0)
    for i, (cp, dfn, res) in enumerate([(300, 0.7, 0), (60, 1.3, 2), (120, 2.9, 0), (45, 0.35, 1), (100, 1.15, 2), (80, 3.3, 0), (100, 0.05, 1),
                                        (75, 1.7, 0), (110, 2.45, 2), (50, 0.9, 1), (280, 1.1, 0)], start=1):
        t.node_control_points[i], t.node_defense[i], t.node_resource[i] = cp, dfn, res
    for u, (h, d, s, c, k) in enumerate([(5, 1, 2, 3, 2), (3, 2, 1, 1, 3), (2, 3, 3, 2, 1)]):
        t.unit_health[u], t.unit_damage[u], t.unit_speed[u], t.unit_control[u], t.unit_cost[u] = h, d, s, c, k
    t.max_turns = 97
    # a board flip that is NOT its own inverse (DemoMap's is): slot s of player 1's view shows node map[s], so node n sits in slot
    # inverse[n], while a group's location and an order's node id go through map itself (server.py:437-439, :485-486, :233-234)
    for i, v in enumerate([0, 11, 8, 9, 10, 6, 7, 5, 2, 3, 4, 1]):
        t.p1_node_map[i] = v
    ot = oracle_mod.Tables()
    assert C.sizeof(ot) == C.sizeof(t)
    C.memmove(C.byref(ot), C.byref(t), C.sizeof(t))
    return t, ot


def test_random_tables_vs_oracle(evg, oracle_mod, tmp_path):
    """HIP against the oracle on RANDOM configurations of the domain include/evg.h states (oracle/custom_configs.random_config: random directed graphs,
    control points 1..511, two-decimal defenses, random resources / bases / unit files) -- the same generator with which tests/test_oracle_vs_live_reference.py
    holds the ORACLE against the live reference where it is mounted (150 configurations checked at the time of writing).  Per configuration: the files go through
    tables_from_json -> evg_create; 110 turns one launch per turn (random and wild orders, auto-reset) compared every turn; the persistent form of a small
    batch (four-lane kernel) and of a 65 568-env batch (two-lane kernel, one whole round + a ragged remainder) against the oracle's final state."""
    import json
    import custom_configs as cc
    from gen_policies import policy_actions
    fast = 0
    for c in range(8):
        rng = np.random.default_rng([20261009, c])
        mobj, uobj = cc.random_config(rng)
        mp, up = tmp_path / ("m%d.json" % c), tmp_path / ("u%d.json" % c)
        mp.write_text(json.dumps(mobj)); up.write_text(json.dumps(uobj))
        tables = evg.tables_from_json(str(mp), str(up))
        ot = oracle_mod.tables_from_json_text(json.dumps(mobj), json.dumps(uobj))
        assert bytes(tables) == bytes(ot)
        N, seed = 160, 400 + c
        env = evg.EvergladesVecEnv(N, seed=seed, auto_reset=True, tables=tables)
        ora = oracle_mod.Oracle(N, seed=seed, auto_reset=True, tables=ot)
        obs = _np(env.reset()).astype(np.float64)
        assert np.array_equal(obs, ora.reset()), ("reset", c)
        for t in range(110):
            a = policy_actions("wild", obs, t, rng) if (t // 25) % 2 else _np(env.random_actions()).copy()
            o, rew, done, info = env.step(a)
            o_obs, o_rew, o_done, o_info = ora.step(a)
            obs = _np(o).astype(np.float64)
            assert np.array_equal(obs, o_obs), ("obs", c, t)
            assert np.array_equal(_np(info["scores"]), o_info["scores"]) and np.array_equal(_np(info["status"]), o_info["status"]), (c, t)
            assert np.array_equal(_np(done), o_done) and np.allclose(_np(rew), o_rew, rtol=0, atol=REWARD_ATOL), (c, t)
        check_state(env, ora.get_state(), ("stepwise", c))
        assert np.array_equal(_np(env.fog_of_war()), ora.fog_of_war()) and np.array_equal(_np(env.knowledge()), ora.knowledge()), c
        env.rollout_random(90, turns_per_launch=90)                          # persistent form, small batch: the four-lane kernel
        for t in range(90):
            ora.step_noobs(ora.random_actions())
        check_state(env, ora.get_state(), ("persistent, four lanes", c))
        env.close()
        if c < 3:                                                            # ... and the two-lane persistent kernel (one whole round + a remainder)
            NB = 65536 + 32
            env = evg.EvergladesVecEnv(NB, seed=seed, auto_reset=True, tables=tables)
            ora = oracle_mod.Oracle(NB, seed=seed, auto_reset=True, tables=ot)
            env.reset(); ora.reset()
            env.rollout_random(45, turns_per_launch=45)
            for t in range(45):
                ora.step_noobs(ora.random_actions())
            check_state(env, ora.get_state(), ("persistent, two lanes", c))
            assert np.array_equal(env.episode_stats()["totals"], ora.episode_stats()["totals"])
            env.close()
        fast += 1
    assert fast == 8


@pytest.mark.parametrize("force_ieee_div", [False, True])
def test_custom_tables_vs_oracle(evg, oracle_mod, force_ieee_div):
    """Every runtime table changed (map distances, control points, non-dyadic defenses, resources, unit stats, turn limit):
    brawl and random play with auto-reset stay bit-equal to the oracle built from the same tables -- once with the exact
    table-reciprocal quotient (validated per table set at evg_create; the product library) and once with the kernel's
    true-division branch, which a table set that fails that validation would run: no realistic table does (a search over
    2e7 denominators found none), so the branch is forced through the diagnostic library's evg_diag_configure."""
    from gen_policies import policy_actions
    t, ot = _custom_tables(evg, oracle_mod)
    N, seed = 224, 31
    extra = dict(library=evg._lib.DIAG_LIB_PATH, diag=dict(force_ieee_div=True)) if force_ieee_div else {}
    env = evg.EvergladesVecEnv(N, seed=seed, auto_reset=True, tables=t, **extra)
    ora = oracle_mod.Oracle(N, seed=seed, auto_reset=True, tables=ot)
    obs = _np(env.reset()).astype(np.float64)
    assert np.array_equal(obs, ora.reset())
    rng = np.random.default_rng(2)
    for tt in range(230):
        a = policy_actions("brawl", obs, tt, rng) if (tt // 40) % 2 == 0 else _np(env.random_actions()).copy()
        o, rew, done, info = env.step(a)
        o_obs, o_rew, o_done, o_info = ora.step(a)
        obs = _np(o).astype(np.float64)
        assert np.array_equal(obs, o_obs), ("obs", tt)
        assert np.array_equal(_np(info["scores"]), o_info["scores"]) and np.array_equal(_np(info["status"]), o_info["status"])
        assert np.array_equal(_np(done), o_done) and np.allclose(_np(rew), o_rew, rtol=0, atol=REWARD_ATOL)
        if tt % 46 == 45:
            check_state(env, ora.get_state(), tt)
            assert np.array_equal(_np(env.fog_of_war()), ora.fog_of_war()) and np.array_equal(_np(env.sightings()), ora.sightings())
    assert env.episode_stats()["totals"][0] >= 2 * N
    # the same table set through the persistent form (at this batch size: the four-lanes-per-env kernel)
    env.rollout_random(120, turns_per_launch=150)
    for tt in range(120):
        a = ora.random_actions()
        o_obs, _, _, _ = ora.step(a)
    assert np.array_equal(_np(env.obs).astype(np.float64), o_obs) and np.array_equal(_np(env._actions), a)
    check_state(env, ora.get_state(), "persistent")
    # bots evaluated from the ON-CHIP state (ChipView: board slots and own-numbering locations of player 1 go through this table set's NON-involutive
    # p1_node_map and its inverse) against the oracle's agents, which read observations: the learner-seat turn on both seats, then the fused rollout
    P = evg.EvergladesVecEnv.POLICIES
    oa = np.zeros((N, 2, 7, 2), np.int32)
    for seat, pol in ((0, "cycle_target_node11P2"), (1, "cycle_rush_turn25"), (0, "cycle_target_node1")):
        env.scripted_reset(); ora.scripted_reset()           # fresh agent objects for every bot
        for tt in range(70):
            rows = env.random_actions_seat(seat)
            ora.scripted_actions(P[pol], 1 - seat, o_obs, oa)
            oa[:, seat] = _np(rows)
            so, _, _, info = env.step_vs(pol, rows, seat=seat)
            o_obs, _, _, o_info = ora.step(oa)
            assert np.array_equal(_np(so).astype(np.float64), o_obs[:, seat]) and np.array_equal(_np(info["scores"]),
                                                                                                 o_info["scores"]), ("step_vs, custom tables", pol, seat, tt)
    env.scripted_reset(); ora.scripted_reset()
    env.rollout_policies(90, "cycle_target_node11P2", "cycle_target_node", fused=True, turns_per_launch=30)
    for tt in range(90):
        ora.scripted_actions(P["cycle_target_node11P2"], 0, o_obs, oa)
        ora.scripted_actions(P["cycle_target_node"], 1, o_obs, oa)
        o_obs, _, _, _ = ora.step(oa)
    assert np.array_equal(_np(env.obs).astype(np.float64), o_obs) and np.array_equal(_np(env._actions), oa)
    check_state(env, ora.get_state(), "fused bots, custom tables")
    env.close()


def test_long_soak_persistent_vs_oracle(evg, oracle_mod):
    """Fourteen consecutive auto-reset episodes (2 100 turns) of the full 65 536-env batch in the persistent form: a 320-env
    window in the middle of the range stays bit-equal to the oracle after every episode (observations, orders, episode
    results), and the batch-wide win counters add up."""
    N, seed, K, first, EPISODES = 65536, 606, 320, 40000, 14
    env = evg.EvergladesVecEnv(N, seed=seed, auto_reset=True)
    env.reset()
    oracle_mod.lib().evo_set_num_threads(2)                  # 320 envs: more threads only add fork/join time to 2 100 tiny calls
    ora = oracle_mod.Oracle(K, seed=seed, env_id_base=first, auto_reset=True)
    ora.reset()
    sl = slice(first, first + K)
    for ep in range(EPISODES):
        env.rollout_random(150, turns_per_launch=150)
        for _ in range(150):
            a = ora.random_actions()
            o_obs, _, _, _ = ora.step(a)
        assert np.array_equal(_np(env.obs[sl]).astype(np.float64), o_obs), ("obs", ep)
        assert np.array_equal(_np(env._actions[sl]), a), ("orders", ep)
        st, ost = env.episode_stats(), ora.episode_stats()
        assert np.array_equal(st["winner"][sl], ost["winner"]) and np.array_equal(st["length"][sl], ost["length"]), ep
        assert np.allclose(st["returns"][sl], ost["returns"], rtol=0, atol=1e-4)
    tot = env.episode_stats()["totals"]
    assert tot[0] == tot[1] + tot[2] + tot[3] and tot[0] >= EPISODES * N
    oracle_mod.lib().evo_set_num_threads(min(16, len(os.sched_getaffinity(0))))
    env.close()


# ---------------------------------------------------------------------------------------------------------------
# round 2: the north-star acceptance sample, whole-batch comparisons at the headline size, the ABI's error paths on a box
# that has a device, and the HIP path sharded over processes
# ---------------------------------------------------------------------------------------------------------------
def test_ten_thousand_reference_matches_on_device(evg):
    """BASELINE north_star: bit-identical win counts vs the CPU reference over 10 000 seeded matches.  ONE 10 000-env handle plays
    env ids 0..9999 of tests/golden/matches_10k.npz (games played by the imported reference) in the persistent form -- a single
    launch of 150 turns, finished games frozen -- and reproduces every game: length, final scores, status, terminal rewards,
    winner, episode returns, the checksums of both final observations, units alive; hence the win/tie counts, bit for bit."""
    from test_oracle_golden import check_matches
    d = load_golden("matches_10k.npz")
    n = len(d["length"])
    env = evg.EvergladesVecEnv(n, seed=int(d["seed"][0]), env_id_base=0, obs_dtype="float64", auto_reset=False)
    env.reset()
    obs, rew, done, info = env.rollout_random(150, turns_per_launch=150)
    assert int(done.sum()) == n
    st = env.episode_stats()
    o = _np(obs)
    got = dict(length=st["length"].astype(np.int16), scores=_np(info["scores"]), status=_np(info["status"]), reward=_np(rew).astype(np.float64),
               returns=st["returns"].astype(np.float64), winner=st["winner"], obs_final_sum=o.sum(axis=2).astype(np.int32),
               alive_final=o[:, :, 49::5].sum(axis=2).astype(np.int16))
    assert np.array_equal(_np(info["winner"]), d["winner"])
    check_matches(got, d)
    assert st["totals"].tolist() == [n] + d["wins_p0_p1_tie"].tolist()
    env.close()
    # the same games one launch per turn through evg_step, on a second handle: identical outcome
    env = evg.EvergladesVecEnv(n, seed=int(d["seed"][0]), env_id_base=0, auto_reset=False)
    env.reset()
    for _ in range(150):
        obs, rew, done, info = env.step(env.random_actions())
    st2 = env.episode_stats()
    assert np.array_equal(st2["winner"], d["winner"]) and np.array_equal(st2["length"], d["length"]) and np.array_equal(_np(info["scores"]), d["scores"])
    env.close()
    # and through the learner-seat turn: the caller's seat takes its orders from a tensor (evg_random_actions_seat), the other seat is the on-device
    # random_actions bot inside the step kernel (evg_step_vs_policy) -- on either seat the same 10 000 reference-played games
    for seat in (0, 1):
        env = evg.EvergladesVecEnv(n, seed=int(d["seed"][0]), env_id_base=0, auto_reset=False)
        env.reset()
        for _ in range(150):
            so, rew, done, info = env.step_vs("random_actions", env.random_actions_seat(seat), seat=seat)
        st3 = env.episode_stats()
        assert int(done.sum()) == n and st3["totals"].tolist() == [n] + d["wins_p0_p1_tie"].tolist(), seat
        assert np.array_equal(st3["winner"], d["winner"]) and np.array_equal(st3["length"], d["length"]) and np.array_equal(_np(info["scores"]),
                                                                                                                            d["scores"]), seat
        env.close()


def test_ten_thousand_reference_matches_config5_on_device(evg):
    """BASELINE config 5 against the reference itself: the 10 000 matches of tests/golden/matches_config5_10k.npz (the reference's
    own Cycle_BRush_Turn25 and SwarmAgent classes on the reference's server) replayed by ONE 10 000-env handle with both bots fused
    into the step kernel (persistent form, one launch of 150 turns, finished games frozen): every game and the win counts, bit for
    bit; then once more with the standalone agent kernel and one evg_step per turn."""
    from test_oracle_golden import check_matches
    d = load_golden("matches_config5_10k.npz")
    n = len(d["length"])
    env = evg.EvergladesVecEnv(n, seed=int(d["seed"][0]), env_id_base=0, obs_dtype="float64", auto_reset=False)
    env.reset()
    obs, rew, done, info = env.rollout_policies(150, "cycle_rush_turn25", "swarm", turns_per_launch=150)
    assert int(done.sum()) == n
    st = env.episode_stats()
    o = _np(obs)
    got = dict(length=st["length"].astype(np.int16), scores=_np(info["scores"]), status=_np(info["status"]), reward=_np(rew).astype(np.float64),
               returns=st["returns"].astype(np.float64), winner=st["winner"], obs_final_sum=o.sum(axis=2).astype(np.int32),
               alive_final=o[:, :, 49::5].sum(axis=2).astype(np.int16))
    check_matches(got, d)
    assert st["totals"].tolist() == [n] + d["wins_p0_p1_tie"].tolist()
    env.close()
    env = evg.EvergladesVecEnv(n, seed=int(d["seed"][0]), env_id_base=0, auto_reset=False)
    env.reset()
    for _ in range(150):
        env.scripted_actions("cycle_rush_turn25", 0)
        obs, rew, done, info = env.step(env.scripted_actions("swarm", 1))
    st2 = env.episode_stats()
    assert np.array_equal(st2["winner"], d["winner"]) and np.array_equal(st2["length"], d["length"]) and np.array_equal(_np(info["scores"]), d["scores"])
    env.close()
    # ... and through the learner-seat turn (evg_step_vs_policy), in both arrangements the reference's scripts use: the "learner" -- here the other bot's
    # orders, computed by the standalone agent kernel from the ONE-SEAT observation tensor the turn returns -- on seat 0 with SwarmAgent inside the step
    # kernel, and on seat 1 with Cycle_BRush_Turn25 inside.  Same 10 000 reference-played games, same winners, lengths and final scores.
    import torch
    for seat, learner, bot in ((0, "cycle_rush_turn25", "swarm"), (1, "swarm", "cycle_rush_turn25")):
        env = evg.EvergladesVecEnv(n, seed=int(d["seed"][0]), env_id_base=0, auto_reset=False)
        env.reset()
        full = torch.zeros((n, 2, 105), dtype=torch.float32, device=env.device)
        so = env.observe_seat(seat)
        for _ in range(150):
            full[:, seat] = so                                   # the agent kernel reads rows [:, seat] of a [N, 2, 105] tensor
            rows = env.scripted_actions(learner, seat, obs=full)[:, seat].contiguous()
            so, rew, done, info = env.step_vs(bot, rows, seat=seat)
        st3 = env.episode_stats()
        assert int(done.sum()) == n and st3["totals"].tolist() == [n] + d["wins_p0_p1_tie"].tolist(), (seat, st3["totals"])
        assert np.array_equal(st3["winner"], d["winner"]) and np.array_equal(st3["length"], d["length"]) and np.array_equal(_np(info["scores"]),
                                                                                                                            d["scores"]), seat
        env.close()


def _compare_whole_batch(env, ora, o_obs, what):
    """every env of the batch: packed state (groups incl. arrival stamps, nodes, float64 health, turn/status/episode), observations and
    the results of the last finished episode"""
    s, os_ = env.get_state(), ora.get_state()
    for k in ("groups", "nodes", "health", "env"):
        assert np.array_equal(s[k], os_[k]), (what, k, int((s[k] != os_[k]).reshape(len(s[k]), -1).any(axis=1).sum()), "envs differ")
    assert np.array_equal(_np(env.obs).astype(np.float64), o_obs), (what, "observations")
    st, ost = env.episode_stats(), ora.episode_stats()
    assert np.array_equal(st["winner"], ost["winner"]) and np.array_equal(st["length"], ost["length"]) and np.array_equal(st["totals"], ost["totals"]), what
    assert np.allclose(st["returns"], ost["returns"], rtol=0, atol=1e-4), what


@pytest.mark.parametrize("form", ["persistent", "one_launch_per_turn"])
def test_config3_whole_batch_vs_oracle(evg, oracle_mod, form):
    """BASELINE config 3 (65 536 concurrent games, random vs random, auto-reset) compared with the oracle on ALL 65 536 envs --
    not a window: after 310 turns (two resets per env) the final packed state incl. float64 health, the observations, the orders of
    the last turn and the per-env episode results are bit-equal, in both launch forms of the step kernel."""
    N, seed, steps = 65536, 90210, 310
    env = evg.EvergladesVecEnv(N, seed=seed, auto_reset=True)
    ora = oracle_mod.Oracle(N, seed=seed, auto_reset=True)
    env.reset(); ora.reset()
    env.rollout_random(steps, turns_per_launch=150 if form == "persistent" else 1)
    for t in range(steps - 1):
        ora.step_noobs(ora.random_actions())
    a = ora.random_actions()
    o_obs, _, _, _ = ora.step(a)
    assert np.array_equal(_np(env._actions), a), "orders of the last turn"
    _compare_whole_batch(env, ora, o_obs, form)
    env.close()


def test_config5_whole_batch_vs_oracle(evg, oracle_mod):
    """BASELINE config 5 (65 536 games, Cycle_BRush_Turn25 vs SwarmAgent fused into the step kernel, persistent form, auto-reset;
    games end by BaseCapture after 84-94 turns so the envs desynchronise) compared with the oracle's agents + env on ALL envs."""
    N, seed, steps = 65536, 515, 200
    seats = ("cycle_rush_turn25", "swarm")
    pid = [evg.EvergladesVecEnv.POLICIES[s] for s in seats]
    env = evg.EvergladesVecEnv(N, seed=seed, auto_reset=True)
    ora = oracle_mod.Oracle(N, seed=seed, auto_reset=True)
    env.reset()
    o_obs = ora.reset()
    env.rollout_policies(steps, seats[0], seats[1], fused=True, turns_per_launch=150)
    oa = np.zeros((N, 2, 7, 2), np.int32)
    for t in range(steps):
        ora.scripted_actions(pid[0], 0, o_obs, oa)
        ora.scripted_actions(pid[1], 1, o_obs, oa)
        o_obs, _, _, _ = ora.step(oa)
    assert np.array_equal(_np(env._actions), oa), "orders of the last turn"
    _compare_whole_batch(env, ora, o_obs, "config5")
    assert env.episode_stats()["totals"][0] >= 2 * N
    env.close()


def test_abi_error_paths_on_a_device(evg):
    """SURVEY 8(b) "Errors": with a device present, every misuse of the C-ABI comes back as a negative status plus a message --
    no crash, no exception across the ABI, no effect on a live handle."""
    import ctypes as C
    lib = evg.load_library()
    L = evg._lib

    def cfg_default(**kw):
        cfg = L.EvgConfig()
        cfg.struct_size, cfg.abi_version, cfg.num_envs, cfg.device_id = C.sizeof(L.EvgConfig), L.ABI_VERSION, 8, 0
        cfg.tables = evg.default_tables()
        for k, v in kw.items():
            setattr(cfg, k, v)
        return cfg

    def create(cfg):
        h = C.c_void_p()
        rc = lib.evg_create(C.byref(cfg), C.byref(h))
        assert (rc == 0) == bool(h.value)
        return rc, h, lib.evg_last_error().decode()

    for bad in (dict(struct_size=12), dict(abi_version=L.ABI_VERSION + 7), dict(num_envs=0), dict(num_envs=-5), dict(obs_dtype=9), dict(rng_mode=5),
                dict(env_id_base=2 ** 32 - 3), dict(env_id_base=2 ** 50)):
        rc, h, msg = create(cfg_default(**bad))
        assert rc == -1 and msg, bad                                  # EVG_ERR_INVALID
    rc, h, msg = create(cfg_default(device_id=1000))
    assert rc == -2 and "device" in msg                               # EVG_ERR_NO_DEVICE

    def table_case(edit):
        cfg = cfg_default()
        edit(cfg.tables)
        return create(cfg)

    def set_dist(t): t.node_dist[1][2] = 9
    def set_diag(t): t.node_dist[3][3] = 2
    def set_cp(t): t.node_control_points[5] = 600
    def set_def(t): t.node_defense[4] = -1.0
    def set_start(t): t.node_team_start[3] = 0
    def set_map(t): t.p1_node_map[2] = 11
    def set_types(t): t.num_unit_types = 7
    def set_dmg(t): t.unit_damage[1] = 40
    def set_gt(t): t.group_type[1][3] = 3
    def set_size(t): t.group_size[0][0] = 10
    def set_turns(t): t.max_turns = 400
    for edit in (set_dist, set_diag, set_cp, set_def, set_start, set_map, set_types, set_dmg, set_gt, set_size, set_turns):
        rc, h, msg = table_case(edit)
        assert rc == -1 and msg, edit.__name__
    assert lib.evg_create(None, None) == -1

    rc, h, msg = create(cfg_default())
    assert rc == 0, msg
    try:
        import torch
        obs = torch.zeros((8, 2, 105), device="cuda")
        act = torch.zeros((8, 2, 7, 2), dtype=torch.int32, device="cuda")
        rew = torch.zeros((8, 2), device="cuda")
        done = torch.zeros(8, dtype=torch.uint8, device="cuda")
        p = lambda t: C.c_void_p(t.data_ptr())
        assert lib.evg_reset(None, None, None, None) == -1
        assert lib.evg_step(None, p(act), p(obs), p(rew), p(done), None, None, None, None) == -1
        assert lib.evg_step(h, None, p(obs), p(rew), p(done), None, None, None, None) == -1 and b"required" in lib.evg_last_error()
        assert lib.evg_step(h, p(act), p(obs), None, p(done), None, None, None, None) == -1
        assert lib.evg_step(h, p(act), p(obs), p(rew), None, None, None, None, None) == -1
        assert lib.evg_observe(h, None, None) == -1 and lib.evg_random_actions(h, None, None) == -1
        assert lib.evg_fog_of_war(h, None, None, None) == -1 and lib.evg_sightings(h, None, None) == -1
        assert lib.evg_scripted_actions(h, 99, 0, p(obs), p(act), None) == -1 and lib.evg_scripted_actions(h, 1, 2, p(obs), p(act), None) == -1
        assert lib.evg_smart_state(h, 3, p(obs), p(obs), None) == -1
        pk = torch.zeros((9, 4), device="cuda")
        assert lib.evg_pack_episode_results(h, None, None) == -1 and lib.evg_pack_episode_results(None, p(pk), None) == -1
        assert lib.evg_pack_episode_results(h, C.c_void_p(pk.data_ptr() + 4), None) == -1 and b"aligned" in lib.evg_last_error()
        # alignment (include/evg.h "Conventions"): the kernels move observations, orders, rewards and scores with 8- / 16-byte vector accesses, so a
        # misaligned device buffer is refused by every entry point that takes one -- before a launch, with the pointer named
        off = lambda t, b=4: C.c_void_p(t.data_ptr() + b)
        sc = torch.zeros((8, 2), dtype=torch.int32, device="cuda")
        big_obs = torch.zeros((9, 2, 105), device="cuda")
        big_act = torch.zeros((9, 2, 7, 2), dtype=torch.int32, device="cuda")

        def refused(rc, name, need=b"16-byte aligned"):
            assert rc == -1 and need in lib.evg_last_error() and name in lib.evg_last_error(), (rc, lib.evg_last_error())
        refused(lib.evg_reset(h, None, off(big_obs), None), b"obs_out")
        refused(lib.evg_observe(h, off(big_obs), None), b"obs_out")
        refused(lib.evg_step(h, p(act), off(big_obs), p(rew), p(done), None, None, None, None), b"obs_out")
        refused(lib.evg_step(h, off(big_act, 8), p(obs), p(rew), p(done), None, None, None, None), b"actions")
        # rewards and scores are one float2 / int2 per env: 8 bytes suffice (an odd env offset into an [N][2] tensor is a legal buffer), 4 do not
        refused(lib.evg_step(h, p(act), p(obs), off(big_obs, 4), p(done), None, None, None, None), b"reward_out", b"8-byte aligned")
        refused(lib.evg_step(h, p(act), p(obs), p(rew), p(done), None, off(sc, 4), None, None), b"scores_out", b"8-byte aligned")
        big_sc = torch.zeros((9, 2), dtype=torch.int32, device="cuda")
        assert lib.evg_step(h, p(act), p(obs), off(big_obs, 8), p(done), None, off(big_sc, 8), None, None) == 0, lib.evg_last_error()
        torch.cuda.synchronize()
        refused(lib.evg_random_actions(h, off(big_act, 8), None), b"actions_out")
        refused(lib.evg_scripted_actions(h, 1, 0, p(obs), off(big_act, 8), None), b"actions_out")
        refused(lib.evg_rollout_random(h, 3, 1, p(act), off(big_obs), p(rew), p(done), None, None, None, None, None), b"obs_out")
        refused(lib.evg_rollout_random(h, 3, 150, off(big_act, 8), p(obs), p(rew), p(done), None, None, None, None, None), b"actions_buf")
        refused(lib.evg_rollout_policies(h, 3, 1, 1, 3, p(act), off(big_obs), p(rew), p(done), None, None, None, None, None), b"obs_out")
        refused(lib.evg_step_vs_policy(h, 0, p(act), 1, 3, off(big_obs), p(rew), p(done), None, None, None, None), b"obs_seat_out")
        refused(lib.evg_observe_seat(h, 0, off(big_obs), None), b"obs_seat_out")
        refused(lib.evg_smart_state(h, 0, p(obs), off(torch.zeros((9, 12, 59), device="cuda")), None), b"features_out")
        ms = C.c_float()
        assert lib.evg_rollout_random(h, 0, 1, p(act), p(obs), p(rew), p(done), None, None, None, C.byref(ms), None) == -1
        assert lib.evg_rollout_policies(h, 5, 1, 77, 0, p(act), p(obs), p(rew), p(done), None, None, None, None, None) == -1
        assert lib.evg_rollout_policies(h, 5, 0, 1, 3, p(act), None, p(rew), p(done), None, None, None, None, None) == -1
        assert lib.evg_seed_stock_entropy(h, None, None) == -1 and b"STOCK" in lib.evg_last_error()      # handle is in the keyed mode
        assert lib.evg_get_stock_entropy(h, None) == -1 and lib.evg_set_stock_entropy(h, None) == -1
        g = np.zeros((8, 2, 12, 8), np.int32)
        assert lib.evg_set_state(h, g.ctypes.data_as(C.c_void_p), None, None, None) == -1
        # out-of-domain state is refused as a whole (nothing is written)
        env_state = np.zeros((8, 4), np.int32)
        nodes, health = np.zeros((8, 11, 2), np.int32), np.zeros((8, 2, 100), np.float64)
        assert lib.evg_set_state(h, g.ctypes.data_as(C.c_void_p), nodes.ctypes.data_as(C.c_void_p), health.ctypes.data_as(C.c_void_p),
                                 env_state.ctypes.data_as(C.c_void_p)) == -1 and b"out-of-domain" in lib.evg_last_error()
        # the handle still works after all of that
        assert lib.evg_reset(h, None, p(obs), None) == 0 and lib.evg_random_actions(h, p(act), None) == 0
        assert lib.evg_step(h, p(act), p(obs), p(rew), p(done), None, None, None, None) == 0
        torch.cuda.synchronize()
        assert float(obs[0, 0, 0]) == 1.0 and int(done.sum()) == 0
    finally:
        lib.evg_destroy(h)
    lib.evg_destroy(None)                                             # a null handle is ignored
    # the Python mirror validates caller tensors before their pointers reach a kernel
    env = evg.EvergladesVecEnv(8, seed=1)
    env.reset()
    import torch
    with pytest.raises(ValueError):
        env.random_actions(out=torch.zeros((8, 2, 7, 2), dtype=torch.int64, device="cuda"))
    with pytest.raises(ValueError):
        env.fog_of_war(out=torch.zeros((8, 2, 12), dtype=torch.uint8, device="cuda"))
    with pytest.raises(ValueError):
        env.sightings(out=torch.zeros((8, 2, 12, 4), dtype=torch.int8))                    # host tensor
    with pytest.raises(ValueError):
        env.smart_state(0, obs=env.obs[:, :, ::1].transpose(0, 1))                          # wrong shape / strides
    with pytest.raises(ValueError):
        env.scripted_actions("swarm", 0, out=torch.zeros((8, 2, 7, 4), dtype=torch.int32, device="cuda")[..., ::2])   # strided view
    with pytest.raises(ValueError):
        env.step(torch.zeros((7, 2, 7, 2), dtype=torch.int32, device="cuda"))
    with pytest.raises(ValueError):
        env.packed_episode_results(out=torch.zeros((8, 3), device="cuda"))
    pk = env.packed_episode_results()
    assert pk.shape == (8, 4) and float(pk[:, 2].max()) == -1.0       # EVG_WINNER_NONE: no episode finished yet
    env.close()


_SHARD_CHILD = r"""
import json, os, sys
rank, world, total, seed, steps, out_dir, root = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6], sys.argv[7]
sys.path.insert(0, root)
import numpy as np
import torch
import torch.distributed as dist
import everglades_amd as evg
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
first, cnt = evg.shard_range(total, world, rank)
env = evg.EvergladesVecEnv(cnt, device="cuda:0", seed=seed, env_id_base=first, auto_reset=True)
env.reset()
env.rollout_random(steps, turns_per_launch=150)
sd = env.episode_stats_device()
g = evg.gather_episode_results(sd["returns"], sd["length"], sd["winner"], total)     # the path's one collective (convenience form: all ranks)
packed = env.packed_episode_results()                                                # evg_pack_episode_results: the payload bench.py gathers
assert torch.equal(packed[:, :2], sd["returns"]) and torch.equal(packed[:, 2].to(torch.int8), sd["winner"]) and torch.equal(packed[:, 3].to(torch.int32), sd["length"])
full = evg.ResultGather(cnt, total, "cuda:0")(packed)                                # the preallocated form: rank 0 only
assert (full is None) == (rank != 0)
if rank == 0:
    d = evg.ResultGather.split(full)
    assert torch.equal(d["winner"], g["winner"]) and torch.equal(d["length"], g["length"]) and torch.equal(d["returns"], g["returns"])
    assert evg.ResultGather.win_counts(full) == tuple(g["wins"])
s = env.get_state()
np.savez(os.path.join(out_dir, "rank%d.npz" % rank), first=first, cnt=cnt, groups=s["groups"], nodes=s["nodes"], health=s["health"], env=s["env"],
         obs=env.obs.cpu().numpy(), g_returns=g["returns"].cpu().numpy(), g_winner=g["winner"].cpu().numpy(), g_length=g["length"].cpu().numpy(),
         wins=np.array(g["wins"]), totals=env.episode_stats()["totals"])
env.close()
dist.barrier()
dist.destroy_process_group()
"""


def test_hip_path_sharded_over_two_processes_equals_one_handle(evg, tmp_path):
    """SURVEY 8(e) on the PRODUCT path: two freshly spawned processes (started before anything touched the GPU in them) each own a
    HIP handle for their contiguous shard (env_id_base from shard_range) on GPU 0 and gather episode results with the path's one
    collective (gloo here; RCCL needs one GPU per rank).  State, observations, gathered per-env results and win counts equal
    those of ONE handle of the full size: results do not depend on the sharding."""
    import subprocess
    import sys
    from conftest import ROOT
    total, seed, steps, world = 5000, 4242, 320, 2                     # uneven shards of 2500 (not a multiple of 32 envs per wave)
    script = tmp_path / "shard_child.py"
    script.write_text(_SHARD_CHILD)
    port = 29600 + os.getpid() % 300
    envv = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script), str(r), str(world), str(total), str(seed), str(steps), str(tmp_path), ROOT], env=envv)
             for r in range(world)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    one = evg.EvergladesVecEnv(total, seed=seed, auto_reset=True)
    one.reset()
    one.rollout_random(steps, turns_per_launch=150)
    s, st = one.get_state(), one.episode_stats()
    parts = [np.load(tmp_path / ("rank%d.npz" % r)) for r in range(world)]
    assert [int(p["first"]) for p in parts] == [0, 2500] and sum(int(p["cnt"]) for p in parts) == total
    for k in ("groups", "nodes", "health", "env"):
        assert np.array_equal(np.concatenate([p[k] for p in parts]), s[k]), k
    assert np.array_equal(np.concatenate([p["obs"] for p in parts]), _np(one.obs))
    for p in parts:                                                   # every rank holds the results of ALL envs, in global order
        assert np.array_equal(p["g_winner"], st["winner"]) and np.array_equal(p["g_length"], st["length"])
        assert np.allclose(p["g_returns"], st["returns"], rtol=0, atol=1e-6)
        assert p["wins"].tolist() == [int((st["winner"] == k).sum()) for k in (0, 1, 2)] + [int((st["winner"] < 0).sum())]
    assert np.array_equal(sum(p["totals"] for p in parts), st["totals"])
    one.close()


def test_four_lanes_per_env_variant_matches_oracle(evg, oracle_mod):
    """The four-lanes-per-env mapping of the step kernel (csrc/evg_step4.inc: 16 envs per wavefront, a side shared by two lanes):
    the product launches it for persistent rollouts of small batches (faster there), the diagnostic library
    can select it in both launch forms.  It must give the same results: caller-supplied orders incl. negative / aliased / duplicate
    ids, fused random and scripted rollouts in both launch forms, a partial last workgroup."""
    from gen_policies import policy_actions
    D = dict(library=evg._lib.DIAG_LIB_PATH, diag=dict(lanes=4))
    N, seed = 333, 31
    env = evg.EvergladesVecEnv(N, seed=seed, auto_reset=True, **D)
    ora = oracle_mod.Oracle(N, seed=seed, auto_reset=True)
    obs = _np(env.reset()).astype(np.float64)
    assert np.array_equal(obs, ora.reset())
    rng = np.random.default_rng(5)
    for t in range(190):
        a = policy_actions("wild" if t % 3 else "brawl", obs, t, rng) if t < 120 else _np(env.random_actions()).copy()
        o, rew, done, info = env.step(a)
        o_obs, o_rew, o_done, o_info = ora.step(a)
        obs = _np(o).astype(np.float64)
        assert np.array_equal(obs, o_obs), ("obs", t)
        assert np.array_equal(_np(info["scores"]), o_info["scores"]) and np.array_equal(_np(done), o_done)
        assert np.allclose(_np(rew), o_rew, rtol=0, atol=REWARD_ATOL)
    check_state(env, ora.get_state(), "stepwise")
    for tpl in (1, 150):
        env.rollout_random(170, turns_per_launch=tpl)
        for t in range(170):
            a = ora.random_actions()
            o_obs, _, _, _ = ora.step(a)
        assert np.array_equal(_np(env.obs).astype(np.float64), o_obs) and np.array_equal(_np(env._actions), a), tpl
        check_state(env, ora.get_state(), ("rollout", tpl))
    assert np.array_equal(env.episode_stats()["totals"], ora.episode_stats()["totals"])
    env.close()
    seats = ("cycle_target_node11P2", "swarm")
    pid = [evg.EvergladesVecEnv.POLICIES[s] for s in seats]
    env = evg.EvergladesVecEnv(N, seed=seed, auto_reset=True, **D)
    ora = oracle_mod.Oracle(N, seed=seed, auto_reset=True)
    env.reset()
    o_obs = ora.reset()
    env.rollout_policies(230, seats[0], seats[1], fused=True, turns_per_launch=150)
    oa = np.zeros((N, 2, 7, 2), np.int32)
    for t in range(230):
        ora.scripted_actions(pid[0], 0, o_obs, oa)
        ora.scripted_actions(pid[1], 1, o_obs, oa)
        o_obs, _, _, _ = ora.step(oa)
    assert np.array_equal(_np(env.obs).astype(np.float64), o_obs)
    check_state(env, ora.get_state(), "scripted")
    env.close()


@pytest.mark.parametrize("N", [4096, 32769, 49153])       # 32 769: the first size of the three-waves build; 49 153: the first two-lane size
def test_small_batch_persistent_rollout_equals_the_two_lane_kernel(evg, oracle_mod, N):
    """What the PRODUCT library launches for a persistent rollout depends on the batch size (four lanes per env up to 49 152 envs --
    built for two waves per SIMD up to 32 768 envs and for three above --, two lanes beyond).  At 4 096 envs (BASELINE config 2) and
    at 32 769 and 49 153: the product's persistent rollout, the two-lane kernel forced through the diagnostic library (lanes = 64) and the
    oracle end in the same state, observations, orders and episode results -- random orders and the scripted bots of config 5."""
    seed, steps = 77, 180
    oracle_mod.lib().evo_set_num_threads(min(16, len(os.sched_getaffinity(0))))
    ora = oracle_mod.Oracle(N, seed=seed, auto_reset=True)
    ora.reset()
    for t in range(steps):
        a = ora.random_actions()
        o_obs, _, _, _ = ora.step(a)
    for kw in (dict(), dict(library=evg._lib.DIAG_LIB_PATH, diag=dict(lanes=64))):
        env = evg.EvergladesVecEnv(N, seed=seed, auto_reset=True, **kw)
        env.reset()
        env.rollout_random(steps, turns_per_launch=150)
        assert np.array_equal(_np(env._actions), a), kw
        _compare_whole_batch(env, ora, o_obs, ("random", tuple(kw)))
        env.close()
    seats = ("cycle_rush_turn25", "swarm")
    pid = [evg.EvergladesVecEnv.POLICIES[s] for s in seats]
    ora = oracle_mod.Oracle(N, seed=seed, auto_reset=True)
    o_obs = ora.reset()
    oa = np.zeros((N, 2, 7, 2), np.int32)
    for t in range(steps):
        ora.scripted_actions(pid[0], 0, o_obs, oa)
        ora.scripted_actions(pid[1], 1, o_obs, oa)
        o_obs, _, _, _ = ora.step(oa)
    for kw in (dict(), dict(library=evg._lib.DIAG_LIB_PATH, diag=dict(lanes=64))):
        env = evg.EvergladesVecEnv(N, seed=seed, auto_reset=True, **kw)
        env.reset()
        env.rollout_policies(steps, seats[0], seats[1], fused=True, turns_per_launch=150)
        _compare_whole_batch(env, ora, o_obs, ("scripted", tuple(kw)))
        env.close()


def test_config4_eight_shards_of_65536_vs_oracle(evg, oracle_mod):
    """BASELINE config 4 at its full size on the HIP path: 524 288 games = 8 shards of 65 536 with global env ids r * 65 536 + e
    (contiguous shards, global ids key the random streams), played here one shard after the other on the one GPU of the
    box -- persistent rollout of 170 turns with auto-reset, so every env finishes an episode and starts the next.  The packed
    16-byte result rows concatenated in rank order (what the RCCL gather delivers to rank 0), the win counts and every shard's
    final state (groups, nodes, float64 health) and orders must equal ONE oracle run over all 524 288 envs (evaluate.py:155-160
    win rule).  Only the transport between GPUs is not exercised here."""
    import torch
    R, n, seed, steps = 8, 65536, 20261004, 170
    total = R * n
    oracle_mod.lib().evo_set_num_threads(min(16, len(os.sched_getaffinity(0))))
    ora = oracle_mod.Oracle(total, seed=seed, auto_reset=True)
    ora.reset()
    for t in range(steps):
        a = ora.random_actions()
        ora.step_noobs(a)
    ost, os_ = ora.episode_stats(), ora.get_state()
    assert (ost["length"] == 150).mean() > 0.98 and ost["totals"][0] >= total            # every env finished (at least) one episode
    rows, totals = [], np.zeros(4, np.int64)
    for r in range(R):
        first, cnt = evg.shard_range(total, R, r)
        assert (first, cnt) == (r * n, n)
        env = evg.EvergladesVecEnv(cnt, seed=seed, env_id_base=first, auto_reset=True)
        env.reset()
        env.rollout_random(steps, turns_per_launch=150)
        rows.append(env.packed_episode_results().clone())
        st = env.episode_stats()
        totals += st["totals"]
        s = env.get_state()
        sl = slice(first, first + cnt)
        for k in ("groups", "nodes", "health", "env"):
            assert np.array_equal(s[k], os_[k][sl]), ("shard", r, k, int((s[k] != os_[k][sl]).reshape(cnt, -1).any(axis=1).sum()), "envs differ")
        assert np.array_equal(_np(env._actions), a[sl]), ("orders of the last turn, shard", r)
        env.close()
    full = _np(torch.cat(rows, dim=0))                       # [524 288, 4] in global env order: return p0, return p1, winner, length
    assert full.shape == (total, 4)
    assert np.array_equal(full[:, 2].astype(np.int8), ost["winner"]) and np.array_equal(full[:, 3].astype(np.int32), ost["length"])
    assert np.allclose(full[:, :2], ost["returns"], rtol=0, atol=1e-4)
    assert np.array_equal(totals, ost["totals"])
    wins = evg.ResultGather.win_counts(torch.cat(rows, dim=0))
    w = ost["winner"]
    assert wins == (int((w == 0).sum()), int((w == 1).sum()), int((w == 2).sum()), 0) and sum(wins) == total


def _melee_state(N, melee, node=6, turn=10):
    """State arrays (evg_set_state layout): in the envs of `melee` all 24 groups stand at `node` at full health, idle, with mixed
    arrival stamps (list order != gid order) and the node held by one of the players in part of the envs (structure defence applies to
    that side's units, server.py:592-597); every other env is in the game_init position."""
    groups = np.zeros((N, 2, 12, 8), np.int32)
    nodes = np.zeros((N, 11, 2), np.int32)
    health = np.full((N, 2, 100), 100.0)
    env = np.zeros((N, 4), np.int32)
    for p in range(2):
        for k in range(12):
            groups[:, p, k] = [1 if p == 0 else 11, -1, 0, 0, 0, 0, 8 if k < 11 else 12, 0]
    nodes[:, :, 1] = -1
    nodes[:, 0] = [500, 0]
    nodes[:, 10] = [-500, 1]
    for e in np.flatnonzero(melee):
        for p in range(2):
            for k in range(12):
                groups[e, p, k, 0] = node
                groups[e, p, k, 7] = (k * 5 + 3 * p + e) % 9          # arrival turn < `turn`
        nodes[e, node - 1] = [[0, -1], [100, 0], [-100, 1], [37, -1]][e % 4]
        env[e] = [turn, 0, 0, 0]
    return groups, nodes, health, env


def _pool_words(groups, envs_per_wave):
    """Words of the shared damage pool every wavefront of the step kernel needs this turn (csrc/evg_kernels.hip stage 0/1: one byte
    per alive fighting unit, rounded up to a word per side and contested node), from the evg_get_state group rows."""
    loc, moving, cnt = groups[..., 0], groups[..., 4], groups[..., 6]
    fight = (cnt > 0) & (moving == 0)
    words = np.zeros(groups.shape[0], np.int64)
    for n in range(1, 12):
        u = (((loc == n) & fight) * cnt).sum(axis=2)                  # [N, 2] alive fighting units per side at node n
        words += ((u[:, 0] > 0) & (u[:, 1] > 0)) * ((u[:, 0] + 3) // 4 + (u[:, 1] + 3) // 4)
    pad = (-len(words)) % envs_per_wave
    return np.concatenate([words, np.zeros(pad, np.int64)]).reshape(-1, envs_per_wave).sum(axis=1)


def test_damage_pool_overflow_takes_two_passes_two_lane_kernel(evg, oracle_mod):
    """The branch `npass == 2` of the two-lane step kernel (csrc/evg_kernels.hip: a wavefront whose fights need more than
    DP_CAP = 1 536 pool words resolves envs 0..15 and 16..31 in two passes; reference semantics at stake: server.py:549-566,
    573-644).  Random games never get there (armies bleed before all 200 units meet), so the position is planted: all 24 groups
    at node 6 at full health in 64 consecutive envs = 2 x 100 units = 50 words per env = 1 600 words per wavefront.  Through
    evg_step (single-turn form), turn by turn against the oracle; the precondition is asserted from the state before each turn."""
    N = 96                                                   # wavefronts 0 and 1: melee; wavefront 2: ordinary openings
    melee = np.arange(N) < 64
    st0 = _melee_state(N, melee)
    env = evg.EvergladesVecEnv(N, seed=99, auto_reset=False)
    ora = oracle_mod.Oracle(N, seed=99, auto_reset=False)
    env.reset(); ora.reset()
    env.set_state(*st0); ora.set_state(*st0)
    assert np.array_equal(_np(env.observe()).astype(np.float64), ora.observe())
    two_pass_turns = 0
    for t in range(6):
        words = _pool_words(env.get_state()["groups"], 32)
        if t < 2:
            assert words[0] > 1536 and words[1] > 1536 and words[2] <= 1536, words
        two_pass_turns += int(words[0] > 1536) + int(words[1] > 1536)
        a = np.zeros((N, 2, 7, 2), np.int32) if t < 3 else _np(env.random_actions()).copy()    # hold (node 0 is no node), then random orders
        obs, rew, done, info = env.step(a)
        o_obs, o_rew, o_done, o_info = ora.step(a)
        assert np.array_equal(_np(obs).astype(np.float64), o_obs), ("obs", t)
        assert np.array_equal(_np(info["scores"]), o_info["scores"]) and np.array_equal(_np(done), o_done)
        check_state(env, ora.get_state(), ("two-pass combat, turn", t))
    assert two_pass_turns >= 6
    h = env.get_state()["health"]
    assert (h[:64] < 100.0).mean() > 0.5                     # the melee really happened
    env.close()
    # the persistent instantiation of the same kernel (MULTI = true; forced through the diagnostic library at this batch size)
    env = evg.EvergladesVecEnv(N, seed=99, auto_reset=True, library=evg._lib.DIAG_LIB_PATH, diag=dict(lanes=64))
    ora = oracle_mod.Oracle(N, seed=99, auto_reset=True)
    env.reset(); ora.reset()
    env.set_state(*st0); ora.set_state(*st0)
    assert _pool_words(ora.get_state()["groups"], 32)[0] > 1536
    env.rollout_random(3, turns_per_launch=3)                # turn 1 overflows (turn 2 no longer: the kernel-drawn orders send ~1.3 groups per side away)
    for t in range(3):
        a = ora.random_actions()
        o_obs, _, _, _ = ora.step(a)
    assert np.array_equal(_np(env.obs).astype(np.float64), o_obs) and np.array_equal(_np(env._actions), a)
    check_state(env, ora.get_state(), "two-pass combat, persistent two-lane form")
    env.close()


def test_damage_pool_overflow_takes_two_passes_four_lane_kernel(evg, oracle_mod):
    """The same branch of the four-lanes-per-env kernel (csrc/evg_step4.inc: 16 envs per wavefront, DP_CAP = 640 words, passes over
    envs 0..7 and 8..15), which the PRODUCT library runs for persistent rollouts of up to 49 152 envs: the planted melee needs
    16 x 50 = 800 words per wavefront.  One 3-turn persistent launch with kernel-drawn random orders against the oracle; the
    precondition of turns 1 and 2 is asserted from the oracle's states, which the final comparison ties to the device's."""
    N = 80                                                   # wavefronts 0..2 melee, 3 ordinary, 4 mixed (half melee: 400 words, one pass)
    melee = (np.arange(N) < 48) | ((np.arange(N) >= 64) & (np.arange(N) < 72))
    st0 = _melee_state(N, melee)
    env = evg.EvergladesVecEnv(N, seed=7, auto_reset=True)
    ora = oracle_mod.Oracle(N, seed=7, auto_reset=True)
    env.reset(); ora.reset()
    env.set_state(*st0); ora.set_state(*st0)
    w = _pool_words(env.get_state()["groups"], 16)
    assert (w[:3] > 640).all() and w[3] <= 640 and 0 < w[4] <= 640, w
    env.rollout_random(3, turns_per_launch=3)                # persistent form at N <= 49 152: evg_step4_kernel
    for t in range(3):
        a = ora.random_actions()
        o_obs, _, _, _ = ora.step(a)
        if t == 0:
            assert (_pool_words(ora.get_state()["groups"], 16)[:3] > 640).all()       # turn 2 of the launch overflows as well
    assert np.array_equal(_np(env.obs).astype(np.float64), o_obs) and np.array_equal(_np(env._actions), a)
    check_state(env, ora.get_state(), "two-pass combat, four-lane kernel")
    assert (env.get_state()["health"][:48] < 100.0).mean() > 0.5
    env.close()


def test_launch_plan_follows_the_device_and_chunked_rollouts_match_oracle(evg, oracle_mod):
    """The kernel selection is derived from what the device holds (compute units from hipDeviceProp_t x the kernels' own occupancy;
    evg_launch_plan reports both), not from 256-CU literals, and a persistent rollout of a batch BEYOND what the device holds at once is
    planned so that no remainder runs alone at low occupancy (csrc/evg_kernels.hip, plan_step): whole rounds as a plain launch; the
    last whole round together with a remainder of up to 60 % of a round as ONE CHUNKED launch -- as many workgroups as the device
    holds, each taking units (set of 32 envs) x (chunk of 25 turns) from its XCD's queue and handing the set on through HBM --; a
    larger remainder as its own launch.  On a whole MI355X: 98 304 envs = 3 072 sets x 6 chunks.  Random orders and the scripted bots
    of config 5 (agent objects are handed on too), against the forced plain two-lane launch (diagnostic library) and the oracle,
    every env."""
    import re
    import torch
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    probe = evg.EvergladesVecEnv(64, seed=1)
    n, text = probe.launch_plan(150)
    m = re.search(r"device: (\d+) CUs, resident wavefronts two-lane (\d+), four-lane (\d+) / (\d+)", text)
    assert n == 1 and m, text
    d_cus, s2, s4a, s4b = (int(x) for x in m.groups())
    assert d_cus == cus and s2 == 8 * cus and s4a == 8 * cus and s4b == 12 * cus, text       # 2 / 2 / 3 waves per SIMD
    assert probe.launch_plan(1)[0] == 1 and "single-turn" in probe.launch_plan(1)[1]
    probe.close()
    cap2, cap4a, cap4b = 32 * s2, 16 * s4a, 16 * s4b
    for N, want in ((cap4a, ["four lanes per env, built for 2"]), (cap4a + 1, ["four lanes per env, built for 3"]),
                    (cap4b + 1, ["two lanes per env, persistent>"]),
                    (cap2, ["two lanes per env, persistent>"]), (2 * cap2, ["two lanes per env, persistent>[envs 0..%d:" % (2 * cap2)]),
                    (cap2 + 1, ["chunked>[envs 0..%d: %d sets of 32 envs x 6 chunks of 25 turns" % (cap2 + 1, s2 + 1)]),
                    (cap2 + cap4a, ["chunked>[envs 0..%d: %d sets of 32 envs x 6 chunks" % (cap2 + cap4a, s2 + s4a // 2)]),
                    (2 * cap2 + 4480 + 7, ["persistent>[envs 0..%d:" % cap2, "chunked>[envs %d..%d:" % (cap2, 2 * cap2 + 4480 + 7)]),
                    (cap2 + cap4b, ["persistent>[envs 0..%d:" % cap2, "four lanes per env, built for 3 waves per SIMD>[envs %d..%d:" % (cap2, cap2 + cap4b)])):
        e = evg.EvergladesVecEnv(N, seed=1)
        n, text = e.launch_plan(150)
        parts = text.split(" | ")[0].split(" + ")
        assert n == len(want) == len(parts) and all(w in p_ for w, p_ in zip(want, parts)), (N, text)
        assert "chunked" not in e.launch_plan(25)[1] and "chunked" not in e.launch_plan(1)[1]      # nothing to hand on in a launch of one chunk
        e.close()
    seed, steps = 1234, 195                                  # 150-turn launch (6 chunks) + 45-turn launch (2 chunks: 25 + 20)
    for N in (cap2 + cap4a, 2 * cap2 + 4480 + 7, cap2 + cap4b):      # chunked | plain round + chunked (ragged last set) | plain round + four-lane remainder
        ora = oracle_mod.Oracle(N, seed=seed, auto_reset=True)
        ora.reset()
        for t in range(steps):
            a = ora.random_actions()
            o_obs, _, _, _ = ora.step(a)
        for kw in (dict(), dict(library=evg._lib.DIAG_LIB_PATH, diag=dict(lanes=64))):
            if kw and N != cap2 + cap4a:
                continue                                     # the forced plain launch is compared at the first size
            env = evg.EvergladesVecEnv(N, seed=seed, auto_reset=True, **kw)
            assert ("chunked" in env.launch_plan(150)[1]) == (not kw and N != cap2 + cap4b)
            env.reset()
            env.rollout_random(steps, turns_per_launch=150)
            assert np.array_equal(_np(env._actions), a), (N, kw)
            _compare_whole_batch(env, ora, o_obs, ("planned rollout", N, tuple(kw)))      # episode_stats also fails on a chunk hand-over fault
            if not kw and N == cap2 + cap4a:
                # the state a chunked launch leaves in HBM is what every other entry point continues from: two evg_step calls with
                # caller-supplied orders, then a masked reset, against the oracle
                ora2 = oracle_mod.Oracle(N, seed=seed, auto_reset=True)
                ora2.reset()
                ora2.set_state(*[ora.get_state()[k] for k in ("groups", "nodes", "health", "env")])
                for t in range(2):
                    a2 = _np(env.random_actions()).copy()
                    obs2, _, _, info2 = env.step(a2)
                    o2, _, _, oi2 = ora2.step(a2)
                    assert np.array_equal(_np(obs2).astype(np.float64), o2) and np.array_equal(_np(info2["scores"]),
                                                                                               oi2["scores"]), ("evg_step after a chunked launch", t)
                mask = (np.arange(N) % 7 == 0).astype(np.uint8)
                assert np.array_equal(_np(env.reset(mask=mask)).astype(np.float64)[mask != 0], ora2.reset(mask=mask)[mask != 0])
                check_state(env, ora2.get_state(), "after chunked launch + steps + masked reset")
                del ora2
            env.close()
        del ora
    N = cap2 + 8192
    seats = ("cycle_rush_turn25", "swarm")
    pid = [evg.EvergladesVecEnv.POLICIES[s_] for s_ in seats]
    ora = oracle_mod.Oracle(N, seed=seed, auto_reset=True)
    o_obs = ora.reset()
    oa = np.zeros((N, 2, 7, 2), np.int32)
    for t in range(150):
        ora.scripted_actions(pid[0], 0, o_obs, oa)
        ora.scripted_actions(pid[1], 1, o_obs, oa)
        o_obs, _, _, _ = ora.step(oa)
    for kw in (dict(), dict(library=evg._lib.DIAG_LIB_PATH, diag=dict(lanes=2))):      # lanes = 2: the chunked form forced over the whole batch
        env = evg.EvergladesVecEnv(N, seed=seed, auto_reset=True, **kw)
        assert "chunked" in env.launch_plan(150)[1]
        env.reset()
        env.rollout_policies(150, seats[0], seats[1], fused=True, turns_per_launch=150)
        assert np.array_equal(_np(env._actions), oa)
        _compare_whole_batch(env, ora, o_obs, ("chunked scripted rollout", tuple(kw)))
        env.close()


def test_a_lost_chunk_hand_over_ends_the_launch_and_is_reported(evg):
    """The safety net of the chunked form: if a set's chunk is never published (simulated through the diagnostic library), the workgroup
    that took the set's next chunk does not spin for ever -- its wait is bounded in time (5 s of s_memrealtime), it flags the handle and
    plays on, every workgroup leaves when the queues are empty -- and the fault is visible WHEREVER results leave the handle:
    evg_check_fault, evg_episode_stats, evg_episode_stats_device, evg_get_state and a timed rollout fail with EVG_ERR_FAULT,
    evg_pack_episode_results (what the multi-GPU gather sends) writes poisoned rows that the win bookkeeping refuses.  The word is sticky."""
    import time
    import torch
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    N = 32 * 8 * cus + 2048
    env = evg.EvergladesVecEnv(N, seed=3, auto_reset=True, library=evg._lib.DIAG_LIB_PATH, diag=dict(ablate=64))
    assert "chunked" in env.launch_plan(150)[1]
    env.reset()
    assert env.check_fault() == 0
    t0 = time.perf_counter()
    env.rollout_random(150, turns_per_launch=150)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert 3.0 < dt < 30.0, dt                                   # one bounded wait, then the grid drained
    with pytest.raises(evg.EvgFault, match="hand a set of envs on"):
        env.check_fault()
    with pytest.raises(evg.EvgFault, match="hand a set of envs on"):
        env.episode_stats()
    with pytest.raises(evg.EvgFault):
        env.episode_stats_device()
    with pytest.raises(evg.EvgFault):
        env.get_state()
    counts = torch.zeros(4, dtype=torch.int64, device=env.device)
    rows = env.packed_episode_results(counts=counts)
    w = _np(rows)
    assert (w[:, 2] == -2).all() and np.isnan(w[:, :2]).all() and (w[:, 3] == -1).all()        # poisoned: winner -2 is no EVG_WINNER_* value
    assert _np(counts).tolist() == [-1, -1, -1, -1]
    with pytest.raises(evg.EvgFault, match="poisoned"):
        evg.ResultGather.win_counts(rows)
    with pytest.raises(evg.EvgFault, match="poisoned"):
        evg.win_counts(evg.ResultGather.split(rows))
    env.reset()                                                  # sticky: a reset does not make the handle valid again
    with pytest.raises(evg.EvgFault):
        env.rollout_random(3, turns_per_launch=1, time_kernel=True)        # a timed call synchronises and reports
    env.close()
    ok = evg.EvergladesVecEnv(N, seed=3, auto_reset=True, library=evg._lib.DIAG_LIB_PATH)      # the same library without the knob: no fault
    ok.reset()
    out = ok.rollout_random(150, turns_per_launch=150, time_kernel=True)
    assert out[-1] > 0 and ok.check_fault() == 0
    assert ok.episode_stats()["totals"][0] >= N
    counts = torch.zeros(4, dtype=torch.int64, device=ok.device)
    w = _np(ok.packed_episode_results(counts=counts))[:, 2]
    assert (w >= 0).all() and _np(counts).tolist() == [int((w == k).sum()) for k in (0, 1, 2)] + [0]      # the pack kernel's own win bookkeeping of its rows
    ok.close()


def test_a_stale_chunk_hand_over_is_reported(evg, oracle_mod):
    """The chunked form's hand-over is cheaper than what the memory model asks for (store drain + relaxed flag inside one XCD's L2 instead of an agent-scope
    release; step_kernel.inc says what it relies on).  If that ever failed to deliver the producer's LATEST words, the consumer must notice: every lane hands
    on a checksum over (chunk number, every state word it stored) and the lane that takes the set's next chunk recomputes it over the words it LOADED.
    Simulated through the diagnostic library (ablate bit 24): the first chunk of the launch's first set keeps its group words to itself -- memory still holds
    the words of the chunk's start, exactly what a stale line would serve -- while flag and checksum are published as usual.  The consumer flags the handle
    (fault bit 3 = 8; sticky; visible wherever results leave the handle); without the knob the same launches are clean and equal the oracle's games."""
    import torch
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    N = 32 * 8 * cus + 2048
    env = evg.EvergladesVecEnv(N, seed=5, auto_reset=True, library=evg._lib.DIAG_LIB_PATH, diag=dict(ablate=1 << 24))
    assert "chunked" in env.launch_plan(150)[1]
    env.reset()
    assert env.check_fault() == 0
    env.rollout_random(150, turns_per_launch=150)
    torch.cuda.synchronize()
    with pytest.raises(evg.EvgFault, match="stale state words"):
        env.check_fault()
    word = C.c_uint32(0)
    assert env.L.evg_check_fault(env._h, C.byref(word)) == evg._lib.ERR_FAULT and word.value == 8           # only the stale-data bit
    with pytest.raises(evg.EvgFault):
        env.get_state()
    assert (_np(env.packed_episode_results())[:, 2] == -2).all()                                            # poisoned rows: nothing leaves unnoticed
    env.close()
    # the same library and launches without the knob: no fault, and the games are the oracle's -- also for the scripted bots, whose agent words are part of
    # what is handed on and summed
    for policies in (None, ("cycle_rush_turn25", "swarm")):
        ok = evg.EvergladesVecEnv(N, seed=5, auto_reset=True, library=evg._lib.DIAG_LIB_PATH)
        ora = oracle_mod.Oracle(N, seed=5, auto_reset=True)
        ok.reset()
        o_obs = ora.reset()
        P = evg.EvergladesVecEnv.POLICIES
        oa = np.zeros((N, 2, 7, 2), np.int32)
        if policies is None:
            ok.rollout_random(60, turns_per_launch=60)
            for _ in range(60):
                ora.step_noobs(ora.random_actions())
        else:
            ok.rollout_policies(60, policies[0], policies[1], fused=True, turns_per_launch=60)
            for _ in range(60):
                ora.scripted_actions(P[policies[0]], 0, o_obs, oa)
                ora.scripted_actions(P[policies[1]], 1, o_obs, oa)
                o_obs, _, _, _ = ora.step(oa)
        assert ok.check_fault() == 0
        check_state(ok, ora.get_state(), "chunked hand-over, checksum clean")
        ok.close()


@pytest.mark.parametrize("variant", ["float64", "int16", "float32-release"])
def test_chunked_form_other_observation_dtypes_and_a_real_release_vs_oracle(evg, oracle_mod, variant):
    """The chunked persistent kernel's float64 and int16 instantiations (the write-out template and the launch plan's footprint rule both
    depend on the dtype): cap2 + 2 048 envs, 160 turns with auto-reset (a 150-turn launch of 6 chunks + a 10-turn launch), every env
    against the oracle.  Third variant: float32 with the chunk published by an agent-scope RELEASE store (diagnostic library, ablate
    bit 7) -- what the memory model asks for; the product's cheaper hand-over (csrc/evg_kernels.hip, "WHAT THIS RELIES ON") must give
    the same result as that and as the oracle."""
    import torch
    cap2 = 32 * 8 * torch.cuda.get_device_properties(0).multi_processor_count
    N, seed, steps = cap2 + 2048, 4242, 160
    kw = dict(obs_dtype=variant) if variant in ("float64", "int16") else dict(library=evg._lib.DIAG_LIB_PATH, diag=dict(ablate=128))
    env = evg.EvergladesVecEnv(N, seed=seed, auto_reset=True, **kw)
    n, text = env.launch_plan(150)
    assert "chunked" in text and n == 1, text
    ora = oracle_mod.Oracle(N, seed=seed, auto_reset=True)
    env.reset(); ora.reset()
    env.rollout_random(steps, turns_per_launch=150)
    for t in range(steps - 1):
        ora.step_noobs(ora.random_actions())
    a = ora.random_actions()
    o_obs, _, _, _ = ora.step(a)
    assert env.obs.dtype == {"float64": torch.float64, "int16": torch.int16}.get(variant, torch.float32)
    assert np.array_equal(_np(env._actions), a), "orders of the last turn"
    _compare_whole_batch(env, ora, o_obs, ("chunked", variant))
    assert env.check_fault() == 0
    env.close()


def test_chunked_form_under_uneven_load_vs_oracle(evg, oracle_mod):
    """Hand-overs under UNEVEN load: sets of 32 envs in a planted melee (all 24 groups on node 6: the most expensive turns the game has, two
    damage-pool passes) alternate with sets in the opening position (no combat for ~10 turns), so the producers of neighbouring sets finish
    their chunks at very different times and consumers find their predecessor anywhere between long done and still running; every
    word of every env (state incl. float64 health, observations, episode results) against the oracle after 150 turns with auto-reset."""
    import torch
    cap2 = 32 * 8 * torch.cuda.get_device_properties(0).multi_processor_count
    N, seed = cap2 + 4096 + 13, 77                            # ragged last set
    melee = ((np.arange(N) // 32) % 3 == 0)
    g, n, h, e = _melee_state(64, np.arange(64) < 32)            # one melee set + one idle set as templates
    reps = np.where(melee, np.arange(N) % 32, 32 + np.arange(N) % 32)
    st0 = (g[reps].copy(), n[reps].copy(), h[reps].copy(), e[reps].copy())
    env = evg.EvergladesVecEnv(N, seed=seed, auto_reset=True)
    assert "chunked" in env.launch_plan(150)[1]
    ora = oracle_mod.Oracle(N, seed=seed, auto_reset=True)
    env.reset(); ora.reset()
    env.set_state(*st0); ora.set_state(*st0)
    w = _pool_words(env.get_state()["groups"], 32)
    assert (w[0::3][:-1] > 1536).all() and (w[1::3] == 0).all()   # every third wavefront starts in the two-pass melee, the others idle
    env.rollout_random(150, turns_per_launch=150)
    for t in range(149):
        ora.step_noobs(ora.random_actions())
    a = ora.random_actions()
    o_obs, _, _, _ = ora.step(a)
    assert np.array_equal(_np(env._actions), a)
    _compare_whole_batch(env, ora, o_obs, "chunked, uneven load")
    assert env.check_fault() == 0
    env.close()


def test_rollouts_without_observations_play_the_same_games(evg, oracle_mod):
    """evg_rollout_*(obs_out = NULL, actions_buf = NULL): the step kernel skips the observation image and its write-out and does not record
    the orders; state, rewards, episode results and win counters are those of the observing rollout (and of the oracle), the observation
    and order tensors keep their old content.  Both kernel mappings (four-lane at 3 000 envs, two-lane at 65 536 + chunked beyond), random
    and scripted orders; the evaluation harness uses this form."""
    import torch
    cap2 = 32 * 8 * torch.cuda.get_device_properties(0).multi_processor_count
    for N, steps in ((3000, 170), (cap2, 170), (cap2 + 2048, 160)):
        for scripted in (False, True):
            a = evg.EvergladesVecEnv(N, seed=11, auto_reset=True)
            b = evg.EvergladesVecEnv(N, seed=11, auto_reset=True)
            a.reset(); b.reset()
            obs_before, act_before = b.obs.clone(), b._actions.clone()
            if scripted:
                a.rollout_policies(steps, "cycle_rush_turn25", "swarm", turns_per_launch=150)
                b.rollout_policies(steps, "cycle_rush_turn25", "swarm", turns_per_launch=150, observe=False, record_actions=False)
            else:
                a.rollout_random(steps, turns_per_launch=150)
                b.rollout_random(steps, turns_per_launch=150, observe=False, record_actions=False)
            sa, sb = a.get_state(), b.get_state()
            for k in ("groups", "nodes", "health", "env"):
                assert np.array_equal(sa[k], sb[k]), (N, scripted, k)
            ea, eb = a.episode_stats(), b.episode_stats()
            for k in ("returns", "length", "winner", "totals"):
                assert np.array_equal(ea[k], eb[k]), (N, scripted, k)
            assert torch.equal(a.reward, b.reward) and torch.equal(a.done, b.done) and torch.equal(a.scores, b.scores)
            assert torch.equal(b.obs, obs_before) and torch.equal(b._actions, act_before)          # untouched
            assert torch.equal(a.observe(), b.observe())                                              # and recoverable from the state
            if N == 3000 and not scripted:
                ora = oracle_mod.Oracle(N, seed=11, auto_reset=True)
                ora.reset()
                for t in range(steps):
                    ora.step_noobs(ora.random_actions())
                check_state(b, ora.get_state(), "no-observation rollout vs oracle")
            a.close(); b.close()
    with pytest.raises(ValueError):
        e = evg.EvergladesVecEnv(64, seed=1)
        e.reset()
        e.rollout_random(3, fused=False, observe=False)


def _wild_seat_rows(rng, N):
    """[N, 7, 2] orders with out-of-domain / duplicate / negative ids (the `wild` policy's ingredients) mixed into valid ones"""
    a = np.stack([rng.integers(-13, 13, size=(N, 7)), rng.integers(-13, 13, size=(N, 7))], axis=-1).astype(np.int32)
    plain = np.stack([rng.integers(0, 12, size=(N, 7)), rng.integers(1, 12, size=(N, 7))], axis=-1).astype(np.int32)
    return np.where(rng.random((N, 1, 1)) < 0.3, a, plain)


@pytest.mark.parametrize("pol", list(range(15)))
def test_step_vs_policy_every_bot_both_seats_vs_oracle(evg, oracle_mod, pol):
    """evg_step_vs_policy -- the loop every training / evaluation script of the reference runs: a caller on one seat, a scripted bot
    on the other (evaluate.py:143-152; dqn_smart_state_training.py:114-122) -- for all 15 bots on both seats at 1 000 envs with
    auto-reset, 230 turns (agents alive across episodes): the caller's orders are random rows with out-of-domain ids mixed in; the
    caller's observation, rewards, scores, done flags every turn and the final state == oracle (scripted_actions for the bot's seat
    from ITS observation + step).  The caller's rows arrive as [N, 7, 2] in one seat and as rows of a [N, 2, 7, 2] tensor in the other."""
    import torch
    N, steps = 1000, 230
    for seat in (0, 1):
        seed = 7000 + 10 * pol + seat
        rng = np.random.default_rng(seed)
        env = evg.EvergladesVecEnv(N, seed=seed, auto_reset=True)
        ora = oracle_mod.Oracle(N, seed=seed, auto_reset=True)
        env.reset()
        o_obs = ora.reset()
        assert np.array_equal(_np(env.observe_seat(seat)).astype(np.float64), o_obs[:, seat])
        both = torch.zeros((N, 2, 7, 2), dtype=torch.int32, device=env.device)
        for t in range(steps):
            rows = _wild_seat_rows(rng, N)
            oa = np.zeros((N, 2, 7, 2), np.int32)
            ora.scripted_actions(pol, 1 - seat, o_obs, oa)
            oa[:, seat] = rows
            if seat == 0:
                obs, rew, done, info = env.step_vs(pol, torch.as_tensor(rows, device=env.device), seat=seat)
            else:
                both[:, seat] = torch.as_tensor(rows, device=env.device)
                both[:, 1 - seat] = -7                           # the other seat's rows of the tensor are ignored
                obs, rew, done, info = env.step_vs(pol, both, seat=seat)
            o_obs, o_rew, o_done, o_info = ora.step(oa)
            assert obs.shape == (N, 105)
            assert np.array_equal(_np(obs).astype(np.float64), o_obs[:, seat]), ("obs", pol, seat, t)
            assert np.array_equal(_np(info["scores"]), o_info["scores"]) and np.array_equal(_np(done), o_done), (pol, seat, t)
            assert np.array_equal(_np(info["winner"]), o_info["winner"]) and np.array_equal(_np(info["status"]), o_info["status"])
            assert np.allclose(_np(rew), o_rew, rtol=0, atol=1e-6)
        check_state(env, ora.get_state(), ("step_vs", pol, seat))
        st, ost = env.episode_stats(), ora.episode_stats()
        assert np.array_equal(st["totals"], ost["totals"]) and np.array_equal(st["winner"], ost["winner"]) and st["totals"][0] >= N
        env.close()


@pytest.mark.parametrize("pol,seat,dtype",
                         [("swarm", 0, "float32"), ("cycle_rush_turn25", 1, "float32"), ("swarm", 1, "float64"), ("cycle_rush_turn25", 0, "int16")])
def test_step_vs_policy_whole_batch_vs_oracle(evg, oracle_mod, pol, seat, dtype):
    """evg_step_vs_policy at the headline size: 65 536 envs + a ragged tail, 200 turns with auto-reset, the caller's orders from the
    on-device generator (evg_random_actions_seat == the seat's rows of evg_random_actions); every env's state incl. float64 health, the
    caller's observation and the episode results against the oracle; all three observation dtypes across the cases."""
    import torch
    N, seed, steps = 65536 + 37, 31337 + seat, 200
    pid = evg.EvergladesVecEnv.POLICIES[pol]
    env = evg.EvergladesVecEnv(N, seed=seed, auto_reset=True, obs_dtype=dtype)
    ora = oracle_mod.Oracle(N, seed=seed, auto_reset=True)
    env.reset()
    o_obs = ora.reset()
    oa = np.zeros((N, 2, 7, 2), np.int32)
    for t in range(steps):
        a = env.random_actions_seat(seat)
        if t in (0, 57, steps - 1):
            assert torch.equal(a, env.random_actions()[:, seat])
        ora.scripted_actions(pid, 1 - seat, o_obs, oa)
        oa[:, seat] = ora.random_actions()[:, seat]
        if t in (0, 57, steps - 1):
            assert np.array_equal(_np(a), oa[:, seat])
        obs, rew, done, info = env.step_vs(pol, a, seat=seat)
        o_obs, _, _, _ = ora.step(oa)
    assert np.array_equal(_np(obs).astype(np.float64), o_obs[:, seat]), "caller's observation"
    s, os_ = env.get_state(), ora.get_state()
    for k in ("groups", "nodes", "health", "env"):
        assert np.array_equal(s[k], os_[k]), (k, int((s[k] != os_[k]).reshape(N, -1).any(axis=1).sum()), "envs differ")
    st, ost = env.episode_stats(), ora.episode_stats()
    assert np.array_equal(st["winner"], ost["winner"]) and np.array_equal(st["length"], ost["length"]) and np.array_equal(st["totals"], ost["totals"])
    assert st["totals"][0] >= N
    # the full observation of the same state and the one-seat one agree; smart-state features from the one-seat tensor == from the full one
    full = env.observe()
    assert torch.equal(full[:, seat], env.observe_seat(seat))
    assert torch.equal(env.smart_state(seat), env.smart_state(seat, obs=env.observe_seat(seat)))
    env.close()


def test_step_vs_policy_equals_two_launch_path_without_auto_reset(evg):
    """evg_step_vs_policy == evg_scripted_actions(bot) + evg_step on the same orders, on frozen (finished, not reset) envs too: a finished
    game's bot is not consulted and its agent object does not advance; argument validation of the new entry points."""
    import torch
    N, seed = 2048, 5
    a_env = evg.EvergladesVecEnv(N, seed=seed, auto_reset=False)
    b_env = evg.EvergladesVecEnv(N, seed=seed, auto_reset=False)
    for seat, pol in ((0, "cycle_rush_turn25"), (1, "cycle_target_node11P2")):
        a_env.reset(); b_env.reset()
        a_env.scripted_reset(); b_env.scripted_reset()
        for t in range(150):
            rows = a_env.random_actions_seat(seat).clone()
            oa, _, da, ia = a_env.step_vs(pol, rows, seat=seat)
            acts = b_env.scripted_actions(pol, 1 - seat)
            acts[:, seat] = rows
            ob, _, db, ib = b_env.step(acts)
            assert torch.equal(oa, ob[:, seat]) and torch.equal(da, db) and torch.equal(ia["scores"], ib["scores"]), (seat, t)
        sa, sb = a_env.get_state(), b_env.get_state()
        for k in ("groups", "nodes", "health", "env"):
            assert np.array_equal(sa[k], sb[k]), k
        assert bool(a_env.done.any())                            # games did end (time expired at the latest): frozen envs were stepped
    # the native driver of the same loop (evg_rollout_vs_policy: per turn evg_random_actions_seat into a tensor, then evg_step_vs_policy)
    a_env.reset(); b_env.reset()
    a_env.scripted_reset(); b_env.scripted_reset()
    out = a_env.rollout_vs(60, "swarm", seat=1, time_kernel=True)
    assert out[-1] > 0
    for t in range(60):
        ob, _, db, ib = b_env.step_vs("swarm", b_env.random_actions_seat(1), seat=1)
    assert torch.equal(out[0], ob) and torch.equal(a_env.scores, b_env.scores) and torch.equal(a_env._actions_seat, b_env._actions_seat)
    sa, sb = a_env.get_state(), b_env.get_state()
    assert all(np.array_equal(sa[k], sb[k]) for k in sa)
    unfinished = evg.EvergladesVecEnv(100, seed=2)                 # rows of envs that have not finished an episode count as such
    unfinished.reset()
    cnt = torch.zeros(4, dtype=torch.int64, device=unfinished.device)
    unfinished.packed_episode_results(counts=cnt)
    assert _np(cnt).tolist() == [0, 0, 0, 100]
    unfinished.close()
    with pytest.raises(ValueError):
        a_env.step_vs("swarm", torch.zeros((N, 7), dtype=torch.int32, device=a_env.device))
    with pytest.raises(evg.EvgError):
        a_env.step_vs(99, a_env.random_actions_seat(0))
    with pytest.raises(evg.EvgError):
        a_env.step_vs("swarm", a_env.random_actions_seat(0), seat=2)
    mt = evg.EvergladesVecEnv(8, seed=1, rng_mode="mt19937")
    mt.reset()
    with pytest.raises(evg.EvgError, match="keyed-Philox"):
        mt.step_vs("swarm", mt.random_actions_seat(0))
    mt.close()
    a_env.close(); b_env.close()


def test_evaluate_harness_learner_seat_uses_step_vs(evg):
    """everglades_amd.evaluate with a callable on one seat and an on-device bot on the other (the reference's evaluate.py with a learned
    agent against a scripted one) plays the games of the all-native pairing when the callable gives the native bot's orders, on either seat;
    the callable receives ITS seat's observation [N, 105] like players[p].get_action(obs[p])."""
    import torch
    N, seed = 96, 21
    seen = []

    def same_commands(obs):
        seen.append(tuple(obs.shape))
        return torch.tensor([[i + 1, i + 1] for i in range(7)], dtype=torch.int32, device=obs.device).expand(N, 7, 2)

    a = evg.evaluate(same_commands, "swarm", N, num_envs=N, seed=seed)
    b = evg.evaluate("same_commands", "swarm", N, num_envs=N, seed=seed)
    assert np.array_equal(a["winners"], b["winners"]) and a["mean_length"] == b["mean_length"]
    c = evg.evaluate("cycle_rush_turn25", same_commands, N, num_envs=N, seed=seed)
    d = evg.evaluate("cycle_rush_turn25", "same_commands", N, num_envs=N, seed=seed)
    assert np.array_equal(c["winners"], d["winners"]) and c["mean_length"] == d["mean_length"]
    assert set(seen) == {(N, 105)}


def test_pipelined_halves_equal_one_handle_and_the_oracle(evg, oracle_mod):
    """PipelinedVecEnv (the double-buffered consumer, everglades_env.py:32-73 called from evaluate.py:143-152): two half-batch handles on
    two streams with global env ids preserved, driven in the overlapped pattern -- wait for half i, run the 'policy' (the on-device
    random_actions generator) on the caller's stream, enqueue half i's step on its own stream -- for 300 turns with auto-reset.  The
    union of the halves equals ONE full handle and the oracle on every env: state incl. float64 health, observations, last orders,
    episode results."""
    import torch
    N, seed, turns = 65536, 99, 300
    pipe = evg.PipelinedVecEnv(N, pipeline=2, seed=seed, auto_reset=True)
    assert [r for r in pipe.ranges] == [(0, N // 2), (N // 2, N // 2)]
    ora = oracle_mod.Oracle(N, seed=seed, auto_reset=True)
    assert np.array_equal(_np(pipe.reset()).astype(np.float64), ora.reset())
    for t in range(turns):
        for i in range(pipe.pipeline):
            obs_i = pipe.wait_part(i)[0]
            assert obs_i.data_ptr() == pipe.obs[pipe.ranges[i][0]:].data_ptr()       # a view of the full-batch tensor, no copy
            pipe.step_part(i, pipe.random_actions_part(i))
    pipe.wait_all()
    torch.cuda.synchronize()
    for t in range(turns - 1):
        ora.step_noobs(ora.random_actions())
    a = ora.random_actions()
    o_obs, _, _, _ = ora.step(a)
    assert np.array_equal(_np(pipe._actions), a), "orders of the last turn"
    _compare_whole_batch(pipe, ora, o_obs, "pipelined halves")
    one = evg.EvergladesVecEnv(N, seed=seed, auto_reset=True)
    one.reset()
    one.rollout_random(turns, turns_per_launch=150)
    assert torch.equal(one.obs, pipe.obs) and torch.equal(one.reward, pipe.reward) and torch.equal(one.scores, pipe.scores) and torch.equal(one.done, pipe.done)
    s1, s2 = one.get_state(), pipe.get_state()
    assert all(np.array_equal(s1[k], s2[k]) for k in s1)
    one.close()
    # the free-running benchmark form continues the same games; the joined step() and a masked reset over part boundaries
    ms = pipe.rollout_random_free(40, time_kernel=True)
    assert len(ms) == 2 and all(m > 0 for m in ms)
    for t in range(40):
        ora.step_noobs(ora.random_actions())
    a = _np(torch.cat([pipe.random_actions_part(i) for i in range(2)])).copy()
    obs, rew, done, info = pipe.step(a)
    o_obs, o_rew, o_done, o_info = ora.step(a)
    assert np.array_equal(_np(obs).astype(np.float64), o_obs) and np.array_equal(_np(info["scores"]), o_info["scores"]) and np.array_equal(_np(done), o_done)
    mask = (np.arange(N) % 5 == 0).astype(np.uint8)
    assert np.array_equal(_np(pipe.reset(mask=mask)).astype(np.float64)[mask != 0], ora.reset(mask=mask)[mask != 0])
    check_state(pipe, ora.get_state(), "pipelined: free-running + joined step + masked reset")
    pipe.close()


def test_pipelined_learner_seat_parts_vs_oracle(evg, oracle_mod):
    """The overlapped pattern with the learner-seat turn (step_vs_part -> evg_step_vs_policy) on three parts of unequal size, int16
    observations: the caller's one-seat observation tensor, scores and final state against the oracle."""
    import torch
    N, seed, turns, seat, pol = 8192 + 24, 17, 170, 1, "cycle_rush_turn25"
    pid = evg.EvergladesVecEnv.POLICIES[pol]
    pipe = evg.PipelinedVecEnv(N, pipeline=3, seed=seed, auto_reset=True, obs_dtype="int16")
    assert pipe.ranges == [(0, 2752), (2752, 2752), (5504, 2712)]
    ora = oracle_mod.Oracle(N, seed=seed, auto_reset=True)
    pipe.reset()
    o_obs = ora.reset()
    assert np.array_equal(_np(pipe.observe_seat(seat)).astype(np.float64), o_obs[:, seat])
    oa = np.zeros((N, 2, 7, 2), np.int32)
    shared = torch.zeros((N, 34), device=pipe.device)                      # full-batch feature tensors: every part's launch fills its own rows
    swarm = torch.zeros((N, 12, 13), device=pipe.device)
    for t in range(turns):
        for i in range(pipe.pipeline):
            pipe.wait_part(i, seat=seat)
            pipe.step_vs_part(i, pol, pipe.random_actions_part(i, seat=seat), seat=seat, features=(shared, swarm) if t >= turns - 3 else None)
        ora.scripted_actions(pid, 1 - seat, o_obs, oa)
        oa[:, seat] = ora.random_actions()[:, seat]
        o_obs, _, _, o_info = ora.step(oa)
    pipe.wait_all()
    torch.cuda.synchronize()
    assert np.array_equal(_np(pipe.obs_seat).astype(np.float64), o_obs[:, seat]) and np.array_equal(_np(pipe.scores), o_info["scores"])
    assert np.array_equal(_np(evg.EvergladesVecEnv.expand_smart_state(shared, swarm)), oracle_mod.smart_state(o_obs[:, seat]).astype(np.float32)), "fused features of the parts"
    check_state(pipe, ora.get_state(), "pipelined learner seat")
    ms = pipe.rollout_vs_free(30, pol, seat=seat, time_kernel=True)        # the free-running benchmark form continues the same games
    assert len(ms) == 3 and all(m > 0 for m in ms)
    for t in range(30):
        ora.scripted_actions(pid, 1 - seat, o_obs, oa)
        oa[:, seat] = ora.random_actions()[:, seat]
        o_obs, _, _, o_info = ora.step(oa)
    assert np.array_equal(_np(pipe.obs_seat).astype(np.float64), o_obs[:, seat]) and np.array_equal(_np(pipe._actions_seat), oa[:, seat])
    check_state(pipe, ora.get_state(), "pipelined learner seat, free-running")
    st, ost = pipe.episode_stats(), ora.episode_stats()
    assert np.array_equal(st["totals"], ost["totals"]) and np.array_equal(st["winner"], ost["winner"])
    pipe.close()


def test_graph_replay_build_plays_the_same_games(evg, oracle_mod):
    """libevg_graphs.so (-DEVG_REPLAY_GRAPHS: every rollout launch plan captured once into a library-owned hipGraph and replayed -- the build the
    driver-shape A/B measured slower than plain launches, tools/driver_shape_ab.sh): the chunked plan (memset + kernel + queue check as one graph)
    and the plain two-lane plan, prepared ahead and replayed over several launches, against the oracle on every env."""
    import os
    import torch
    if not os.path.exists(evg._lib.GRAPHS_LIB_PATH):
        pytest.skip("libevg_graphs.so not built (make -C everglades-ai-wargame_amd/csrc graphs)")
    cap2 = 32 * 8 * torch.cuda.get_device_properties(0).multi_processor_count
    for N in (cap2 + 2048, 5000):
        env = evg.EvergladesVecEnv(N, seed=77, auto_reset=True, library=evg._lib.GRAPHS_LIB_PATH)
        ora = oracle_mod.Oracle(N, seed=77, auto_reset=True)
        env.reset(); ora.reset()
        env.rollout_random(60, turns_per_launch=60, prepare=True)          # captures, plays nothing
        check_state(env, ora.get_state(), "prepare plays nothing")
        for _ in range(3):
            env.rollout_random(60, turns_per_launch=60)                    # one capture, three replays
        env.rollout_random(25, turns_per_launch=60)                        # another shape: its own graph
        for t in range(204):
            ora.step_noobs(ora.random_actions())
        a = ora.random_actions()
        o_obs, _, _, _ = ora.step(a)
        assert np.array_equal(_np(env._actions), a)
        _compare_whole_batch(env, ora, o_obs, ("graph replay build", N))
        assert env.check_fault() == 0
        env.close()


def test_smart_state_loop_example_runs_and_its_compact_network_equals_the_expanded_one(evg, oracle_mod):
    """examples/smart_state_loop.py: the Smart_State learner's acting loop with everything but the network on the device.  The example evaluates its stand-in
    network on the COMPACT features; here the same weights on the expanded [N, 12, 59] matrix give the same Q values (to float32 rounding of the re-associated
    sums), and the loop plays whole episodes."""
    import importlib.util
    import torch
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("evg_example_smart", os.path.join(ROOT, "examples", "smart_state_loop.py"))
    ex = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ex)
    st = ex.main(num_envs=2048 + 5, turns=170, epsilon=0.25, opponent="cycle_rush_turn25", seat=1, seed=4)
    assert int(st["totals"][0]) >= 2048 and int(st["totals"][1:].sum()) == int(st["totals"][0])
    env = evg.EvergladesVecEnv(777, seed=2, auto_reset=True)
    env.reset()
    env.rollout_random(50, turns_per_launch=50)
    sh, sw = env.smart_state_compact(0)
    full = evg.EvergladesVecEnv.expand_smart_state(sh, sw)                      # [N, 12, 59]
    net = ex.make_network(env.device, seed=0)
    g = torch.Generator(device="cpu").manual_seed(0)
    w1 = (torch.randn((60, 59), generator=g) * 0.2).to(env.device)
    w2 = (torch.randn((60, 60), generator=g) * 0.2).to(env.device)
    w3 = (torch.randn((5, 60), generator=g) * 0.2).to(env.device)
    want = torch.relu(torch.relu(full @ w1.T) @ w2.T) @ w3.T
    assert torch.allclose(net(sh, sw), want, rtol=1e-4, atol=1e-4)
    env.close()
