#!/usr/bin/env python3
"""One-off soak (not a test): HIP against the oracle on MANY random configurations (oracle/custom_configs.random_config), every launch form.
usage (on the GPU box): python tools/random_tables_soak.py [first] [count] [envs]     (envs > 4096: fewer stepwise turns; persistent launches run the four-lane kernel for 2 / 3
wavefronts per SIMD up to 32 768 / 49 152 envs and the two-lane kernel above)"""
import json, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import everglades_amd as evg
import oracle as om
import custom_configs as cc
from gen_policies import policy_actions

first, count = (int(sys.argv[1]) if len(sys.argv) > 1 else 100), (int(sys.argv[2]) if len(sys.argv) > 2 else 60)
ENVS = int(sys.argv[3]) if len(sys.argv) > 3 else 192
STEPWISE, BOTWISE = (160, 60) if ENVS <= 4096 else (30, 12)
tmp = tempfile.mkdtemp()
om.lib().evo_set_num_threads(16)
fast = 0
for c in range(first, first + count):
    rng = np.random.default_rng([20261010, c])
    mobj, uobj = cc.random_config(rng)
    mp, up = os.path.join(tmp, "m.json"), os.path.join(tmp, "u.json")
    open(mp, "w").write(json.dumps(mobj)); open(up, "w").write(json.dumps(uobj))
    tables = evg.tables_from_json(mp, up)
    ot = om.tables_from_json_text(json.dumps(mobj), json.dumps(uobj))
    N, seed = ENVS, 900 + c
    env = evg.EvergladesVecEnv(N, seed=seed, auto_reset=True, tables=tables)
    ora = om.Oracle(N, seed=seed, auto_reset=True, tables=ot)
    obs = env.reset().cpu().numpy().astype(np.float64)
    assert np.array_equal(obs, ora.reset()), ("reset", c)
    for t in range(STEPWISE):
        a = policy_actions("wild", obs, t, rng) if (t // 20) % 3 == 1 else env.random_actions().cpu().numpy().copy()
        o, rew, done, info = env.step(a)
        o_obs, o_rew, o_done, o_info = ora.step(a)
        obs = o.cpu().numpy().astype(np.float64)
        assert np.array_equal(obs, o_obs), ("obs", c, t)
        assert np.array_equal(info["scores"].cpu().numpy(), o_info["scores"]) and np.array_equal(done.cpu().numpy(), o_done), (c, t)
    s, os_ = env.get_state(), ora.get_state()
    assert all(np.array_equal(s[k], os_[k]) for k in ("groups", "nodes", "health", "env")), ("state", c)
    assert np.array_equal(env.fog_of_war().cpu().numpy(), ora.fog_of_war()) and np.array_equal(env.knowledge().cpu().numpy(), ora.knowledge()), ("fog", c)
    assert np.array_equal(env.sightings().cpu().numpy(), ora.sightings()), ("sightings", c)
    env.rollout_random(100, turns_per_launch=100)
    for t in range(100):
        ora.step_noobs(ora.random_actions())
    s, os_ = env.get_state(), ora.get_state()
    assert all(np.array_equal(s[k], os_[k]) for k in ("groups", "nodes", "health", "env")), ("persistent state", c)
    # ... and the scripted bots on the same tables: a random pairing through evg_scripted_actions (from the observation tensor) and then fused into the
    # persistent kernel (from the on-chip state), against the oracle's bots (held to the reference's agent classes on random maps by
    # tests/test_oracle_vs_live_reference.py)
    P = evg.EvergladesVecEnv.POLICIES
    names = [n for n in evg._lib.POLICY_NAMES if n != "no_action"]
    pa, pb = (names[int(i)] for i in rng.integers(0, len(names), 2))
    env.scripted_reset(); ora.scripted_reset()
    o_obs = ora.observe()
    env.observe()
    oa = np.zeros((N, 2, 7, 2), np.int32)
    for t in range(BOTWISE):
        env.scripted_actions(pa, 0)
        a = env.scripted_actions(pb, 1)
        ora.scripted_actions(P[pa], 0, o_obs, oa); ora.scripted_actions(P[pb], 1, o_obs, oa)
        assert np.array_equal(a.cpu().numpy(), oa), ("bot orders", c, pa, pb, t)
        o, rew, done, info = env.step(a)
        o_obs, _, _, _ = ora.step(oa)
        assert np.array_equal(o.cpu().numpy().astype(np.float64), o_obs), ("bot obs", c, pa, pb, t)
    env.rollout_policies(70, pa, pb, fused=True, turns_per_launch=70)
    for t in range(70):
        ora.scripted_actions(P[pa], 0, o_obs, oa); ora.scripted_actions(P[pb], 1, o_obs, oa)
        o_obs, _, _, _ = ora.step(oa)
    s, os_ = env.get_state(), ora.get_state()
    assert all(np.array_equal(s[k], os_[k]) for k in ("groups", "nodes", "health", "env")), ("fused bots state", c, pa, pb)
    assert np.array_equal(env._actions.cpu().numpy(), oa), ("fused bots orders", c, pa, pb)
    env.close()
    if c % 10 == 0 or ENVS > 4096:
        print("config", c, "ok", flush=True)
print("soak ok: %d random configurations, %d games each" % (count, ENVS))
