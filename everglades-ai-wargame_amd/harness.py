"""Batched mirror of the reference's evaluation harness loops (evaluate.py:127-181, evaluate_all.py:140-200): play
`episodes` games between two agents and keep the win bookkeeping of the harness -- player 0 wins when
`reward[0] > reward[1]`, a tie when equal, else a loss (evaluate.py:155-160) -- with the normal-approximation confidence
interval the scripts print (statsmodels `proportion_confint(wins, n, alpha, 'normal')`).

The games run as `num_envs` concurrent envs on the GPU, `episodes / num_envs` consecutive episodes each.  Like the
reference's agent objects, which are built once outside the episode loop (evaluate.py:85-93), the on-device scripted
agents keep their state across the consecutive episodes of an env.  Agents are the on-device bots of
agents/State_Machine/ (by file name, see EvergladesVecEnv.scripted_actions) or a callable `policy(obs) -> actions` that
maps ITS SEAT's observations [N, 105] -- the batched `players[p].get_action(obs[p])` of evaluate.py:143-147 -- to this seat's
orders, an int32 tensor [N, 7, 2] on the same device.  The case every script of the reference runs, one callable (the
learner) against one on-device bot, takes ONE launch per turn (EvergladesVecEnv.step_vs -> evg_step_vs_policy: the bot is
evaluated inside the step kernel and only the learner's observation is written).
"""
import math

from ._lib import POLICY_ALIASES, POLICY_NAMES
from .vec_env import EvergladesVecEnv


def proportion_confint_normal(count, nobs, alpha=0.05):
    """statsmodels.stats.proportion.proportion_confint(count, nobs, alpha, method='normal'), clipped to [0, 1]."""
    if nobs <= 0:
        return (0.0, 1.0)
    q = count / float(nobs)
    # inverse normal CDF at 1 - alpha/2 (Acklam's rational approximation, |rel. error| < 1.2e-9; no scipy dependency)
    p = 1.0 - alpha / 2.0
    a = [-3.969683028665376e+01, 2.209460984245205e+02, -2.759285104469687e+02, 1.383577518672690e+02, -3.066479806614716e+01, 2.506628277459239e+00]
    b = [-5.447609879822406e+01, 1.615858368580409e+02, -1.556989798598866e+02, 6.680131188771972e+01, -1.328068155288572e+01]
    c = [-7.784894002430293e-03, -3.223964580411365e-01, -2.400758277161838e+00, -2.549732539343734e+00, 4.374664141464968e+00, 2.938163982698783e+00]
    d = [7.784695709041462e-03, 3.224671290700398e-01, 2.445134137142996e+00, 3.754408661907416e+00]
    if p > 1 - 0.02425:
        t = math.sqrt(-2.0 * math.log(1.0 - p))
        z = -(((((c[0] * t + c[1]) * t + c[2]) * t + c[3]) * t + c[4]) * t + c[5]) / ((((d[0] * t + d[1]) * t + d[2]) * t + d[3]) * t + 1.0)
    else:
        t = p - 0.5
        r = t * t
        z = (((((a[0] * r + a[1]) * r + a[2]) * r + a[3]) * r + a[4]) * r + a[5]) * t / (((((b[0] * r + b[1]) * r + b[2]) * r + b[3]) * r + b[4]) * r + 1.0)
    half = z * math.sqrt(q * (1.0 - q) / nobs)
    return (max(0.0, q - half), min(1.0, q + half))


def _is_device_policy(p):
    return isinstance(p, str) and (p in POLICY_NAMES or p in POLICY_ALIASES)


def evaluate(player0, player1, episodes, num_envs=4096, seed=0, device=None, alpha=0.05, env=None, turns_per_launch=150):
    """Play `episodes` games (rounded up to a multiple of num_envs) of player0 (seat 0) against player1 (seat 1).
    Returns dict(games, wins, ties, losses, win_rate, confint, mean_length, winners) from seat 0's point of view;
    `winners` is the int8 array [rounds, num_envs] of EVG_WINNER_* per game."""
    import numpy as np
    import torch
    num_envs = int(min(num_envs, max(1, episodes)))
    rounds = (int(episodes) + num_envs - 1) // num_envs
    own = env is None
    if own:
        env = EvergladesVecEnv(num_envs, device=device, seed=seed, auto_reset=False)
    assert env.num_envs == num_envs and not env.auto_reset, "evaluate() needs a handle without auto-reset"
    both_native = _is_device_policy(player0) and _is_device_policy(player1)
    max_turns = int(env.tables.max_turns)          # the handle's own turn limit (150 for the reference's tables, server.py:321)
    winners, lengths = [], []
    for _ in range(rounds):
        obs = env.reset()
        if both_native:
            # finished envs stay frozen; the loop reads only the episode results, so no observations are written and no orders recorded
            env.rollout_policies(max_turns, player0, player1, fused=True, turns_per_launch=turns_per_launch, observe=False, record_actions=False)
        elif _is_device_policy(player0) != _is_device_policy(player1):
            # a learner on one seat, an on-device bot on the other: one launch per turn, the learner's observation only
            seat = 1 if _is_device_policy(player0) else 0
            learner, bot = (player0, player1) if seat == 0 else (player1, player0)
            seat_obs = env.observe_seat(seat)
            for _t in range(max_turns):
                a = torch.as_tensor(learner(seat_obs), device=env.device).to(torch.int32).reshape(num_envs, 7, 2).contiguous()
                seat_obs, _, done, _ = env.step_vs(bot, a, seat=seat)
                if bool(done.all()):
                    break
        else:
            for _t in range(max_turns):
                acts = env._actions
                for seat, pl in ((0, player0), (1, player1)):
                    acts[:, seat] = torch.as_tensor(pl(obs[:, seat]), device=env.device).to(torch.int32).reshape(num_envs, 7, 2)
                obs, _, done, _ = env.step(acts)
                if bool(done.all()):
                    break
        st = env.episode_stats()
        winners.append(st["winner"].copy())
        lengths.append(st["length"].copy())
    if own:
        env.close()
    w = np.stack(winners)
    games = int(w.size)
    wins, losses, ties = int((w == 0).sum()), int((w == 1).sum()), int((w == 2).sum())
    assert wins + losses + ties == games, "an env did not finish its episode"
    return dict(games=games, wins=wins, ties=ties, losses=losses, win_rate=wins / games, confint=proportion_confint_normal(wins, games, alpha),
                mean_length=float(np.stack(lengths).mean()), winners=w)


def evaluate_all(player0, opponents, episodes, **kw):
    """evaluate_all.py: `player0` against every opponent of a pool; {opponent: evaluate(...) result}."""
    return {name: evaluate(player0, name, episodes, **kw) for name in opponents}
