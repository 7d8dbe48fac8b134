#!/usr/bin/env python3
"""The Smart_State learner's acting loop (agents/Smart_State/training_scripts/dqn_smart_state_training.py:114-122 over DQNAgent.get_action, DQNAgent.py:130-300)
with everything but the network on the device, one launch each:

    features --(the consumer's QNetwork 59-60-60-5)--> Q [N, 12, 5]
             --evg_smart_get_action (epsilon coin, get_random_actions / get_best_actions)--> orders [N, 7, 2] (+ directions for the replay memory)
             --evg_step_vs_policy_smart (the scripted opponent inside the step kernel)--> observation, reward, done AND the next features

The network here is a stand-in with random weights, evaluated on the COMPACT features: features[e, s] = cat(shared[e], swarm[e, s], onehot(s)), so the first
layer is W[:, :34] @ shared + W[:, 34:47] @ swarm + W[:, 47 + s] -- a quarter of the bytes of the expanded [N, 12, 59] matrix.

    python examples/smart_state_loop.py [envs] [turns] [epsilon]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import everglades_amd as evg


def make_network(device, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    w1, b1 = torch.randn((60, 59), generator=g) * 0.2, torch.zeros(60)
    w2, b2 = torch.randn((60, 60), generator=g) * 0.2, torch.zeros(60)
    w3, b3 = torch.randn((5, 60), generator=g) * 0.2, torch.zeros(5)
    w1, b1, w2, b2, w3, b3 = (t.to(device) for t in (w1, b1, w2, b2, w3, b3))

    def q_values(shared, swarm):
        # first layer on the compact pair: [N, 1, 60] + [N, 12, 60] + the column of the swarm's one-hot id
        h = (shared @ w1[:, :34].T).unsqueeze(1) + swarm @ w1[:, 34:47].T + w1[:, 47:59].T.unsqueeze(0) + b1
        h = torch.relu(h)
        h = torch.relu(h @ w2.T + b2)
        return (h @ w3.T + b3).contiguous()                          # [N, 12, 5]
    return q_values


def main(num_envs=8192, turns=200, epsilon=0.1, opponent="swarm_agent", seat=0, seed=1):
    env = evg.EvergladesVecEnv(num_envs, seed=seed, auto_reset=True)
    net = make_network(env.device)
    env.reset()
    obs = env.observe_seat(seat)                                      # [N, 105]
    shared, swarm = env.smart_state_compact(-1, obs)                  # the first features of the loop; afterwards the step launch refills them
    directions = torch.zeros((num_envs, 7, 2), dtype=torch.int32, device=env.device)
    explored = torch.zeros(num_envs, dtype=torch.uint8, device=env.device)
    torch.cuda.synchronize()
    t0, ret = time.perf_counter(), torch.zeros((num_envs, 2), device=env.device)
    for _ in range(turns):
        q = net(shared, swarm)
        actions = env.smart_get_action(q, epsilon, seat=seat, obs=obs, directions=directions, explored=explored)
        obs, reward, done, info = env.step_vs(opponent, actions, seat=seat, features=(shared, swarm))
        ret += reward                                                  # (a learner would push (features, directions, reward, done) into its replay memory here)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = env.episode_stats()
    print("%d envs x %d turns in %.3f s = %.1f M env-steps/s (network included); episodes finished %d, wins seat0 / seat1 / ties %s; explored %.1f %% of the last turn"
          % (num_envs, turns, dt, num_envs * turns / dt / 1e6, int(st["totals"][0]), st["totals"][1:].tolist(), 100.0 * explored.float().mean().item()))
    env.close()
    return st


if __name__ == "__main__":
    a = sys.argv[1:]
    main(int(a[0]) if a else 8192, int(a[1]) if len(a) > 1 else 200, float(a[2]) if len(a) > 2 else 0.1)
