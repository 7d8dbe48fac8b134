#!/usr/bin/env python3
"""Experiment (VERDICT r02 item 4 i): does splitting the batch into two half-batch handles on two streams shorten a turn of the
one-launch-per-turn form, because one half's write-out overlaps the other half's compute?

  A   one handle, 65 536 envs, N_TURNS single-turn launches back to back (evg_rollout_random, orders drawn in the kernel)
  B1  two handles of 32 768 envs on two streams, JOINED every turn (each stream waits for the other's turn-t event before turn t+1):
      what a synchronous Gym consumer would get -- it needs all observations of turn t before it can produce the orders of t+1
  B2  the same two handles FREE-RUNNING (each stream plays its N_TURNS turns on its own, enqueued from two host threads): what a
      double-buffered consumer gets (policy on half A while half B steps)
Times are wall clock per turn of the WHOLE 65 536-env batch, device drained before and after."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import everglades_amd as evg

N, N_TURNS = 65536, 300


def desync(env, first):
    ids = torch.arange(first, first + env.num_envs, device=env.device)
    for j in range(150):
        env.rollout_random(1)
        env.reset(mask=((((ids * 2654435761) & 0xFFFFFFFF) >> 8) % 150 == j).to(torch.uint8))
    env.rollout_random(150)


def timed(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / N_TURNS * 1e6)
    return best


one = evg.EvergladesVecEnv(N, seed=1, auto_reset=True)
one.reset(); desync(one, 0)
a = timed(lambda: one.rollout_random(N_TURNS))
one.close()

halves = [evg.EvergladesVecEnv(N // 2, seed=1, env_id_base=i * (N // 2), auto_reset=True) for i in range(2)]
streams = [torch.cuda.Stream() for _ in range(2)]
for i, h in enumerate(halves):
    h.reset(); desync(h, i * (N // 2))
torch.cuda.synchronize()


def joined():
    evs = [[torch.cuda.Event() for _ in range(N_TURNS)] for _ in range(2)]
    for t in range(N_TURNS):
        for i in range(2):
            with torch.cuda.stream(streams[i]):
                if t:
                    streams[i].wait_event(evs[1 - i][t - 1])
                halves[i].rollout_random(1)
                evs[i][t].record(streams[i])


def free():
    def run(i):
        with torch.cuda.stream(streams[i]):
            halves[i].rollout_random(N_TURNS)
    th = [threading.Thread(target=run, args=(i,)) for i in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]


def one_half_alone():
    with torch.cuda.stream(streams[0]):
        halves[0].rollout_random(N_TURNS)


b1, b2, h1 = timed(joined), timed(free), timed(one_half_alone)
print("A   one handle of %d envs, one launch per turn:                         %6.2f us per turn" % (N, a))
print("B1  two handles of %d envs on two streams, joined every turn (python):  %6.2f us per turn" % (N // 2, b1))
print("B2  two handles of %d envs on two streams, free-running:                %6.2f us per turn of the whole batch" % (N // 2, b2))
print("    one half alone (32 768 envs, one launch per turn):                     %6.2f us per turn" % h1)
