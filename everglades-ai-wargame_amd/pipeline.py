"""PipelinedVecEnv -- the double-buffered consumer of the Gym-shaped path.

A synchronous `step(actions) -> obs` over the whole batch cannot overlap anything: the orders of turn t + 1 depend on all
observations of turn t, so every launch starts when the previous one has retired, and half of a single-turn launch is fixed cost --
dispatch, one memory round trip for the state, the serial chain of one wavefront's turn, end of kernel (DESIGN.md section 6:
14.5 of 27 us at 65 536 envs).  A consumer that splits its batch in halves CAN overlap: while its policy network runs on the
observations of half A, half B steps, and one half's fixed costs run under the other half's compute (two halves free-running:
25.5 us of stream time per turn of the whole batch instead of 28.2).  This class is that pattern as an API instead of a recipe: `pipeline` handles of num_envs / pipeline envs each (global
env ids preserved: part i owns ids env_id_base + i * n ...), each with its own HIP stream, all writing into contiguous slices of ONE
set of full-batch tensors, plus the two event waits per part that make the hand-over between the caller's stream (where the policy
runs) and a part's stream correct.

    env = PipelinedVecEnv(65536, pipeline=2)
    env.reset()
    for t in range(turns):
        for i in range(env.pipeline):
            obs_i = env.wait_part(i)[0]            # the caller's stream waits for part i's last step; views of the full tensors
            a_i = policy(obs_i)                     # runs on the caller's stream while the OTHER part steps on its own
            env.step_part(i, a_i)                   # part i's stream waits for a_i, then steps (returns at once)

Reference loop being served: gym_everglades/envs/everglades_env.py:32-73 called from evaluate.py:143-152; with
`step_vs_part` the learner-vs-bot turn of evg_step_vs_policy.  Results are those of one handle of the full size, env by env.
"""
import threading

import numpy as np

from . import _lib
from .vec_env import EvergladesVecEnv


class PipelinedVecEnv(object):
    def __init__(self, num_envs, pipeline=2, device=None, seed=0, env_id_base=0, obs_dtype="float32", auto_reset=True, **kw):
        import torch
        self.pipeline = int(pipeline)
        if self.pipeline < 1 or int(num_envs) < self.pipeline:
            raise ValueError("pipeline must be >= 1 and <= num_envs")
        self.num_envs = int(num_envs)
        # (first env, count) of every part: boundaries at multiples of 32 envs (whole wavefronts; every slice of every tensor 16-byte aligned)
        per = ((self.num_envs + self.pipeline - 1) // self.pipeline + 31) // 32 * 32
        self.ranges = [(i * per, min(per, self.num_envs - i * per)) for i in range(self.pipeline)]
        if self.ranges[-1][1] < 1:
            raise ValueError("%d envs do not split into %d parts of whole wavefronts" % (self.num_envs, self.pipeline))
        self.parts = [EvergladesVecEnv(cnt, device=device, seed=seed, env_id_base=int(env_id_base) + lo, obs_dtype=obs_dtype, auto_reset=auto_reset, **kw)
                      for lo, cnt in self.ranges]
        p0 = self.parts[0]
        self.device, self.obs_dtype, self.auto_reset = p0.device, p0.obs_dtype, p0.auto_reset
        N = self.num_envs
        with torch.cuda.device(self.device):
            self.obs = torch.zeros((N, 2, _lib.OBS_LEN), dtype=self.obs_dtype, device=self.device)
            self.obs_seat = torch.zeros((N, _lib.OBS_LEN), dtype=self.obs_dtype, device=self.device)
            self.reward = torch.zeros((N, 2), dtype=torch.float32, device=self.device)
            self.done = torch.zeros((N,), dtype=torch.uint8, device=self.device)
            self.winner = torch.full((N,), -1, dtype=torch.int8, device=self.device)
            self.scores = torch.zeros((N, 2), dtype=torch.int32, device=self.device)
            self.status = torch.zeros((N,), dtype=torch.uint8, device=self.device)
            self._actions = torch.zeros((N, 2, _lib.NUM_ACTIONS, 2), dtype=torch.int32, device=self.device)
            self._actions_seat = torch.zeros((N, _lib.NUM_ACTIONS, 2), dtype=torch.int32, device=self.device)
            self.streams = [torch.cuda.Stream(device=self.device) for _ in self.parts]
        # the parts write straight into their slices of the full-batch tensors (contiguous: env-major layouts).  16-byte alignment of a
        # slice's first row holds for any part boundary that is a multiple of 8 envs (int16 one-seat rows are 210 B); other boundaries are refused here
        for (lo, cnt), part in zip(self.ranges, self.parts):
            if lo % 8:
                raise ValueError("part boundaries must be multiples of 8 envs (16-byte aligned slices of every tensor); got a part starting at env %d" % lo)
            part._adopt_buffers(obs=self.obs[lo:lo + cnt], reward=self.reward[lo:lo + cnt], done=self.done[lo:lo + cnt], winner=self.winner[lo:lo + cnt],
                                scores=self.scores[lo:lo + cnt], status=self.status[lo:lo + cnt], actions=self._actions[lo:lo + cnt],
                                obs_seat=self.obs_seat[lo:lo + cnt], actions_seat=self._actions_seat[lo:lo + cnt])
        self._stepped = [torch.cuda.Event() for _ in self.parts]       # recorded on part i's stream behind its last enqueued work
        self._ready = [torch.cuda.Event() for _ in self.parts]         # recorded on the caller's stream when part i's orders are complete
        self._info = dict(winner=self.winner, scores=self.scores, status=self.status)
        torch.cuda.synchronize(self.device)

    # ------------------------------------------------------------------ plumbing
    def _torch(self):
        import torch
        return torch

    def _on_part(self, i, fn, after_caller=True):
        """run fn() with part i's stream current; the part's stream first waits for what the caller's stream has enqueued so far (the orders)"""
        torch = self._torch()
        cur = torch.cuda.current_stream(self.device)
        if after_caller:
            self._ready[i].record(cur)
            self.streams[i].wait_event(self._ready[i])
        with torch.cuda.stream(self.streams[i]):
            out = fn()
            self._stepped[i].record(self.streams[i])
        return out

    def part_views(self, i, seat=None):
        lo, cnt = self.ranges[i]
        sl = slice(lo, lo + cnt)
        info = dict(winner=self.winner[sl], scores=self.scores[sl], status=self.status[sl])
        return (self.obs[sl] if seat is None else self.obs_seat[sl]), self.reward[sl], self.done[sl], info

    def wait_part(self, i, seat=None):
        """The caller's current stream waits for everything part i has enqueued (its last step); returns part i's views
        (obs [n_i, 2, 105] -- or the one-seat tensor [n_i, 105] with seat=0|1 --, reward, done, info) of the full-batch tensors."""
        self._torch().cuda.current_stream(self.device).wait_event(self._stepped[i])
        return self.part_views(i, seat)

    def wait_all(self):
        for i in range(self.pipeline):
            self.wait_part(i)

    def close(self):
        for p in getattr(self, "parts", []):
            p.close()
        self.parts = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ Gym-style API over the whole batch (joins the parts: no overlap)
    def reset(self, mask=None):
        torch = self._torch()
        m = None if mask is None else torch.as_tensor(mask, device=self.device).to(torch.uint8).contiguous()
        for i, ((lo, cnt), part) in enumerate(zip(self.ranges, self.parts)):
            self._on_part(i, lambda part=part, lo=lo, cnt=cnt: part.reset(None if m is None else m[lo:lo + cnt]))
        self.wait_all()
        return self.obs

    def step(self, actions):
        """One turn of the whole batch: every part steps on its stream, the caller's stream waits for all of them (the synchronous
        Gym turn; use wait_part / step_part for the overlapped pattern)."""
        torch = self._torch()
        a = torch.as_tensor(actions, device=self.device)
        if tuple(a.shape) != (self.num_envs, 2, _lib.NUM_ACTIONS, 2):
            raise ValueError("actions must have shape [N, 2, 7, 2], got %s" % (tuple(a.shape),))
        a = (a if a.dtype == torch.int32 else a.to(torch.int32)).contiguous()
        for i, (lo, cnt) in enumerate(self.ranges):
            self.step_part(i, a[lo:lo + cnt])
        self.wait_all()
        return self.obs, self.reward, self.done, self._info

    # ------------------------------------------------------------------ the overlapped pattern
    def step_part(self, i, actions):
        """Enqueue one turn of part i (actions [n_i, 2, 7, 2], e.g. a slice of a full-batch tensor) on ITS stream, behind everything the
        caller's current stream has enqueued so far (the policy that produced `actions`).  Returns at once with part i's views; call
        wait_part(i) before reading them on the caller's stream."""
        part = self.parts[i]
        self._on_part(i, lambda: part.step(actions))
        return self.part_views(i)

    def step_vs_part(self, i, policy, actions, seat=0, features=None):
        """The learner-seat turn (EvergladesVecEnv.step_vs) of part i: actions [n_i, 7, 2] or [n_i, 2, 7, 2]; the caller seat's
        observation lands in obs_seat[part i].  features=(shared [N, 34], swarm [N, 12, 13]): FULL-batch float32 tensors whose rows of part i the launch
        also fills with the Smart_State features of the new observation (evg_step_vs_policy_smart; parts are whole wavefronts, so the slices are aligned)."""
        part = self.parts[i]
        feats = None
        if features is not None:
            lo, n = self.ranges[i]
            feats = (features[0][lo:lo + n], features[1][lo:lo + n])
        self._on_part(i, lambda: part.step_vs(policy, actions, seat=seat, features=feats))
        return self.part_views(i, seat)

    def observe_seat(self, seat=0):
        for i, part in enumerate(self.parts):
            self._on_part(i, lambda part=part: part.observe_seat(seat))
        self.wait_all()
        return self.obs_seat

    def random_actions_part(self, i, seat=None):
        """Stand-in for a policy on the CALLER's stream (after wait_part(i)): random_actions of part i's envs, [n_i, 2, 7, 2] or, with
        seat=0|1, that seat's rows [n_i, 7, 2]."""
        return self.parts[i].random_actions() if seat is None else self.parts[i].random_actions_seat(seat)

    def rollout_vs_free(self, steps, policy, seat=0, time_kernel=False):
        """The same free-running benchmark for the learner-seat turn: every part plays `steps` turns of EvergladesVecEnv.rollout_vs (per turn the
        caller seat's stand-in orders into a tensor, then step_vs with the on-device bot `policy`) on its own stream."""
        return self._free(lambda part: part.rollout_vs(steps, policy, seat=seat, time_kernel=time_kernel), time_kernel)

    def rollout_random_free(self, steps, time_kernel=False):
        """Benchmark of the overlap: every part plays `steps` turns of random vs random, one launch per turn (orders drawn in the step
        kernel), FREE-RUNNING on its own stream -- what the double-buffered pattern converges to when the policy is cheap.  The native
        loops are started from one host thread per part (they only enqueue).  Returns the per-part stream time per turn in ms
        (time_kernel=True; synchronises) or None."""
        return self._free(lambda part: part.rollout_random(steps, time_kernel=time_kernel, fused=True, turns_per_launch=1), time_kernel)

    def _free(self, play, time_kernel):
        torch = self._torch()
        cur = torch.cuda.current_stream(self.device)
        for i in range(self.pipeline):
            self._ready[i].record(cur)
            self.streams[i].wait_event(self._ready[i])
        ms = [None] * self.pipeline
        err = [None] * self.pipeline

        def run(i):
            try:
                with torch.cuda.device(self.device), torch.cuda.stream(self.streams[i]):
                    out = play(self.parts[i])
                    ms[i] = out[-1] if time_kernel else None
                    self._stepped[i].record(self.streams[i])
            except Exception as ex:          # re-raised on the calling thread
                err[i] = ex

        th = [threading.Thread(target=run, args=(i,)) for i in range(self.pipeline)]
        [t.start() for t in th]
        [t.join() for t in th]
        for e in err:
            if e is not None:
                raise e
        self.wait_all()
        return ms if time_kernel else None

    # ------------------------------------------------------------------ state / results of the whole batch, in global env order
    def get_state(self):
        for s in self.streams:
            s.synchronize()
        st = [p.get_state() for p in self.parts]
        return {k: np.concatenate([s[k] for s in st], axis=0) for k in st[0]}

    def episode_stats(self):
        for s in self.streams:
            s.synchronize()
        st = [p.episode_stats() for p in self.parts]
        out = {k: np.concatenate([s[k] for s in st], axis=0) for k in ("returns", "length", "winner")}
        out["totals"] = np.sum([s["totals"] for s in st], axis=0)
        return out

    def launch_plan(self, turns_per_launch=1):
        return [p.launch_plan(turns_per_launch) for p in self.parts]
