"""EvergladesVecEnv -- N concurrent two-player Everglades games on one MI355X.

Host-side mirror of the reference's Gym interface (gym_everglades/envs/everglades_env.py) over the
C-ABI of include/evg.h: `reset()` and `step()` have the reference's meaning, vectorised over a leading
env axis, with all I/O in caller-visible torch tensors on the GPU.  PyTorch is used only for device
memory and streams; the game itself runs in the HIP kernels of csrc/evg_kernels.hip.
"""
import ctypes as C
import os

import numpy as np

from . import _lib
from .tables import default_tables, tables_from_json


def _torch():
    import torch
    return torch


class EvergladesVecEnv(object):
    # constants of the reference environment (everglades_env.py:17-22)
    num_turns = 150
    num_units = _lib.NUM_UNITS
    num_groups = _lib.NUM_GROUPS
    num_nodes = _lib.NUM_NODES
    num_actions_per_turn = _lib.NUM_ACTIONS
    unit_classes = ["controller", "striker", "tank"]
    obs_len = _lib.OBS_LEN

    def __init__(self, num_envs, device=None, seed=0, env_id_base=0, obs_dtype="float32", auto_reset=True, tables=None,
                 map_file=None, unit_file=None, config_dir=None, rng_mode="philox", library=None, diag=None, cache_mib=0):
        """rng_mode "philox" (default, the fast keyed draws of DESIGN.md section 4) or "mt19937": every env owns numpy's legacy
        generator seeded like np.random.seed((seed + env_id_base + e) & 0xFFFFFFFF) and consumes it in the reference's
        order, so a game replays the UNMODIFIED reference process bit for bit (validation mode, sequential draws).
        library / diag: diagnostics only -- path of a diagnostic build of the library (default: the product libevg.so) and, for
        libevg_diag.so, dict(ablate=bits, lanes=0|64|32|4, force_ieee_div=bool) passed to its evg_diag_configure.
        cache_mib: 0 = derive the memory-side cache budget of chunked rollout launches from the device (evg_config.cache_mib)."""
        torch = _torch()
        self.L = _lib.load(library)
        if not torch.cuda.is_available():
            raise _lib.EvgError("EvergladesVecEnv needs a HIP device (MI355X); there is no CPU fallback")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if self.device.type != "cuda":
            raise _lib.EvgError("device must be a cuda (HIP) device, got %s" % (self.device,))
        dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", dev_index)
        self.num_envs = int(num_envs)
        self.seed, self.env_id_base, self.auto_reset = int(seed), int(env_id_base), bool(auto_reset)
        names = {"float32": (_lib.OBS_F32, torch.float32), "float64": (_lib.OBS_F64, torch.float64), "int16": (_lib.OBS_I16, torch.int16)}
        key = str(obs_dtype).replace("torch.", "")
        if key not in names:
            raise ValueError("obs_dtype must be float32, float64 or int16")
        self._obs_code, self.obs_dtype = names[key]
        if tables is None:
            tables = default_tables() if (map_file is None and unit_file is None) else tables_from_json(map_file, unit_file, config_dir)
        self.tables = tables
        cfg = _lib.EvgConfig()
        cfg.struct_size, cfg.abi_version = C.sizeof(_lib.EvgConfig), _lib.ABI_VERSION
        cfg.num_envs, cfg.device_id = self.num_envs, dev_index
        cfg.seed, cfg.env_id_base = self.seed & 0xFFFFFFFFFFFFFFFF, self.env_id_base
        cfg.obs_dtype, cfg.auto_reset = self._obs_code, int(self.auto_reset)
        modes = {"philox": _lib.RNG_KEYED_PHILOX, "mt19937": _lib.RNG_STOCK_MT19937}
        if rng_mode not in modes:
            raise ValueError("rng_mode must be 'philox' or 'mt19937'")
        self.rng_mode = rng_mode
        cfg.rng_mode = modes[rng_mode]
        cfg.cache_mib = int(cache_mib)
        cfg.tables = tables
        h = C.c_void_p()
        self._check(self.L.evg_create(C.byref(cfg), C.byref(h)))
        self._h = h
        if diag:
            if not hasattr(self.L, "evg_diag_configure"):
                raise _lib.EvgError("diag options need a diagnostic library (library=_lib.DIAG_LIB_PATH); the product library has none")
            self._check(self.L.evg_diag_configure(h, int(diag.get("ablate", 0)), int(diag.get("lanes", 0)), int(bool(diag.get("force_ieee_div", False)))))
        N = self.num_envs
        with torch.cuda.device(self.device):
            self.obs = torch.zeros((N, 2, _lib.OBS_LEN), dtype=self.obs_dtype, device=self.device)
            self.reward = torch.zeros((N, 2), dtype=torch.float32, device=self.device)
            self.done = torch.zeros((N,), dtype=torch.uint8, device=self.device)
            self.winner = torch.full((N,), -1, dtype=torch.int8, device=self.device)
            self.scores = torch.zeros((N, 2), dtype=torch.int32, device=self.device)
            self.status = torch.zeros((N,), dtype=torch.uint8, device=self.device)
            self._actions = torch.zeros((N, 2, _lib.NUM_ACTIONS, 2), dtype=torch.int32, device=self.device)
        # per-call fast path: raw pointers of the env's own buffers, the info dict and the raw-stream getter are cached
        self._p = {k: C.c_void_p(getattr(self, k).data_ptr()) for k in ("obs", "reward", "done", "winner", "scores", "status", "_actions")}
        self._info = dict(winner=self.winner, scores=self.scores, status=self.status)
        self._act_shape = (N, 2, _lib.NUM_ACTIONS, 2)
        self._raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
        self._int32 = torch.int32

    # ------------------------------------------------------------------ plumbing
    def _check(self, rc):
        if rc:
            _lib.check(rc, self.L)

    def _stream(self):
        if self._raw_stream is not None:        # raw hipStream_t of torch's current stream without building a Stream object
            return C.c_void_p(self._raw_stream(self.device.index))
        return C.c_void_p(_torch().cuda.current_stream(self.device).cuda_stream)

    @staticmethod
    def _ptr(t):
        return None if t is None else C.c_void_p(t.data_ptr())

    def _user(self, t, shape, dtype, name):
        """A caller-supplied tensor reaches the kernels as a raw pointer: it must be exactly what the kernel assumes."""
        torch = _torch()
        if not isinstance(t, torch.Tensor):
            raise ValueError("%s must be a torch tensor on %s" % (name, self.device))
        if t.device != self.device or t.dtype != dtype or tuple(t.shape) != tuple(shape) or not t.is_contiguous():
            raise ValueError("%s must be a contiguous %s tensor of shape %s on %s, got %s %s on %s%s" % (
                name, dtype, tuple(shape), self.device, t.dtype, tuple(t.shape), t.device, "" if t.is_contiguous() else " (not contiguous)"))
        return t

    def _adopt_buffers(self, obs, reward, done, winner, scores, status, actions, obs_seat=None, actions_seat=None):
        """Use caller-owned tensors (e.g. contiguous slices of full-batch tensors: PipelinedVecEnv) as this env's output / order
        buffers instead of its own."""
        torch = _torch()
        N = self.num_envs
        self.obs = self._user(obs, (N, 2, _lib.OBS_LEN), self.obs_dtype, "obs")
        self.reward = self._user(reward, (N, 2), torch.float32, "reward")
        self.done = self._user(done, (N,), torch.uint8, "done")
        self.winner = self._user(winner, (N,), torch.int8, "winner")
        self.scores = self._user(scores, (N, 2), torch.int32, "scores")
        self.status = self._user(status, (N,), torch.uint8, "status")
        self._actions = self._user(actions, self._act_shape, torch.int32, "actions")
        for t in (self.obs, self.reward, self.scores, self._actions):
            if t.data_ptr() % 16:
                raise ValueError("adopted buffers must be 16-byte aligned")
        self._p = {k: C.c_void_p(getattr(self, k).data_ptr()) for k in ("obs", "reward", "done", "winner", "scores", "status", "_actions")}
        self._info = dict(winner=self.winner, scores=self.scores, status=self.status)
        if obs_seat is not None:
            self._obs_seat = self._user(obs_seat, (N, _lib.OBS_LEN), self.obs_dtype, "obs_seat")
            self._actions_seat = self._user(actions_seat, (N, _lib.NUM_ACTIONS, 2), torch.int32, "actions_seat")
            if self._obs_seat.data_ptr() % 16 or self._actions_seat.data_ptr() % 16:
                raise ValueError("adopted buffers must be 16-byte aligned")
            self._p["obs_seat"] = C.c_void_p(self._obs_seat.data_ptr())

    def close(self):
        if getattr(self, "_h", None):
            _torch().cuda.synchronize(self.device)
            self.L.evg_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ Gym-style API
    def reset(self, mask=None):
        """Start a new episode in every env (or in envs where mask != 0).  Returns obs [N, 2, 105]
        (everglades_env.py:75-116)."""
        torch = _torch()
        m = None
        if mask is not None:
            m = torch.as_tensor(mask, device=self.device).to(torch.uint8).contiguous()
            assert m.shape == (self.num_envs,)
        self._check(self.L.evg_reset(self._h, self._ptr(m), self._ptr(self.obs), self._stream()))
        return self.obs

    def _as_actions(self, actions):
        torch = _torch()
        if isinstance(actions, dict):       # reference-style {player: array(7, 2)} per env is handled by EvergladesEnv
            raise TypeError("pass a [N, 2, 7, 2] tensor; dict actions are the single-env API (EvergladesEnv)")
        a = torch.as_tensor(actions, device=self.device)
        if a.dtype != torch.int32:
            a = a.to(torch.int32)           # truncation, like action.astype(int) (server.py:232)
        a = a.contiguous()
        if a.shape != (self.num_envs, 2, _lib.NUM_ACTIONS, 2):
            raise ValueError("actions must have shape [N, 2, 7, 2], got %s" % (tuple(a.shape),))
        return a

    def step(self, actions):
        """One turn of every game.  Returns (obs [N,2,105], reward [N,2] f32, done [N] u8, info) where info has
        winner [N] i8, scores [N,2] i32, status [N] u8 (everglades_env.py:32-73).  The tensors are the env's own
        output buffers and are overwritten by the next call."""
        a = actions
        if not (type(a) is _torch().Tensor and a.dtype is self._int32 and a.device == self.device and a.shape == self._act_shape
                and a.is_contiguous()):
            a = self._as_actions(actions)
        p = self._p
        rc = self.L.evg_step(self._h, C.c_void_p(a.data_ptr()), p["obs"], p["reward"], p["done"], p["winner"], p["scores"], p["status"],
                             self._stream())
        if rc:
            self._check(rc)
        return self.obs, self.reward, self.done, self._info

    def observe(self):
        self._check(self.L.evg_observe(self._h, self._ptr(self.obs), self._stream()))
        return self.obs

    # ------------------------------------------------------------------ one caller seat against an on-device bot
    def _seat_buffers(self):
        torch = _torch()
        if getattr(self, "_obs_seat", None) is None:
            with torch.cuda.device(self.device):
                self._obs_seat = torch.zeros((self.num_envs, _lib.OBS_LEN), dtype=self.obs_dtype, device=self.device)
                self._actions_seat = torch.zeros((self.num_envs, _lib.NUM_ACTIONS, 2), dtype=torch.int32, device=self.device)
            self._p["obs_seat"] = C.c_void_p(self._obs_seat.data_ptr())
        return self._obs_seat

    def step_vs(self, policy, actions, seat=0, out=None, features=None):
        """One turn of the loop the reference's training and evaluation scripts run (evaluate.py:143-152;
        agents/Smart_State/training_scripts/dqn_smart_state_training.py:114-122): the caller plays `seat` with `actions` -- int32
        [N, 7, 2], or a [N, 2, 7, 2] tensor whose rows [:, seat] are used --, the on-device scripted bot `policy` (a name from
        _lib.POLICY_NAMES or an EVG_POLICY_* id) plays the other seat, evaluated INSIDE the step kernel from the on-chip state
        (evg_step_vs_policy: one launch, no opponent observations or orders through HBM).  Returns (obs_seat [N, 105] -- the caller's
        seat only --, reward [N, 2], done [N], info) like step(); bit-identical to scripted_actions(policy, 1 - seat) + step().
        features=(shared [N, 34], swarm [N, 12, 13]) float32 tensors: the launch ALSO writes the Smart_State features of the new observation into them
        (evg_step_vs_policy_smart: what smart_state_compact(-1, obs_seat, shared, swarm) would compute afterwards, without that kernel)."""
        torch = _torch()
        pid = self.POLICIES[policy] if isinstance(policy, str) else int(policy)
        a = actions
        if not (type(a) is torch.Tensor and a.dtype is self._int32 and a.device == self.device and a.is_contiguous()):
            a = torch.as_tensor(actions, device=self.device)
            a = (a if a.dtype == torch.int32 else a.to(torch.int32)).contiguous()      # truncation, like action.astype(int) (server.py:232)
        if tuple(a.shape) == self._act_shape:
            both = 1
        elif tuple(a.shape) == (self.num_envs, _lib.NUM_ACTIONS, 2):
            both = 0
        else:
            raise ValueError("actions must have shape [N, 7, 2] (the caller's seat) or [N, 2, 7, 2], got %s" % (tuple(a.shape),))
        obs = self._seat_buffers() if out is None else self._user(out, (self.num_envs, _lib.OBS_LEN), self.obs_dtype, "out")
        p = self._p
        if features is not None:
            shared, swarm = features
            self._user(shared, (self.num_envs, 34), torch.float32, "features[0] (shared)")
            self._user(swarm, (self.num_envs, _lib.NUM_GROUPS, 13), torch.float32, "features[1] (swarm)")
            rc = self.L.evg_step_vs_policy_smart(self._h, int(seat), C.c_void_p(a.data_ptr()), both, pid, C.c_void_p(obs.data_ptr()),
                                                 C.c_void_p(shared.data_ptr()), C.c_void_p(swarm.data_ptr()), p["reward"], p["done"], p["winner"], p["scores"],
                                                 p["status"], self._stream())
        else:
            rc = self.L.evg_step_vs_policy(self._h, int(seat), C.c_void_p(a.data_ptr()), both, pid, C.c_void_p(obs.data_ptr()), p["reward"], p["done"],
                                           p["winner"], p["scores"], p["status"], self._stream())
        if rc:
            self._check(rc)
        return obs, self.reward, self.done, self._info

    def rollout_vs(self, steps, policy, seat=0, time_kernel=False):
        """`steps` turns of the learner-seat loop driven from native code (evg_rollout_vs_policy): per turn the on-device random_actions
        generator writes the caller seat's rows into a tensor (the stand-in for a policy network's output), then step_vs(policy) runs.
        Returns like step_vs(); with time_kernel=True also the stream time per turn in ms (synchronises)."""
        pid = self.POLICIES[policy] if isinstance(policy, str) else int(policy)
        obs = self._seat_buffers()
        ms = C.c_float(0.0)
        p = self._p
        self._check(self.L.evg_rollout_vs_policy(self._h, int(steps), int(seat), pid, self._ptr(self._actions_seat), p["obs_seat"], p["reward"], p["done"],
                                                 p["winner"],
                                                 p["scores"], p["status"], C.byref(ms) if time_kernel else None, self._stream()))
        out = (obs, self.reward, self.done, self._info)
        return out + (float(ms.value),) if time_kernel else out

    def observe_seat(self, seat=0, out=None):
        """obs [N, 105] of one seat from the current state (evg_observe_seat): what a step_vs() loop starts from after reset()."""
        obs = self._seat_buffers() if out is None else self._user(out, (self.num_envs, _lib.OBS_LEN), self.obs_dtype, "out")
        self._check(self.L.evg_observe_seat(self._h, int(seat), self._ptr(obs), self._stream()))
        return obs

    def random_actions_seat(self, seat=0, out=None):
        """int32 [N, 7, 2]: the rows [:, seat] of random_actions() (a stand-in for a learner's policy output)."""
        self._seat_buffers()
        out = self._actions_seat if out is None else self._user(out, (self.num_envs, _lib.NUM_ACTIONS, 2), self._int32, "out")
        self._check(self.L.evg_random_actions_seat(self._h, int(seat), self._ptr(out), self._stream()))
        return out

    def fog_of_war(self, out=None):
        """uint8 [N, 2, 11]: the fog-of-war mask the reference computes in board_state and discards (server.py:402-425)."""
        torch = _torch()
        if out is None:
            if getattr(self, "_fog", None) is None:
                self._fog = torch.zeros((self.num_envs, 2, _lib.NUM_NODES), dtype=torch.uint8, device=self.device)
            out = self._fog
        self._user(out, (self.num_envs, 2, _lib.NUM_NODES), torch.uint8, "out")
        self._check(self.L.evg_fog_of_war(self._h, self._ptr(out), None, self._stream()))
        return out

    def knowledge(self, out=None):
        """uint8 [N, 2, 11]: node knowledge levels 0/1/2 of build_knowledge_output (server.py:779-832)."""
        torch = _torch()
        if out is None:
            if getattr(self, "_know", None) is None:
                self._know = torch.zeros((self.num_envs, 2, _lib.NUM_NODES), dtype=torch.uint8, device=self.device)
            out = self._know
        self._user(out, (self.num_envs, 2, _lib.NUM_NODES), torch.uint8, "out")
        self._check(self.L.evg_fog_of_war(self._h, None, self._ptr(out), self._stream()))
        return out

    def sightings(self, out=None):
        """int8 [N, 2, 12, 4]: what player p knows of opposing group g -- (seen, node id, destination key, units alive) --
        the `opp_k` of build_knowledge_output (server.py:845-907); key -1 = staying, else the destination's index (ID - 1)."""
        torch = _torch()
        if out is None:
            if getattr(self, "_sight", None) is None:
                self._sight = torch.zeros((self.num_envs, 2, _lib.NUM_GROUPS, 4), dtype=torch.int8, device=self.device)
            out = self._sight
        self._user(out, (self.num_envs, 2, _lib.NUM_GROUPS, 4), torch.int8, "out")
        self._check(self.L.evg_sightings(self._h, self._ptr(out), self._stream()))
        return out

    def smart_state(self, player, obs=None, out=None):
        """float32 [N, 12, 59]: the per-swarm input of the reference's Smart_State agents (DQNAgent.create_swarm_obs) for
        seat `player`, computed on device from `obs` (default: the env's observation buffer)."""
        torch = _torch()
        obs = self.obs if obs is None else obs
        if out is None:
            out = torch.empty((self.num_envs, _lib.NUM_GROUPS, 59), dtype=torch.float32, device=self.device)
        self._user(out, (self.num_envs, _lib.NUM_GROUPS, 59), torch.float32, "out")
        if isinstance(obs, torch.Tensor) and obs.dim() == 2:          # a one-seat observation tensor [N, 105] (step_vs / observe_seat): `player` is not needed
            self._user(obs, (self.num_envs, _lib.OBS_LEN), self.obs_dtype, "obs")
            self._check(self.L.evg_smart_state_seat(self._h, self._ptr(obs), self._ptr(out), self._stream()))
            return out
        self._user(obs, (self.num_envs, 2, _lib.OBS_LEN), self.obs_dtype, "obs")
        self._check(self.L.evg_smart_state(self._h, int(player), self._ptr(obs), self._ptr(out), self._stream()))
        return out

    def smart_state_compact(self, player, obs=None, shared=None, swarm=None):
        """The Smart_State features without their redundancy (evg_smart_state_compact): (shared float32 [N, 34], swarm float32 [N, 12, 13]) with
        features[e, s] == cat(shared[e], swarm[e, s], onehot(s)) value for value -- 760 B per env instead of 2 832.  `obs` as in smart_state():
        the env's observation buffer, a [N, 2, 105] tensor (rows of seat `player`) or a one-seat tensor [N, 105]."""
        torch = _torch()
        obs = self.obs if obs is None else obs
        if shared is None:
            shared = torch.empty((self.num_envs, 34), dtype=torch.float32, device=self.device)
        if swarm is None:
            swarm = torch.empty((self.num_envs, _lib.NUM_GROUPS, 13), dtype=torch.float32, device=self.device)
        self._user(shared, (self.num_envs, 34), torch.float32, "shared")
        self._user(swarm, (self.num_envs, _lib.NUM_GROUPS, 13), torch.float32, "swarm")
        if isinstance(obs, torch.Tensor) and obs.dim() == 2:
            self._user(obs, (self.num_envs, _lib.OBS_LEN), self.obs_dtype, "obs")
            player = -1
        else:
            self._user(obs, (self.num_envs, 2, _lib.OBS_LEN), self.obs_dtype, "obs")
        self._check(self.L.evg_smart_state_compact(self._h, int(player), self._ptr(obs), self._ptr(shared), self._ptr(swarm), self._stream()))
        return shared, swarm

    def smart_actions(self, q, player=0, obs=None, out=None, directions=None):
        """Network output -> orders on the device (evg_smart_actions = DQNAgent.get_best_actions): `q` float32 [N, 12, 5], the policy network's Q values of
        every swarm for the directions left / right / up / down / stay; `obs` the env's observation buffer, a [N, 2, 105] tensor (rows of seat `player`) or a
        one-seat tensor [N, 105] (step_vs / observe_seat).  Returns int32 [N, 7, 2] {swarm, node}: what step_vs() takes as `actions` -- the seven swarms with
        the LOWEST best Q, like the reference (ascending stable sort, first seven).  `directions`: an int32 [N, 7, 2] tensor that also receives {swarm, direction}."""
        torch = _torch()
        obs = self.obs if obs is None else obs
        self._user(q, (self.num_envs, _lib.NUM_GROUPS, 5), torch.float32, "q")
        if out is None:
            self._seat_buffers()
            out = self._actions_seat
        self._user(out, (self.num_envs, _lib.NUM_ACTIONS, 2), self._int32, "out")
        if directions is not None:
            self._user(directions, (self.num_envs, _lib.NUM_ACTIONS, 2), self._int32, "directions")
        if isinstance(obs, torch.Tensor) and obs.dim() == 2:
            self._user(obs, (self.num_envs, _lib.OBS_LEN), self.obs_dtype, "obs")
            player = -1
        else:
            self._user(obs, (self.num_envs, 2, _lib.OBS_LEN), self.obs_dtype, "obs")
        rc = self.L.evg_smart_actions(self._h, int(player), C.c_void_p(obs.data_ptr()), C.c_void_p(q.data_ptr()), C.c_void_p(out.data_ptr()),
                                      self._ptr(directions), self._stream())
        if rc:
            self._check(rc)
        return out

    def smart_get_action(self, q, epsilon, seat=0, obs=None, out=None, directions=None, explored=None):
        """DQNAgent.get_action on the device (evg_smart_get_action): per env the epsilon coin, then get_random_actions (7 distinct swarms, 7 directions with
        replacement, get_move) or get_best_actions (smart_actions).  `epsilon`: a float in [0, 1] for every env, or a float32 tensor [N] with one per env.
        `seat` is the agent's player number (it keys the draws); `obs` as in smart_actions.  `explored`: a uint8 tensor [N] that receives 1 where the random
        branch ran.  Returns int32 [N, 7, 2] {swarm, node}, what step_vs() takes as `actions`."""
        torch = _torch()
        obs = self.obs if obs is None else obs
        self._user(q, (self.num_envs, _lib.NUM_GROUPS, 5), torch.float32, "q")
        if out is None:
            self._seat_buffers()
            out = self._actions_seat
        self._user(out, (self.num_envs, _lib.NUM_ACTIONS, 2), self._int32, "out")
        if directions is not None:
            self._user(directions, (self.num_envs, _lib.NUM_ACTIONS, 2), self._int32, "directions")
        if explored is not None:
            self._user(explored, (self.num_envs,), torch.uint8, "explored")
        one_seat = isinstance(obs, torch.Tensor) and obs.dim() == 2
        self._user(obs, (self.num_envs, _lib.OBS_LEN) if one_seat else (self.num_envs, 2, _lib.OBS_LEN), self.obs_dtype, "obs")
        eps_env = None
        if isinstance(epsilon, torch.Tensor):
            eps_env = self._user(epsilon, (self.num_envs,), torch.float32, "epsilon")
            epsilon = 0.0
        rc = self.L.evg_smart_get_action(self._h, int(seat), int(one_seat), C.c_void_p(obs.data_ptr()), C.c_void_p(q.data_ptr()), float(epsilon),
                                         self._ptr(eps_env), C.c_void_p(out.data_ptr()), self._ptr(directions), self._ptr(explored), self._stream())
        if rc:
            self._check(rc)
        return out

    @staticmethod
    def expand_smart_state(shared, swarm):
        """[N, 12, 59] from the compact pair (for checks; a consumer would rather split its first layer's weights)."""
        torch = _torch()
        n = shared.shape[0]
        eye = torch.eye(_lib.NUM_GROUPS, dtype=shared.dtype, device=shared.device).expand(n, -1, -1)
        return torch.cat([shared[:, None, :].expand(-1, _lib.NUM_GROUPS, -1), swarm, eye], dim=2)

    @staticmethod
    def move_table():
        """int32 [11, 5]: Move_Translation.get_move(node0, direction) as a lookup table (directions left, right, up, down, stay)."""
        t = np.zeros((11, 5), np.int32)
        _lib.load().evg_move_table(t.ctypes.data_as(C.c_void_p))
        return t

    def random_actions(self, out=None):
        """On-device equivalent of agents/State_Machine/random_actions.py for both players of every env."""
        out = self._actions if out is None else self._user(out, self._act_shape, self._int32, "out")
        self._check(self.L.evg_random_actions(self._h, self._ptr(out), self._stream()))
        return out

    POLICIES = dict({n: i for i, n in enumerate(_lib.POLICY_NAMES)}, **_lib.POLICY_ALIASES)

    def scripted_actions(self, policy, player, obs=None, out=None):
        """Orders of the on-device scripted agent `policy` (a name from _lib.POLICY_NAMES = the file names under
        agents/State_Machine/, or an EVG_POLICY_* id)
        playing seat `player` in every env, from the observations `obs` (default: the env's own obs buffer).  Writes
        rows [:, player] of `out` (default: the env's action buffer) and returns it.  Agent objects persist across
        episodes like the reference's; scripted_reset() re-creates them."""
        obs = self.obs if obs is None else obs
        out = self._actions if out is None else out
        pid = self.POLICIES[policy] if isinstance(policy, str) else int(policy)
        self._user(obs, (self.num_envs, 2, _lib.OBS_LEN), self.obs_dtype, "obs")
        self._user(out, self._act_shape, self._int32, "out")
        self._check(self.L.evg_scripted_actions(self._h, pid, int(player), self._ptr(obs), self._ptr(out), self._stream()))
        return out

    def scripted_reset(self):
        self._check(self.L.evg_scripted_reset(self._h, self._stream()))

    def rollout_random(self, steps, time_kernel=False, fused=True, turns_per_launch=1, observe=True, record_actions=True, prepare=False):
        """`steps` turns of random-vs-random play driven from native code (evg_rollout_random): per turn the
        on-device random_actions generator fills self._actions, then the step kernel runs (fused=True: the step kernel draws the same orders itself and stores them in
        self._actions -- one launch per turn; turns_per_launch > 1: persistent form, each launch plays that many
        consecutive turns per wavefront, outputs still written every turn).  observe=False / record_actions=False (fused forms only):
        no observations are written (self.obs keeps its old content) / the orders are not stored -- for loops that read only rewards,
        done flags and episode results.  Returns the outputs of
        the last turn like step(); with time_kernel=True also the STREAM time per turn in ms (synchronises): persistent form -- the duration
        of each launch (plan) between two events, summed, over the turns played; one launch per turn -- two events around the whole loop,
        i.e. step kernel + launch gap (+ the action kernel with fused=False).  The kernel alone is in the rocprofv3 traces under profiles/.
        A timed call also reports a chunk hand-over fault of the handle (EvgFault).
        prepare=True: nothing is played -- the hipGraphs of the launches such a call would make (persistent form) are captured and
        instantiated now, so that the first real call pays no one-time cost (e.g. before a timed region)."""
        ms = C.c_float(0.0)
        if prepare:
            steps, time_kernel = -abs(int(steps)), False
        p = self._p                     # cached raw pointers of the env's own buffers: the call itself is the only host work before the launch
        if not fused and not (observe and record_actions):
            raise ValueError("observe=False / record_actions=False need the fused forms (the unfused form passes the orders through self._actions)")
        rc = self.L.evg_rollout_random(self._h, int(steps), (max(1, int(turns_per_launch)) if fused else 0), p["_actions"] if record_actions else None,
                                       p["obs"] if observe else None, p["reward"],
                                       p["done"], p["winner"], p["scores"], p["status"], C.byref(ms) if time_kernel else None, self._stream())
        if rc:
            self._check(rc)
        out = (self.obs, self.reward, self.done, self._info)
        return out + (float(ms.value),) if time_kernel else out

    def rollout_policies(self, steps, policy0, policy1, time_kernel=False, fused=True, turns_per_launch=1, observe=True, record_actions=True, prepare=False):
        """`steps` turns of on-device policy0 (seat 0) vs policy1 (seat 1), driven from native code (evg_rollout_policies).
        fused=False: two agent launches per turn read self.obs (which must hold the current observations; it does after
        reset()/step()); fused=True: the step kernel evaluates both agents from the on-chip state, one launch per turn or --
        turns_per_launch > 1 -- the persistent form.  Identical results.  observe=False / record_actions=False (fused forms only): no
        observations are written / the orders are not stored (the evaluation harness: it reads only the episode results)."""
        ms = C.c_float(0.0)
        if prepare:                                      # see rollout_random
            steps, time_kernel = -abs(int(steps)), False
        p0 = self.POLICIES[policy0] if isinstance(policy0, str) else int(policy0)
        p1 = self.POLICIES[policy1] if isinstance(policy1, str) else int(policy1)
        if not fused and not (observe and record_actions):
            raise ValueError("observe=False / record_actions=False need the fused forms (the agents of the unfused form read self.obs)")
        self._check(self.L.evg_rollout_policies(self._h, int(steps), (max(1, int(turns_per_launch)) if fused else 0), p0, p1,
                                                self._ptr(self._actions) if record_actions else None,
                                               self._ptr(self.obs) if observe else None, self._ptr(self.reward),
                                               self._ptr(self.done), self._ptr(self.winner), self._ptr(self.scores), self._ptr(self.status),
                                               C.byref(ms) if time_kernel else None, self._stream()))
        out = (self.obs, self.reward, self.done, self._info)
        return out + (float(ms.value),) if time_kernel else out

    # ------------------------------------------------------------------ state exchange / stats
    def get_state(self):
        N = self.num_envs
        s = dict(groups=np.zeros((N, 2, 12, 8), np.int32), nodes=np.zeros((N, 11, 2), np.int32),
                 health=np.zeros((N, 2, 100), np.float64), env=np.zeros((N, 4), np.int32))
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        self._check(self.L.evg_get_state(self._h, p(s["groups"]), p(s["nodes"]), p(s["health"]), p(s["env"])))
        return s

    def set_state(self, groups, nodes, health, env):
        N = self.num_envs
        g = np.ascontiguousarray(groups, np.int32)
        n = np.ascontiguousarray(nodes, np.int32)
        h = np.ascontiguousarray(health, np.float64)
        e = np.ascontiguousarray(env, np.int32)
        if g.shape != (N, 2, 12, 8) or n.shape != (N, 11, 2) or h.shape != (N, 2, 100) or e.shape != (N, 4):
            raise ValueError("set_state: wrong array shapes")
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        self._check(self.L.evg_set_state(self._h, p(g), p(n), p(h), p(e)))

    def get_run_state(self):
        """What get_state() does not carry and a resumed run needs (evg_get_run_state): the scripted agents' objects uint32 [N, 2, 3], the running episode
        returns float32 [N, 2], the results of the last finished episodes (returns, length, winner) and the win counters (totals)."""
        N = self.num_envs
        d = dict(agents=np.zeros((N, 2, 3), np.uint32), running_returns=np.zeros((N, 2), np.float32), returns=np.zeros((N, 2), np.float32),
                 length=np.zeros(N, np.int32), winner=np.zeros(N, np.int8), totals=np.zeros(4, np.int64))
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        self._check(self.L.evg_get_run_state(self._h, p(d["agents"]), p(d["running_returns"]), p(d["returns"]), p(d["length"]), p(d["winner"]), p(d["totals"])))
        return d

    def set_run_state(self, agents=None, running_returns=None, returns=None, length=None, winner=None, totals=None):
        N = self.num_envs

        def arr(a, dtype, shape, name):
            if a is None:
                return None
            a = np.ascontiguousarray(a, dtype)
            if a.shape != shape:
                raise ValueError("set_run_state: %s must have shape %s, got %s" % (name, shape, a.shape))
            return a
        keep = [arr(agents, np.uint32, (N, 2, 3), "agents"), arr(running_returns, np.float32, (N, 2), "running_returns"), arr(returns, np.float32, (N, 2), "returns"),
                arr(length, np.int32, (N,), "length"), arr(winner, np.int8, (N,), "winner"), arr(totals, np.int64, (4,), "totals")]
        self._check(self.L.evg_set_run_state(self._h, *[None if a is None else a.ctypes.data_as(C.c_void_p) for a in keep]))

    def checkpoint(self):
        """Everything a handle created with the same arguments needs to continue this run bit for bit: get_state() + get_run_state() (+ the generators of the
        stock-entropy mode).  Host arrays (numpy): np.savez-able."""
        ck = dict(state=self.get_state(), run=self.get_run_state())
        if self.rng_mode == "mt19937":
            ck["entropy"] = self.get_stock_entropy()
        return ck

    def restore(self, ck):
        s = ck["state"]
        self.set_state(s["groups"], s["nodes"], s["health"], s["env"])
        self.set_run_state(**ck["run"])
        if "entropy" in ck:
            self.set_stock_entropy(ck["entropy"])

    def seed_stock_entropy(self, seeds=None):
        """rng_mode="mt19937": np.random.seed(seeds[e]) for every env (None: the create-time rule seed + env_id_base + e)."""
        a = None
        if seeds is not None:
            a = np.ascontiguousarray(np.asarray(seeds, np.uint64) & 0xFFFFFFFF, np.uint32)
            if a.shape != (self.num_envs,):
                raise ValueError("seeds must have one entry per env")
        self._check(self.L.evg_seed_stock_entropy(self._h, None if a is None else a.ctypes.data_as(C.c_void_p), self._stream()))

    def get_stock_entropy(self):
        """uint32 [N, 625]: the 624 key words and the position of every env's MT19937 (np.random.get_state()[1:3])."""
        a = np.zeros((self.num_envs, 625), np.uint32)
        self._check(self.L.evg_get_stock_entropy(self._h, a.ctypes.data_as(C.c_void_p)))
        return a

    def set_stock_entropy(self, state):
        a = np.ascontiguousarray(state, np.uint32)
        if a.shape != (self.num_envs, 625):
            raise ValueError("state must be uint32 [N, 625]")
        self._check(self.L.evg_set_stock_entropy(self._h, a.ctypes.data_as(C.c_void_p)))

    def episode_stats(self):
        N = self.num_envs
        r, ln, w, tot = np.zeros((N, 2), np.float32), np.zeros(N, np.int32), np.zeros(N, np.int8), np.zeros(4, np.int64)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        self._check(self.L.evg_episode_stats(self._h, p(r), p(ln), p(w), p(tot)))
        return dict(returns=r, length=ln, winner=w, totals=tot)

    def episode_stats_device(self):
        """Zero-copy torch views of the per-env results of the last finished episode (for the multi-GPU gather).  They alias
        memory owned by the handle: they are valid until close() and must not be used afterwards."""
        torch = _torch()
        pr, pl, pw = C.c_void_p(), C.c_void_p(), C.c_void_p()
        self._check(self.L.evg_episode_stats_device(self._h, C.byref(pr), C.byref(pl), C.byref(pw)))
        N = self.num_envs

        def view(ptr, shape, dtype, itemsize):
            n = int(np.prod(shape))
            iface = dict(shape=(n,), typestr={4: "<f4" if dtype == torch.float32 else "<i4", 1: "|i1"}[itemsize],
                         data=(ptr.value, False), version=2)
            holder = type("_Dev", (), {"__cuda_array_interface__": iface})()
            return torch.as_tensor(holder, device=self.device).view(*shape)

        return dict(returns=view(pr, (N, 2), torch.float32, 4), length=view(pl, (N,), torch.int32, 4),
                    winner=view(pw, (N,), torch.int8, 1))

    def packed_episode_results(self, out=None, counts=None):
        """The per-env results of the last finished episode as ONE float32 [N, 4] tensor {return p0, return p1, winner, length}
        (evg_pack_episode_results: one small kernel on the current stream): the payload of the path's single exchange between
        GPUs (`distributed.ResultGather`).  `out`: a float32 [N, 4] CUDA tensor to fill (default: a buffer owned by the env).
        `counts`: an int64 [4] CUDA tensor that the same kernel fills with the rows' win bookkeeping (p0, p1, ties, unfinished) --
        what a multi-GPU run all-reduces next to the gather."""
        torch = _torch()
        if counts is not None:
            self._user(counts, (4,), torch.int64, "counts")
        if out is None:
            if getattr(self, "_packed", None) is None:
                self._packed = torch.empty((self.num_envs, 4), dtype=torch.float32, device=self.device)
            out = self._packed
        else:
            out = self._user(out, (self.num_envs, 4), torch.float32, "out")
        if counts is not None:
            self._check(self.L.evg_pack_episode_results_counted(self._h, self._ptr(out), self._ptr(counts), self._stream()))
        else:
            self._check(self.L.evg_pack_episode_results(self._h, self._ptr(out), self._stream()))
        return out

    def check_fault(self):
        """Synchronises the device and raises EvgFault when the handle's fault word is set (evg_check_fault: a chunked rollout launch failed
        to hand a set of envs on -- never expected; the handle must then be destroyed).  get_state(), episode_stats(),
        episode_stats_device() and timed rollouts check it themselves; packed_episode_results() returns poisoned rows (winner -2)."""
        w = C.c_uint32(0)
        self._check(self.L.evg_check_fault(self._h, C.byref(w)))
        return int(w.value)

    def launch_plan(self, turns_per_launch=150):
        """(number of kernel launches, description) of what one rollout launch of `turns_per_launch` turns runs for this batch on
        this device (evg_launch_plan): the step-kernel mapping per env range and the device capacity the plan was derived from."""
        buf = C.create_string_buffer(1024)
        n = self.L.evg_launch_plan(self._h, int(turns_per_launch), buf, len(buf))
        if n < 0:
            self._check(n)
        return n, buf.value.decode("utf-8", "replace")

    @property
    def state_bytes_per_env(self):
        return int(self.L.evg_state_bytes_per_env(self._h))
