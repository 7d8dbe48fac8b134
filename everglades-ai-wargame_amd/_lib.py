"""ctypes binding of libevg.so (include/evg.h).  This is the whole FFI surface; no torch types cross it.

The library is built in-tree by `__graft_entry__.build()` / `make -C everglades-ai-wargame_amd/csrc`.
There is no CPU fallback: a missing library or a missing gfx950 device raises.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libevg.so")                # the product library; nothing in the environment changes that
DIAG_LIB_PATH = os.path.join(HERE, "libevg_diag.so")      # `make -C csrc diag`: phase ablation, 16-envs-per-wave variant, forced IEEE division
# `make -C csrc graphs`: rollout launch plans replayed as library-owned hipGraphs (A/B build; measured slower)
GRAPHS_LIB_PATH = os.path.join(HERE, "libevg_graphs.so")
STAMPS_LIB_PATH = os.path.join(HERE, "libevg_stamps.so")  # `make -C csrc stamps`: in-kernel phase stamps (tools/stamps.py)

NUM_PLAYERS, NUM_GROUPS, NUM_NODES, NUM_UNITS, NUM_ACTIONS, OBS_LEN = 2, 12, 11, 100, 7, 105
MAX_SCORE = 3700
OBS_F32, OBS_F64, OBS_I16 = 0, 1, 2
ABI_VERSION = 6
ERR_COMM = -6         # EVG_ERR_COMM: RCCL missing or one of its calls failed (evg_comm_*, evg_gather_returns)
COMM_ID_BYTES = 128
ERR_FAULT = -5        # EVG_ERR_FAULT: the handle's fault word is set (evg_check_fault)
RNG_KEYED_PHILOX, RNG_STOCK_MT19937 = 0, 1
POLICY_NAMES = ["random", "cycle_rush_turn25", "cycle_rush_turn50", "swarm", "all_cycle", "base_rush_v1", "bull_rush",
                "cycle_target_node", "cycle_target_node1", "cycle_target_node11", "cycle_target_node11P2", "dfs_attack", "no_action",
                "random_actions_delay", "same_commands"]      # index = EVG_POLICY_* of include/evg.h (agents/State_Machine/<name>.py)
POLICY_ALIASES = {"random_actions": 0, "random_actions_2": 0, "swarm_agent": 3, "same_commands_2": 14}

EXPORTS = ["evg_default_tables", "evg_create", "evg_destroy", "evg_reset", "evg_step", "evg_observe", "evg_step_vs_policy", "evg_step_vs_policy_smart",
           "evg_observe_seat",
           "evg_random_actions_seat", "evg_smart_state_seat", "evg_smart_state_compact", "evg_check_fault", "evg_rollout_vs_policy", "evg_fog_of_war",
           "evg_sightings", "evg_smart_state", "evg_smart_actions", "evg_smart_get_action", "evg_move_table", "evg_random_actions", "evg_rollout_random", "evg_rollout_policies",
           "evg_scripted_actions", "evg_scripted_reset",
           "evg_get_state", "evg_set_state", "evg_get_run_state", "evg_set_run_state", "evg_seed_stock_entropy", "evg_get_stock_entropy", "evg_set_stock_entropy", "evg_episode_stats",
           "evg_episode_stats_device", "evg_pack_episode_results", "evg_pack_episode_results_counted", "evg_comm_unique_id", "evg_comm_init",
           "evg_gather_returns", "evg_comm_destroy", "evg_launch_plan", "evg_num_envs",
           "evg_state_bytes_per_env", "evg_last_error", "evg_abi_version"]


def kernel_source_files():
    """The files the product's device code is compiled from: every header and include of csrc/ (kernels, phases, RNG) plus the public header;
    NOT the host side (evg_abi.hip).  Sorted by name, so the hash does not depend on directory order."""
    import glob
    csrc = os.path.join(HERE, "csrc")
    files = sorted(f for pat in ("*.h", "*.inc", "evg_kernels.hip") for f in glob.glob(os.path.join(csrc, pat)))
    return files + [os.path.join(os.path.dirname(HERE), "include", "evg.h")]


_C_TOKEN = None


def _code_only(text):
    """C / C++ source without comments and with runs of white space collapsed (string and character literals kept as they are): what the compiler sees."""
    global _C_TOKEN
    import re
    if _C_TOKEN is None:
        _C_TOKEN = re.compile(r'''("(?:\\.|[^"\\])*"|'(?:\\.|[^'\\])*')|(/\*.*?\*/|//[^\n]*)''', re.S)
    text = _C_TOKEN.sub(lambda m: m.group(1) if m.group(1) is not None else " ", text)
    return " ".join(text.split())


def kernel_source_hash():
    """Identifies the kernel CODE a libevg.so was built from: the counter passes committed under profiles/ carry it, and bench.py uses
    only figures whose hash equals the one of the tree it runs in (the ONE definition: bench.py and tools/_prof.py call this).  Comments and white space
    are not part of it (round 6: a documentation change in include/evg.h used to invalidate every counter pass)."""
    import hashlib
    h = hashlib.sha256()
    for f in kernel_source_files():
        h.update(os.path.basename(f).encode())
        h.update(_code_only(open(f, encoding="utf-8", errors="replace").read()).encode())
    return h.hexdigest()[:16]


class EvgTables(C.Structure):
    _fields_ = [
        ("node_dist", (C.c_int32 * 12) * 12), ("node_control_points", C.c_int32 * 12),
        ("node_defense", C.c_double * 12), ("node_resource", C.c_int32 * 12),
        ("node_team_start", C.c_int32 * 12), ("p1_node_map", C.c_int32 * 12),
        ("num_unit_types", C.c_int32), ("unit_health", C.c_int32 * 4), ("unit_damage", C.c_int32 * 4),
        ("unit_speed", C.c_int32 * 4), ("unit_control", C.c_int32 * 4), ("unit_cost", C.c_int32 * 4),
        ("group_type", (C.c_int32 * 12) * 2), ("group_size", (C.c_int32 * 12) * 2), ("max_turns", C.c_int32),
    ]


class EvgConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("abi_version", C.c_uint32), ("num_envs", C.c_int32), ("device_id", C.c_int32),
        ("seed", C.c_uint64), ("env_id_base", C.c_uint64), ("obs_dtype", C.c_int32), ("auto_reset", C.c_int32),
        ("rng_mode", C.c_int32), ("cache_mib", C.c_int32), ("tables", EvgTables),
    ]


class EvgError(RuntimeError):
    pass


class EvgFault(EvgError):
    """EVG_ERR_FAULT: a chunked rollout launch of the handle failed to hand a set of envs on; its state and results are not valid."""


_libs = {}


def _hip_runtimes_mapped():
    """Paths of the libamdhip64 images mapped into this process."""
    found = set()
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                path = line.rsplit(None, 1)[-1]
                if "libamdhip64" in os.path.basename(path):
                    found.add(os.path.realpath(path))
    except OSError:
        pass
    return sorted(found)


def load(path=None):
    """Load libevg.so (or, for diagnostics, the library at `path`: an explicit argument, never an environment variable);
    fail loudly when it has not been built (there is no fallback path)."""
    path = os.path.abspath(path or LIB_PATH)
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise EvgError("%s not found -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                       "or `make -C everglades-ai-wargame_amd/csrc`; this package has no CPU fallback" % path)
    # libevg.so shares streams and device pointers with PyTorch, so both must run on ONE HIP runtime
    # instance: import torch first, so that its libamdhip64.so.7 is the one already in the process when
    # the loader resolves libevg's NEEDED entry of the same SONAME.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(path)
    rts = _hip_runtimes_mapped()
    if len(rts) > 1:
        raise EvgError("two HIP runtimes are mapped into this process (%s): torch's stream handles and device pointers would be "
                       "invalid inside libevg.so -- rebuild libevg.so against the libamdhip64 that torch loads" % ", ".join(rts))
    vp = C.c_void_p
    if hasattr(L, "evg_abi_version"):
        L.evg_abi_version.restype = C.c_int
        if L.evg_abi_version() != ABI_VERSION:
            raise EvgError("%s: ABI version %d, binding expects %d (an older build of the library? rebuild it)" % (path, L.evg_abi_version(), ABI_VERSION))
    missing = [n for n in EXPORTS if not hasattr(L, n)]
    if missing:       # an older build of the library (e.g. a baseline kept for tools/ab.sh): say what it lacks instead of an AttributeError
        raise EvgError("%s does not export %s -- it was built from older sources than this binding (ABI %d); rebuild it" % (path, ", ".join(missing),
                                                                                                                            ABI_VERSION))
    L.evg_last_error.restype = C.c_char_p
    L.evg_abi_version.restype = C.c_int
    L.evg_default_tables.argtypes = [C.POINTER(EvgTables)]
    L.evg_default_tables.restype = None
    L.evg_create.argtypes = [C.POINTER(EvgConfig), C.POINTER(vp)]
    L.evg_destroy.argtypes = [vp]
    L.evg_destroy.restype = None
    L.evg_reset.argtypes = [vp, vp, vp, vp]
    L.evg_step.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.evg_observe.argtypes = [vp, vp, vp]
    L.evg_step_vs_policy.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp]
    L.evg_get_run_state.argtypes = [vp] * 7
    L.evg_set_run_state.argtypes = [vp] * 7
    L.evg_step_vs_policy_smart.argtypes = [vp, C.c_int, vp, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.evg_observe_seat.argtypes = [vp, C.c_int, vp, vp]
    L.evg_random_actions_seat.argtypes = [vp, C.c_int, vp, vp]
    L.evg_smart_state_seat.argtypes = [vp, vp, vp, vp]
    L.evg_smart_state_compact.argtypes = [vp, C.c_int, vp, vp, vp, vp]
    L.evg_check_fault.argtypes = [vp, C.POINTER(C.c_uint32)]
    L.evg_rollout_vs_policy.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, C.POINTER(C.c_float), vp]
    L.evg_fog_of_war.argtypes = [vp, vp, vp, vp]
    L.evg_sightings.argtypes = [vp, vp, vp]
    L.evg_smart_state.argtypes = [vp, C.c_int, vp, vp, vp]
    L.evg_smart_actions.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp]
    L.evg_smart_get_action.argtypes = [vp, C.c_int, C.c_int, vp, vp, C.c_float, vp, vp, vp, vp, vp]
    L.evg_move_table.argtypes = [vp]
    L.evg_move_table.restype = None
    L.evg_random_actions.argtypes = [vp, vp, vp]
    L.evg_rollout_random.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, C.POINTER(C.c_float), vp]
    L.evg_rollout_policies.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, C.POINTER(C.c_float), vp]
    L.evg_scripted_actions.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp]
    L.evg_scripted_reset.argtypes = [vp, vp]
    L.evg_get_state.argtypes = [vp, vp, vp, vp, vp]
    L.evg_set_state.argtypes = [vp, vp, vp, vp, vp]
    L.evg_seed_stock_entropy.argtypes = [vp, vp, vp]
    L.evg_get_stock_entropy.argtypes = [vp, vp]
    L.evg_set_stock_entropy.argtypes = [vp, vp]
    L.evg_episode_stats.argtypes = [vp, vp, vp, vp, vp]
    L.evg_episode_stats_device.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    L.evg_pack_episode_results.argtypes = [vp, vp, vp]
    L.evg_pack_episode_results_counted.argtypes = [vp, vp, vp, vp]
    L.evg_comm_unique_id.argtypes = [vp]
    L.evg_comm_init.argtypes = [vp, vp, C.c_int, C.c_int, vp]
    L.evg_gather_returns.argtypes = [vp, C.c_int, vp, vp]
    L.evg_comm_destroy.argtypes = [vp]
    L.evg_launch_plan.argtypes = [vp, C.c_int, C.c_char_p, C.c_int]
    L.evg_num_envs.argtypes = [vp]
    L.evg_state_bytes_per_env.argtypes = [vp]
    if hasattr(L, "evg_diag_configure"):      # diagnostic libraries only
        L.evg_diag_configure.argtypes = [vp, C.c_uint32, C.c_int, C.c_int]
    _libs[path] = L
    return L


def check(rc, lib=None):
    if rc != 0:
        raise (EvgFault if rc == ERR_FAULT else EvgError)("libevg error %d: %s" % (rc, (lib or load()).evg_last_error().decode("utf-8", "replace")))
