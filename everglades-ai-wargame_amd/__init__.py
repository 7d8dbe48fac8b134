"""MI355X-native batched Everglades environment.

Drop-in for the turn loop of everglades-server and step()/reset() of gym_everglades
(jlehett/everglades-ai-wargame): Python host code over a thin C-ABI (include/evg.h, libevg.so)
whose kernels are hand-written HIP for gfx950.  Import as `everglades_amd` (alias module at the
repository root; this directory's name is not a legal Python identifier).
"""
from . import _lib
from ._lib import EvgError, EvgFault, load as load_library
from .tables import default_tables, tables_from_json
from .vec_env import EvergladesVecEnv
from .pipeline import PipelinedVecEnv
from .env import EvergladesEnv, canonical_actions
from .distributed import shard_range, gather_episode_results, win_counts, ResultGather, NativeGather
from .harness import evaluate, evaluate_all, proportion_confint_normal

__all__ = ["EvergladesVecEnv", "PipelinedVecEnv", "EvergladesEnv", "EvgError", "EvgFault", "load_library", "default_tables", "tables_from_json",
           "canonical_actions", "shard_range", "gather_episode_results", "win_counts", "ResultGather", "NativeGather", "evaluate", "evaluate_all",
           "proportion_confint_normal"]



def _register_with_gym():
    """The reference's gym id (gym_everglades/__init__.py:3-6: register(id='everglades-v0', entry_point=...)), so that `gym.make('everglades-v0')`
    builds this class.  gym is optional: without it there is nothing to register (ImportError only -- any other failure of the registration is a
    real error and propagates)."""
    try:
        from gym.envs.registration import register
    except ImportError:
        return False
    register(id="everglades-v0", entry_point="everglades_amd:EvergladesEnv")
    return True


GYM_REGISTERED = _register_with_gym()
