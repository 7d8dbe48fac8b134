import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    d = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return {k: d[k] for k in d.files}


def golden_initial_state(d, g, t=0):
    """Canonical state arrays (evg_set_state layout) of game g at turn t of a trajectory fixture."""
    groups = d["groups"][g, t].astype(np.int32)[None]
    nodes = d["nodes"][g, t].astype(np.int32)[None]
    health = d["health"][g, t][None].copy()
    status = 0 if t == 0 else int(d["status"][g, t - 1])
    env = np.array([[t, status, int(d["episode"][g]), 0]], np.int32)
    return groups, nodes, health, env


TRAJ_FILES = ["traj_random.npz", "traj_wild.npz", "traj_rush.npz", "traj_brawl.npz", "traj_brawl_v_random.npz",
              "traj_rush_v_random.npz", "kat_nocombat.npz"]


# the imported reference on NON-default map / unit files (oracle/custom_configs.py; EvergladesEnv.reset(map_file=, unit_file=))
CUSTOM_FILES = ["custom_varA.npz", "custom_varB.npz", "custom_varC.npz"]


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle as om
    om.build()
    return om
