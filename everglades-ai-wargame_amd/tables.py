"""Constant tables of the environment: what the reference parses at every reset from
config/DemoMap.json and config/UnitDefinitions.json (server.py:40-131) plus the fixed army of
everglades_env.py:145-156, flattened once into the `evg_tables` struct of include/evg.h.
"""
import json
import os

from . import _lib

UNIT_CLASSES = ["controller", "striker", "tank"]        # everglades_env.py:22
P1_NODE_MAP_DEMO = [0, 11, 8, 9, 10, 5, 6, 7, 2, 3, 4, 1]   # server.py:89 (hand-made for DemoMap)
RESOURCE_BITS = {"DEFENSE": 1, "OBSERVE": 2}


def default_tables():
    """DemoMap + UnitDefinitions + default army, as compiled into libevg (evg_default_tables)."""
    t = _lib.EvgTables()
    _lib.load().evg_default_tables(t)
    return t


def _resolve(config_dir, name):
    # server.py:24-38 accepts either a path that exists or config_dir + name
    if name is None:
        return None
    if os.path.exists(name):
        return name
    if config_dir is not None and os.path.exists(os.path.join(config_dir, name)):
        return os.path.join(config_dir, name)
    if config_dir is not None and os.path.exists(config_dir + name):
        return config_dir + name
    raise FileNotFoundError("cannot find %r (config_dir=%r)" % (name, config_dir))


def tables_from_json(map_file=None, unit_file=None, config_dir=None, p1_node_map=None, num_units=100, num_groups=12,
                     max_turns=150):
    """Build tables from files in the reference's JSON schema; anything not given keeps the DemoMap default."""
    t = default_tables()
    map_path, unit_path = _resolve(config_dir, map_file), _resolve(config_dir, unit_file)
    if map_path is not None:
        with open(map_path) as fid:
            dat = json.load(fid)
        nodes = dat["nodes"]
        if len(nodes) != _lib.NUM_NODES or sorted(n["ID"] for n in nodes) != list(range(1, _lib.NUM_NODES + 1)):
            raise ValueError("this build supports maps with node IDs 1..11 (DemoMap-shaped), got %d nodes" % len(nodes))
        for a in range(12):
            for b in range(12):
                t.node_dist[a][b] = 0
            t.node_resource[a] = 0
            t.node_team_start[a] = -1
        for n in nodes:
            i = int(n["ID"])
            t.node_control_points[i] = int(n["ControlPoints"])
            t.node_defense[i] = float(n["StructureDefense"])
            t.node_team_start[i] = int(n["TeamStart"])
            for r in n.get("Resource", []):
                t.node_resource[i] |= RESOURCE_BITS.get(r, 0)
            for c in n["Connections"]:
                t.node_dist[i][int(c["ConnectedID"])] = int(c["Distance"])
    pm = P1_NODE_MAP_DEMO if p1_node_map is None else list(p1_node_map)
    for i, v in enumerate(pm):
        t.p1_node_map[i] = int(v)
    if unit_path is not None:
        with open(unit_path) as fid:
            dat = json.load(fid)
        units = dat["units"]
        if len(units) > 4:
            raise ValueError("at most 4 unit types are supported")
        names = {}
        t.num_unit_types = len(units)
        for uid, u in enumerate(units):                 # unit id = JSON order (server.py:113-130)
            names[u["Name"].lower()] = uid
            t.unit_health[uid], t.unit_damage[uid], t.unit_speed[uid] = int(u["Health"]), int(u["Damage"]), int(u["Speed"])
            t.unit_control[uid], t.unit_cost[uid] = int(u["Control"]), int(u["Cost"])
        per = num_units // num_groups
        for p in range(2):                              # _build_groups, everglades_env.py:145-156
            for g in range(num_groups):
                t.group_type[p][g] = names[UNIT_CLASSES[g % len(UNIT_CLASSES)]]
                t.group_size[p][g] = per if g < num_groups - 1 else num_units - per * (num_groups - 1)
    t.max_turns = int(max_turns)
    return t
