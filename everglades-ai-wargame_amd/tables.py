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


def _whole(value, what):
    """The tables are integer tables: a JSON number with a fraction would be silently truncated here but used as a float by the
    reference (server.py:57-79, :116-123), so it is refused."""
    if isinstance(value, bool) or not isinstance(value, (int, float)) or int(value) != value:
        raise ValueError("%s must be a whole number, got %r" % (what, value))
    return int(value)


def tables_from_json(map_file=None, unit_file=None, config_dir=None, p1_node_map=None, num_units=100, num_groups=12,
                     max_turns=150):
    """Build tables from files in the reference's JSON schema (what EvergladesEnv.reset(map_file=, unit_file=, config_dir=) hands to
    server.py:24-131); anything not given keeps the DemoMap default.

    Pinned against the reference itself by tests/golden/custom_*.npz (oracle/custom_configs.py): directed connections with
    distances 2..7, one-way edges, control points up to 511, any non-negative StructureDefense, DEFENSE / OBSERVE anywhere, bases on
    any two nodes, unit files in any order with up to four types.  Refused, because the reference would behave differently from
    these tables: node lists that are not in ascending ID order (the reference's own fog mask then mixes list positions with IDs,
    server.py:409-418), a 'DEFEND' resource (it would switch the otherwise dead fortress bonus on, server.py:595), fractional values
    where the tables hold integers.  `p1_node_map` has no file in the reference (hard-coded for DemoMap, server.py:89)."""
    t = default_tables()
    map_path, unit_path = _resolve(config_dir, map_file), _resolve(config_dir, unit_file)
    if map_path is not None:
        with open(map_path) as fid:
            dat = json.load(fid)
        nodes = dat["nodes"]
        if len(nodes) != _lib.NUM_NODES or [n["ID"] for n in nodes] != list(range(1, _lib.NUM_NODES + 1)):
            raise ValueError("this build supports maps with the node IDs 1..11 listed in ascending order (DemoMap-shaped), got %r"
                             % [n.get("ID") for n in nodes])
        for a in range(12):
            for b in range(12):
                t.node_dist[a][b] = 0
            t.node_resource[a] = 0
            t.node_team_start[a] = -1
        for n in nodes:
            i = int(n["ID"])
            t.node_control_points[i] = _whole(n["ControlPoints"], "ControlPoints of node %d" % i)
            t.node_defense[i] = float(n["StructureDefense"])
            t.node_team_start[i] = _whole(n["TeamStart"], "TeamStart of node %d" % i)
            if "DEFEND" in n["Resource"]:
                raise ValueError("node %d lists the resource 'DEFEND': the reference would apply its fortress bonus there "
                                 "(server.py:595), which this build does not model" % i)
            for r in n["Resource"]:                     # any other string has no effect in the reference either (server.py:415, :442-443)
                t.node_resource[i] |= RESOURCE_BITS.get(r, 0)
            for c in n["Connections"]:
                b = _whole(c["ConnectedID"], "ConnectedID at node %d" % i)
                if not 1 <= b <= _lib.NUM_NODES:
                    raise ValueError("node %d is connected to the unknown node %d" % (i, b))
                if t.node_dist[i][b] == 0:              # the first listed connection to a node wins (server.py:245-249: break)
                    t.node_dist[i][b] = _whole(c["Distance"], "Distance %d -> %d" % (i, b))
        if sorted(int(t.node_team_start[i]) for i in range(1, 12) if t.node_team_start[i] != -1) != [0, 1]:
            raise ValueError("exactly one node must have TeamStart 0 and one TeamStart 1")
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
            for field, key in ((t.unit_health, "Health"), (t.unit_damage, "Damage"), (t.unit_speed, "Speed"), (t.unit_control, "Control"),
                               (t.unit_cost, "Cost")):
                field[uid] = _whole(u[key], "%s of unit %r" % (key, u["Name"]))
        per = num_units // num_groups
        for p in range(2):                              # _build_groups, everglades_env.py:145-156
            for g in range(num_groups):
                t.group_type[p][g] = names[UNIT_CLASSES[g % len(UNIT_CLASSES)]]
                t.group_size[p][g] = per if g < num_groups - 1 else num_units - per * (num_groups - 1)
    t.max_turns = int(max_turns)
    return t
