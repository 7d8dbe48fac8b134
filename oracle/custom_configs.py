"""Non-default configurations in the reference's JSON schema (TEST INFRASTRUCTURE; the build's own data, not reference text).

`EvergladesEnv.reset(map_file=, unit_file=, config_dir=)` (everglades_env.py:75-106 -> server.py:24-38, board_init :40-100,
unitTypes_init :103-131) parses two JSON files at every reset.  The variants below change every table those files carry, so that
oracle/gen_golden.py can have the imported reference play on them (fixtures tests/golden/custom_*.npz) and the runtime-table path
of the build (tables_from_json -> evg_create, incl. the tabulated-reciprocal / true-division choice) is pinned by the reference
and not only by the oracle.

  varA  bases where DemoMap has them; directed distances in 2..7 incl. odd ones, one edge with different lengths in the two
        directions (2->3: 3, 3->2: 4), one one-way edge (6->5); control points 45..511; non-dyadic StructureDefense (1.3, 2.1, ...),
        one 0 and one integer; DEFENSE / OBSERVE moved, one node with both, one unknown resource string; unit file in another
        order (unit ids follow the file order, server.py:113-130) with every stat changed, speed 3 (arrival mid-edge).
  varB  bases moved to nodes 4 and 8 (not a mirror pair of the reference's hard-coded p1_node_map, server.py:89: player 1 sees
        its base as node 2 and the enemy's as node 10); four unit types, the first one unused by the army; mixed-case names.
  varC  DemoMap + stock units; the reference object's `p1_node_map` attribute is replaced after construction by a board flip that
        is not its own inverse (an EDIT of the reference object, like edit_annihilation; the unmodified server has one map only).
"""
import copy
import json

P1MAP = [0, 11, 8, 9, 10, 5, 6, 7, 2, 3, 4, 1]            # server.py:89
P1MAP_C = [0, 11, 8, 9, 10, 6, 7, 5, 2, 3, 4, 1]          # varC: 5 -> 6 -> 7 -> 5


def _node(i, conn, cp, dfn, res, team=-1):
    return {"Connections": [{"ConnectedID": b, "Distance": d} for b, d in conn.items()], "ControlPoints": cp, "ID": i,
            "Radius": 1.0 + i / 1000.0, "Resource": list(res), "StructureDefense": dfn, "X": (i - 6) // 3, "Y": (i - 6) % 3 - 1,
            "TeamStart": team}


def _map(name, conn, cp, dfn, res, bases):
    return {"__type": "Map:#Everglades_MapJSONDef", "MapName": name,
            "nodes": [_node(i, conn[i], cp[i], dfn[i], res.get(i, []), bases.get(i, -1)) for i in range(1, 12)]}


def _units(rows):
    return {"__type": "Units", "units": [dict(Name=n, Health=h, Damage=d, Speed=s, Control=c, Cost=k) for n, h, d, s, c, k in rows]}


CONN_A = {1: {2: 5, 4: 7}, 2: {1: 5, 3: 3, 5: 6}, 3: {2: 4, 4: 2, 5: 4, 6: 5, 7: 4}, 4: {1: 7, 3: 2, 7: 6},
          5: {2: 6, 3: 4, 8: 3, 9: 7}, 6: {3: 5, 9: 2, 5: 2}, 7: {3: 4, 4: 6, 9: 3, 10: 5}, 8: {5: 3, 9: 6, 11: 4},
          9: {5: 7, 6: 2, 7: 3, 8: 6, 10: 2}, 10: {7: 5, 9: 2, 11: 7}, 11: {8: 4, 10: 7}}
MAP_A = _map("VariantA", CONN_A,
             cp={1: 350, 2: 60, 3: 130, 4: 45, 5: 100, 6: 511, 7: 77, 8: 75, 9: 110, 10: 50, 11: 420},
             dfn={1: 1.3, 2: 2.1, 3: 0.7, 4: 1.9, 5: 1.15, 6: 3.3, 7: 0, 8: 1.7, 9: 2.45, 10: 2, 11: 1.1},
             res={2: ["DEFENSE"], 3: ["OBSERVE"], 5: ["OBSERVE", "DEFENSE"], 6: ["DEFENSE"], 7: ["OBSERVE"], 9: ["SUPPLY"], 10: ["OBSERVE"]},
             bases={1: 0, 11: 1})
UNITS_A = _units([("Striker", 2, 3, 3, 1, 2), ("Tank", 5, 1, 1, 2, 3), ("Controller", 3, 2, 2, 3, 1)])

CONN_B = {1: {2: 3, 4: 2}, 2: {1: 3, 3: 7, 5: 5}, 3: {2: 7, 4: 5, 5: 2, 6: 4, 7: 3}, 4: {1: 2, 3: 5, 7: 7},
          5: {2: 5, 3: 2, 8: 6, 9: 4}, 6: {3: 4, 9: 6}, 7: {3: 3, 4: 7, 9: 5, 10: 2}, 8: {5: 6, 9: 3, 11: 5},
          9: {5: 4, 6: 6, 7: 5, 8: 3, 10: 7}, 10: {7: 2, 9: 7, 11: 3}, 11: {8: 5, 10: 3}}
MAP_B = _map("VariantB", CONN_B,
             cp={1: 90, 2: 150, 3: 64, 4: 300, 5: 33, 6: 200, 7: 125, 8: 275, 9: 81, 10: 140, 11: 70},
             dfn={1: 0.6, 2: 1.45, 3: 2.7, 4: 1.25, 5: 0.33, 6: 1.8, 7: 2.2, 8: 0.95, 9: 3.1, 10: 1.05, 11: 4},
             res={1: ["OBSERVE"], 3: ["DEFENSE"], 4: ["OBSERVE"], 6: ["OBSERVE"], 8: ["DEFENSE"], 9: ["DEFENSE", "OBSERVE"], 11: ["DEFENSE"]},
             bases={4: 0, 8: 1})
UNITS_B = _units([("Scout", 1, 1, 4, 1, 1), ("controller", 4, 1, 3, 2, 2), ("TANK", 7, 2, 2, 1, 4), ("Striker", 3, 2, 1, 3, 3)])

VARIANTS = {
    "varA": dict(map=MAP_A, units=UNITS_A, p1_node_map=None, brawl_node=9, rush_target={0: 11, 1: 11}),
    "varB": dict(map=MAP_B, units=UNITS_B, p1_node_map=None, brawl_node=3, rush_target={0: 8, 1: 10}),
    "varC": dict(map=None, units=None, p1_node_map=P1MAP_C, brawl_node=6, rush_target={0: 11, 1: 11}),
}


def json_text(obj):
    return json.dumps(obj, indent=1, sort_keys=True)


def adjacency(map_obj):
    """{node id: {neighbour id: distance}} of a map object (directed, as the file lists it)."""
    return {int(n["ID"]): {int(c["ConnectedID"]): int(c["Distance"]) for c in n["Connections"]} for n in map_obj["nodes"]}


def own_view_hops(adj, own_to_real, real_to_own):
    """Next hop [src][dst] (own numbering of one player) along shortest directed paths of `adj` (real numbering).  An order's node id goes
    through own_to_real (server.py:233-234), a location is shown through real_to_own (server.py:485-486; the reference uses ONE table for both)."""
    import heapq
    hop = [[0] * 12 for _ in range(12)]
    for src in range(1, 12):
        dist, prev, pq = {src: 0}, {}, [(0, src)]
        while pq:
            d, u = heapq.heappop(pq)
            if d > dist.get(u, 1e9):
                continue
            for v, w in adj[u].items():
                if d + w < dist.get(v, 1e9):
                    dist[v], prev[v] = d + w, u
                    heapq.heappush(pq, (d + w, v))
        for dst in range(1, 12):
            if dst == src or dst not in prev:
                continue
            v = dst
            while prev[v] != src:
                v = prev[v]
            hop[real_to_own[src]][real_to_own[dst]] = own_to_real.index(v) if v in own_to_real[1:] else 0
    return hop


def variant_objects(name, demo_map, demo_units):
    """(map object, unit object, p1 map) of a variant; None entries fall back to the given DemoMap / UnitDefinitions objects."""
    v = VARIANTS[name]
    return (copy.deepcopy(v["map"] or demo_map), copy.deepcopy(v["units"] or demo_units), list(v["p1_node_map"] or P1MAP))


def random_config(rng):
    """A random configuration inside the domain include/evg.h states (numpy Generator `rng`): a random DIRECTED graph over the 11 nodes (1-5 outbound edges per
    node, distances 1..7, possibly disconnected), control points 1..511, StructureDefense with up to two decimals incl. 0, random resource sets, bases on any
    two nodes, 3-4 unit types in random file order with random stats (an army's total damage <= 255).  Returns (map object, unit object) in the reference's
    JSON schema.  Used by tests/test_oracle_vs_live_reference.py (oracle against the live reference) and by the GPU parity tests (HIP against the oracle)."""
    import numpy as np
    nodes = []
    bases = rng.choice(np.arange(1, 12), 2, replace=False)
    for i in range(1, 12):
        others = [j for j in range(1, 12) if j != i]
        outs = rng.choice(others, int(rng.integers(1, 6)), replace=False)
        res = [r for r in ("DEFENSE", "OBSERVE", "SUPPLY") if rng.random() < 0.3]
        nodes.append({"ID": i, "Radius": 1.0, "X": 0, "Y": 0, "Resource": res, "ControlPoints": int(rng.integers(1, 512)),
                      "StructureDefense": float(rng.choice([0, 0.05, 0.5, 1, 1.3, 1.75, 2.45, 3.3, round(float(rng.random() * 4), 2)])),
                      "TeamStart": 0 if i == bases[0] else (1 if i == bases[1] else -1),
                      "Connections": [{"ConnectedID": int(j), "Distance": int(rng.integers(1, 8))} for j in outs]})
    names = ["Tank", "Controller", "Striker"] + (["Scout"] if rng.random() < 0.5 else [])
    while True:
        order = list(rng.permutation(len(names)))
        units = [dict(Name=names[k], Health=int(rng.integers(1, 10)), Damage=int(rng.integers(1, 4)), Speed=int(rng.integers(1, 8)),
                      Control=int(rng.integers(1, 16)), Cost=int(rng.integers(1, 16))) for k in order]
        dmg = {u["Name"].lower(): u["Damage"] for u in units}
        if 32 * dmg["controller"] + 32 * dmg["striker"] + 36 * dmg["tank"] <= 255:
            break
    return {"MapName": "fuzz", "nodes": nodes}, {"units": units}
