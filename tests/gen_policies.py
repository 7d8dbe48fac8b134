"""Vectorised host-side action policies used as INPUT generators by the parity tests (numpy, any N).
They read observations [N, 2, 105] in each player's own numbering, like the reference's scripted agents."""
import heapq

import numpy as np

ADJ = {1: {2: 6, 4: 6}, 2: {1: 6, 3: 4, 5: 4}, 3: {2: 4, 4: 4, 5: 6, 6: 3, 7: 6}, 4: {1: 6, 3: 4, 7: 4},
       5: {2: 4, 3: 6, 8: 4, 9: 6}, 6: {3: 3, 9: 3}, 7: {3: 6, 4: 4, 9: 6, 10: 4}, 8: {5: 4, 9: 4, 11: 6},
       9: {5: 6, 6: 3, 7: 6, 8: 4, 10: 4}, 10: {7: 4, 9: 4, 11: 6}, 11: {8: 6, 10: 6}}


def _hops():
    hop = np.zeros((12, 12), np.int32)
    for src in ADJ:
        dist, prev, pq = {src: 0}, {}, [(0, src)]
        while pq:
            d, u = heapq.heappop(pq)
            if d > dist.get(u, 1e9):
                continue
            for v, w in ADJ[u].items():
                if d + w < dist.get(v, 1e9):
                    dist[v], prev[v] = d + w, u
                    heapq.heappush(pq, (d + w, v))
        for dst in ADJ:
            v = dst
            while v != src and prev[v] != src:
                v = prev[v]
            hop[src, dst] = v
    return hop


HOP = _hops()


def march_actions(obs, target, rot):
    """Each idle, alive group not yet at `target` is sent to the next hop; the first 7 (rotated by `rot`) get a row."""
    N = obs.shape[0]
    out = np.zeros((N, 2, 7, 2), np.int32)
    for p in range(2):
        loc = obs[:, p, 45::5].astype(np.int64)
        mov = obs[:, p, 48::5].astype(np.int64)
        alive = obs[:, p, 49::5].astype(np.int64)
        order = (np.arange(12) + rot) % 12
        want = (mov[:, order] == 0) & (alive[:, order] > 0) & (loc[:, order] != target)
        rank = np.cumsum(want, axis=1) - 1
        for r in range(7):
            sel = want & (rank == r)
            has = sel.any(axis=1)
            col = sel.argmax(axis=1)
            g = order[col]
            nxt = HOP[np.clip(loc[np.arange(N), g], 1, 11), target]
            out[:, p, r, 0] = np.where(has, g, 0)
            out[:, p, r, 1] = np.where(has, nxt, 0)
    return out


def wild_actions(N, rng):
    a = rng.integers(-2, 14, size=(N, 2, 7, 2)).astype(np.int32)      # includes out-of-domain ids
    dup = rng.random((N, 2)) < 0.5
    a[..., 1:3, 0] = np.where(dup[..., None], a[..., 0:1, 0], a[..., 1:3, 0])   # duplicate group ids
    return a


def policy_actions(policy, obs, t, rng):
    if policy == "rush":
        return march_actions(obs, 11, t)
    if policy == "brawl":
        return march_actions(obs, 6, 0)
    if policy == "wild":
        return wild_actions(obs.shape[0], rng)
    raise KeyError(policy)


class NumpyGlobalRandomAgent(object):
    """What agents/State_Machine/random_actions.py:38-46 does, as a test stand-in (the reference is not importable on the
    GPU box): two draws without replacement from numpy's GLOBAL generator per call -- the same stream the reference's
    server draws combat targets from, so agent and server draws interleave.  Pinned by the action streams recorded in
    tests/golden/config1_stock.npz from the reference's own agent class."""

    def get_action(self, obs):
        a = np.zeros((7, 2))
        a[:, 0] = np.random.choice(12, 7, replace=False)
        a[:, 1] = np.random.choice(list(range(1, 12)), 7, replace=False)
        return a


def sighting_rows(records, rank, types):
    """Per-group sightings of one observer (int8 [12][4] = seen, node id, destination key, count; oracle / device layout) ->
    the rows of the reference's `opp_k` dict flattened in ITS iteration order (tests/golden `sight`: node id, destination
    key, unit type id, count; -2 rows after the last): nodes ascending, destination keys in order of first appearance along
    the node's group list, groups in list order.  rank[g] = position of opposing group g in its node's list, types[g] its
    unit type id."""
    out = np.full((12, 4), -2, np.int8)
    by_node = {}
    for g in sorted((g for g in range(12) if records[g][0]), key=lambda g: (int(records[g][1]), int(rank[g]))):
        by_node.setdefault(int(records[g][1]), {}).setdefault(int(records[g][2]), []).append(g)
    i = 0
    for nid in sorted(by_node):
        for dst, gs in by_node[nid].items():
            for g in gs:
                out[i] = (nid, dst, int(types[g]), int(records[g][3]))
                i += 1
    return out
