"""RNG contract of the build (TEST INFRASTRUCTURE -- lives under oracle/, never imported by the product).

The reference draws combat targets from numpy's *global* MT19937 stream
(`everglades_server/server.py:562`, plus two unobservable focus draws at `:205` and `:338`)
and never seeds it.  A per-lane GPU generator cannot replay that stream, so "identical seeds"
is defined against the reference with its entropy source injected:
every draw is a pure function of (seed, env_id, episode, turn, node, attacking player, attacker
ordinal).  This file is the plain-Python statement of that function; the C oracle
(`evg_oracle.c`), the HIP kernels (`csrc/evg_rng.h`) and the RNG proxy that `gen_golden.py`
plants into the imported reference all implement exactly this.

Generator: Philox4x32-10 (Salmon et al., SC'11), 64-bit key, 128-bit counter.

  key     = (seed & 0xffffffff, seed >> 32)
  counter = (block | domain << 28,  turn | node << 8 | player << 12 | group << 16,  episode,
             env_id & 0xffffffff)

  A block's four 32-bit words are used as EIGHT 16-bit draws ("halves"): half h of a block is bits
  16*(h & 1) .. 16*(h & 1) + 15 of word h >> 1.  A draw below n is (half * n) >> 16; n never exceeds 100
  here, so the deviation from uniform stays below n / 65536 = 0.15 % -- and one Philox block (the expensive
  part on the device) serves a whole 8-unit group instead of half of it.

  combat  (domain 0): the j-th alive unit (j = the loop index of server.py:561) of attacking group
                      `group` of `player` at `node` on `turn` uses half (j & 7) of block (j >> 3);
                      target index uid = (half * n) >> 16 with n = opposing alive units at the node.
                      (Keyed by group and unit, not by a node-wide ordinal, so that a GPU lane can
                      draw for its group without knowing the groups listed before it.)
  actions (domain 1): the on-device stand-in for agents/State_Machine/random_actions.py:38-46
                      (7 distinct groups of 12, 7 distinct nodes of 1..11): halves 0..6 of block 0 pick
                      the groups, halves 0..6 of block 1 the nodes; partial Fisher-Yates, see
                      `random_action_rows`.
  swarm   (domain 2): shuffle of the SwarmAgent attack list (swarm_agent.py:86-87), see agents.
  delay   (domain 3): the `random.random() > 0.68` coin of random_actions_delay.py: word 0 of block 0 as a
                      fraction of 2^32.
  explore (domain 4): DQNAgent.get_action of the Smart_State agent (agents/Smart_State/DQNAgent.py:130-173), one agent call =
                      (env, episode, turn = obs[0], player): halves 0..6 of block 0 pick the 7 distinct swarms of
                      np.random.choice(12, 7, replace=False) (partial Fisher-Yates as in `random_action_rows`), halves 0..6 of
                      block 1 the 7 directions of np.random.choice(5, 7, replace=True) as (half * 5) >> 16, and the two spare
                      halves make the epsilon coin: random.random() = (half 7 of block 0 << 16 | half 7 of block 1) / 2^32.
"""

M0, M1 = 0xD2511F53, 0xCD9E8D57
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = 0xFFFFFFFF

DOMAIN_COMBAT, DOMAIN_ACTION, DOMAIN_SWARM, DOMAIN_DELAY, DOMAIN_EXPLORE = 0, 1, 2, 3, 4


def philox4x32_10(ctr, key):
    c0, c1, c2, c3 = (int(x) & MASK for x in ctr)
    k0, k1 = (int(x) & MASK for x in key)
    for r in range(10):
        if r:
            k0 = (k0 + W0) & MASK
            k1 = (k1 + W1) & MASK
        p0 = M0 * c0
        p1 = M1 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & MASK, p1 & MASK, ((p0 >> 32) ^ c3 ^ k1) & MASK, p0 & MASK
    return c0, c1, c2, c3


def _key(seed):
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    return seed & MASK, seed >> 32


def _ctr(domain, block, turn, node, player, episode, env_id, group=0):
    return ((block & 0x0FFFFFFF) | (domain << 28),
            (turn & 0xFF) | ((node & 0xF) << 8) | ((player & 1) << 12) | ((group & 0xF) << 16),
            episode & MASK,
            env_id & MASK)


def halves(words):
    """The eight 16-bit draws of one block."""
    return [(words[h >> 1] >> (16 * (h & 1))) & 0xFFFF for h in range(8)]


def combat_draw(seed, env_id, episode, turn, node, player, group, j, n):
    """Target index in [0, n) for the j-th alive unit of attacking `group` (see module docstring)."""
    w = philox4x32_10(_ctr(DOMAIN_COMBAT, j >> 3, turn, node, player, episode, env_id, group), _key(seed))
    return (halves(w)[j & 7] * int(n)) >> 16


def random_action_rows(seed, env_id, episode, turn, player):
    """7 rows (group, node) -- distinct groups from 0..11, distinct nodes from 1..11 (player's own numbering)."""
    hg = halves(philox4x32_10(_ctr(DOMAIN_ACTION, 0, turn, 0, player, episode, env_id), _key(seed)))
    hn = halves(philox4x32_10(_ctr(DOMAIN_ACTION, 1, turn, 0, player, episode, env_id), _key(seed)))
    g = list(range(12))
    n = list(range(1, 12))
    for i in range(7):
        j = i + ((hg[i] * (12 - i)) >> 16)
        g[i], g[j] = g[j], g[i]
    for i in range(7):
        j = i + ((hn[i] * (11 - i)) >> 16)
        n[i], n[j] = n[j], n[i]
    return [(g[i], n[i]) for i in range(7)]


def swarm_shuffle(seed, env_id, episode, turn, player, items):
    """Fisher-Yates (numpy.random.shuffle order: i = n-1 .. 1, j = draw(i+1)) of `items`, 8 entries or fewer."""
    items = list(items)
    words = []
    for b in range(2):
        words += philox4x32_10(_ctr(DOMAIN_SWARM, b, turn, 0, player, episode, env_id), _key(seed))
    for w, i in enumerate(range(len(items) - 1, 0, -1)):
        j = (words[w] * (i + 1)) >> 32
        items[i], items[j] = items[j], items[i]
    return items


def delay_uniform(seed, env_id, episode, turn, player):
    """Stand-in for random.random() in random_actions_delay.get_action: a float in [0, 1)."""
    w = philox4x32_10(_ctr(DOMAIN_DELAY, 0, turn, 0, player, episode, env_id), _key(seed))
    return w[0] / 4294967296.0


def explore_draws(seed, env_id, episode, turn, player):
    """One DQNAgent.get_action call: (coin as a 32-bit integer -- random.random() = coin / 2^32 --, the 7 swarms, the 7 directions)."""
    h0 = halves(philox4x32_10(_ctr(DOMAIN_EXPLORE, 0, turn, 0, player, episode, env_id), _key(seed)))
    h1 = halves(philox4x32_10(_ctr(DOMAIN_EXPLORE, 1, turn, 0, player, episode, env_id), _key(seed)))
    g = list(range(12))
    for i in range(7):
        j = i + ((h0[i] * (12 - i)) >> 16)
        g[i], g[j] = g[j], g[i]
    return (h0[7] << 16) | h1[7], g[:7], [(h1[i] * 5) >> 16 for i in range(7)]


if __name__ == "__main__":
    # Random123 known-answer vectors for philox4x32-10
    assert philox4x32_10((0, 0, 0, 0), (0, 0)) == (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)
    assert philox4x32_10((MASK,) * 4, (MASK,) * 2) == (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)
    assert philox4x32_10((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0)) == \
        (0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1)
    print("philox KAT ok")
