// evg_kernels.hip -- hand-written HIP kernels for gfx950 (MI355X / CDNA4).
//
// One fused kernel per env-step replaces EvergladesEnv.step -> EvergladesGame.game_turn ->
// board_state/player_state of the reference (everglades_env.py:32-73, server.py:211-501).
//
// Mapping (DESIGN.md "Kernel design"): one env per LANE, one wavefront (64 envs) per workgroup.
// The game rules are short, branchy integer programs with almost no intra-env parallelism
// (7 ordered orders per player, ~1 contested node per turn), so a lane-per-env mapping executes
// 64 envs per instruction, where a wave-per-env mapping would leave most lanes idle.  Everything
// a lane indexes dynamically (its 24 group words, 11 node words, damage accumulators) lives in
// LDS as [index][lane] columns: lane-private and bank-conflict-free by construction
// (bank = lane mod 32, halves of the wave never collide).  The SoA state arrays are env-fastest,
// so every state load/store is a fully coalesced 256-byte wave access; the observation block of
// the wave's 64 envs (64 x 840 B, contiguous) is produced through an LDS transpose and written
// with 16-byte-per-lane coalesced stores.  Constant map/unit tables are staged in LDS.
// No MFMA: there is no dense contraction anywhere on this path.
#include <hip/hip_runtime.h>
#include "evg_device.h"
#include "evg_rng.h"

namespace evg {

struct CombatLds {
    uint32_t D[26][WG];                  // damage per target index of the half-item a lane is processing, 4 x u8 per word
    uint32_t SNAP[24][WG];               // pre-combat snapshot per group: bit31 fights | list-order prefix << 16 | node << 12 | alive mask
    uint32_t ACC[6][WG];                 // alive units of the fighting groups per (player, node): [p*3 + node/4], 8 bits per node
    uint32_t TURN[WG], EPI[WG];          // per-env scalars a lane needs when it works on another lane's env
    uint16_t W[WG * 2 * NN];             // work list of half-items: env lane | node << 6 | attacking player << 10
};

struct __align__(16) StepLds {
    uint32_t G[24][WG];                  // group words, lane-private columns
    uint32_t NW[12][WG];                 // node words by node ID
    union {                              // phases are disjoint in time (one wavefront per workgroup)
        CombatLds c;
        uint32_t A[24][WG];              // capture: per (player,node) points | units << 16
        uint32_t R[WG * REC_WORDS];      // observation records, [env][word], odd stride
    } u;
    uint64_t adj[12];
    double   defense[12];
    uint16_t desc[DESC_MAX];
};

// 12-input sorting network (tools/gen_sort12.py: 42 compare-exchanges, verified with the 0-1 principle)
#define EVG_SORT12_CES(CE) \
    CE(0, 1) CE(2, 3) CE(4, 5) CE(6, 7) CE(8, 9) CE(10, 11) CE(0, 2) CE(1, 3) \
    CE(4, 6) CE(5, 7) CE(8, 10) CE(9, 11) CE(1, 2) CE(5, 6) CE(9, 10) CE(0, 4) \
    CE(1, 5) CE(2, 6) CE(3, 7) CE(2, 4) CE(3, 5) CE(1, 2) CE(3, 4) CE(5, 6) \
    CE(9, 10) CE(0, 8) CE(1, 9) CE(2, 10) CE(3, 11) CE(4, 8) CE(5, 9) CE(6, 10) \
    CE(7, 11) CE(2, 4) CE(3, 5) CE(6, 8) CE(7, 9) CE(1, 2) CE(3, 4) CE(5, 6) \
    CE(7, 8) CE(9, 10)

// numpy's pairwise summation of a short contiguous float64 vector (np.sum at server.py:481):
// ((a0+a1)+(a2+a3))+((a4+a5)+(a6+a7)), then the tail sequentially.
__device__ __forceinline__ double np_sum8(const double* h) {
    return ((h[0] + h[1]) + (h[2] + h[3])) + ((h[4] + h[5]) + (h[6] + h[7]));
}

template <typename OT>
__device__ __forceinline__ void store_obs_vec(OT* dst, const int (&v)[16 / sizeof(OT)]);
template <>
__device__ __forceinline__ void store_obs_vec<float>(float* dst, const int (&v)[4]) {
    *reinterpret_cast<float4*>(dst) = make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}
template <>
__device__ __forceinline__ void store_obs_vec<double>(double* dst, const int (&v)[2]) {
    *reinterpret_cast<double2*>(dst) = make_double2((double)v[0], (double)v[1]);
}
template <>
__device__ __forceinline__ void store_obs_vec<int16_t>(int16_t* dst, const int (&v)[8]) {
    uint4 o;
    o.x = (uint32_t)(v[0] & 0xFFFF) | ((uint32_t)v[1] << 16);
    o.y = (uint32_t)(v[2] & 0xFFFF) | ((uint32_t)v[3] << 16);
    o.z = (uint32_t)(v[4] & 0xFFFF) | ((uint32_t)v[5] << 16);
    o.w = (uint32_t)(v[6] & 0xFFFF) | ((uint32_t)v[7] << 16);
    *reinterpret_cast<uint4*>(dst) = o;
}

// ---------------------------------------------------------------------------------------------
// fused env-step
// ---------------------------------------------------------------------------------------------
template <typename OT>
__global__ void __launch_bounds__(WG) evg_step_kernel(DevState S, StepIO io) {
    __shared__ StepLds L;
    const int lane = threadIdx.x;
    const int e0 = blockIdx.x * WG;
    const int nvalid = min(WG, S.N - e0);
    const bool valid = lane < nvalid;
    const int e = valid ? e0 + lane : e0;
    const size_t N = (size_t)S.N;
    const DevTables* __restrict__ T = S.T;

    // ---- stage constant tables in LDS
    if (lane < 12) {
        L.adj[lane] = T->adj_row[lane];
        L.defense[lane] = T->defense[lane];
    }
    for (int i = lane; i < DESC_MAX / 2; i += WG)
        reinterpret_cast<uint32_t*>(L.desc)[i] = reinterpret_cast<const uint32_t*>(T->obs_desc)[i];

    // ---- load state (coalesced, env fastest)
    const uint32_t envw = S.env[e];
    int turn = (int)(envw & 0xFFu);
    int status = (int)((envw >> 8) & 3u);
    uint32_t episode = S.episode[e];
    uint32_t st[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) st[j] = S.stamp[(size_t)j * N + e];
#pragma unroll
    for (int k = 0; k < 24; ++k) L.G[k][lane] = S.grp[(size_t)k * N + e];
#pragma unroll
    for (int n = 1; n <= NN; ++n) L.NW[n][lane] = S.node[(size_t)(n - 1) * N + e];
    __syncthreads();

    const bool observe_only = io.observe_only != 0;
    const bool frozen = status != 0;                    // finished, not auto-reset: repeat terminal outputs
    const bool play = valid && !frozen && !observe_only;
    const uint64_t p1nib = T->p1map_nib;
    const int max_turns = T->max_turns;

    const uint32_t abl = io.ablate;         // diagnostic only (EVG_ABLATE): skips phases to price them; 0 in production
    if (play) {
        turn += 1;                                                               // server.py:214
        // ---------------- orders (server.py:218-271)
        if (!(abl & 1u)) {
        const int4* ap = reinterpret_cast<const int4*>(io.actions) + (size_t)e * 7;
        int4 a[7];
#pragma unroll
        for (int j = 0; j < 7; ++j) a[j] = ap[j];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            uint32_t used = 0;
#pragma unroll
            for (int i = 0; i < NA; ++i) {
                const int r = p * NA + i;
                int gid = (r & 1) ? a[r >> 1].z : a[r >> 1].x;
                int nid = (r & 1) ? a[r >> 1].w : a[r >> 1].y;
                const bool ok = (uint32_t)gid < 12u && (uint32_t)nid < 12u;     // build-defined domain
                gid = ok ? gid : 0;
                nid = ok ? nid : 0;
                if (p == 1) nid = (int)((p1nib >> (4 * nid)) & 15u);             // :233-234
                const uint32_t w = L.G[p * 12 + gid][lane];
                const int loc = (int)(w & G_LOC_M);
                const uint32_t d = (uint32_t)((L.adj[loc] >> (4 * nid)) & 15u);  // test3 + distance, :245-250
                const bool accept = ok && !((used >> gid) & 1u) && ((w & G_MODE_M) >> G_MODE_S) != MODE_MOVING && d != 0;
                used |= (accept ? 1u : 0u) << gid;
                const uint32_t nw_ = (w & ~(G_DEST_M | G_DIST_M | G_MODE_M)) | ((uint32_t)nid << G_DEST_S) | (d << G_DIST_S) |
                                     (MODE_READY << G_MODE_S);                   // :267-270
                L.G[p * 12 + gid][lane] = accept ? nw_ : w;
            }
        }
        }

    }

    // ---------------- combat (server.py:503-654)
    // Stage 0 (lane = env): pre-combat snapshot.  A group fights at its node if it is alive and not moving (:525)
    // and the node holds such groups of both players (:539).  The reference walks node.groups[p] in list order,
    // which is (arrival stamp, gid) order; target index uid counts alive units along that
    // order, so each fighting group gets the prefix `base` of alive units listed before it.
    uint32_t g[24];
#pragma unroll
    for (int k = 0; k < 24; ++k) g[k] = L.G[k][lane];
    uint32_t occ0 = 0, occ1 = 0;
#pragma unroll
    for (int k = 0; k < 24; ++k) {
        const uint32_t w = g[k];
        const bool elig = (w & G_MASK_M) != 0 && ((w & G_MODE_M) >> G_MODE_S) != MODE_MOVING;
        const uint32_t bit = (elig ? 1u : 0u) << (w & G_LOC_M);
        if (k < 12) occ0 |= bit; else occ1 |= bit;
    }
    const uint32_t contested = (play && !(abl & 2u)) ? (occ0 & occ1) : 0u;
    if (__any(contested != 0)) {                      // wave-uniform: skip when none of the 64 envs fights
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            uint32_t key[12];
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                const uint32_t w = g[p * 12 + k];
                const uint32_t mask = (w & G_MASK_M) >> G_MASK_S;
                const uint32_t elig = (mask != 0 && ((w & G_MODE_M) >> G_MODE_S) != MODE_MOVING) ? 1u : 0u;
                const uint32_t stamp = (st[(p * 12 + k) >> 2] >> (8 * ((p * 12 + k) & 3))) & 0xFFu;
                key[k] = (stamp << 21) | ((uint32_t)k << 17) | ((w & G_LOC_M) << 13) | (mask << 1) | elig;
            }
#define EVG_CE(a, b) { const uint32_t lo_ = min(key[a], key[b]); key[b] = max(key[a], key[b]); key[a] = lo_; }
            EVG_SORT12_CES(EVG_CE)
#undef EVG_CE
            uint32_t a0 = 0, a1 = 0, a2 = 0;
#pragma unroll
            for (int i = 0; i < 12; ++i) {            // list order
                const uint32_t kk = key[i];
                const uint32_t gid = (kk >> 17) & 15u, loc = (kk >> 13) & 15u, mask = (kk >> 1) & 0xFFFu;
                const bool fights = (kk & 1u) && ((contested >> loc) & 1u);
                const uint32_t idx = loc >> 2, sh = (loc & 3u) * 8;
                const uint32_t cur = idx == 0 ? a0 : (idx == 1 ? a1 : a2);
                const uint32_t base = (cur >> sh) & 0xFFu;
                const uint32_t add = fights ? (uint32_t)__popc(mask) << sh : 0u;
                a0 += idx == 0 ? add : 0u;
                a1 += idx == 1 ? add : 0u;
                a2 += idx == 2 ? add : 0u;
                L.u.c.SNAP[p * 12 + gid][lane] = fights ? (0x80000000u | (base << 16) | (loc << 12) | mask) : 0u;
            }
            L.u.c.ACC[p * 3 + 0][lane] = a0;
            L.u.c.ACC[p * 3 + 1][lane] = a1;
            L.u.c.ACC[p * 3 + 2][lane] = a2;
        }
        L.u.c.TURN[lane] = (uint32_t)turn;
        L.u.c.EPI[lane] = episode;

        // Stage 1: wave-wide work list of half-items (env, node, attacking player), by prefix scan over lanes
        const int nitems = 2 * __popc(contested);
        int incl = nitems;
#pragma unroll
        for (int d = 1; d < WG; d <<= 1) {
            const int t = __shfl_up(incl, d);
            if (lane >= d) incl += t;
        }
        const int total = __shfl(incl, WG - 1);
        {
            int off = incl - nitems;
            uint32_t c = contested;
            while (c) {
                const uint32_t node = (uint32_t)__ffs(c) - 1u;
                c &= c - 1;
                L.u.c.W[off++] = (uint16_t)((uint32_t)lane | (node << 6));
                L.u.c.W[off++] = (uint16_t)((uint32_t)lane | (node << 6) | (1u << 10));
            }
        }
        __syncthreads();

        // Stage 2: lanes take half-items round-robin, whichever env they belong to (balanced over the wave).
        // Half-item (env, node, p): the alive units of p's fighting groups each draw one target among the
        // q-side's alive units at the node (:549-566); the summed damage is then applied to q (:573-644).
        // Both half-items of a node read only the snapshot, so the two directions are simultaneous.
        const uint64_t tn0 = T->type_nib[0], tn1 = T->type_nib[1];
        const uint32_t dmg_nib = T->damage_nib, armor_byte = T->armor_byte;
        for (int it = lane; it < total; it += WG) {
            const uint32_t item = L.u.c.W[it];
            const int EL = (int)(item & 63u), node = (int)((item >> 6) & 15u), p = (int)(item >> 10), q = 1 - p;
            const uint64_t tn_p = p ? tn1 : tn0, tn_q = q ? tn1 : tn0;
            const int tot_q = (int)((L.u.c.ACC[q * 3 + (node >> 2)][EL] >> ((node & 3) * 8)) & 0xFFu);
            uint32_t pm = 0, qm = 0;
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                const uint32_t sp = L.u.c.SNAP[p * 12 + k][EL], sq = L.u.c.SNAP[q * 12 + k][EL];
                pm |= ((sp >> 31) && (int)((sp >> 12) & 15u) == node) ? (1u << k) : 0u;
                qm |= ((sq >> 31) && (int)((sq >> 12) & 15u) == node) ? (1u << k) : 0u;
            }
            const int nwords = (tot_q + 3) >> 2;
            for (int i = 0; i < nwords; ++i) L.u.c.D[i][lane] = 0;
            const int turn_e = (int)L.u.c.TURN[EL];
            const uint32_t epi_e = L.u.c.EPI[EL], env_id_e = S.env_id_base + (uint32_t)(e0 + EL);
            // draw phase
            while (pm) {
                const int gid = __ffs(pm) - 1;
                pm &= pm - 1;
                const int cnt = __popc(L.u.c.SNAP[p * 12 + gid][EL] & 0xFFFu);
                const uint32_t type = (uint32_t)((tn_p >> (4 * gid)) & 15u);
                const uint32_t dmg = (dmg_nib >> (4 * type)) & 15u;
                for (int b = 0; b * 4 < cnt; ++b) {
                    const uint4 x = rng_block(S.seed_lo, S.seed_hi, env_id_e, epi_e, RNG_COMBAT, (uint32_t)b, turn_e, node, p, gid);
                    const uint32_t xs[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (b * 4 + i < cnt) {
                            const uint32_t uid = __umulhi(xs[i], (uint32_t)tot_q);                  // :562
                            L.u.c.D[uid >> 2][lane] += dmg << (8 * (uid & 3u));                    // :563-566
                        }
                    }
                }
            }
            // apply phase: uid-th alive unit of the snapshot, groups in list order (prefix `base`)
            const uint32_t nword = L.NW[node][EL];
            const int ctrl_by = (int)((nword >> 10) & 3u) - 1;
            const double ndef = L.defense[node];
            while (qm) {
                const int gid = __ffs(qm) - 1;
                qm &= qm - 1;
                const uint32_t sq = L.u.c.SNAP[q * 12 + gid][EL];
                const uint32_t mask = sq & 0xFFFu;
                int idx = (int)((sq >> 16) & 0xFFu);
                uint32_t dm[3] = {0, 0, 0}, any = 0;
#pragma unroll
                for (int sl = 0; sl < 12; ++sl) {
                    if ((mask >> sl) & 1u) {
                        const uint32_t d = (L.u.c.D[idx >> 2][lane] >> (8 * (idx & 3))) & 0xFFu;
                        ++idx;
                        dm[sl >> 2] |= d << (8 * (sl & 3));
                        any |= d;
                    }
                }
                if (any) {
                    double* row = S.health + (size_t)(e0 + EL) * (2 * NU) + q * NU + gid * 8;
                    double h[12];
                    const double2* r2 = reinterpret_cast<const double2*>(row);
#pragma unroll
                    for (int sl = 0; sl < 4; ++sl) { const double2 v = r2[sl]; h[2 * sl] = v.x; h[2 * sl + 1] = v.y; }
                    if (gid == 11) {
#pragma unroll
                        for (int sl = 4; sl < 6; ++sl) { const double2 v = r2[sl]; h[2 * sl] = v.x; h[2 * sl + 1] = v.y; }
                    } else {
                        h[8] = h[9] = h[10] = h[11] = 0.0;
                    }
                    const uint32_t type = (uint32_t)((tn_q >> (4 * gid)) & 15u);
                    const double armor = (double)((armor_byte >> (8 * type)) & 0xFFu);
                    const double denom = armor + (ctrl_by == q ? ndef : 0.0);                       // :592-597 (fort bonus dead)
                    uint32_t newmask = mask;
#pragma unroll
                    for (int sl = 0; sl < 12; ++sl) {
                        const uint32_t d = (dm[sl >> 2] >> (8 * (sl & 3))) & 0xFFu;
                        if (d) {
                            const double loss = (10.0 * (double)d) / denom;                           // :601
                            double hv = h[sl] - loss;                                                 // :609
                            if (hv <= 0.0) { hv = 0.0; newmask &= ~(1u << sl); }                      // :615-618
                            h[sl] = hv;
                        }
                    }
                    double2* w2 = reinterpret_cast<double2*>(row);
#pragma unroll
                    for (int sl = 0; sl < 4; ++sl) w2[sl] = make_double2(h[2 * sl], h[2 * sl + 1]);
                    double sum = np_sum8(h);
                    if (gid == 11) {
#pragma unroll
                        for (int sl = 4; sl < 6; ++sl) w2[sl] = make_double2(h[2 * sl], h[2 * sl + 1]);
                        sum = (((sum + h[8]) + h[9]) + h[10]) + h[11];
                    }
                    const int alive = __popc(newmask);
                    const uint32_t avg = alive ? (uint32_t)(int)(sum / (double)alive) : 0u;          // :491 truncation
                    const uint32_t w = L.G[q * 12 + gid][EL];
                    L.G[q * 12 + gid][EL] = (w & ~(G_MASK_M | G_AVG_M)) | (newmask << G_MASK_S) | (avg << G_AVG_S);
                }
            }
        }
        __syncthreads();
    }

    if (play) {
        // ---------------- movement (server.py:656-706)
        if (!(abl & 4u))
#pragma unroll
        for (int k = 0; k < 24; ++k) {
            uint32_t w = L.G[k][lane];
            const uint32_t mode = (w & G_MODE_M) >> G_MODE_S;
            if ((w & G_MASK_M) != 0) {                                            // not destroyed, :663
                if (mode == MODE_READY) {
                    w = (w & ~G_MODE_M) | (MODE_MOVING << G_MODE_S);              // :664-667
                } else if (mode == MODE_MOVING) {
                    const int type = T->group_type[k / 12][k % 12];
                    const int nd = (int)((w & G_DIST_M) >> G_DIST_S) - T->unit_speed[type];   // :671
                    if (nd <= 0) {                                                // arrived, :678-695
                        const uint32_t dest = (w & G_DEST_M) >> G_DEST_S;
                        w = (w & ~(G_LOC_M | G_DEST_M | G_DIST_M | G_MODE_M)) | dest;
                        st[k >> 2] = (st[k >> 2] & ~(0xFFu << (8 * (k & 3)))) | ((uint32_t)turn << (8 * (k & 3)));
                    } else {
                        w = (w & ~G_DIST_M) | ((uint32_t)nd << G_DIST_S);
                    }
                }
                L.G[k][lane] = w;
            }
        }
    }

    // ---------------- per-node aggregates of the (post-movement) state
#pragma unroll
    for (int i = 0; i < 24; ++i) L.u.A[i][lane] = 0;
    int unit_score[2] = {0, 0}, units_alive[2] = {0, 0};
    uint32_t gw[24];
#pragma unroll
    for (int k = 0; k < 24; ++k) {
        const uint32_t w = L.G[k][lane];
        gw[k] = w;
        const int cnt = __popc(w & G_MASK_M);
        if (cnt) {
            const int type = T->group_type[k / 12][k % 12];
            const bool elig = ((w & G_MODE_M) >> G_MODE_S) != MODE_MOVING;                      // :720
            const uint32_t add = (elig ? (uint32_t)(cnt * T->unit_control[type]) : 0u) | ((uint32_t)cnt << 16);
            L.u.A[(k / 12) * 12 + (w & G_LOC_M)][lane] += add;
            unit_score[k / 12] += cnt * T->unit_cost[type];                                     // :315-317
            units_alive[k / 12] += cnt;
        }
    }

    // ---------------- capture (server.py:708-767), scores and status (server.py:281-348)
    int score[2] = {unit_score[0], unit_score[1]};
    bool base_captured = false;
    int cs_arr[12];
    uint32_t units_w[12][2];
#pragma unroll
    for (int n = 1; n <= NN; ++n) {
        const uint32_t a0 = L.u.A[n][lane], a1 = L.u.A[12 + n][lane];
        uint32_t nword = L.NW[n][lane];
        int cs = (int)(nword & 0x3FFu) - 512;
        int cb = (int)((nword >> 10) & 3u) - 1;
        const int cp = T->control_points[n];
        if (play) {
            const int pts0 = (int)(a0 & 0xFFFFu), pts1 = (int)(a1 & 0xFFFFu);
            const bool c0 = pts0 > 0, c1 = pts1 > 0;                               // ctr >= 1 (control >= 1)
            if (c0 != c1) {                                                        // exactly one controller, :729
                const int pid = c0 ? 0 : 1;
                if (abs(cs) < cp || pid != cb) {                                   // :731-732
                    const int pxer = pid == 0 ? 1 : -1;
                    const int old_sign = cs < 0;
                    cs += (pid == 0 ? pts0 : pts1) * pxer;                         // :748 (turn > 0 here)
                    const bool neutralize = old_sign != (cs < 0);                  // :747-750
                    if (abs(cs) >= cp) { cs = cp * pxer; cb = pid; }               // :763-765
                    if (cb != -1 && neutralize) cb = -1;                           // :766-767
                    nword = (uint32_t)(cs + 512) | ((uint32_t)(cb + 1) << 10);
                    L.NW[n][lane] = nword;
                }
            }
        }
        const int ts = T->team_start[n];
        if (ts != -1 && cb != -1 && cb != ts) {                                    // :299-304
            base_captured = true;
            score[cb == 0 ? 0 : 1] += 1000;
        }
        if (cs != 0) {                                                             // :305-310
            const int pts = abs(cs) == cp ? 2 * cp : abs(cs);
            if (cs > 0) score[0] += pts; else score[1] += pts;
        }
        cs_arr[n] = cs;
        units_w[n][0] = a0 >> 16;
        units_w[n][1] = a1 >> 16;
    }
    if (play) {
        if (turn >= max_turns) status = EVG_TIME_EXPIRED;                          // :321
        else if (units_alive[0] + units_alive[1] == 0) status = EVG_ANNIHILATION;  // :324
        else if (base_captured) status = EVG_BASE_CAPTURE;                         // :327
    }

    // ---------------- reward / done / winner (everglades_env.py:37-61, evaluate.py:155-160)
    float rew0, rew1;
    int winner = EVG_WINNER_NONE;
    const bool done = status != 0;
    if (done) {
        winner = score[0] > score[1] ? EVG_WINNER_P0 : (score[1] > score[0] ? EVG_WINNER_P1 : EVG_WINNER_TIE);
        rew0 = score[0] > score[1] ? 1.f : 0.f;
        rew1 = score[1] > score[0] ? 1.f : (score[0] > score[1] ? -1.f : 0.f);
    } else {
        rew0 = (float)((double)score[0] / (double)EVG_MAX_SCORE);
        rew1 = (float)((double)score[1] / (double)EVG_MAX_SCORE);
    }
    if (valid && !observe_only) {
        reinterpret_cast<float2*>(io.reward)[e] = make_float2(rew0, rew1);
        io.done[e] = done ? 1 : 0;
        if (io.winner) io.winner[e] = (int8_t)winner;
        if (io.scores) reinterpret_cast<int2*>(io.scores)[e] = make_int2(score[0], score[1]);
        if (io.status) io.status[e] = (uint8_t)status;
    }

    // ---------------- episode bookkeeping + auto-reset
    bool do_reset = false;
    if (play) {
        float r0 = S.ep_ret[e] + rew0, r1 = S.ep_ret[N + e] + rew1;
        if (done) {
            reinterpret_cast<float2*>(S.fin_ret)[e] = make_float2(r0, r1);
            S.fin_len[e] = turn;
            S.fin_win[e] = (int8_t)winner;
            if (S.auto_reset) { do_reset = true; r0 = r1 = 0.f; }
        }
        S.ep_ret[e] = r0;
        S.ep_ret[N + e] = r1;
    }
    {
        const bool fin = play && done;
        const uint64_t mf = __ballot(fin);
        if (mf) {
            const uint64_t m0 = __ballot(fin && winner == EVG_WINNER_P0), m1 = __ballot(fin && winner == EVG_WINNER_P1);
            if (lane == 0) {
                const int nf = __popcll(mf), n0 = __popcll(m0), n1 = __popcll(m1);
                atomicAdd(&S.totals[0], (unsigned long long)nf);
                if (n0) atomicAdd(&S.totals[1], (unsigned long long)n0);
                if (n1) atomicAdd(&S.totals[2], (unsigned long long)n1);
                if (nf - n0 - n1) atomicAdd(&S.totals[3], (unsigned long long)(nf - n0 - n1));
            }
        }
    }

    // ---------------- observation record (board_state :382-455, player_state :457-501)
    __syncthreads();        // all lanes have read their A columns before anyone overwrites the union with records
    uint32_t* rec = &L.u.R[lane * REC_WORDS];
    if (do_reset) {
        // new episode: state of game_init (server.py:133-209), observation of everglades_env.py:75-116
        turn = 0; status = 0; episode += 1u;
#pragma unroll
        for (int j = 0; j < 6; ++j) st[j] = 0;
#pragma unroll
        for (int k = 0; k < 24; ++k) gw[k] = T->init_grp[k];
#pragma unroll
        for (int n = 1; n <= NN; ++n) L.NW[n][lane] = T->init_node[n];
#pragma unroll
        for (int wd = 0; wd < REC_WORDS; ++wd) rec[wd] = T->reset_rec[wd];
    } else {
        // fields: 0 turn | 1..11 controlState | 12..22 p1 units at node | 23..33 p0 units at node |
        //         34 + 4*(p*12+k) + {0 location (own numbering), 1 avg health, 2 moving, 3 alive}
        int f[34];
        f[0] = turn;
#pragma unroll
        for (int n = 1; n <= NN; ++n) { f[n] = cs_arr[n]; f[11 + n] = (int)units_w[n][1]; f[22 + n] = (int)units_w[n][0]; }
#pragma unroll
        for (int j = 0; j < 17; ++j) rec[j] = ((uint32_t)f[2 * j] & 0xFFFFu) | ((uint32_t)f[2 * j + 1] << 16);
#pragma unroll
        for (int k = 0; k < 24; ++k) {
            const uint32_t w = gw[k];
            uint32_t loc = w & G_LOC_M;
            if (k >= 12) loc = (uint32_t)((p1nib >> (4 * loc)) & 15u);             // :485-486
            const uint32_t avg = (w & G_AVG_M) >> G_AVG_S;
            const uint32_t moving = ((w & G_MODE_M) >> G_MODE_S) == MODE_MOVING ? 1u : 0u;
            const uint32_t alive = __popc(w & G_MASK_M);
            rec[17 + 2 * k] = loc | (avg << 16);
            rec[17 + 2 * k + 1] = moving | (alive << 16);
        }
    }

    // ---------------- store state (coalesced)
    if (valid && !observe_only && (play || do_reset) && !(abl & 32u)) {
#pragma unroll
        for (int k = 0; k < 24; ++k) S.grp[(size_t)k * N + e] = gw[k];
#pragma unroll
        for (int j = 0; j < 6; ++j) S.stamp[(size_t)j * N + e] = st[j];
#pragma unroll
        for (int n = 1; n <= NN; ++n) S.node[(size_t)(n - 1) * N + e] = (uint16_t)L.NW[n][lane];
        S.env[e] = (uint32_t)turn | ((uint32_t)status << 8);
        if (do_reset) S.episode[e] = episode;
    }
    __syncthreads();        // records visible to the whole wave; combat's health stores drained

    // ---------------- observation write-out: 64 envs x 2 x 105 elements, 16 bytes per lane, coalesced
    if (io.obs && !(abl & 16u)) {
        constexpr int EP = 16 / (int)sizeof(OT);          // elements per 16-byte vector
        constexpr int U = EP == 2 ? 1 : (EP == 4 ? 2 : 4); // envs per descriptor unit: U*210 % EP == 0
        constexpr int VPU = U * 2 * OBS / EP;              // = 105 vectors per unit
        const int nvec = (WG / U) * VPU;
        const int limit = nvalid * 2 * OBS;
        OT* out = reinterpret_cast<OT*>(io.obs) + (size_t)e0 * (2 * OBS);
#pragma unroll 4
        for (int v = lane; v < nvec; v += WG) {
            const int unit = v / VPU, r = v - unit * VPU;
            int vals[EP];
#pragma unroll
            for (int j = 0; j < EP; ++j) {
                const uint32_t d = L.desc[r * EP + j];
                const int el = unit * U + (int)((d >> 8) & 3u);
                const uint32_t fld = d & 0xFFu;
                const uint32_t word = L.u.R[el * REC_WORDS + (fld >> 1)];
                const int fv = (int)(int16_t)(word >> (16 * (fld & 1u)));
                vals[j] = (d & 0x8000u) ? (int)(d & 0x7FFFu) : fv;
            }
            const int elem0 = v * EP;
            if (elem0 + EP <= limit) {
                store_obs_vec<OT>(out + elem0, vals);
            } else {
#pragma unroll
                for (int j = 0; j < EP; ++j)
                    if (elem0 + j < limit) out[elem0 + j] = (OT)vals[j];
            }
        }
    }

    // ---------------- health of envs that start a new episode: 1600 B each, written by the whole wave
    uint64_t rm = __ballot(do_reset);
    while (rm) {
        const int l = __ffsll((unsigned long long)rm) - 1;
        rm &= rm - 1;
        double2* dst = reinterpret_cast<double2*>(S.health + (size_t)(e0 + l) * (2 * NU));
        for (int i = lane; i < NU; i += WG) dst[i] = make_double2(100.0, 100.0);
    }
}

// ---------------------------------------------------------------------------------------------
// reset (everglades_env.py:75-116 -> server.py:133-209): masked, per env
// ---------------------------------------------------------------------------------------------
template <typename OT>
__global__ void __launch_bounds__(WG) evg_reset_kernel(DevState S, const uint8_t* mask, void* obs) {
    const int lane = threadIdx.x;
    const int e0 = blockIdx.x * WG;
    const int e = e0 + lane;
    const size_t N = (size_t)S.N;
    const DevTables* __restrict__ T = S.T;
    const bool sel = e < S.N && (mask == nullptr || mask[e] != 0);
    if (sel) {
#pragma unroll
        for (int k = 0; k < 24; ++k) S.grp[(size_t)k * N + e] = T->init_grp[k];
#pragma unroll
        for (int j = 0; j < 6; ++j) S.stamp[(size_t)j * N + e] = 0;
#pragma unroll
        for (int n = 1; n <= NN; ++n) S.node[(size_t)(n - 1) * N + e] = (uint16_t)T->init_node[n];
        S.env[e] = 0;
        S.episode[e] += 1u;                 // 0xFFFFFFFF at create -> episode 0 on the first reset
        S.ep_ret[e] = 0.f;
        S.ep_ret[N + e] = 0.f;
    }
    uint64_t rm = __ballot(sel);
    while (rm) {
        const int l = __ffsll((unsigned long long)rm) - 1;
        rm &= rm - 1;
        double2* dst = reinterpret_cast<double2*>(S.health + (size_t)(e0 + l) * (2 * NU));
        for (int i = lane; i < NU; i += WG) dst[i] = make_double2(100.0, 100.0);
        if (obs) {
            OT* o = reinterpret_cast<OT*>(obs) + (size_t)(e0 + l) * (2 * OBS);
            for (int i = lane; i < 2 * OBS; i += WG) o[i] = (OT)T->reset_obs[i];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// random_actions stand-in (agents/State_Machine/random_actions.py:38-46): 7 distinct groups of 12,
// 7 distinct nodes of 1..11 per player, partial Fisher-Yates on nibble-packed permutations
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) evg_random_actions_kernel(DevState S, int32_t* actions) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= 2 * S.N) return;
    const int e = idx >> 1, p = idx & 1;
    const int turn = (int)(S.env[e] & 0xFFu);
    const uint32_t episode = S.episode[e];
    const uint32_t env_id = S.env_id_base + (uint32_t)e;
    uint32_t w[16];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const uint4 x = rng_block(S.seed_lo, S.seed_hi, env_id, episode, RNG_ACTION, (uint32_t)b, turn, 0, p, 0);
        w[4 * b] = x.x; w[4 * b + 1] = x.y; w[4 * b + 2] = x.z; w[4 * b + 3] = x.w;
    }
    uint64_t gp = 0xBA9876543210ull;      // nibble i = i
    uint64_t np_ = 0xBA987654321ull;      // nibble i = i + 1
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int j = i + (int)__umulhi(w[i], (uint32_t)(12 - i));
        const uint64_t x = ((gp >> (4 * i)) ^ (gp >> (4 * j))) & 15ull;
        gp ^= (x << (4 * i)) ^ (x << (4 * j));
    }
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int j = i + (int)__umulhi(w[8 + i], (uint32_t)(11 - i));
        const uint64_t x = ((np_ >> (4 * i)) ^ (np_ >> (4 * j))) & 15ull;
        np_ ^= (x << (4 * i)) ^ (x << (4 * j));
    }
    int2* out = reinterpret_cast<int2*>(actions) + (size_t)idx * NA;
#pragma unroll
    for (int i = 0; i < NA; ++i) out[i] = make_int2((int)((gp >> (4 * i)) & 15ull), (int)((np_ >> (4 * i)) & 15ull));
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
int launch_step(const DevState& S, const StepIO& io, int obs_dtype, void* stream) {
    const dim3 grid((S.N + WG - 1) / WG), block(WG);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    switch (obs_dtype) {
        case EVG_OBS_F32: hipLaunchKernelGGL(evg_step_kernel<float>, grid, block, 0, s, S, io); break;
        case EVG_OBS_F64: hipLaunchKernelGGL(evg_step_kernel<double>, grid, block, 0, s, S, io); break;
        case EVG_OBS_I16: hipLaunchKernelGGL(evg_step_kernel<int16_t>, grid, block, 0, s, S, io); break;
        default: return -1;
    }
    return (int)hipGetLastError();
}

int launch_reset(const DevState& S, const uint8_t* mask, void* obs, int obs_dtype, void* stream) {
    const dim3 grid((S.N + WG - 1) / WG), block(WG);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    switch (obs_dtype) {
        case EVG_OBS_F32: hipLaunchKernelGGL(evg_reset_kernel<float>, grid, block, 0, s, S, mask, obs); break;
        case EVG_OBS_F64: hipLaunchKernelGGL(evg_reset_kernel<double>, grid, block, 0, s, S, mask, obs); break;
        case EVG_OBS_I16: hipLaunchKernelGGL(evg_reset_kernel<int16_t>, grid, block, 0, s, S, mask, obs); break;
        default: return -1;
    }
    return (int)hipGetLastError();
}

int launch_random_actions(const DevState& S, int32_t* actions, void* stream) {
    const dim3 grid((2 * S.N + 255) / 256), block(256);
    hipLaunchKernelGGL(evg_random_actions_kernel, grid, block, 0, reinterpret_cast<hipStream_t>(stream), S, actions);
    return (int)hipGetLastError();
}

}  // namespace evg
