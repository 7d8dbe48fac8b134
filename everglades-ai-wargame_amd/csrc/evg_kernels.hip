// evg_kernels.hip -- hand-written HIP kernels for gfx950 (MI355X / CDNA4).
//
// One fused kernel per env-step replaces EvergladesEnv.step -> EvergladesGame.game_turn ->
// board_state/player_state of the reference (everglades_env.py:32-73, server.py:211-501).
//
// Mapping: lane = (env slot, player); one wavefront = 32 envs per workgroup.
// The game rules are short, branchy integer programs with almost no parallelism inside one env
// (7 orders per player, ~1 contested node per turn), so envs are spread over LANES: one instruction
// serves 32 envs, where a wave-per-env mapping would leave most lanes idle.  Two lanes per env halve
// every per-player loop and give 2 waves per SIMD at 65 536 envs (latency hiding); the pair exchanges
// values with one DPP move.  Everything a lane indexes dynamically (its 12 group words, the node
// words, per-node accumulators) lives in LDS as [index][lane] columns: lane-private and
// bank-conflict-free by construction (bank = lane mod 32).  Combat is rebalanced over the whole wave:
// a prefix scan over lanes builds one work item per fighting group, damage accumulates in a shared
// LDS pool with integer LDS atomics (order-free, deterministic).  The SoA state arrays are
// env-fastest (coalesced); the wave's observation block (32 x 840 B, contiguous in the output) is
// assembled in LDS in output order and streamed out with 16-byte-per-lane coalesced stores.
// Uniform map/unit tables sit in SGPRs as nibble-packed words, per-lane-indexed ones in LDS.
// No MFMA: there is no dense contraction anywhere on this path.
#include <hip/hip_runtime.h>
#include <type_traits>
#include "evg_device.h"
#include "evg_rng.h"
#include "evg_mt.h"

namespace evg {

// Diagnostics are compiled only into the separate libraries libevg_diag.so (-DEVG_DIAG: phase ablation, the 16-envs-per-wave
// variant, a switch that forces the IEEE-division branch) and libevg_stamps.so (-DEVG_DIAG -DEVG_STAMPS: lane 0 of every
// wave stores s_memtime at phase boundaries into a debug buffer of its own).  The product library libevg.so contains
// none of it: no ablation branch, no stamp, no environment variable.
#ifdef EVG_DIAG
#define ABLATED(bit) ((io.ablate & (bit)) != 0u)
#else
#define ABLATED(bit) false
#endif
#ifdef EVG_STAMPS
#define STAMP(i)                                                                        \
    do {                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                              \
        const unsigned long long t_ = clock64();                                        \
        __builtin_amdgcn_sched_barrier(0);                                              \
        if (lane == 0 && io.stamps) io.stamps[(size_t)blockIdx.x * 16 + (i)] = t_;      \
    } while (0)
// whole-launch stamps of the wave: s_memrealtime (constant 100 MHz) at its start and end plus where it ran (HW_ID, XCC_ID)
#define STAMP_WAVE_BEGIN() const unsigned long long wave_t0_ = __builtin_amdgcn_s_memrealtime()
#define STAMP_WAVE_END()                                                                                                   \
    do {                                                                                                                   \
        const unsigned long long t1_ = __builtin_amdgcn_s_memrealtime();                                                   \
        if (threadIdx.x == 0 && A->io_.stamps) {                                                                            \
            A->io_.stamps[(size_t)blockIdx.x * 16 + 14] = (wave_t0_ & 0xFFFFFFFFull) | ((unsigned long long)__builtin_amdgcn_s_getreg(63492) << 32); \
            A->io_.stamps[(size_t)blockIdx.x * 16 + 15] = (t1_ & 0xFFFFFFFFull) | ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32);      \
        }                                                                                                                  \
    } while (0)
#else
#define STAMP(i)
#define STAMP_WAVE_BEGIN()
#define STAMP_WAVE_END()
#endif

// Phase markers (stamps build only).
#define PHASE(i) STAMP(i)

#include "step_common.inc"      // LDS layout of a wavefront's envs, phase fence, sorting network, small device helpers
#include "step_agents.inc"      // the scripted opponents (one device function over a view)
#include "step_kernel.inc"      // evg_step_kernel: skeleton + the phases of a turn (step_orders / step_combat / step_move_capture / step_outputs .inc)

#include "evg_step4.inc"      // the four-lanes-per-env mapping: what persistent launches of SMALL batches run (launch_step)

#undef S
#undef io

#include "side_kernels.inc"     // pack, chunk-queue check, reset, seeding, action generators, fog planes, Smart_State features

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
template <int LPW, bool MULTI>
static int launch_step_variant(const DevState& S, const StepIO& io, int obs_dtype, hipStream_t s) {
    const int n = io.env_hi - io.env_lo;
    const int nsets = (n + LPW / 2 - 1) / (LPW / 2);
    const dim3 grid((unsigned)nsets), block(WG);
    const StepArgs args{S, io};
    switch (obs_dtype) {
        case EVG_OBS_F32: hipLaunchKernelGGL((evg_step_kernel<float, LPW, MULTI>), grid, block, 0, s, args); break;
        case EVG_OBS_F64: hipLaunchKernelGGL((evg_step_kernel<double, LPW, MULTI>), grid, block, 0, s, args); break;
        case EVG_OBS_I16: hipLaunchKernelGGL((evg_step_kernel<int16_t, LPW, MULTI>), grid, block, 0, s, args); break;
        default: return -1;
    }
    return (int)hipGetLastError();
}

#ifdef EVG_DIAG
// round-5 experiment (diagnostic library, lanes = 256): the single-turn form with FOUR independent wavefronts per 256-thread workgroup
static int launch_step_wg256(const DevState& S, const StepIO& io, int obs_dtype, hipStream_t s) {
    const int nsets = (io.env_hi - io.env_lo + WG / 2 - 1) / (WG / 2);
    const dim3 grid((unsigned)((nsets + 3) / 4)), block(4 * WG);
    const StepArgs args{S, io};
    switch (obs_dtype) {
        case EVG_OBS_F32: hipLaunchKernelGGL((evg_step_kernel<float, WG, false, false, false, false, 4>), grid, block, 0, s, args); break;
        case EVG_OBS_F64: hipLaunchKernelGGL((evg_step_kernel<double, WG, false, false, false, false, 4>), grid, block, 0, s, args); break;
        case EVG_OBS_I16: hipLaunchKernelGGL((evg_step_kernel<int16_t, WG, false, false, false, false, 4>), grid, block, 0, s, args); break;
        default: return -1;
    }
    return (int)hipGetLastError();
}
#endif

// the chunked form of the persistent two-lane kernel: as many workgroups as the device holds, each taking units from its XCD's queue
static int launch_step_chunked(const DevState& S, const StepIO& io, int obs_dtype, hipStream_t s) {
    const dim3 grid((unsigned)io.grid_slots), block(WG);
    const StepArgs args{S, io};
    switch (obs_dtype) {
        case EVG_OBS_F32: hipLaunchKernelGGL((evg_step_kernel<float, WG, true, false, true>), grid, block, 0, s, args); break;
        case EVG_OBS_F64: hipLaunchKernelGGL((evg_step_kernel<double, WG, true, false, true>), grid, block, 0, s, args); break;
        case EVG_OBS_I16: hipLaunchKernelGGL((evg_step_kernel<int16_t, WG, true, false, true>), grid, block, 0, s, args); break;
        default: return -1;
    }
    return (int)hipGetLastError();
}

// Persistent launches of small and medium batches: the four-lanes-per-env mapping of evg_step4.inc -- 16 envs per wavefront,
// twice the wavefronts, 2 213 instead of 3 164 vector instructions per wave-turn.  While a SIMD holds few wavefronts, what counts is
// how long ONE wavefront needs for a turn, not the total instruction count.  Persistent form, us per turn on a whole MI355X,
// two-lane / four-lane (A/B of two builds on one box): 8 192 envs 11.0 / 8.3, 16 384 envs 11.8 / 8.3, 24 576 envs 12.1 / 9.6,
// 32 768 envs 12.1 / 10.4 (2 048 four-lane wavefronts = two per SIMD, the 160-VGPR build at 2 waves per SIMD); 40 960 envs
// 14.0 / 12.9, 49 152 envs 14.5 / 13.7 (up to 3 072 wavefronts = three per SIMD: the same kernel built for three, still
// unspilled).  Beyond that the four-lane grid is no longer resident at once (4 waves per SIMD need <= 128 VGPRs: 28 spilled) and
// the two-lane kernel wins.  Same state in HBM, same results (the persistent form of every small-batch test runs this kernel and
// is compared with the two-lane single-turn form and with the oracle).  The two thresholds are what the DEVICE holds
// (DeviceCaps::slots4_w2 / slots4_w3 wavefronts of 16 envs), not literals.
template <typename OT, bool MULTI, int WPE>
static void launch_step4_t(const DevState& S, const StepIO& io, hipStream_t s) {
    const StepArgs args{S, io};
    const dim3 grid((io.env_hi - io.env_lo + 15) / 16), block(WG);
    hipLaunchKernelGGL((evg_step4_kernel<OT, MULTI, WPE>), grid, block, 0, s, args);
}
template <bool MULTI, int WPE>
static int launch_step4(const DevState& S, const StepIO& io, int obs_dtype, hipStream_t s) {
    switch (obs_dtype) {
        case EVG_OBS_F32: launch_step4_t<float, MULTI, WPE>(S, io, s); break;
        case EVG_OBS_F64: launch_step4_t<double, MULTI, WPE>(S, io, s); break;
        case EVG_OBS_I16: launch_step4_t<int16_t, MULTI, WPE>(S, io, s); break;
        default: return -1;
    }
    return (int)hipGetLastError();
}

// What the device holds at once: compute units from hipDeviceProp_t, resident workgroups per CU from the occupancy of the kernels
// themselves.  On a whole MI355X: 256 CUs, 8 two-lane workgroups per CU (LDS: 8 x 20 208 B of 160 KiB; 2 waves per SIMD) = 2 048
// wavefronts = 65 536 envs; four-lane kernel 2 / 3 waves per SIMD = 2 048 / 3 072 wavefronts = 32 768 / 49 152 envs.
template <typename OT>
static int query_caps_t(DeviceCaps* c) {
    int b2m = 0, b2s = 0, b4w2 = 0, b4w3 = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&b2m, (const void*)evg_step_kernel<OT, WG, true>, WG, 0);
    if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&b2s, (const void*)evg_step_kernel<OT, WG, false>, WG, 0);
    if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&b4w2, (const void*)evg_step4_kernel<OT, true, 2>, WG, 0);
    if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&b4w3, (const void*)evg_step4_kernel<OT, true, 3>, WG, 0);
    if (e != hipSuccess) return (int)e;
    const int b2 = b2m < b2s ? b2m : b2s;
    if (b2 < 1 || b4w2 < 1 || b4w3 < 1) return (int)hipErrorLaunchOutOfResources;
    c->slots2 = c->cus * b2;
    // the four-lane builds aim at 2 / 3 waves per SIMD (amdgpu_waves_per_eu); the hardware may hold more of them, the plan does not use that
    c->slots4_w2 = c->cus * (b4w2 < 8 ? b4w2 : 8);
    c->slots4_w3 = c->cus * (b4w3 < 12 ? b4w3 : 12);
    return 0;
}

int query_device_caps(int device_id, int obs_dtype, DeviceCaps* caps) {
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, device_id);
    if (e != hipSuccess) return (int)e;
    caps->cus = prop.multiProcessorCount;
    caps->simds = 4 * caps->cus;
    // MI355X: 256 MiB of Infinity Cache behind 256 CUs; a partition gets its share (evg_config::cache_mib overrides)
    caps->cache_bytes = (256ll << 20) * caps->cus / 256;
    switch (obs_dtype) {
        case EVG_OBS_F32: return query_caps_t<float>(caps);
        case EVG_OBS_F64: return query_caps_t<double>(caps);
        case EVG_OBS_I16: return query_caps_t<int16_t>(caps);
        default: return -1;
    }
}

// Which kernel plays which envs.  Single-turn launches (evg_step) and the stock-entropy mode: one launch of the two-lane kernel.
// Persistent form: a batch up to what the device holds at once runs the four-lane kernel while its grid is resident at two or
// three waves per SIMD and the two-lane kernel above that (the thresholds are DeviceCaps, not literals).  A LARGER batch used to be
// one launch of ceil(N / 32) workgroups, each playing all its turns: whole rounds of resident workgroups one after the other -- which
// is good, every round's working set (171 MB at 65 536 envs) stays inside the 256 MB Infinity Cache -- but the REMAINDER behind the
// last whole round ran alone at low occupancy for a whole launch (98 304 envs: 30.9 us per turn where 1.5 x 17.0 = 25.5 would be
// proportional; 65 536 + 4 480 envs cost 28.0 instead of 18.0).  Now:
//   * the whole rounds but the last: one plain launch, as before;
//   * the last whole round TOGETHER WITH the remainder, while the two fit the Infinity Cache (kChunkFootprintMax): one CHUNKED launch -- as many
//     workgroups as the device holds, each taking units (set of 32 envs) x (chunk of kChunkTurns turns) from its XCD's queue and handing
//     the set on through HBM (see the kernel's prologue): all slots stay busy until the queues run dry (98 304 envs: 26 us per turn,
//     70 016: 18.0);
//   * a larger remainder: its own plain launch behind the whole rounds (four-lane kernel up to 49 152 envs).  Chunking it too would
//     cycle the launch through nearly two rounds' worth of envs, more than the Infinity Cache holds, and every turn then runs ~30 %
//     slower (measured: 131 104 envs chunked 46 us per turn, plain 44; 262 144 chunked 90, plain 67).
#ifndef EVG_CHUNK_TURNS
// turns per chunk of a chunked launch (a build-time knob so that it can be re-measured with two builds: tools/scaling_lib.py)
#define EVG_CHUNK_TURNS 25
#endif
[[maybe_unused]] constexpr int kChunkTurns = EVG_CHUNK_TURNS;
// A chunked launch cycles through ALL its envs every few chunks, so its working set -- state, and what THIS rollout writes: observations,
// orders, results; 2.74 KB per env with float32 observations -- must fit the device's share of the Infinity Cache (DeviceCaps::cache_bytes:
// 256 MiB on a whole MI355X, + 2 %): measured, us per turn, chunked / the alternative: 98 304 envs (270 MB) 25.8 / 27.5; 104 448 envs
// (287 MB) 32.5 / ~29.5.
long long rollout_bytes_per_env(const StepIO& io, int obs_dtype) {
    long long b = kStateBytesPerEnv + 13 /* fin_ret, fin_len, fin_win */ + 19 /* reward, done, winner, scores, status */ + 8 /* hand-over checksums */;
    if (io.obs) b += OBS2 * (obs_dtype == EVG_OBS_F64 ? 8 : (obs_dtype == EVG_OBS_I16 ? 2 : 4));
    if (io.actions_out) b += 2 * NA * 2 * 4;
    if (io.gen_actions == 2) b += 24;                    // the scripted agents' objects
    return b;
}
LaunchPlan plan_step(const DevState& S, const StepIO& io, int obs_dtype, const DeviceCaps& caps) {
    LaunchPlan p;
    p.n = 1;
    p.piece[0] = LaunchPiece{0, 0, S.N, 0};
    p.piece[1] = LaunchPiece{0, 0, 0, 0};
    const bool multi = io.turns > 1;
    if (S.mt_key || !multi) return p;
    [[maybe_unused]] const long long cap2 = 32ll * caps.slots2, cap4_2 = 16ll * caps.slots4_w2, cap4_3 = 16ll * caps.slots4_w3;
    [[maybe_unused]] const long long N = S.N;
#ifdef EVG_DIAG
    // experiment: the chunked form over the WHOLE batch, whatever its size (a working set beyond the Infinity Cache)
    if (io.lanes_per_wave == 2) {
        if (N > cap2 && io.turns > kChunkTurns) p.piece[0].chunk_turns = kChunkTurns;
        return p;
    }
    if (io.lanes_per_wave != 0) return p;          // a forced kernel variant plays the whole batch in one plain launch
#endif
#ifdef EVG_STAMPS
    return p;                                       // the stamp buffer is indexed by workgroup: one plain launch
#else
    if (N <= cap4_2) { p.piece[0].four_lane_wpe = 2; return p; }
    if (N <= cap4_3) { p.piece[0].four_lane_wpe = 3; return p; }
    if (N <= cap2) return p;
    const long long full = N / cap2, rem = N - full * cap2;
    if (rem == 0) return p;                         // whole rounds only: one plain launch
    if ((cap2 + rem) * rollout_bytes_per_env(io, obs_dtype) <= caps.cache_bytes + caps.cache_bytes / 50 && io.turns > kChunkTurns) {
        p.n = 0;
        if (full >= 2) p.piece[p.n++] = LaunchPiece{0, 0, (int32_t)((full - 1) * cap2), 0};
        p.piece[p.n++] = LaunchPiece{0, (int32_t)((full - 1) * cap2), (int32_t)N, kChunkTurns};
        return p;
    }
    p.n = 2;
    p.piece[0] = LaunchPiece{0, 0, (int32_t)(full * cap2), 0};
    p.piece[1] = LaunchPiece{rem <= cap4_2 ? 2 : (rem <= cap4_3 ? 3 : 0), (int32_t)(full * cap2), (int32_t)N, 0};
    return p;
#endif
}

int launch_step(const DevState& S, const StepIO& io_in, int obs_dtype, const DeviceCaps& caps, void* stream) {
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    StepIO io = io_in;
    const bool multi = io.turns > 1;
    io.env_lo = 0; io.env_hi = S.N; io.flags = 0; io.nsets = 0; io.chunk_turns = 0; io.progress_base = 0; io.grid_slots = 0;
    const int grid2 = (S.N + WG / 2 - 1) / (WG / 2);
    if (!multi && grid2 <= caps.slots2) io.flags |= STEP_F_STAGGER;
    if (S.mt_key) {                       // stock-entropy mode: single-turn launches of the sequential-draw instantiation
        if (multi) return -1;
        const dim3 grid(grid2), block(WG);
        const StepArgs args{S, io};
        switch (obs_dtype) {
            case EVG_OBS_F32: hipLaunchKernelGGL((evg_step_kernel<float, WG, false, true>), grid, block, 0, s, args); break;
            case EVG_OBS_F64: hipLaunchKernelGGL((evg_step_kernel<double, WG, false, true>), grid, block, 0, s, args); break;
            case EVG_OBS_I16: hipLaunchKernelGGL((evg_step_kernel<int16_t, WG, false, true>), grid, block, 0, s, args); break;
            default: return -1;
        }
        return (int)hipGetLastError();
    }
#ifdef EVG_DIAG
    // the two-lane kernel at any size
    if (io.lanes_per_wave == 64) return multi ? launch_step_variant<64, true>(S, io, obs_dtype, s) : launch_step_variant<64, false>(S, io, obs_dtype, s);
    if (io.lanes_per_wave == 256) return multi ? launch_step_variant<64, true>(S, io, obs_dtype, s) : launch_step_wg256(S, io, obs_dtype, s);
    if (io.lanes_per_wave == 4) return multi ? launch_step4<true, 4>(S, io, obs_dtype, s) : launch_step4<false, 4>(S, io, obs_dtype, s);
    if (io.lanes_per_wave == 32) return multi ? launch_step_variant<32, true>(S, io, obs_dtype, s) : launch_step_variant<32, false>(S, io, obs_dtype, s);
#endif
    if (!multi) return launch_step_variant<64, false>(S, io, obs_dtype, s);
    const LaunchPlan plan = plan_step(S, io, obs_dtype, caps);
    for (int i = 0; i < plan.n; ++i) {
        const LaunchPiece& pc = plan.piece[i];
        io.env_lo = pc.env_lo; io.env_hi = pc.env_hi;
        io.flags = (pc.four_lane_wpe && (pc.env_hi - pc.env_lo + 15) / 16 > caps.simds) ? STEP_F_SHARED_SIMD : 0;
        io.nsets = 0; io.chunk_turns = 0;
        if (pc.chunk_turns > 0) {
            io.nsets = (pc.env_hi - pc.env_lo + WG / 2 - 1) / (WG / 2);
            io.chunk_turns = pc.chunk_turns;
            io.progress_base = 0;
            io.grid_slots = io.nsets < caps.slots2 ? io.nsets : caps.slots2;
            // every XCD's queue starts at unit 0 and every set's progress flag at "no chunk finished": ONE memset on the stream (the two arrays are one
            // allocation, queue first).  Nothing of a chunked launch lives on the host, so a captured launch can be replayed (hipGraph).
            const hipError_t me = hipMemsetAsync(S.queue, 0, (1024 + ((size_t)S.N + 31) / 32 + 1) * sizeof(uint32_t), s);
            if (me != hipSuccess) return (int)me;
        }
        int rc;
        if (pc.four_lane_wpe == 2) rc = launch_step4<true, 2>(S, io, obs_dtype, s);
        else if (pc.four_lane_wpe == 3) rc = launch_step4<true, 3>(S, io, obs_dtype, s);
        else if (io.nsets > 0) rc = launch_step_chunked(S, io, obs_dtype, s);
        else rc = launch_step_variant<64, true>(S, io, obs_dtype, s);
        if (rc) return rc;
        if (io.nsets > 0) {
            // ... and behind it, on the same stream: every XCD's queue handed out all its units (sets x chunks), or the handle is flagged
            ChunkUnits want;
            const uint32_t nchunks = (uint32_t)((io.turns + io.chunk_turns - 1) / io.chunk_turns);
            for (int x = 0; x < 16; ++x) want.per_xcd[x] = x < S.nxcd ? (uint32_t)((io.nsets - x + S.nxcd - 1) / S.nxcd) * nchunks : 0u;
            hipLaunchKernelGGL(evg_chunk_verify_kernel, dim3(1), dim3(WG), 0, s, S, want);
            rc = (int)hipGetLastError();
            if (rc) return rc;
        }
    }
    return 0;
}

// evg_step_vs_policy / evg_observe_seat: one launch of the one-seat instantiation of the single-turn two-lane kernel
int launch_step_seat(const DevState& S, const StepIO& io_in, int obs_dtype, const DeviceCaps& caps, void* stream) {
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (S.mt_key) return -1;
    StepIO io = io_in;
    io.turns = 1; io.env_lo = 0; io.env_hi = S.N; io.flags = 0; io.nsets = 0; io.chunk_turns = 0; io.progress_base = 0; io.grid_slots = 0;
    const int grid2 = (S.N + WG / 2 - 1) / (WG / 2);
    if (grid2 <= caps.slots2) io.flags |= STEP_F_STAGGER;
    const dim3 grid(grid2), block(WG);
    const StepArgs args{S, io};
    switch (obs_dtype) {
        case EVG_OBS_F32: hipLaunchKernelGGL((evg_step_kernel<float, WG, false, false, false, true>), grid, block, 0, s, args); break;
        case EVG_OBS_F64: hipLaunchKernelGGL((evg_step_kernel<double, WG, false, false, false, true>), grid, block, 0, s, args); break;
        case EVG_OBS_I16: hipLaunchKernelGGL((evg_step_kernel<int16_t, WG, false, false, false, true>), grid, block, 0, s, args); break;
        default: return -1;
    }
    return (int)hipGetLastError();
}

// which XCC ids does this device have?  (evg_create: 1 024 one-wave workgroups report where they ran)
__global__ void __launch_bounds__(WG) evg_xcd_probe_kernel(uint32_t* out) {
    if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_getreg(63508) & 15u;      // HW_REG_XCC_ID
}
int launch_xcd_probe(uint32_t* out, void* stream) {
    hipLaunchKernelGGL(evg_xcd_probe_kernel, dim3(1024), dim3(WG), 0, reinterpret_cast<hipStream_t>(stream), out);
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

int launch_scripted_actions(const DevState& S, int policy, int player, const void* obs, int32_t* actions, int obs_dtype, void* stream) {
    const dim3 grid((S.N + 255) / 256), block(256);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    switch (obs_dtype) {
        case EVG_OBS_F32: hipLaunchKernelGGL(evg_scripted_actions_kernel<float>, grid, block, 0, s, S, policy, player, (const float*)obs, actions); break;
        case EVG_OBS_F64: hipLaunchKernelGGL(evg_scripted_actions_kernel<double>, grid, block, 0, s, S, policy, player, (const double*)obs, actions); break;
        case EVG_OBS_I16: hipLaunchKernelGGL(evg_scripted_actions_kernel<int16_t>, grid, block, 0, s, S, policy, player, (const int16_t*)obs, actions); break;
        default: return -1;
    }
    return (int)hipGetLastError();
}

int launch_mt_seed(const DevState& S, const uint32_t* seeds_dev, void* stream) {
    hipLaunchKernelGGL(evg_mt_seed_kernel, dim3((S.N + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), S, seeds_dev);
    return (int)hipGetLastError();
}

int launch_scripted_reset(const DevState& S, void* stream) {
    hipLaunchKernelGGL(evg_scripted_reset_kernel, dim3((2 * S.N + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), S);
    return (int)hipGetLastError();
}

template <typename OT>
static void launch_smart_state_t(int n, const dim3 grid, hipStream_t s, int player, int seat_only, const void* obs, float* out, float* out_swarm) {
    if (out_swarm) hipLaunchKernelGGL((evg_smart_state_kernel<OT, true>), grid, dim3(256), 0, s, n, player, seat_only, (const OT*)obs, out, out_swarm);
    else hipLaunchKernelGGL((evg_smart_state_kernel<OT, false>), grid, dim3(256), 0, s, n, player, seat_only, (const OT*)obs, out, out_swarm);
}
// out_swarm non-NULL: compact form, out = shared [N][34]
int launch_smart_state(const DevState& S, int player, const void* obs, int seat_only, float* out, float* out_swarm, int obs_dtype, void* stream) {
    const int blocks = (S.N + 3) / 4;                       // one wavefront per env and pass; 8 blocks per CU resident, further envs in passes
    const dim3 grid((unsigned)(blocks < 2048 ? blocks : 2048));
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    switch (obs_dtype) {
        case EVG_OBS_F32: launch_smart_state_t<float>(S.N, grid, s, player, seat_only, obs, out, out_swarm); break;
        case EVG_OBS_F64: launch_smart_state_t<double>(S.N, grid, s, player, seat_only, obs, out, out_swarm); break;
        case EVG_OBS_I16: launch_smart_state_t<int16_t>(S.N, grid, s, player, seat_only, obs, out, out_swarm); break;
        default: return -1;
    }
    return (int)hipGetLastError();
}

// network output -> orders: one DPP row (16 lanes) per env
int launch_smart_actions(const DevState& S, int player, const void* obs, int seat_only, const float* q, int32_t* actions, int32_t* directions, int obs_dtype,
                         void* stream, const SmartExplore* ex) {
    // (the exploring form draws per env in the first wavefront of 1 024-thread workgroups = 64 envs: side_kernels.inc)
    const unsigned threads = ex ? 1024u : 256u;
    const dim3 grid((unsigned)(((size_t)S.N * 16 + threads - 1) / threads)), block(threads);
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    int2* a = reinterpret_cast<int2*>(actions);
    int2* d = reinterpret_cast<int2*>(directions);
    ExploreArgs X{};
    if (ex) X = ExploreArgs{S.seed_lo, S.seed_hi, S.env_id_base, S.episode, ex->seat, ex->eps, ex->eps_env, ex->explored};
#define EVG_LAUNCH_SMART(OT)                                                                                                                     \
    if (ex) hipLaunchKernelGGL((evg_smart_actions_kernel<OT, true>), grid, block, 0, s, S.N, player, seat_only, (const OT*)obs, q, a, d, X);      \
    else hipLaunchKernelGGL((evg_smart_actions_kernel<OT, false>), grid, block, 0, s, S.N, player, seat_only, (const OT*)obs, q, a, d, X)
    switch (obs_dtype) {
        case EVG_OBS_F32: EVG_LAUNCH_SMART(float); break;
        case EVG_OBS_F64: EVG_LAUNCH_SMART(double); break;
        case EVG_OBS_I16: EVG_LAUNCH_SMART(int16_t); break;
        default: return -1;
    }
#undef EVG_LAUNCH_SMART
    return (int)hipGetLastError();
}

int launch_fog(const DevState& S, uint8_t* fog, uint8_t* know, int8_t* sight, void* stream) {
    hipLaunchKernelGGL(evg_fog_kernel, dim3((2 * S.N + 255) / 256), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), S, fog, know, sight);
    return (int)hipGetLastError();
}

int launch_pack_results(const DevState& S, float* out, long long* counts, void* stream) {
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (counts) {
        const hipError_t me = hipMemsetAsync(counts, 0, 4 * sizeof(long long), s);
        if (me != hipSuccess) return (int)me;
    }
    hipLaunchKernelGGL(evg_pack_results_kernel, dim3((S.N + 255) / 256), dim3(256), 0, s, S, reinterpret_cast<float4*>(out), counts);
    return (int)hipGetLastError();
}

int launch_random_actions(const DevState& S, int32_t* actions, int seat, void* stream) {
    // (a latency chain of two Philox blocks per thread; 128-thread blocks for the half-sized one-seat form are no faster: 5.4 against 5.2 us)
    const dim3 grid(((seat < 0 ? 2 : 1) * S.N + 255) / 256), block(256);
    hipLaunchKernelGGL(evg_random_actions_kernel, grid, block, 0, reinterpret_cast<hipStream_t>(stream), S, actions, seat);
    return (int)hipGetLastError();
}

}  // namespace evg
