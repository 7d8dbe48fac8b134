/* c_client.c -- the drop-in boundary used from plain C: no Python, no torch, only include/evg.h and the HIP runtime for
 * the caller-owned device buffers.  Plays `turns` turns of N random-vs-random DemoMap games (the loop of the reference's
 * demo/random_demo.py:90-113, vectorised) and prints the win counters plus a checksum of the last observations; then the same number of turns of
 * the learner-seat loop (evg_step_vs_policy: caller on seat 0, on-device swarm_agent on seat 1) with its own counters and checksum.
 *
 *   gcc -O2 -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include examples/c_client.c \
 *       -Leverglades-ai-wargame_amd -levg -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/everglades-ai-wargame_amd -Wl,-rpath,/opt/rocm/lib -o c_client
 *   ./c_client 4096 300 7
 *
 * Sharded over GPUs the same program runs once per GPU -- `c_client N turns seed WORLD RANK IDFILE`: rank r plays the contiguous global env ids
 * [r N, (r + 1) N) on device r and, at the end, every rank's per-env episode results travel to rank 0 in ONE exchange: evg_comm_unique_id (rank 0 writes the 128
 * bytes to IDFILE, the other ranks read it), evg_comm_init, evg_gather_returns -- RCCL opened by libevg.so itself, no framework.  Without the three extra
 * arguments it is a one-rank communicator (the same calls, nothing on the wire).
 */
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
#include <unistd.h>
#include <hip/hip_runtime_api.h>
#include "evg.h"

#define CHECK(x) do { int rc_ = (x); if (rc_ != 0) { fprintf(stderr, "%s failed (%d): %s\n", #x, rc_, evg_last_error()); return 1; } } while (0)
#define HIPCHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 4096, turns = argc > 2 ? atoi(argv[2]) : 300;
    const uint64_t seed = argc > 3 ? strtoull(argv[3], NULL, 0) : 7;
    const int world = argc > 6 ? atoi(argv[4]) : 1, rank = argc > 6 ? atoi(argv[5]) : 0;
    const char* idfile = argc > 6 ? argv[6] : NULL;
    if (world < 1 || rank < 0 || rank >= world) { fprintf(stderr, "bad world / rank\n"); return 1; }
    if (evg_abi_version() != EVG_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 1; }

    evg_config cfg = {0};
    cfg.struct_size = sizeof(cfg); cfg.abi_version = EVG_ABI_VERSION;
    cfg.num_envs = n; cfg.device_id = rank; cfg.seed = seed; cfg.env_id_base = (uint64_t)rank * (uint64_t)n;     /* one GPU and one contiguous shard per rank */
    cfg.obs_dtype = EVG_OBS_F32; cfg.auto_reset = 1; cfg.rng_mode = EVG_RNG_KEYED_PHILOX;
    evg_default_tables(&cfg.tables);
    evg_handle* h = NULL;
    CHECK(evg_create(&cfg, &h));
    HIPCHECK(hipSetDevice(rank));                            /* the caller-owned buffers below live on the handle's device */

    int32_t* actions; float *obs, *reward; uint8_t* done;
    const size_t obs_elems = (size_t)n * EVG_NUM_PLAYERS * EVG_OBS_LEN;
    HIPCHECK(hipMalloc((void**)&actions, (size_t)n * EVG_NUM_PLAYERS * EVG_NUM_ACTIONS * 2 * sizeof(int32_t)));
    HIPCHECK(hipMalloc((void**)&obs, obs_elems * sizeof(float)));
    HIPCHECK(hipMalloc((void**)&reward, (size_t)n * 2 * sizeof(float)));
    HIPCHECK(hipMalloc((void**)&done, (size_t)n));

    CHECK(evg_reset(h, NULL, obs, NULL));
    for (int t = 0; t < turns; ++t) {                       /* everything is enqueued on the default stream */
        CHECK(evg_random_actions(h, actions, NULL));
        CHECK(evg_step(h, actions, obs, reward, done, NULL, NULL, NULL, NULL));
    }
    HIPCHECK(hipDeviceSynchronize());

    int64_t totals[4];
    CHECK(evg_episode_stats(h, NULL, NULL, NULL, totals));
    float* host = (float*)malloc(obs_elems * sizeof(float));
    HIPCHECK(hipMemcpy(host, obs, obs_elems * sizeof(float), hipMemcpyDeviceToHost));
    long long sum = 0;
    for (size_t i = 0; i < obs_elems; ++i) sum += (long long)host[i] * (long long)(1 + i % 7);
    printf("envs %d turns %d episodes %lld p0 %lld p1 %lld tie %lld obs_checksum %lld\n", n, turns, (long long)totals[0],
           (long long)totals[1], (long long)totals[2], (long long)totals[3], sum);
    free(host);

    /* the turn the reference's scripts run (evaluate.py:85-93,143-152): a caller on seat 0 -- its 7 rows arrive in a tensor, here from the library's
     * stand-in generator --, the on-device swarm_agent bot on seat 1 inside the step kernel, the caller's observation [N][105] only */
    int32_t* seat_rows; float* seat_obs;
    HIPCHECK(hipMalloc((void**)&seat_rows, (size_t)n * EVG_NUM_ACTIONS * 2 * sizeof(int32_t)));
    HIPCHECK(hipMalloc((void**)&seat_obs, (size_t)n * EVG_OBS_LEN * sizeof(float)));
    CHECK(evg_observe_seat(h, 0, seat_obs, NULL));
    for (int t = 0; t < turns; ++t) {
        CHECK(evg_random_actions_seat(h, 0, seat_rows, NULL));
        CHECK(evg_step_vs_policy(h, 0, seat_rows, 0, EVG_POLICY_SWARM, seat_obs, reward, done, NULL, NULL, NULL, NULL));
    }
    uint32_t fault = 0;
    CHECK(evg_check_fault(h, &fault));                      /* synchronises */
    CHECK(evg_episode_stats(h, NULL, NULL, NULL, totals));
    const size_t seat_elems = (size_t)n * EVG_OBS_LEN;
    host = (float*)malloc(seat_elems * sizeof(float));
    HIPCHECK(hipMemcpy(host, seat_obs, seat_elems * sizeof(float), hipMemcpyDeviceToHost));
    sum = 0;
    for (size_t i = 0; i < seat_elems; ++i) sum += (long long)host[i] * (long long)(1 + i % 7);
    printf("vs_episodes %lld vs_p0 %lld vs_p1 %lld vs_tie %lld vs_obs_checksum %lld\n", (long long)totals[0], (long long)totals[1], (long long)totals[2],
           (long long)totals[3], sum);
    free(host);

    /* the path's one exchange between GPUs (win bookkeeping of evaluate.py:155-181 on rank 0): every rank's packed rows {return p0, return p1, winner, length} */
    {
        unsigned char id[EVG_COMM_ID_BYTES];
        if (rank == 0) {
            CHECK(evg_comm_unique_id(id));
            if (idfile) {
                char tmp[4096];
                snprintf(tmp, sizeof(tmp), "%s.tmp", idfile);
                FILE* f = fopen(tmp, "wb");
                if (!f || fwrite(id, 1, sizeof(id), f) != sizeof(id) || fclose(f) != 0 || rename(tmp, idfile) != 0) { fprintf(stderr, "cannot write %s\n", idfile); return 1; }
            }
        } else {
            FILE* f = NULL;
            for (int i = 0; i < 2400 && !(f = fopen(idfile, "rb")); ++i) usleep(50000);
            if (!f || fread(id, 1, sizeof(id), f) != sizeof(id)) { fprintf(stderr, "no communicator id in %s\n", idfile); return 1; }
            fclose(f);
        }
        int32_t* counts = (int32_t*)malloc(sizeof(int32_t) * (size_t)world);
        for (int r = 0; r < world; ++r) counts[r] = n;
        CHECK(evg_comm_init(h, id, world, rank, counts));
        float* rows = NULL;
        const size_t total = (size_t)world * (size_t)n;
        if (rank == 0) HIPCHECK(hipMalloc((void**)&rows, total * 4 * sizeof(float)));
        CHECK(evg_gather_returns(h, 0, rows, NULL));
        HIPCHECK(hipDeviceSynchronize());
        if (rank == 0) {
            float* hr = (float*)malloc(total * 4 * sizeof(float));
            HIPCHECK(hipMemcpy(hr, rows, total * 4 * sizeof(float), hipMemcpyDeviceToHost));
            long long w[4] = {0, 0, 0, 0}, len = 0;
            for (size_t e = 0; e < total; ++e) { const int k = (int)hr[4 * e + 2]; w[k < 0 ? 3 : k] += 1; len += (long long)hr[4 * e + 3]; }
            printf("gathered_rows %lld gathered_p0 %lld gathered_p1 %lld gathered_tie %lld gathered_unfinished %lld gathered_length_sum %lld\n", (long long)total, w[0], w[1], w[2],
                   w[3], len);
            free(hr);
            hipFree(rows);
        }
        free(counts);
        CHECK(evg_comm_destroy(h));
    }
    hipFree(seat_rows); hipFree(seat_obs);
    hipFree(actions); hipFree(obs); hipFree(reward); hipFree(done);
    evg_destroy(h);
    return 0;
}
