/* c_client.c -- the drop-in boundary used from plain C: no Python, no torch, only include/evg.h and the HIP runtime for
 * the caller-owned device buffers.  Plays `turns` turns of N random-vs-random DemoMap games (the loop of the reference's
 * demo/random_demo.py:90-113, vectorised) and prints the win counters plus a checksum of the last observations; then the same number of turns of
 * the learner-seat loop (evg_step_vs_policy: caller on seat 0, on-device swarm_agent on seat 1) with its own counters and checksum.
 *
 *   gcc -O2 -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include examples/c_client.c \
 *       -Leverglades-ai-wargame_amd -levg -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/everglades-ai-wargame_amd -Wl,-rpath,/opt/rocm/lib -o c_client
 *   ./c_client 4096 300 7
 */
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <hip/hip_runtime_api.h>
#include "evg.h"

#define CHECK(x) do { int rc_ = (x); if (rc_ != 0) { fprintf(stderr, "%s failed (%d): %s\n", #x, rc_, evg_last_error()); return 1; } } while (0)
#define HIPCHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 4096, turns = argc > 2 ? atoi(argv[2]) : 300;
    const uint64_t seed = argc > 3 ? strtoull(argv[3], NULL, 0) : 7;
    if (evg_abi_version() != EVG_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 1; }

    evg_config cfg = {0};
    cfg.struct_size = sizeof(cfg); cfg.abi_version = EVG_ABI_VERSION;
    cfg.num_envs = n; cfg.device_id = 0; cfg.seed = seed; cfg.env_id_base = 0;
    cfg.obs_dtype = EVG_OBS_F32; cfg.auto_reset = 1; cfg.rng_mode = EVG_RNG_KEYED_PHILOX;
    evg_default_tables(&cfg.tables);
    evg_handle* h = NULL;
    CHECK(evg_create(&cfg, &h));

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
    hipFree(seat_rows); hipFree(seat_obs);
    hipFree(actions); hipFree(obs); hipFree(reward); hipFree(done);
    evg_destroy(h);
    return 0;
}
