// hooks.hip -- device-side accumulators behind the episode hooks, so that `run(policy, env, stop, hook)` with
// hooks attached never synchronises the stream per step.
//
// Replaces the per-step host work of
//   TotalRewardPerEpisode      RLCore/src/core/hooks.jl:146-196  (reward accumulated per step, pushed at episode end)
//   BatchStepsPerEpisode       hooks.jl:202-231                  (per-instance step counters of a vector env)
//   StepsPerEpisode            hooks.jl:64-101
// In the reference these hooks read reward(env) / is_terminated(env) on the host every step.  Here one launch per
// vec-step keeps (steps, return) per env instance in HBM and appends one record per finished episode to a device
// log; the host reads the log when it is asked for the lists (end of the experiment, or on demand).
// Record = {vec_step, env, steps, return (Float64 bits)}.  Slots are claimed with an atomic counter, so the order
// inside one vec-step is arbitrary -- the host sorts by (vec_step, env), which makes the lists deterministic.
#include "common.h"

namespace rlhip {

__global__ __launch_bounds__(256) void episode_stats_kernel(const float* __restrict__ reward,
                                                            const uint8_t* __restrict__ done, int64_t n,
                                                            uint32_t vec_step, int32_t* __restrict__ steps_acc,
                                                            double* __restrict__ return_acc,
                                                            rlhip_episode_record* __restrict__ log, uint32_t log_cap,
                                                            uint32_t* __restrict__ log_count) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int32_t st = steps_acc[i] + 1;
    double rt = return_acc[i] + (double)reward[i];
    if (done[i]) {
        uint32_t slot = atomicAdd(log_count, 1u);  // keeps counting past the capacity: the host sees the overflow
        if (slot < log_cap) {
            rlhip_episode_record rec;
            rec.vec_step = vec_step;
            rec.env = (uint32_t)i;
            rec.steps = st;
            rec.pad = 0;
            rec.total_reward = rt;
            log[slot] = rec;
        }
        st = 0;
        rt = 0.0;
    }
    steps_acc[i] = st;
    return_acc[i] = rt;
}

}  // namespace rlhip

using namespace rlhip;

extern "C" int32_t rlhip_hook_episode_stats(const float* reward, const uint8_t* done, int64_t n, uint32_t vec_step,
                                            int32_t* steps_acc, double* return_acc, rlhip_episode_record* log,
                                            uint32_t log_capacity, uint32_t* log_count, rlhip_stream_t stream) {
    RLHIP_REQUIRE(n >= 0, "negative size");
    if (n == 0) return RLHIP_OK;
    RLHIP_REQUIRE(reward && done && steps_acc && return_acc && log && log_count, "NULL array");
    hipLaunchKernelGGL(episode_stats_kernel, dim3((int)((n + 255) / 256)), dim3(256), 0, as_stream(stream), reward, done,
                       n, vec_step, steps_acc, return_acc, log, log_capacity, log_count);
    RLHIP_LAUNCH_CHECK();
    return RLHIP_OK;
}
