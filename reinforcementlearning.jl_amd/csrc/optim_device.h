// optim_device.h -- device helpers shared by optim.hip and the fused learner tails (dqn.hip): the block-wide Float64
// sum and the Adam step, so that a fused "reduce -> clip -> Adam" kernel reproduces rlhip_clip_adam_f32 bit for bit.
#pragma once
#include "common.h"

namespace rlhip {

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// block-wide sum (result valid in every thread); scratch: >= 16 doubles of LDS
__device__ __forceinline__ double block_sum(double v, double* scratch) {
    int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    v = wave_sum(v);
    if (lane == 0) scratch[wid] = v;
    __syncthreads();
    int nw = (blockDim.x + 63) >> 6;
    double t = 0.0;
    for (int w = 0; w < nw; ++w) t += scratch[w];
    __syncthreads();
    return t;
}

// Optimisers.Adam: mt = b1*mt + (1-b1)*dx; vt = b2*vt + (1-b2)*dx^2;
//                  dx' = mt / (1 - b1^t) / (sqrt(vt / (1 - b2^t)) + eps) * eta;  x -= dx'
__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, float lr, float b1, float b2,
                                      float eps, float c1, float c2) {
    float mi = b1 * m + (1.0f - b1) * g;
    float vi = b2 * v + (1.0f - b2) * (g * g);
    m = mi;
    v = vi;
    float d = mi / c1 / (sqrtf(vi / c2) + eps) * lr;
    p = p - d;
}

}  // namespace rlhip
