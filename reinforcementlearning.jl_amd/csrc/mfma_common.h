// mfma_common.h -- bf16 helpers and the gfx950 MFMA fragment types shared by the MFMA learners (dqn3.hip, ppo3.hip, ppo3w.hip).
//
// v_mfma_f32_32x32x16_bf16: D(32x32) += A(32x16) * B(16x32), one wave.  Register layout (MI355X guide
// section 3; pinned by every parity test of the MFMA learners and by tools/micro/mfma_f32_l1.hip for the f32 form):
//   A: lane l holds A[row = l & 31][k = 8 * (l >> 5) .. +7]          (8 bf16 = one 16-byte load)
//   B: lane l holds B[k = 8 * (l >> 5) .. +7][col = l & 31]
//   D: lane l holds D[row = (q & 3) + 8 * (q >> 2) + 4 * (l >> 5)][col = l & 31],  q = 0..15
#pragma once
#include "common.h"

namespace rlhip {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;

// f32 -> bf16, round to nearest even: gfx950 has the conversion in hardware (v_cvt_pk_bf16_f32)
__device__ __forceinline__ uint32_t pack2_bf16(float lo, float hi) {
    f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ uint16_t f32_to_bf16_rne(float f) { return (uint16_t)(pack2_bf16(f, 0.0f) & 0xFFFFu); }
__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }

// row of MFMA accumulator register q for a lane in half kb = lane >> 5
__device__ __forceinline__ int mfma_row(int q, int kb) { return (q & 3) + 8 * (q >> 2) + 4 * kb; }

}  // namespace rlhip
