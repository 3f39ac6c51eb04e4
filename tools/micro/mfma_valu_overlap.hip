// mfma_valu_overlap.hip -- do a SIMD's f32 VALU stream and its MFMA stream run BESIDE each other on gfx950, or do they add?
//
// Round-4 rewrite.  The round-3 version was built with plain -O3: the SLP vectorizer turned its "32 v_fma_f32" into 16
// v_pk_fma_f32 (the one VALU form that is known to be slow beside MFMAs) and the hazard recognizer put an s_nop 11 into the
// f32-MFMA loops, so its conclusion ("the streams add") was not a statement about v_fma_f32.  This version
//   * issues the VALU stream as inline-asm v_fma_f32 (nothing can pack or reorder it),
//   * gives every MFMA stream FOUR independent accumulators (no dependent back-to-back MFMAs, hence no hazard s_nop),
//   * is built with -fno-slp-vectorize -fno-vectorize, and tests/test_micro_overlap_disasm.py disassembles the timed loops and
//     fails on any v_pk_*_f32 or any s_nop >= 4 in them,
//   * adds the case the round-3 file lacked: VALU-only waves beside bf16-MFMA-only waves on one SIMD,
//   * is run a second time under rocprofv3 --pmc (SQ_VALU_MFMA_BUSY_CYCLES, SQ_ACTIVE_INST_VALU, SQ_WAVE_CYCLES,
//     SQ_BUSY_CYCLES) -- every mode is its own kernel name.
//
// One workgroup per CU (grid 256).  WPS = waves per SIMD = blockDim / 256 (wave w of a workgroup sits on SIMD w % 4).
// Per loop iteration a wave of stream
//   V issues NV independent v_fma_f32                                   (NV x 128 flop)
//   M issues 4 independent v_mfma_f32_32x32x2_f32                       (4 x 4096 flop;  16 passes = 64 cycles each)
//   B issues 4 independent v_mfma_f32_32x32x16_bf16                     (4 x 32768 flop;  8 passes = 32 cycles each)
// Modes (template):  V | M | B alone in every wave;  VM / VB: both streams in EVERY wave (the MFMAs first, the FMAs issued
// under them);  V|M and V|B: the waves of a SIMD split in two halves, one half runs V only, the other the MFMA stream only
// (needs WPS >= 2);  V/2, M/2, B/2: the split kernels with the other half idle (the baselines the split modes compare to).
// "overlap" = the combined time sits at max(t_a, t_b); "add" = at t_a + t_b.
//
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -fno-vectorize -o mfma_valu_overlap.bin mfma_valu_overlap.hip
//   ./mfma_valu_overlap.bin            (table)        ./mfma_valu_overlap.bin pmc   (one launch per mode, for rocprofv3 --pmc)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

enum { SV = 1, SM = 2, SB = 4 };

// STREAMS_A: what the "first half" waves (or every wave when SPLIT == 0) run; STREAMS_B: the second half (SPLIT == 1).
template <int STREAMS_A, int STREAMS_B, int SPLIT, int NV>
__global__ __launch_bounds__(1024) void ovl(float* out, int iters, float s) {
    const int w = threadIdx.x >> 6;
    const int wps = blockDim.x >> 8;
    // wave w sits on SIMD w % 4; its index on that SIMD is w / 4: the lower half of the indices is "first half"
    const bool second = SPLIT && (w >> 2) >= (wps >> 1);
    const int streams = second ? STREAMS_B : STREAMS_A;
    bf16x8 ba, bb;
    for (int i = 0; i < 8; ++i) {
        ba[i] = (__bf16)(0.001f * (threadIdx.x & 7) + s);
        bb[i] = (__bf16)(0.002f * i);
    }
    float a[NV];
    for (int i = 0; i < NV; ++i) a[i] = threadIdx.x * 1e-6f + i;
    f32x16 c[4];
    for (int j = 0; j < 4; ++j)
        for (int q = 0; q < 16; ++q) c[j][q] = q * 1e-3f * (j + 1);
    float x = s, y = s * 0.5f;
    // the loop operands live in VGPRs for the whole kernel (opaque to the compiler: nothing is re-materialised from SGPRs inside a loop)
    asm volatile("" : "+v"(x), "+v"(y), "+v"(ba), "+v"(bb));
    if (streams == SV) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < NV; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y));
        }
    } else if (streams == SM) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, c[j], 0, 0, 0);
        }
    } else if (streams == SB) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 4; ++j) c[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ba, bb, c[j], 0, 0, 0);
        }
    } else if (streams == (SV | SM)) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                c[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, c[j], 0, 0, 0);
#pragma unroll
                for (int i = j * (NV / 4); i < (j + 1) * (NV / 4); ++i)
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y));
            }
        }
    } else if (streams == (SV | SB)) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                c[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ba, bb, c[j], 0, 0, 0);
#pragma unroll
                for (int i = j * (NV / 4); i < (j + 1) * (NV / 4); ++i)
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y));
            }
        }
    }
    float r = 0.f;
    for (int i = 0; i < NV; ++i) r += a[i];
    for (int j = 0; j < 4; ++j)
        for (int q = 0; q < 16; ++q) r += c[j][q];
    out[blockIdx.x * 1024 + threadIdx.x] = r;
}

template <int A, int B, int SPLIT, int NV>
static float run(float* d, int iters, int wps, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((ovl<A, B, SPLIT, NV>), dim3(256), dim3(256 * wps), 0, 0, d, iters, 0.999f);
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((ovl<A, B, SPLIT, NV>), dim3(256), dim3(256 * wps), 0, 0, d, iters, 0.999f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    return best;
}

static const char* verdict(float t, float ta, float tb) {
    const float mx = ta > tb ? ta : tb, sm = ta + tb;
    const float pos = (t - mx) / (sm - mx);  // 0 = perfect overlap, 1 = the streams add
    static char buf[8][96];
    static int k = 0;
    char* b = buf[k++ & 7];
    snprintf(b, 96, "%.2f x max, %.2f x sum, overlap position %.2f (0 = max, 1 = sum)", t / mx, t / sm, pos);
    return b;
}

template <int NV>
static void table(float* d, int iters, int wps, int reps) {
    const double waves = 256.0 * 4 * wps;
    const float tv = run<SV, 0, 0, NV>(d, iters, wps, reps);
    const float tm = run<SM, 0, 0, NV>(d, iters, wps, reps);
    const float tb = run<SB, 0, 0, NV>(d, iters, wps, reps);
    const float tvm = run<SV | SM, 0, 0, NV>(d, iters, wps, reps);
    const float tvb = run<SV | SB, 0, 0, NV>(d, iters, wps, reps);
    printf("-- %d wave(s) per SIMD, NV = %d v_fma_f32 per iteration (4 MFMAs per iteration), %d iterations\n", wps, NV, iters);
    printf("V  alone               %8.3f ms  %7.1f TFLOP/s (v_fma_f32)   %.2f cycles per v_fma_f32 per SIMD at 2.4 GHz\n", tv,
           waves * iters * NV * 128.0 / tv / 1e9, tv * 1e-3 * 2.4e9 / ((double)iters * NV * wps));
    printf("M  alone               %8.3f ms  %7.1f TFLOP/s (v_mfma_f32_32x32x2_f32)\n", tm, waves * iters * 4 * 4096.0 / tm / 1e9);
    printf("B  alone               %8.3f ms  %7.1f TFLOP/s (v_mfma_f32_32x32x16_bf16)\n", tb, waves * iters * 4 * 32768.0 / tb / 1e9);
    printf("V+M in every wave      %8.3f ms  %s\n", tvm, verdict(tvm, tv, tm));
    printf("V+B in every wave      %8.3f ms  %s\n", tvb, verdict(tvb, tv, tb));
    if (wps >= 2) {
        const float hv = run<SV, 0, 1, NV>(d, iters, wps, reps);
        const float hm = run<0, SM, 1, NV>(d, iters, wps, reps);
        const float hb = run<0, SB, 1, NV>(d, iters, wps, reps);
        const float svm = run<SV, SM, 1, NV>(d, iters, wps, reps);
        const float svb = run<SV, SB, 1, NV>(d, iters, wps, reps);
        printf("V/2 (other half idle)  %8.3f ms\n", hv);
        printf("M/2 (other half idle)  %8.3f ms\n", hm);
        printf("B/2 (other half idle)  %8.3f ms\n", hb);
        printf("V | M split waves      %8.3f ms  %s\n", svm, verdict(svm, hv, hm));
        printf("V | B split waves      %8.3f ms  %s\n", svb, verdict(svb, hv, hb));
    }
}

int main(int argc, char** argv) {
    float* d;
    hipMalloc(&d, 256 * 1024 * 4);
    if (argc > 1 && !strcmp(argv[1], "pmc")) {  // one warm-up + one launch per mode at 4 and 2 waves per SIMD: read the counters per kernel name
        const int iters = 20000;
        for (int wps = 4; wps >= 2; wps -= 2) {
            run<SV, 0, 0, 32>(d, iters, wps, 1);
            run<SM, 0, 0, 32>(d, iters, wps, 1);
            run<SB, 0, 0, 32>(d, iters, wps, 1);
            run<SV | SM, 0, 0, 32>(d, iters, wps, 1);
            run<SV | SB, 0, 0, 32>(d, iters, wps, 1);
            run<SV, SM, 1, 32>(d, iters, wps, 1);
            run<SV, SB, 1, 32>(d, iters, wps, 1);
        }
        return 0;
    }
    const int iters = 20000, reps = 3;
    table<32>(d, iters, 4, reps);
    table<32>(d, iters, 2, reps);
    table<32>(d, iters, 1, reps);
    table<16>(d, iters, 2, reps);
    table<16>(d, iters, 4, reps);
    return 0;
}
