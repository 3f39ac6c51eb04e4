"""The instruction-level facts the learner kernels rely on, checked on the GPU by stand-alone programs (tools/micro/*.hip,
cross-compiled by __graft_entry__.build()):

  mfma_f32_l1    v_mfma_f32_32x32x2_f32 with the bias as the accumulator's initial value == the oracle's fmaf chain
                 b1 + w0 x0 + w1 x1 + ... bit for bit, in both operand orders (layer 1 of ppo3w.hip and of the PPO tile)
  mfma_f32_4x4   the same for v_mfma_f32_4x4x1_16b_f32 (operand images + exactness; measured no faster than the VALU, not used)
  tanh_sel       the PPO tile's branch-free tanh == ocml tanhf for all 2^32 float bit patterns
  log_sampling   log_f64_sampling (the Float64 log of the sampling path's log-sum-exp, csrc/select_device.h) == the host libm's log
                 after rounding to Float32 for EVERY Float32 in [1, 64]; within 1 ulp of a long-double reference on 2^24 doubles
  trig_f32arg    sincos_f32arg / jl_mod_2pi_f32arg (csrc/env_device.h: Float64 sin / cos / mod 2 pi of a Float32 argument, ~45 instructions
                 instead of ocml's general routines) == the host libm after rounding, for EVERY Float32 with |x| <= 2^16 (2.4e9 values)
  wave_simd_map  wave w and wave w + 4 of a 512-thread workgroup share a SIMD (a PERFORMANCE premise of the two-wave PPO rollout,
                 csrc/ppo.hip: its actor wave and critic wave interleave on one SIMD; results do not depend on it)
"""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["mfma_f32_l1", "mfma_f32_4x4", "tanh_sel", "wave_simd_map", "log_sampling", "trig_f32arg"])
def test_micro_check(name):
    exe = os.path.join(ROOT, "tools", "micro", name + ".bin")
    assert os.path.exists(exe), f"{exe} is missing: run python -c 'import __graft_entry__ as g; g.build()'"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    print(r.stdout)
    if name == "wave_simd_map" and r.returncode == 1:  # a PERFORMANCE premise (results do not depend on it): report, do not fail
        pytest.xfail("this device does not pair wave w and w + 4 of a workgroup on one SIMD: the two-wave rollout runs slower here\n" + r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr  # every program exits non-zero on the first kind of mismatch it counts
