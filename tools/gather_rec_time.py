"""round 5: the small transition gather of bench.roofline_hbm_side alone (2^20 CartPole samples out of a 256 x 4096 record
ring), for the RLHIP_GATHER_VARIANT A / B (one process per variant: the hook is read once) and the PMC passes"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "reinforcementlearning.jl_amd")); sys.path.insert(0, ROOT)
import torch, rlhip, bench
from rlhip._lib import call
from rlhip.ops import ptr, stream_ptr
from rlhip.trajectory import CircularArraySARTSTraces

lib, s = rlhip._lib.lib, stream_ptr()
n_env, cap = 4096, int(os.environ.get("R5_GATHER_CAP", "256"))
batch = 1 << 20
tr = CircularArraySARTSTraces(capacity=cap, n_env=n_env, obs_dim=4)
tr.records.normal_()
tr.rb.len_sa, tr.rb.len_rt = cap + 1, cap
idx = tr.sample_indices(batch, seed=11, draw_ctr=0)
bufs = tr.gather(idx)
c = [1]


def smp():
    call("rlhip_ring_sample_indices", C.byref(tr.rb), batch, 11, c[0], ptr(idx), s)
    c[0] += 1


def sg():
    smp()
    call("rlhip_ring_gather", C.byref(tr.rb), ptr(idx), batch, ptr(bufs[0]), ptr(bufs[1]), ptr(bufs[2]), ptr(bufs[3]), ptr(bufs[4]), s)


ms = bench.event_time_ms(sg, 10, lib, s, 0.05) - bench.event_time_ms(smp, 10, lib, s)
print(f"variant {os.environ.get('RLHIP_GATHER_VARIANT', '0')} cap {cap}: gather {ms * 1e3:.1f} us per 2^20 samples = "
      f"{82 * batch / (ms * 1e-3) / 1e9:.0f} GB/s algorithmic ({82 * batch / (ms * 1e-3) / 8e12:.3f} of 8 TB/s)")
