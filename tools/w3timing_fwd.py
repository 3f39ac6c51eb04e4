"""per-phase ticks of one steady-state tile of ppo3w_fwd_kernel<actor> from a -DRLHIP_W3_TIMING build (RLHIP_LIB_PATH), for the default
schedule and for the merged-barrier schedule of profiles/attic/r06_ppo3w_merged_barrier.patch (W3_MERGED=1 in the environment selects
the labels).  clock64() = s_memtime (100 MHz): proportions only; thread 0 of workgroup 0 = WAVE 0, the wave that runs the loss line."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "reinforcementlearning.jl_amd")]
import torch, rlhip
env = rlhip.HipVecEnv("pendulum", 4096, seed=7)
pol = rlhip.PPOPolicy(env, update_freq=128, hidden=256, seed=7, layers=3, clip_range=0.1)
pol.rollout_(); pol.update_(); torch.cuda.synchronize()
st = (C.c_longlong * 48)()
fn = rlhip._lib.lib.rlhip_debug_w3_stamps
fn.restype = C.c_int32
assert fn(st) == 0
b = list(st)[:16]
if os.environ.get("W3_MERGED") == "1":
    seq = [(0, 1, "merged barrier A + D (wave 0 arrives last: its loss line + its layer 1 sit in front)"), (1, 2, "head backward of the PREVIOUS tile + dZ2 stores"),
           (2, 4, "MFMA + bias / act + head partials + barrier C"), (4, 5, "loss line (wave 0)"), (5, 6, "layer 1 of the NEXT tile")]
    total = (0, 6)
else:
    seq = [(1, 2, "layer 1 + barrier A"), (2, 3, "MFMA + bias / act"), (3, 4, "head partials + barrier C"), (4, 5, "loss line (wave 0) + barrier D"),
           (5, 6, "head backward + fragment stores"), (6, 7, "row copy-out")]
    total = (1, 7)
for i, j, n in seq:
    print(f"   {n:90s} {b[j] - b[i]:7d}")
print(f"   {'stamped span of the tile':90s} {b[total[1]] - b[total[0]]:7d}")
print(f"   {'prologue':90s} {b[9] - b[8]:7d}   tile loop {b[10] - b[9]:7d}")
