"""stamped timeline of one fused DQN vec-step (act launch, then the one-launch optimise!): thread 0 of every gradient workgroup and of
workgroup 0 of the act kernel, from a -DRLHIP_DQN_TIMING build of csrc/dqn.hip + csrc/dqn_act.hip (RLHIP_LIB_PATH=gpurun_ab/libT.so):
    python tools/dqn_timeline.py [batch]
s_memrealtime ticks (10 ns); the stamps pin the schedule, so proportions."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "reinforcementlearning.jl_amd")); sys.path.insert(0, ROOT)
import torch, rlhip
n = 4096
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 512
env = rlhip.CartPoleEnv(n, seed=5)
net = rlhip.HipApproximator(4, 128, 2, seed=5, layers=2)
learner = rlhip.DQNLearner(rlhip.TargetNetwork(net, sync_freq=100000), batchsize=batch, min_replay_history=n, seed=5)
policy = rlhip.QBasedPolicy(learner, rlhip.EpsilonGreedyExplorer(0.01, kind="exp", decay_steps=500, seed=5))
agent = rlhip.Agent(policy, rlhip.Trajectory(rlhip.CircularArraySARTSTraces(capacity=256, n_env=n, obs_dim=4)))
rlhip.run_fused_dqn(agent, env, rlhip.StopAfterNSteps(500))
torch.cuda.synchronize()
lib = rlhip._lib.lib
for f, t in ((lib.rlhip_debug_dqn_stamps, C.c_longlong * 1024), (lib.rlhip_debug_act_stamps, C.c_longlong * 4)):
    f.restype, f.argtypes = C.c_int32, [C.POINTER(C.c_longlong)]
g, a = (C.c_longlong * 1024)(), (C.c_longlong * 4)()
assert lib.rlhip_debug_dqn_stamps(g) == 0 and lib.rlhip_debug_act_stamps(a) == 0
nb = (batch + 63) // 64
rows = [[g[b * 16 + k] for k in range(16)] for b in range(min(nb, 64))]
t0 = a[0]
last = max(range(len(rows)), key=lambda b: rows[b][12])
print(f"batch {batch}: {nb} gradient workgroups; times in ns since the act kernel's workgroup 0 started")
print(f"  act kernel workgroup 0: start 0, end {(a[1] - t0) * 10}")
names = ["start", "gather + weights staged", "Q / Q_target partials", "TD line", "weight gradients done", "row stores issued", "row stores drained",
         "counted out", "[last] partial rows folded", "[last] norm", "", "[last] Adam stores issued", "[last] drained"]
for b in sorted(set([0, last, len(rows) - 1])):
    r = rows[b]
    print(f"  gradient workgroup {b}{' (departed last)' if b == last else ''}:")
    for k, nm in enumerate(names):
        if nm and r[k] and (k < 8 or b == last):
            print(f"     {nm:32s} {(r[k] - t0) * 10:7d}")
r = rows[last]
print(f"  shader clock over stamps 0 -> 7 of that workgroup: {(r[15] - r[14]) / ((r[7] - r[0]) * 10.0):.3f} GHz (s_memtime ticks per ns of s_memrealtime)")
starts = [(r[0] - t0) * 10 for r in rows]
outs = [(r[7] - t0) * 10 for r in rows]
print(f"  gradient workgroups start {min(starts)} .. {max(starts)}, count out {min(outs)} .. {max(outs)}")
