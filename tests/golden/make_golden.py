#!/usr/bin/env python3
"""Transcription of the reference's own known-answer tests for the hot path into JSON fixtures.

The reference is Julia and cannot be executed in this image (no `julia`), so these vectors are
copied literally from its test files; every case carries the file:line it was copied from
(paths relative to /root/reference/src/ReinforcementLearningCore/test/).  Run this script to
regenerate tests/golden/*.json -- it reads nothing from /root/reference at run time.

Matrix convention: Julia literals `[a b; c d]` are row-major text; they are stored here as nested
row lists ("rows") and converted to column-major by the tests.  `reshape(1:9, 3, 3)` is column-major
fill: rows [[1,4,7],[2,5,8],[3,6,9]].
"""
import json
import os

HERE = os.path.dirname(os.path.abspath(__file__))

R33 = [[1, 4, 7], [2, 5, 8], [3, 6, 9]]  # reshape(1:9, 3, 3)
V43 = [[1, 5, 9], [2, 6, 10], [3, 7, 11], [4, 8, 12]]  # reshape(1:12, 4, 3)
V34 = [[1, 4, 7, 10], [2, 5, 8, 11], [3, 6, 9, 12]]  # reshape(1:12, 3, 4)
TERM33 = [[0, 1, 0], [1, 0, 1], [0, 1, 0]]  # [false true false; true false true; false true false]
INIT3 = [-2.0, 0.0, 2.0]

scans = {
    "source": "utils/base.jl",
    "discount_rewards": [
        # utils/base.jl:23-25
        dict(src="utils/base.jl:24", reward=[1.0], gamma=0.5, expect=[1.0]),
        dict(src="utils/base.jl:25", reward=[1.0], gamma=0.5, init=2.0, expect=[2.0]),
        # :27-35
        dict(src="utils/base.jl:29", reward=[1, 2, 3], gamma=0.5, expect=[2.75, 3.5, 3.0]),
        dict(src="utils/base.jl:30", reward=[1, 2, 3], gamma=0.5, init=4.0, expect=[3.25, 4.5, 5.0]),
        dict(src="utils/base.jl:32-33", reward=[1, 2, 3], gamma=0.5, terminal=[0, 1, 0], init=2.0,
             expect=[2.0, 2.0, 4.0]),
        dict(src="utils/base.jl:34-35", reward=[1, 2, 3], gamma=0.5, terminal=[1, 0, 1], init=2.0,
             expect=[1.0, 3.5, 3.0]),
        # 2D :41-61
        dict(src="utils/base.jl:45", reward_rows=R33, gamma=0.5, expect_error=True),
        dict(src="utils/base.jl:46-47", reward_rows=R33, gamma=0.5, dims=1,
             expect_rows=[[2.75, 8.0, 13.25], [3.5, 8.0, 12.5], [3.0, 6.0, 9.0]]),
        dict(src="utils/base.jl:48-49", reward_rows=R33, gamma=0.5, dims=2,
             expect_rows=[[4.75, 7.5, 7.0], [6.5, 9.0, 8.0], [8.25, 10.5, 9.0]]),
        dict(src="utils/base.jl:50-51", reward_rows=R33, gamma=0.5, dims=1, init=INIT3,
             expect_rows=[[2.5, 8.0, 13.5], [3.0, 8.0, 13.0], [2.0, 6.0, 10.0]]),
        dict(src="utils/base.jl:52-53", reward_rows=R33, gamma=0.5, dims=2, init=INIT3,
             expect_rows=[[4.5, 7.0, 6.0], [6.5, 9.0, 8.0], [8.5, 11.0, 10.0]]),
        dict(src="utils/base.jl:56-57", reward_rows=R33, gamma=0.5, dims=1, terminal_rows=TERM33,
             expect_rows=[[2.0, 4.0, 11.0], [2.0, 8.0, 8.0], [3.0, 6.0, 9.0]]),
        dict(src="utils/base.jl:58-59", reward_rows=R33, gamma=0.5, dims=1, terminal_rows=TERM33,
             init=INIT3, expect_rows=[[2.0, 4.0, 11.0], [2.0, 8.0, 8.0], [2.0, 6.0, 10.0]]),
        dict(src="utils/base.jl:60-61", reward_rows=R33, gamma=0.5, dims=2, terminal_rows=TERM33,
             init=INIT3, expect_rows=[[3.0, 4.0, 6.0], [2.0, 9.0, 8.0], [6.0, 6.0, 10.0]]),
    ],
    "discount_rewards_reduced": [
        dict(src="utils/base.jl:66", reward=[1.0], gamma=0.5, expect=[1.0]),
        dict(src="utils/base.jl:69", reward=[1, 2, 3], gamma=0.5, expect=[2.75]),
        dict(src="utils/base.jl:70", reward=[1, 2, 3], gamma=0.5, init=4.0, expect=[3.25]),
        dict(src="utils/base.jl:71", reward=[1, 2, 3], gamma=0.5, terminal=[0, 1, 0], expect=[2.0]),
        dict(src="utils/base.jl:72-77", reward=[1, 2, 3], gamma=0.5, terminal=[0, 1, 0], init=4.0,
             expect=[2.0]),
        dict(src="utils/base.jl:84", reward_rows=R33, gamma=0.5, expect_error=True),
        dict(src="utils/base.jl:86", reward_rows=R33, gamma=0.5, dims=1, expect=[2.75, 8.0, 13.25]),
        dict(src="utils/base.jl:87", reward_rows=R33, gamma=0.5, dims=2, expect=[4.75, 6.5, 8.25]),
        dict(src="utils/base.jl:88-94", reward_rows=R33, gamma=0.5, dims=1, terminal_rows=TERM33,
             init=INIT3, expect=[2.0, 4.0, 11.0]),
        dict(src="utils/base.jl:95-101", reward_rows=R33, gamma=0.5, dims=2, terminal_rows=TERM33,
             init=INIT3, expect=[3.0, 2.0, 6.0]),
    ],
    "generalized_advantage_estimation": [
        dict(src="utils/base.jl:105-106", reward=[1.0], values=[2.0, 3.0], gamma=0.5, lam=0.3,
             expect=[0.5]),
        dict(src="utils/base.jl:108-109", reward=[1.0, 1.0], values=[1, 2, 3], gamma=0.5, lam=0.3,
             expect=[1.075, 0.5]),
        dict(src="utils/base.jl:111-112", reward=[1, 2, 3], values=[1, 2, 3, 4], gamma=0.5, lam=0.3,
             expect=[1.27, 1.8, 2.0]),
        dict(src="utils/base.jl:114-120", reward=[1, 2, 3], values=[1, 2, 3, 4], gamma=0.5, lam=0.3,
             terminal=[1, 0, 1], expect=[0.0, 1.5, 0.0]),
        dict(src="utils/base.jl:129", reward_rows=R33, values_rows=V43, gamma=0.5, lam=0.3,
             expect_error=True),
        dict(src="utils/base.jl:130-131", reward_rows=R33, values_rows=V43, gamma=0.5, lam=0.3, dims=1,
             expect_rows=[[1.27, 2.4425, 3.615], [1.8, 2.95, 4.1], [2.0, 3.0, 4.0]]),
        dict(src="utils/base.jl:133-135", reward_rows=R33, values_rows=V34, gamma=0.5, lam=0.3, dims=2,
             expect_rows=[[2.6375, 4.25, 5.0], [3.22375, 4.825, 5.5], [3.81, 5.4, 6.0]]),
        dict(src="utils/base.jl:137-147", reward_rows=R33, values_rows=V43, gamma=0.5, lam=0.3, dims=1,
             terminal_rows=TERM33,
             expect_rows=[[1.0, -1.0, 2.7], [0.0, 2.35, -2.0], [2.0, -1.0, 4.0]]),
        dict(src="utils/base.jl:149-151", reward_rows=R33, values_rows=V34, gamma=0.5, lam=0.3, dims=2,
             expect_rows=[[2.6375, 4.25, 5.0], [3.22375, 4.825, 5.5], [3.81, 5.4, 6.0]]),
    ],
}

INF = "inf"
NINF = "-inf"
select = {
    "source": "utils/base.jl, policies/explorers/epsilon_greedy_explorer.jl",
    # indices are Julia 1-based, exactly as written in the reference tests
    "find_all_max": [
        dict(src="utils/base.jl:3", x=[NINF, NINF, NINF], vmax=NINF, idx=[1, 2, 3]),
        dict(src="utils/base.jl:4", x=[NINF, NINF, NINF], mask=[1, 0, 1], vmax=NINF, idx=[1, 3]),
        dict(src="utils/base.jl:6", x=[INF, INF, INF], vmax=INF, idx=[1, 2, 3]),
        dict(src="utils/base.jl:7", x=[INF, INF, INF], mask=[1, 1, 0], vmax=INF, idx=[1, 2]),
        dict(src="utils/base.jl:9", x=[INF, 0, INF], vmax=INF, idx=[1, 3]),
        dict(src="utils/base.jl:10", x=[INF, 0, INF], mask=[0, 1, 0], vmax=0, idx=[2]),
        dict(src="utils/base.jl:12", x=[0, 1, 2, 1, 2, 1, 0], vmax=2, idx=[3, 5]),
        dict(src="utils/base.jl:13", x=[0, 1, 2, 1, 2, 1, 0], mask=[1, 1, 0, 0, 0, 1, 1], vmax=1,
             idx=[2, 6]),
    ],
    "get_eps": [
        # EpsilonGreedyExplorer(kind, eps_init=0.9, eps_stable=0.1, warmup_steps=100, decay_steps=100)
        dict(src="epsilon_greedy_explorer.jl:8", kind="linear", step=50, expect=0.9),
        dict(src="epsilon_greedy_explorer.jl:9", kind="linear", step=100, expect=0.9),
        dict(src="epsilon_greedy_explorer.jl:10", kind="linear", step=150, expect=0.5),
        dict(src="epsilon_greedy_explorer.jl:11", kind="linear", step=200, expect=0.1),
        dict(src="epsilon_greedy_explorer.jl:15", kind="exp", step=50, expect=0.9),
        dict(src="epsilon_greedy_explorer.jl:17", kind="exp", step=150, expect=0.5852245277701068),
        dict(src="epsilon_greedy_explorer.jl:18", kind="exp", step=2000, expect=0.1, atol=1e-2),
    ],
    "get_eps_params": dict(eps_init=0.9, eps_stable=0.1, warmup_steps=100, decay_steps=100),
    "prob": [
        # explorer at step = 1 (<= warmup) so eps = eps_init = 0.9; values = [0.1, 0.5, 0.5, 0.3]
        dict(src="epsilon_greedy_explorer.jl:46-48", values=[0.1, 0.5, 0.5, 0.3], eps=0.9, is_break_tie=1,
             expect=[0.225, 0.275, 0.275, 0.225]),
        dict(src="epsilon_greedy_explorer.jl:53-55", values=[0.1, 0.5, 0.5, 0.3], eps=0.9, is_break_tie=0,
             expect=[0.225, 0.32499999999999996, 0.225, 0.225]),
        # GreedyExplorer == eps 0
        dict(src="epsilon_greedy_explorer.jl:68-70", values=[0.1, 0.5, 0.5, 0.3], eps=0.0, is_break_tie=0,
             expect=[0.0, 1.0, 0.0, 0.0]),
    ],
    "greedy_plan": [
        dict(src="epsilon_greedy_explorer.jl:62-64", values=[0.1, 0.5, 0.5, 0.3], expect=2),
    ],
    "stop_after_n_steps": dict(src="core/stop_conditions.jl:8", n=10, calls=20, n_true=11),
    "target_sync": [
        # policies/learners/target_network.jl:57-72: sync_freq = 2 -> n_optimise 1 then 0
        dict(src="policies/learners/target_network.jl:57-72", sync_freq=2, counters=[1, 0]),
        # :74-102: sync_freq = 3 -> target equals model only after the 3rd optimise!, counter reset
        dict(src="policies/learners/target_network.jl:74-102", sync_freq=3, counters=[1, 2, 0]),
    ],
}

# Random123 known-answer vectors for Philox4x32-10 (kat_vectors of the Random123 distribution;
# Salmon, Moraes, Dror, Shaw, "Parallel random numbers: as easy as 1, 2, 3", SC'11).
philox = {
    "source": "Random123 kat_vectors (philox4x32 10)",
    "cases": [
        dict(ctr=["00000000", "00000000", "00000000", "00000000"], key=["00000000", "00000000"],
             out=["6627e8d5", "e169c58d", "bc57ac4c", "9b00dbd8"]),
        dict(ctr=["ffffffff", "ffffffff", "ffffffff", "ffffffff"], key=["ffffffff", "ffffffff"],
             out=["408f276d", "41c83b0e", "a20bc7c6", "6d5451fd"]),
        dict(ctr=["243f6a88", "85a308d3", "13198a2e", "03707344"], key=["a4093822", "299f31d0"],
             out=["d16cfe09", "94fdcceb", "5001e420", "24126ea1"]),
    ],
}

for name, obj in (("scans", scans), ("select", select), ("philox", philox)):
    with open(os.path.join(HERE, name + ".json"), "w") as f:
        json.dump(obj, f, indent=1)
        f.write("\n")
print("wrote scans.json select.json philox.json")
