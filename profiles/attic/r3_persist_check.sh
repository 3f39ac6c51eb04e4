#!/bin/bash
# round 3: GPU contact of the PPO update paths (tests, A/B bench, timeline)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/${1:-r3a}
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_persist.py -q 2>&1 | tail -25 > $O/tests.log
timeout 600 python -m pytest tests/test_gpu_learners.py -x -q 2>&1 | tail -12 >> $O/tests.log
cat $O/tests.log
RLHIP_PPO_PERSIST=1 timeout 300 python bench.py --steps 200 --warmup 20 --no-extras > $O/bench_persist.json 2> $O/bench_persist.err
RLHIP_PPO_PERSIST=0 timeout 300 python bench.py --steps 200 --warmup 20 --no-extras > $O/bench_twolaunch.json 2> $O/bench_twolaunch.err
python - <<PY
import json
for f in ("bench_persist", "bench_twolaunch"):
    try:
        d = json.loads(open("$O/%s.json" % f).read().strip().splitlines()[-1])
        print(f, d["ms_per_step"], d["value"], d["final_loss"], d["mean_episode_len_last_rollout"])
    except Exception as e:
        print(f, "failed", e)
PY
RLHIP_PPO_PERSIST=1 timeout 300 python tools/persist_timeline.py > $O/timeline.txt 2>&1; grep -v "workgroup 0:" $O/timeline.txt
