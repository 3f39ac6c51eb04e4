#!/bin/bash
# round 4, contact D: split rollout (actor wave + critic wave per SIMD) vs the one-wave kernel; bench leg order A / B
export PYTHONPATH=$GRAFT_REPO_ROOT/reinforcementlearning.jl_amd:$GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r4_d; mkdir -p $O
cd $GRAFT_REPO_ROOT
./tools/micro/wave_simd_map.bin > $O/simd_map.txt 2>&1; cat $O/simd_map.txt
timeout 600 python -m pytest tests/test_gpu_learners.py tests/test_gpu_run.py -q -x -m gpu 2>&1 | tail -4
RLHIP_ROLLOUT_SPLIT=0 timeout 600 python -m pytest tests/test_gpu_learners.py -q -x -m gpu -k "rollout" 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
for sp in 0 1; do
  RLHIP_ROLLOUT_SPLIT=$sp timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_sp$sp -o p -- python $GRAFT_REPO_ROOT/tools/rollout_one.py 60 > $O/prof_sp$sp.log 2>&1
  f=$(find $O/prof_sp$sp -name "*kernel_stats.csv" | head -1)
  echo "split=$sp"; grep -i "rollout" $f | cut -c1-60,200-400 | head -3; grep -i rollout $f | awk -F, '{print $(NF-7), $(NF-6), $(NF-5), $(NF-4)}' | head -2
done
cd $GRAFT_REPO_ROOT
RLHIP_ROLLOUT_SPLIT=0 python bench.py --no-extras --steps 200 --warmup 20 > $O/bench_sp0.json 2>$O/bench_sp0.err; cat $O/bench_sp0.json | cut -c1-200
RLHIP_ROLLOUT_SPLIT=1 python bench.py --no-extras --steps 200 --warmup 20 > $O/bench_sp1.json 2>$O/bench_sp1.err; cat $O/bench_sp1.json | cut -c1-200
RLHIP_BENCH_EXTRAS_FIRST=0 python bench.py --steps 20 --warmup 5 > $O/bench_after.json 2>$O/bench_after.err; cut -c1-200 $O/bench_after.json
RLHIP_BENCH_EXTRAS_FIRST=1 python bench.py --steps 20 --warmup 5 > $O/bench_first.json 2>$O/bench_first.err; cut -c1-200 $O/bench_first.json
python - <<'PY'
import json,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/r4_d/"
for f in ("bench_after.json","bench_first.json"):
    try:
        d=json.loads(open(O+f).read().strip().splitlines()[-1]); print(f, d["ms_per_step"], d["kernels"], d["roofline"]["frac"])
    except Exception as e: print(f, "ERR", e)
PY
