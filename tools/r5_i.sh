#!/bin/bash
# round 5, contact i: the bench in the driver's form (--steps 20 --warmup 5, all legs) with and without the 60 pre-heat steps
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export GRAFT_REPO_ROOT=$PWD TMPDIR=/tmp
O=gpurun_out/r5_i; mkdir -p $O
for v in 0 60 0 60; do
    r=$(RLHIP_BENCH_PREHEAT_STEPS=$v timeout 400 python bench.py --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['preheat_steps'], d['roofline']['frac'])")
    echo "preheat=$v $r"
done | tee $O/ab.txt
timeout 400 python bench.py > $O/bench_default.json 2>/dev/null; python -c "
import json; d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1]); print('default form', d['ms_per_step'], d['value'])"
