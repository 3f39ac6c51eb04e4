#!/bin/bash
# same-box A/B of two builds of the library: gpurun_ab/libA.so vs gpurun_ab/libB.so (copied over the in-tree library in turn)
# usage: tools/ab.sh [bench args]   (default: the headline bench without extras, 300 steps)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
L=reinforcementlearning.jl_amd/lib/librlhip.so
cp $L /tmp/lib_keep.so
for v in A B A B A B; do
    cp gpurun_ab/lib$v.so $L
    r=$(timeout 300 python bench.py --steps 300 --warmup 30 --no-extras "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('final_loss'))")
    echo "$v $r"
done
cp /tmp/lib_keep.so $L
