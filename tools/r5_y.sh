#!/bin/bash
# round 5, contact y: store policy of the gradient kernel's partial rows (A = plain, C = device-scope write-through): the DEFAULT bench
# command per build (the legs in front of the timed region put the device in its steady clock state; --no-extras did not on these boxes)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export GRAFT_REPO_ROOT=$PWD TMPDIR=/tmp
O=gpurun_out/r5_y; mkdir -p $O
L=reinforcementlearning.jl_amd/lib/librlhip.so
cp $L /tmp/lib_keep.so
for v in A C A C A C; do
    cp gpurun_ab/lib$v.so $L
    r=$(timeout 300 python bench.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('final_loss'), (d.get('kernels') or {}).get('per_microbatch_us'), (d.get('kernels') or {}).get('update_us'))")
    echo "$v $r"
done | tee $O/ab.txt
cp /tmp/lib_keep.so $L
