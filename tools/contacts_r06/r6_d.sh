#!/bin/bash
# round 6 contact D: n-step tests, the new edge tests (65+ streams, rejected pushes), how long the plain-C host takes to start
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export GRAFT_REPO_ROOT=$PWD TMPDIR=/tmp
O=gpurun_out/r6_d; mkdir -p $O
rm -f gpurun_out/bench_shape_margins.jsonl
( time timeout 1200 python -m pytest tests/test_gpu_nstep.py tests/test_gpu_edges.py tests/test_gpu_sumtree.py tests/test_gpu_stackframes.py tests/test_gpu_parity.py -q -m gpu --durations=8 2>&1 | tail -40 ) > $O/tests.log 2>&1; cat $O/tests.log
cp gpurun_out/bench_shape_margins.jsonl $O/ 2>/dev/null; cat $O/bench_shape_margins.jsonl | cut -c1-600
( time ./tests/abi_host/abi_host.bin /tmp/abi1.bin > /dev/null ) 2>&1 | grep real
( time ./tests/abi_host/abi_host.bin /tmp/abi2.bin > /dev/null ) 2>&1 | grep real
