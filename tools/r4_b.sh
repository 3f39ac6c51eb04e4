#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r4_b; mkdir -p $O
rm -f gpurun_out/grad_err.jsonl
timeout 900 python -m pytest tests -m gpu -q > $O/tests.log 2>&1; echo "pytest rc=$?"; tail -8 $O/tests.log
cp gpurun_out/grad_err.jsonl $O/ 2>/dev/null
python tools/r4_side.py 2>/dev/null | tail -1
RLHIP_STREAM_NT_STORES=1 python tools/r4_side.py 2>/dev/null | tail -1
RLHIP_STREAM_UNROLL2=1 python tools/r4_side.py 2>/dev/null | tail -1
RLHIP_STREAM_NT_STORES=1 RLHIP_STREAM_UNROLL2=1 RLHIP_GATHER_GENERIC=1 python tools/r4_side.py 2>/dev/null | tail -1
