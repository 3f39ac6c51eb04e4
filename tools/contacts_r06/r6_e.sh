#!/bin/bash
# round 6 contact E: tools/micro/adam_stream (where the streaming Adam's time goes: memory pattern vs arithmetic; stream mix; placement)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export GRAFT_REPO_ROOT=$PWD TMPDIR=/tmp
O=gpurun_out/r6_e; mkdir -p $O
timeout 900 tools/micro/adam_stream.bin 26 2>&1 | grep "mix" | tee $O/adam_stream_26_mix.txt
