#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/gab; mkdir -p $O
(for w in stacked frames gae env; do timeout 300 python tools/gather_ab.py $w 2>&1 | grep "variant\|Error\|error"; done) | tee $O/ab2.txt
