#!/bin/bash
# round 6 contact G: same-box A / B of the Polyak launch (rows per wave x store policy), the Adam micro-benchmark on the same box, and the
# fabric counters of the stream-mix kernels (tools/contacts_r06/r6_f.sh)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export GRAFT_REPO_ROOT=$PWD TMPDIR=/tmp
O=gpurun_out/r6_g; mkdir -p $O
for rep in 1 2 3; do for v in 0 1 2 3; do RLHIP_POLYAK_VARIANT=$v python tools/polyak_ab.py 2>&1 | grep variant | tee -a $O/polyak_ab.txt; done; done
timeout 600 tools/micro/adam_stream.bin 26 2>&1 | grep -v differ | grep "K0\|K1\|Polyak\|Adam: g" | tee $O/adam_same_box.txt
bash tools/contacts_r06/r6_f.sh 2>&1 | tail -40
