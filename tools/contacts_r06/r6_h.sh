#!/bin/bash
# round 6 contact H: does the DATA matter to the stream-mix sweeps? (random floats vs the zeros of hipMemset), same box; Polyak in the library beside it
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export GRAFT_REPO_ROOT=$PWD TMPDIR=/tmp
O=gpurun_out/r6_h; mkdir -p $O
timeout 600 tools/micro/adam_stream.bin 26 2>&1 | grep "mix" | tee $O/mix_random.txt
ADAM_STREAM_ZEROS=1 timeout 600 tools/micro/adam_stream.bin 26 2>&1 | grep "mix" | grep "4352\|mix2" | tee $O/mix_zeros.txt
for v in 0 1 3; do RLHIP_POLYAK_VARIANT=$v python tools/polyak_ab.py 2>&1 | grep variant | tee -a $O/polyak_ab.txt; done
