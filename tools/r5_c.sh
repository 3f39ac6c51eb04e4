#!/bin/bash
# round 5, third contact (short): gather access-pattern probe (variant 4 = one 64-byte line per sample, timing only), request-size
# counters of the gather, Adam fold A / B on one box, the re-barred config-3 rollout test + the new multi-rank tests
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export GRAFT_REPO_ROOT=$PWD TMPDIR=/tmp
R=$PWD; O=gpurun_out/r5_c; mkdir -p $O
for v in 0 4 0 4; do RLHIP_GATHER_VARIANT=$v timeout 120 python tools/r5_gather.py 2>/dev/null | tail -1; done | tee $O/gather_variants.txt
for f in 1 0 1 0; do echo "fold=$f $(RLHIP_ADAM_FOLD=$f timeout 300 python tools/adam_grid_ab.py 2>/dev/null | tail -1)"; done | tee $O/adam_fold.txt
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $R/$O/gather_req -o pmc -- python $R/tools/r5_gather.py > $R/$O/gather_req.log 2>&1)
python3 tools/pmc_last.py $O/gather_req gather_rec_kernel 8 2>&1 | tee $O/gather_req.txt; tail -3 $O/gather_req.log
( time timeout 1500 python -m pytest tests/test_gpu_bench_shapes.py tests/test_gpu_run.py tests/test_gpu_abi_host.py tests/test_gpu_edges.py tests/test_gpu_explorers.py -q -m gpu 2>&1 | tail -30 ) > $O/tests.log 2>&1; tail -30 $O/tests.log
cp gpurun_out/bench_shape_margins.jsonl $O/ 2>/dev/null
