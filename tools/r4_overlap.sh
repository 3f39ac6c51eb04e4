#!/bin/bash
# round 4, VERDICT item 1(a): the corrected VALU || MFMA measurement -- the table, then one PMC pass per counter group
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r4_overlap
B=tools/micro/mfma_valu_overlap.bin
$B > gpurun_out/r4_overlap/table.txt 2>&1
cat gpurun_out/r4_overlap/table.txt
R=$PWD
cd /tmp && export TMPDIR=/tmp
for pass in "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INST_CYCLES_VMEM"; do
    tag=$(echo $pass | cut -d' ' -f1)
    rocprofv3 --kernel-trace --pmc $pass --output-format csv -d $R/gpurun_out/r4_overlap/pmc_$tag -o pmc -- $R/$B pmc > $R/gpurun_out/r4_overlap/pmc_$tag.log 2>&1
    f=$(find $R/gpurun_out/r4_overlap/pmc_$tag -name "*counter_collection.csv" | head -1)
    python3 - "$f" <<'PY' | tee -a $R/gpurun_out/r4_overlap/pmc.txt
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
order=[]
for r in csv.DictReader(open(sys.argv[1])):
    k=(r['Kernel_Name'].split('(')[0], r.get('Workgroup_Size') or r.get('Workgroup_Size_X'))
    if k not in order: order.append(k)
    acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k in order:
    d=acc[k]
    print(k[0][-40:], 'wg', k[1], {c: round(v[-1],1) for c,v in d.items()}, 'n=',len(next(iter(d.values()))))
PY
done
