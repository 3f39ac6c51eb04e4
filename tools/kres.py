#!/usr/bin/env python3
"""Per-kernel register / scratch / occupancy table of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage).
usage: python tools/kres.py csrc/ppo3w.hip [extra hipcc flags]"""
import re
import subprocess
import sys

src = sys.argv[1]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-c", src,
       "-o", "/tmp/kres.o", "-Rpass-analysis=kernel-resource-usage"] + sys.argv[2:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark:\s+(.*?): (.*?) \[-Rpass", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == "Function Name":
        cur = {"name": v}
        rows.append(cur)
    elif cur is not None:
        cur[k] = v
dem = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.splitlines()
print(f"{'kernel':100s} {'VGPR':>5s} {'AGPR':>5s} {'scr':>5s} {'occ':>4s} {'vspill':>6s}")
for r, d in zip(rows, dem):
    d = re.sub(r"\(.*", "", d).replace("rlhip::", "").replace("void ", "")
    print(f"{d[:100]:100s} {r.get('VGPRs','?'):>5s} {r.get('AGPRs','?'):>5s} {r.get('ScratchSize [bytes/lane]','?'):>5s} "
          f"{r.get('Occupancy [waves/SIMD]','?'):>4s} {r.get('VGPRs Spill','?'):>6s}")
