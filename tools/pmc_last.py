"""mean of a PMC counter over the LAST n launches of the kernels whose name contains `pat`:
   python tools/pmc_last.py <counter_collection.csv or dir> <pat> [n=20] [grid_size]
   grid_size (threads, the CSV's Grid_Size column): only launches of that size -- a kernel that the script launches at several
   sizes (the frame gather at batch 4096 / 512 / 32, Adam at 2^22 / 2^26 parameters) is then read at ONE of them"""
import collections, csv, glob, os, sys
path, pat = sys.argv[1], sys.argv[2]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 20
grid = int(sys.argv[4]) if len(sys.argv) > 4 else None
if os.path.isdir(path):
    path = sorted(glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True))[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(path)):
    if pat in r["Kernel_Name"] and (grid is None or int(r["Grid_Size"]) == grid):
        acc[r["Kernel_Name"].split("(")[0][-90:]][r["Counter_Name"]].append((int(r.get("Dispatch_Id", 0)), float(r["Counter_Value"])))
for k, d in acc.items():
    for c, v in d.items():
        v.sort()
        last = [x for _, x in v[-n:]]
        print(f"{k} {c}: mean of last {len(last)} of {len(v)} launches = {sum(last) / len(last):.1f}")
