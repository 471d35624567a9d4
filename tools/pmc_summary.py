"""Summarise rocprofv3 --pmc csv output: per kernel, mean counter value per dispatch.
usage: python tools/pmc_summary.py <dir> [kernel-substring]"""
import csv, glob, sys, collections
d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if sub in k:
            acc[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"    {c:34s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
