"""Summarise rocprofv3 csv output dirs: per-kernel mean of each counter / kernel durations."""
import sys, os, csv, glob, collections
root = sys.argv[1]
for f in sorted(glob.glob(os.path.join(root, "**", "*.csv"), recursive=True)):
    base = os.path.basename(f)
    if base.endswith("counter_collection.csv"):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            if "raymarch" not in k: continue
            print(f, k)
            for c, v in cs.items():
                print(f"   {c:28s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
    elif base.endswith("kernel_stats.csv"):
        print(f)
        for r in csv.DictReader(open(f)):
            print("   ", {k: (v[:70] if isinstance(v, str) else v) for k, v in r.items()})
