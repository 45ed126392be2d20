"""Summarise rocprofv3 --pmc counter CSVs per kernel (sum over dispatches of the named kernel)."""
import csv, glob, sys, collections
path, pat = sys.argv[1], sys.argv[2]
tot = collections.defaultdict(float); nd = collections.defaultdict(int)
for f in glob.glob(path + "/**/*counter_collection.csv", recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            if pat in r["Kernel_Name"]:
                tot[r["Counter_Name"]] += float(r["Counter_Value"]); nd[r["Counter_Name"]] += 1
for k in sorted(tot):
    print(f"{k:28s} dispatches={nd[k]:3d} sum={tot[k]:.6g} per-dispatch={tot[k]/max(nd[k],1):.6g}")
