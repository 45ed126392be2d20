"""Summarise a rocprofv3 kernel_trace.csv: per-kernel duration stats and launch-to-launch gaps."""
import csv, glob, sys, collections
path = sys.argv[1]
files = glob.glob(path + "/**/*kernel_trace.csv", recursive=True)
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
by = collections.defaultdict(list)
for s, e, k in rows:
    by[k.split("(")[0][:70]].append(e - s)
print(f"{len(rows)} dispatches")
for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    v2 = sorted(v)
    print(f"{k:72s} n={len(v):6d} avg={sum(v)/len(v)/1e3:8.2f}us p50={v2[len(v)//2]/1e3:8.2f} p99={v2[int(len(v)*.99)]/1e3:8.2f} max={v2[-1]/1e3:9.2f} total={sum(v)/1e6:9.2f}ms")
# gaps between consecutive train_step kernels
ts = [(s, e) for s, e, k in rows if "train_step" in k]
gaps = sorted(ts[i + 1][0] - ts[i][1] for i in range(len(ts) - 1))
per = sorted(ts[i + 1][0] - ts[i][0] for i in range(len(ts) - 1))
if gaps:
    n = len(gaps)
    print(f"train_step gaps: p10={gaps[n//10]/1e3:.2f} p50={gaps[n//2]/1e3:.2f} p90={gaps[n*9//10]/1e3:.2f} p99={gaps[int(n*.99)]/1e3:.2f} max={gaps[-1]/1e3:.1f} us")
    print(f"train_step start-to-start: p10={per[n//10]/1e3:.2f} p50={per[n//2]/1e3:.2f} p90={per[n*9//10]/1e3:.2f} mean={sum(per)/n/1e3:.2f} us")
    big = [g for g in gaps if g > 50_000]
    print(f"gaps > 50us: {len(big)} totalling {sum(big)/1e6:.2f} ms of {(ts[-1][1]-ts[0][0])/1e6:.2f} ms span")
