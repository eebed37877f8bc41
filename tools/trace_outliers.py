"""Developer aid: from a rocprofv3 --kernel-trace CSV, list kernel launches far slower than their median."""
import csv, sys, collections, statistics
rows = list(csv.DictReader(open(sys.argv[1])))
by = collections.defaultdict(list)
for r in rows:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    by[r["Kernel_Name"][:70]].append((d, int(r["Start_Timestamp"]), r.get("Grid_Size_X", ""), r.get("Workgroup_Size_X", "")))
t0 = min(int(r["Start_Timestamp"]) for r in rows)
out = []
for k, v in by.items():
    med = statistics.median(d for d, *_ in v)
    tot = sum(d for d, *_ in v)
    big = [(d, s, g, w) for d, s, g, w in v if d > 5 * med and d > 200000]
    out.append((sum(d for d, *_ in big), k, len(v), med, tot, big[:6]))
out.sort(reverse=True)
for exc, k, n, med, tot, big in out[:12]:
    print(f"{k:70s} n={n:6d} median={med/1e3:8.1f}us total={tot/1e6:8.2f}ms outliers={exc/1e6:8.2f}ms")
    for d, s, g, w in big:
        print(f"      {d/1e6:8.3f} ms at t={(s-t0)/1e6:9.2f} ms grid={g} wg={w}")
