import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
def short(n):
    n = n.replace("void ", "").replace("(anonymous namespace)::", "")
    return n[:60]
# last third = replays
start = len(names) * 2 // 3
c = collections.Counter()
for i in range(start, len(names) - 1):
    n = names[i]
    if "copyBuffer" in n or "at::native" in n or "fillBuffer" in n:
        c[(short(names[i - 1]), short(n), short(names[i + 1]), str(rows[i].get("Grid_Size") or rows[i].get("Grid_Size_X") or ""))] += 1
for k, v in sorted(c.items(), key=lambda kv: -kv[1]):
    print(v, " | ".join(k))
