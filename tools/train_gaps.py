"""Developer aid: GPU idle gaps of the steady-state training loop from a rocprofv3 --kernel-trace (+ --memory-copy-trace)
CSV directory of `bench.py --train-steps N`: per optimizer step (adamw_ema_kernel to adamw_ema_kernel) the span, the busy
time and the largest gaps with the kernels on either side."""
import csv, glob, os, sys
d = sys.argv[1]
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][-50:]))
for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "")))
rows.sort()
opt = [i for i, r in enumerate(rows) if "adamw_ema_kernel" in r[2]]
opt = opt[-9:-1]
for a, b in zip(opt[:-1], opt[1:]):
    seg = rows[a:b + 1]
    span = seg[-1][1] - seg[0][1]
    busy, end, gaps = 0, seg[0][1], []
    for s, e, n in seg[1:]:
        if s > end:
            gaps.append((s - end, prev, n))
        busy += max(0, e - max(s, end))
        if e > end:
            end, prev = e, n
    gaps.sort(reverse=True)
    print(f"step: span {span / 1e3:8.1f} us  busy {busy / 1e3:8.1f} us  idle {(span - busy) / 1e3:7.1f} us; largest gaps: " +
          "; ".join(f"{g / 1e3:.0f} us [{p[-28:]} -> {n[-28:]}]" for g, p, n in gaps[:4]))
