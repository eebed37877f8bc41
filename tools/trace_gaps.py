"""Developer aid: busy time vs span of the last graph replays in a rocprofv3 --kernel-trace CSV of bench.py."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# find p_sample kernels (one per denoising step) and take the window between the last 40 of them
idx = [i for i, r in enumerate(rows) if "p_sample_kernel" in r["Kernel_Name"]]
sel = idx[-240:-200] if len(idx) > 260 else idx[-40:]
a, b = sel[0], sel[-1]
seg = rows[a + 1:b + 1]
steps = len(sel) - 1
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
span = int(rows[b]["End_Timestamp"]) - int(rows[a]["End_Timestamp"])
print(f"steps {steps}: span {span / steps / 1e3:.1f} us/step, kernel busy {busy / steps / 1e3:.1f} us/step, "
      f"gaps {(span - busy) / steps / 1e3:.1f} us/step over {len(seg) / steps:.1f} kernels/step")
from collections import defaultdict
agg = defaultdict(lambda: [0, 0])
for r in seg:
    k = r["Kernel_Name"].split("(")[0][-60:]
    agg[k][0] += 1; agg[k][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
    print(f"{k:62s} {n / steps:6.1f}/step {t / steps / 1e3:8.1f} us/step  avg {t / n / 1e3:6.1f} us")
