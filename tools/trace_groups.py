import csv, sys, collections, re
f = sys.argv[1]; steps = int(sys.argv[2])
rows = list(csv.DictReader(open(f)))
g = collections.defaultdict(list)
for r in rows:
    name = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])[:70]
    key = (name, r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size'), r.get('LDS_Block_Size', ''))
    g[key].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
out = sorted(g.items(), key=lambda kv: -sum(kv[1]))
tot = sum(sum(v) for v in g.values())
print("total us/step", tot / steps)
for k, v in out[:70]:
    v2 = sorted(v)
    print(f"{k[0]:70s} grid={k[1]:>8s} lds={k[2]:>6s} n/step={len(v)/steps:6.1f} med={v2[len(v2)//2]:8.2f} us/step={sum(v)/steps:8.1f}")
