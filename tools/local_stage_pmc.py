"""Fold the rocprofv3 counter passes of tools/local_stage_bench.py (LOCAL_BENCH_ONLY=<variant>, one counter per pass) into
fabric-side bytes per chain launch and per stage:  local_stage_pmc.py <dir with <variant>_<COUNTER>/ passes> <bench.json> <out.json>
Bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 read correction, MI355X_MICROARCH.md HBM section; both counters in KiB),
averaged over the level_chain_kernel dispatches of the pass.  Algorithmic bytes of a stage: activations in + out, the filter
tensor once, GroupNorm / FiLM parameters."""
import csv, glob, json, os, sys

root, bench, out = sys.argv[1:4]
NST, N, CH = 8, 40, 128


def mean_counter(d, counter):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not f:
        return None
    rows = [r for r in csv.DictReader(open(f[0])) if r["Counter_Name"] == counter and "level_chain_kernel" in r["Kernel_Name"]]
    rows = rows[len(rows) // 4:]          # (skip the first launches: cold filters)
    return sum(float(r["Counter_Value"]) for r in rows) / max(1, len(rows)) if rows else None


res = {"note": "fabric-side bytes per level_chain_kernel launch = (2*FETCH_SIZE + WRITE_SIZE)*1024, separate passes; 8 stages per launch",
       "variants": []}
b = json.load(open(bench))
for v in b["variants"]:
    tag = f"{v['kind']}_{v['H']}" + (f"_{v['row_tiles']}" if v["kind"] == "local" else "")
    fe, wr = mean_counter(os.path.join(root, tag + "_FETCH_SIZE"), "FETCH_SIZE"), mean_counter(os.path.join(root, tag + "_WRITE_SIZE"), "WRITE_SIZE")
    M = v["M"]
    alg = NST * 4 * (2 * M * CH + 9 * CH * CH + 3 * CH + (N // 20) * 2 * CH)
    rec = dict(kind=v["kind"], H=v["H"], M=M, row_tiles=v["row_tiles"], us_per_stage=v["us_per_stage"],
               algorithmic_bytes_per_launch=alg)
    if fe is not None and wr is not None:
        tot = (2 * fe + wr) * 1024
        rec.update(fetch_kib_raw=round(fe, 1), write_kib=round(wr, 1), fabric_bytes_per_launch=round(tot),
                   fabric_bytes_per_stage=round(tot / NST), ratio_to_algorithmic=round(tot / alg, 2))
    res["variants"].append(rec)
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
