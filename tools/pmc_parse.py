"""Aggregate a rocprofv3 --pmc counter_collection.csv: per kernel (name substring), mean counter value and mean duration.
usage: pmc_parse.py <dir> [name-substring]"""
import csv, glob, sys, collections
d = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else "gemm256"
csv.field_size_limit(1 << 30)
for path in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(list); dur = collections.defaultdict(dict)
    for r in csv.DictReader(open(path)):
        if sub not in r["Kernel_Name"]: continue
        key = (r["Kernel_Name"][:80], r["Grid_Size"])
        acc[(key, r["Counter_Name"])].append(float(r["Counter_Value"]))
        dur[key][r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    for key in dur:
        ds = list(dur[key].values())[1:] or list(dur[key].values())
        us = sum(ds) / len(ds)
        out = {c: sum(v[1:] or v) / len(v[1:] or v) for (k, c), v in acc.items() if k == key}
        line = f"{path.split('/')[-2]:>10s} {key[0][:78]} grid={key[1]:>8s} n={len(ds)} avg_us={us:8.1f}"
        if "GRBM_GUI_ACTIVE" in out: line += f" clk={out['GRBM_GUI_ACTIVE'] / 8 / us / 1e3:5.2f}GHz"
        print(line)
        for c, v in sorted(out.items()):
            extra = ""
            if c != "SQ_WAVE_CYCLES" and "SQ_WAVE_CYCLES" in out and c.startswith("SQ_") and out["SQ_WAVE_CYCLES"]:
                extra = f"  ({v / out['SQ_WAVE_CYCLES'] * 100:5.1f}% of SQ_WAVE_CYCLES)"
            print(f"            {c:32s} {v:16.1f}{extra}")
