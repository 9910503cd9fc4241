#!/usr/bin/env python3
"""gpurun_out/pmc_<tag>/{a,b,c} (written by tools/pmc_py.sh) -> profiles/r02_sq_<tag>.json: per kernel instantiation and grid
the mean of every SQ counter over the profiled dispatches.   python tools/condense_sq.py <tag> <kernel name filter> [note]"""
import collections, csv, glob, json, sys
tag, filt = sys.argv[1], sys.argv[2]
note = sys.argv[3] if len(sys.argv) > 3 else ""
out = collections.defaultdict(dict)
for t in "abc":
    f = glob.glob(f"gpurun_out/pmc_{tag}/{t}/**/*counter_collection.csv", recursive=True)
    if not f:
        continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        if filt in r["Kernel_Name"]:
            acc[(r["Kernel_Name"][:120], r.get("Grid_Size", ""))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for (k, g), cs in acc.items():
        e = out[f"{k} grid={g}"]
        for c, v in cs.items():
            e[c] = round(sum(v) / len(v))
        e["dispatches"] = len(next(iter(cs.values())))
doc = {"_note": "per-dispatch means; SQ_ACTIVE_INST_* / SQ_WAIT_* / SQ_WAVE_CYCLES count quad-cycles summed over waves, "
                "SQ_VALU_MFMA_BUSY_CYCLES cycles summed over the 1024 SIMDs, SQ_LDS_IDX_ACTIVE / SQ_LDS_BANK_CONFLICT cycles summed over the "
                "256 CUs, GRBM_GUI_ACTIVE cycles summed over the 8 XCDs. " + note}
doc.update(out)
json.dump(doc, open(f"profiles/r02_sq_{tag}.json", "w"), indent=1)
print(json.dumps({k: v for k, v in list(doc.items())[:3]}, indent=1)[:1500])
