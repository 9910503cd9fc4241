#!/usr/bin/env python3
"""Condenses gpurun_out/prof_<round>/ (written by tools/profile_round.sh on the GPU box) into the committed evidence:

    profiles/<round>_<cfg>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary (top rows, names cut to 200 chars)
    profiles/<round>_bench_under_rocprof.json the JSON lines the profiled runs printed
    profiles/<round>_pmc_traffic.json         HBM-side bytes per launch and kernel instantiation:
                                              (2 * FETCH_SIZE + WRITE_SIZE) * 1024 from separate --pmc passes (FETCH_SIZE is
                                              doubled as MI355X_MICROARCH.md prescribes for gfx950's wide coalesced reads)
    profiles/<round>_cfg2_sq_counters.json    SQ busy / wait counters of the cfg2 kernels at the bench configuration
    profiles/<round>_sq_<cfg>.json            the same counters for the side configs profiled with them (SQCFGS)

    python tools/condense_round.py r03 [gpurun_out/prof_r03] [profiles]
"""
import collections
import csv
import glob
import json
import os
import sys

CONFIGS = ("headline", "cfg2_f32", "cfg1", "cfg3a", "cfg3a_bf16", "cfg3b", "cfg4_r4", "cfg4_r8", "cfg4_r16", "cfg4_eps36", "cfg5")
# the float32 register family's kernels (cfg2_f32): their traffic gets flat keys like the headline's
F32_KEYS = ("eps_fwd_q2f32_k", "eps_bwd_q2f32_k", "eps_q2f32_finish_k")
HEADLINE_KEYS = ("eps_fwd_head_q2reg_t_k", "eps_fwd_head_q2reg_k", "eps_fwd_q2reg_k", "eps_bwd_dcore_q2reg_k", "eps_head_reduce_k", "eps_bwd_dcore_reduce_k",
                 "head_fwd_k")


def find(directory, suffix):
    hits = glob.glob(os.path.join(directory, "**", "*" + suffix), recursive=True)
    return hits[0] if hits else None


def kernel_stats(directory):
    path = find(directory, "kernel_stats.csv")
    if not path:
        return []
    return list(csv.DictReader(open(path)))


def counters(directory):
    """kernel name -> counter -> (mean value, launches)"""
    path = find(directory, "counter_collection.csv")
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    if path:
        for row in csv.DictReader(open(path)):
            acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return {k: {c: (sum(v) / len(v), len(v)) for c, v in cs.items()} for k, cs in acc.items()}


def ours(name):
    return "at::native" not in name and "rocclr" not in name and "Cijk" not in name


def main():
    rnd = sys.argv[1] if len(sys.argv) > 1 else "r05"
    src = sys.argv[2] if len(sys.argv) > 2 else f"gpurun_out/prof_{rnd}"
    dst = sys.argv[3] if len(sys.argv) > 3 else "profiles"
    traffic, lines = {"_note": "HBM-side bytes per launch: (2*FETCH_SIZE + WRITE_SIZE)*1024 (FETCH_SIZE / WRITE_SIZE are KB; FETCH_SIZE is "
                               "doubled as MI355X_MICROARCH.md prescribes for gfx950 coalesced reads); avg_us from the kernel-trace "
                               "pass of the same command.  Per config: the kernel instantiations of this library, longest first."}, {}
    for cfg in CONFIGS:
        d = os.path.join(src, cfg)
        stats = kernel_stats(d)
        if stats:
            with open(os.path.join(dst, f"{rnd}_{'bench_cfg2' if cfg == 'headline' else cfg}_kernel_stats.csv"), "w", newline="") as f:
                w = csv.writer(f)
                w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
                for r in stats[:25]:
                    w.writerow([r["Name"][:200], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
        jl = os.path.join(src, cfg + ".json")
        if os.path.exists(jl):
            txt = open(jl).read().strip().splitlines()
            if txt:
                try:
                    lines[cfg] = json.loads(txt[-1])
                except Exception:
                    pass
        fetch, write = counters(os.path.join(src, cfg + "_fetch")), counters(os.path.join(src, cfg + "_write"))
        avg_us = {r["Name"]: float(r["AverageNs"]) / 1e3 for r in stats}
        entries = []
        for k in sorted(set(fetch) | set(write)):
            if not ours(k):
                continue
            f_kb = fetch.get(k, {}).get("FETCH_SIZE", (0.0, 0))
            w_kb = write.get(k, {}).get("WRITE_SIZE", (0.0, 0))
            us = next((v for n, v in avg_us.items() if n[:150] == k[:150]), None)
            entries.append({"kernel": k[:200], "launches": int(f_kb[1] or w_kb[1]), "avg_us": us,
                            "fetch_bytes": int(2 * f_kb[0] * 1024), "write_bytes": int(w_kb[0] * 1024),
                            "traffic_bytes": int((2 * f_kb[0] + w_kb[0]) * 1024)})
        entries.sort(key=lambda e: -(e["avg_us"] or 0) * e["launches"])
        if entries:
            traffic[cfg] = entries
        if cfg == "cfg2_f32":
            for e in entries:
                for key in F32_KEYS:
                    if key in e["kernel"]:
                        traffic.setdefault(f"{key}:B1024", e["traffic_bytes"])
        if cfg == "headline":
            for e in entries:
                for key in HEADLINE_KEYS:
                    if key + "<" in e["kernel"] or key + "(" in e["kernel"] or ("N_1" in e["kernel"] and key in e["kernel"]):
                        traffic.setdefault(f"{key}:B1024", e["traffic_bytes"])
    # what ties these passes to a build: the sha256 of the library the profiled processes loaded (taken on the GPU box by
    # profile_round.sh) and the git head of the tree it was built from (bench.py nulls `roofline.traffic` on a mismatch)
    meta = {}
    stamp = os.path.join(src, "stamp.json")
    if os.path.exists(stamp):
        meta.update(json.load(open(stamp)))
    try:
        import subprocess

        meta["head"] = subprocess.run(["git", "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True, check=True).stdout.strip()
        dirty = subprocess.run(["git", "status", "--porcelain", "--", "dctn_amd", "include"], capture_output=True, text=True).stdout.strip()
        meta["head_note"] = ("library sources differ from this head (uncommitted changes at condense time)" if dirty
                             else "dctn_amd/ and include/ clean at this head when the passes were condensed")
    except Exception:
        pass
    traffic["_meta"] = meta
    json.dump(traffic, open(os.path.join(dst, f"{rnd}_pmc_traffic.json"), "w"), indent=1)
    json.dump(lines, open(os.path.join(dst, f"{rnd}_bench_under_rocprof.json"), "w"), indent=1)
    sq = {}
    for tag in ("headline_sq1", "headline_sq2"):
        for k, cs in counters(os.path.join(src, tag)).items():
            for key in HEADLINE_KEYS:
                if key in k and ours(k):
                    sq.setdefault(key, {"kernel": k[:160]}).update({c: round(v[0]) for c, v in cs.items()})
    if sq:
        sq["_note"] = ("per-dispatch means, cfg2 bf16 B = 1024, eager launches (--graph 0) under --pmc; SQ_ACTIVE_INST_* / SQ_WAIT_* / "
                       "SQ_WAVE_CYCLES count quad-cycles summed over waves, SQ_VALU_MFMA_BUSY_CYCLES cycles summed over SIMDs, "
                       "GRBM_GUI_ACTIVE cycles summed over the 8 XCDs")
        json.dump(sq, open(os.path.join(dst, f"{rnd}_cfg2_sq_counters.json"), "w"), indent=1)
    # side configs profiled with SQ counters (profile_round.sh SQCFGS): every kernel of this library, per-dispatch means
    for cfg in CONFIGS[1:]:
        per = {}
        for tag in (cfg + "_sq1", cfg + "_sq2"):
            for k, cs in counters(os.path.join(src, tag)).items():
                if ours(k):
                    per.setdefault(k[:160], {}).update({c: round(v[0]) for c, v in cs.items()})
                    per[k[:160]]["launches"] = max(v[1] for v in cs.values())
        if per:
            per["_note"] = ("per-dispatch means under --pmc (two passes); SQ_ACTIVE_INST_* / SQ_WAIT_* / SQ_WAVE_CYCLES count quad-cycles "
                            "summed over waves; GRBM_GUI_ACTIVE cycles summed over the 8 XCDs")
            json.dump(per, open(os.path.join(dst, f"{rnd}_sq_{cfg}.json"), "w"), indent=1)
    print(json.dumps({k: (v if not isinstance(v, list) else [(e["kernel"][:60], e["avg_us"], e["traffic_bytes"]) for e in v[:4]])
                      for k, v in traffic.items() if k not in ("_note", "_meta")}, indent=1))


if __name__ == "__main__":
    main()
