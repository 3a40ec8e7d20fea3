#!/usr/bin/env python
"""Copy the judged subset of gpurun_out/profiles_<tag>/ (scratch, written on the
GPU box by scripts/collect_profiles.sh) into profiles/<tag>/ (tracked) and
refresh profiles/traffic.json.

    python scripts/publish_profiles.py r01
"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(tag):
    src = os.path.join(ROOT, "gpurun_out", "profiles_" + tag)
    dst = os.path.join(ROOT, "profiles", tag)
    os.makedirs(dst, exist_ok=True)
    for name in ("SUMMARY.txt", "microbench.txt", "stagebench.txt", "ablation.txt",
                 "bench_cfg2.json", "bench_cfg3.json", "bench_cfg4_1gpu.json",
                 "bench_cfg5.json", "bo_loop.json"):
        if os.path.exists(os.path.join(src, name)):
            shutil.copy(os.path.join(src, name), os.path.join(dst, name))
    for c in (2, 3):
        f = glob.glob(os.path.join(src, "stats_cfg%d" % c, "*", "*_kernel_stats.csv"))
        if f:
            shutil.copy(max(f, key=os.path.getmtime), os.path.join(dst, "kernel_stats_cfg%d.csv" % c))
    # per (pass, kernel, counter) averages of every PMC pass
    rows = []
    sweep = {}
    for d in sorted(glob.glob(os.path.join(src, "pmc_*_cfg2"))):
        f = glob.glob(os.path.join(d, "*", "*_counter_collection.csv"))
        if not f:
            continue
        agg = collections.defaultdict(list)
        for x in csv.DictReader(open(max(f, key=os.path.getmtime))):
            m = re.search(r"(k_\w+(<[\w, ]+>)?|__amd\w+)", x["Kernel_Name"])
            k = m.group(1) if m else x["Kernel_Name"][:40]
            agg[(k, x["Counter_Name"])].append(
                (float(x["Counter_Value"]), int(x["End_Timestamp"]) - int(x["Start_Timestamp"])))
        for (k, cn), v in sorted(agg.items()):
            rows.append((os.path.basename(d), k, cn, sum(a for a, _ in v) / len(v), len(v),
                         sum(b for _, b in v) / len(v)))
            if "k_sweep" in k:
                sweep[cn] = sum(a for a, _ in v) / len(v)
                sweep["kernel"] = k
    with open(os.path.join(dst, "pmc_cfg2.csv"), "w") as f:
        w = csv.writer(f)
        w.writerow(["pass", "kernel", "counter", "avg_value", "dispatches", "avg_duration_ns"])
        w.writerows(rows)
    if "FETCH_SIZE" in sweep and "WRITE_SIZE" in sweep:
        tj = os.path.join(ROOT, "profiles", "traffic.json")
        t = {"config2": {
            "kernel": sweep["kernel"],
            "FETCH_SIZE_KB": sweep["FETCH_SIZE"], "WRITE_SIZE_KB": sweep["WRITE_SIZE"],
            "hbm_bytes_per_launch": (2 * sweep["FETCH_SIZE"] + sweep["WRITE_SIZE"]) * 1024,
            "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE "
                    "doubled (gfx950 reports 1/2 of wide coalesced reads, MI355X_MICROARCH.md "
                    "section HBM); source profiles/%s/pmc_cfg2.csv" % tag}}
        json.dump(t, open(tj, "w"), indent=1)
    print("published", sorted(os.listdir(dst)))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r01")
