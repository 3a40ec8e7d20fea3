#!/usr/bin/env python
"""Condense a scripts/collect_profiles.sh output directory into one text file."""
import collections
import csv
import glob
import json
import os
import re
import sys


def key(n):
    m = re.search(r"(k_\w+(<[\w, ]+>)?|__amd\w+)", n)
    return m.group(1) if m else n[:40]


def stats(path, title):
    files = glob.glob(os.path.join(path, "*", "*_kernel_stats.csv"))
    if not files:
        return
    print("== rocprofv3 --kernel-trace --stats : %s" % title)
    print("%-30s %6s %12s %12s %8s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for r in csv.DictReader(open(max(files, key=os.path.getmtime))):
        print("%-30s %6s %12.1f %12.2f %8s" % (key(r["Name"]), r["Calls"],
              float(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, r["Percentage"]))
    print()


def pmc(path):
    files = glob.glob(os.path.join(path, "*", "*_counter_collection.csv"))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    if not files:
        return agg
    for x in csv.DictReader(open(max(files, key=os.path.getmtime))):
        agg[key(x["Kernel_Name"])][x["Counter_Name"]].append(
            (float(x["Counter_Value"]), int(x["End_Timestamp"]) - int(x["Start_Timestamp"])))
    return agg


def main(d):
    for name in sorted(glob.glob(os.path.join(d, "bench_*.json"))):
        try:
            j = json.loads(open(name).read().strip().splitlines()[-1])
        except Exception as e:          # noqa
            print("%s: no JSON (%s)" % (os.path.basename(name), e))
            continue
        rf = j["roofline"]
        print("== %s" % os.path.basename(name))
        print("   %s" % j["config"]["workload"])
        print("   value %.4g %s   ms/step %.3f   sweep kernel %.3f ms x %d   %.2f TFLOP/s = %.1f%% of %.1f"
              % (j["value"], j["unit"], j["ms_per_step"], rf["kernel_ms_avg"], rf["launches"],
                 rf["achieved"], 100 * rf["frac"], rf["peak"]))
        if "cpu_baseline" in j:
            cb = j["cpu_baseline"]
            print("   cpu_baseline %.4g %s on %d cores (%s)  -> speedup %.0fx"
                  % (cb["value"], cb["unit"], cb["cores"], cb["kind"], j["speedup_vs_cpu"]))
        if j.get("parity"):
            print("   parity %s" % j["parity"])
        print()
    for c in (2, 3):
        stats(os.path.join(d, "stats_cfg%d" % c), "bench.py --config %d" % c)
    a = pmc(os.path.join(d, "pmc_mfma_cfg2"))
    for k, v in a.items():
        if "sweep" in k or "expander" in k or "mfma_bench" in k:
            c = {n: sum(x[0] for x in vals) / len(vals) for n, vals in v.items()}
            dur = sum(x[1] for x in list(v.values())[0]) / len(list(v.values())[0])
            gui = c.get("GRBM_GUI_ACTIVE", 0) / 8.0          # summed over 8 XCDs
            util = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (gui * 1024) if gui else 0
            print("PMC %-16s dur %.1f us  clock %.2f GHz  MfmaUtil %.1f%% (busy cycles / (GUI_ACTIVE * 1024 SIMDs))"
                  % (k, dur / 1e3, gui / dur if dur else 0, 100 * util))
    f = pmc(os.path.join(d, "pmc_fetch_cfg2"))
    w = pmc(os.path.join(d, "pmc_write_cfg2"))
    for k in f:
        if k in w and ("sweep" in k or "expander" in k or k in ("k_candidates", "k_maximizers", "k_argmax")):
            fs = sum(x[0] for x in f[k]["FETCH_SIZE"]) / len(f[k]["FETCH_SIZE"])
            ws = sum(x[0] for x in w[k]["WRITE_SIZE"]) / len(w[k]["WRITE_SIZE"])
            print("PMC %-16s FETCH_SIZE %.0f KB (x2 gfx950 correction -> %.1f MB)  WRITE_SIZE %.0f KB  "
                  "HBM traffic/launch %.1f MB" % (k, fs, 2 * fs / 1024, ws, (2 * fs + ws) / 1024))
    l = pmc(os.path.join(d, "pmc_lds_cfg2"))
    for k, v in l.items():
        if "sweep" in k:
            print("PMC %-16s %s" % (k, {n: sum(x[0] for x in vals) / len(vals) for n, vals in v.items()}))
    for sub in ("pmc_issue_cfg2", "pmc_coexec_cfg2"):
        for k, v in pmc(os.path.join(d, sub)).items():
            if "sweep" in k:
                print("PMC %-16s %s" % (k, {n: "%.4g" % (sum(x[0] for x in vals) / len(vals))
                                            for n, vals in sorted(v.items())}))
    for name, title in (("stagebench.txt", "scripts/stagebench.py"),
                        ("ablation.txt", "scripts/ablate.sh (instrumented build; results invalid by design)")):
        f = os.path.join(d, name)
        if os.path.exists(f):
            print("\n== %s" % title)
            print(open(f).read())
    mb = os.path.join(d, "microbench.txt")
    if os.path.exists(mb):
        print("\n== scripts/microbench.py")
        print(open(mb).read())


if __name__ == "__main__":
    main(sys.argv[1])
