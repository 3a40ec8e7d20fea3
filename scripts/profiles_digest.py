#!/usr/bin/env python
"""Condense a scripts/collect_profiles.sh output directory (gpurun_out/profiles_<tag>)
into SUMMARY.txt (stdout) and, with --publish <tag>, copy the judged subset into
profiles/<tag>/ (tracked) and refresh profiles/traffic.json for every config."""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def key(n):
    m = re.search(r"(k_\w+(<[\w, ]+>)?|__amd\w+)", n)
    return m.group(1) if m else n[:40]


def newest(pattern):
    f = glob.glob(pattern)
    return max(f, key=os.path.getmtime) if f else None


def kernel_stats(d, c):
    f = newest(os.path.join(d, "stats_cfg%d" % c, "*", "*_kernel_stats.csv"))
    return list(csv.DictReader(open(f))) if f else []


def pmc(d, name):
    f = newest(os.path.join(d, name, "*", "*_counter_collection.csv"))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    if f:
        for x in csv.DictReader(open(f)):
            agg[key(x["Kernel_Name"])][x["Counter_Name"]].append(
                (float(x["Counter_Value"]), int(x["End_Timestamp"]) - int(x["Start_Timestamp"])))
    return agg


def mean(vals):
    return sum(v for v, _ in vals) / len(vals)


def digest(d):
    out, traffic, rows = [], {}, collections.defaultdict(list)
    algo = {}
    for c in (2, 3, 4, 5):
        bj = os.path.join(d, "bench_cfg%d.json" % c)
        if os.path.exists(bj):
            try:
                j = json.loads(open(bj).read().strip().splitlines()[-1])
                rf = j["roofline"]
                algo[c] = (rf.get("algorithmic_hbm_bytes_per_launch"), j["config"])
                out.append("== bench.py --config %d" % c)
                out.append("   %s" % j["config"]["workload"])
                out.append("   value %.4g %s   ms/step %.3f (timed region %.2f s)   k_sweep %.3f ms x %d   "
                           "%.2f TFLOP/s algorithmic = %.1f%% of %.1f"
                           % (j["value"], j["unit"], j["ms_per_step"], j.get("timed_region_s", 0),
                              rf["kernel_ms_avg"], rf["launches"], rf["achieved"], 100 * rf["frac"], rf["peak"]))
                if "cpu_baseline" in j:
                    cb = j["cpu_baseline"]
                    out.append("   cpu_baseline %.4g %s on %d threads of %s (median of %d); 1 thread %.4g; ratio %.0fx"
                               % (cb["value"], cb["unit"], cb["cores"], cb.get("cpu", "?"), cb.get("runs", 1),
                                  cb.get("single_thread", {}).get("value", float("nan")), j["speedup_vs_cpu"]))
                if j.get("parity"):
                    out.append("   parity %s" % j["parity"])
            except Exception as e:      # noqa
                out.append("bench_cfg%d.json: no JSON (%s)" % (c, e))
        st = kernel_stats(d, c)
        if st:
            out.append("   rocprofv3 --kernel-trace --stats (same config; step counts: scripts/collect_profiles.sh):")
            tot = sum(float(r["TotalDurationNs"]) for r in st)
            for r in st[:8]:
                out.append("     %-34s calls %5s  avg %10.2f us  %5.1f%%" % (
                    key(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3,
                    100 * float(r["TotalDurationNs"]) / tot))
            sw = [r for r in st if "k_sweep" in r["Name"]]
            others = [r for r in st if "k_sweep" not in r["Name"]]
            if sw:
                calls = int(sw[0]["Calls"])
                out.append("     non-sweep kernels per sweep launch: %.1f launches, %.1f us" % (
                    sum(int(r["Calls"]) for r in others) / calls,
                    sum(float(r["TotalDurationNs"]) for r in others) / calls / 1e3))
        a = pmc(d, "pmc_mfma_cfg%d" % c)
        f = pmc(d, "pmc_fetch_cfg%d" % c)
        w = pmc(d, "pmc_write_cfg%d" % c)
        for k in a:
            if "k_sweep" not in k:
                continue
            v = a[k]
            cm = {n: mean(vals) for n, vals in v.items()}
            dur = sum(x[1] for x in list(v.values())[0]) / len(list(v.values())[0])
            gui = cm.get("GRBM_GUI_ACTIVE", 0) / 8.0          # summed over 8 XCDs
            util = cm.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (gui * 1024) if gui else 0
            out.append("   PMC %s: %.1f us under the profiler, clock %.2f GHz, MFMA pipe busy %.1f%% of all SIMD cycles"
                       % (k, dur / 1e3, gui / dur if dur else 0, 100 * util))
            for n, val in sorted(cm.items()):
                rows[c].append(("pmc_mfma", k, n, val))
            if k in f and k in w:
                fs, ws = mean(f[k]["FETCH_SIZE"]), mean(w[k]["WRITE_SIZE"])
                out.append("   PMC %s: FETCH_SIZE %.0f KiB (x2 gfx950 correction -> %.1f MB) + WRITE_SIZE %.0f KiB "
                           "(%.1f MB) = %.1f MB (1e6 B) HBM traffic per launch"
                           % (k, fs, 2 * fs * 1024 / 1e6, ws, ws * 1024 / 1e6, (2 * fs + ws) * 1024 / 1e6))
                rows[c] += [("pmc_fetch", k, "FETCH_SIZE", fs), ("pmc_write", k, "WRITE_SIZE", ws)]
                ab, cf = algo.get(c, (None, None))
                if cf and c == 5:           # swarm fitness: particle in, value + flag out
                    ab = cf["rows_per_gpu"] * (8.0 * cf["d"] + 9.0)
                extra = {}
                if ab:
                    # 8d + 16G + 3 bytes per row; with the resident mean / var
                    # arrays the product keeps: + 16G
                    with_mv = ab + 16.0 * cf["G"] * cf["rows_per_gpu"] if cf and c != 5 else ab
                    extra = dict(algorithmic_hbm_bytes_per_launch=ab,
                                 algorithmic_incl_mean_var=with_mv,
                                 executed_over_algorithmic_bytes=(2 * fs + ws) * 1024 / ab,
                                 executed_over_algorithmic_incl_mean_var=(2 * fs + ws) * 1024 / with_mv)
                    out.append("   executed / algorithmic HBM bytes: %.2fx of SURVEY 8d's %.1f MB (%.2fx of the %.1f MB that "
                               "include the resident mean / var arrays)"
                               % (extra["executed_over_algorithmic_bytes"], ab / 1e6,
                                  extra["executed_over_algorithmic_incl_mean_var"], with_mv / 1e6))
                traffic["config%d" % c] = dict(
                    kernel=k, FETCH_SIZE_KB=fs, WRITE_SIZE_KB=ws,
                    hbm_bytes_per_launch=(2 * fs + ws) * 1024, **extra,
                    mfma_pipe_busy=util, clock_ghz_under_profiler=gui / dur if dur else 0,
                    note="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE "
                         "doubled (gfx950 reports 1/2 of wide coalesced reads, MI355X_MICROARCH.md "
                         "section HBM)")
        # a sweep that is several launches (k_sweep_mid in passes of row blocks: config 2 since
        # round 5): the bytes and busy cycles of its passes together
        mids = [k for k in a if "k_sweep_mid" in k and k in f and k in w]
        if len(mids) > 1:
            fs = sum(mean(f[k]["FETCH_SIZE"]) for k in mids)
            ws = sum(mean(w[k]["WRITE_SIZE"]) for k in mids)
            busy = sum(mean(a[k]["SQ_VALU_MFMA_BUSY_CYCLES"]) for k in mids)
            gui = sum(mean(a[k]["GRBM_GUI_ACTIVE"]) for k in mids) / 8.0
            dur = sum(sum(x[1] for x in list(a[k].values())[0]) / len(list(a[k].values())[0]) for k in mids)
            ab, cf = algo.get(c, (None, None))
            extra = {}
            if ab:
                with_mv = ab + 16.0 * cf["G"] * cf["rows_per_gpu"]
                extra = dict(algorithmic_hbm_bytes_per_launch=ab, algorithmic_incl_mean_var=with_mv,
                             executed_over_algorithmic_bytes=(2 * fs + ws) * 1024 / ab,
                             executed_over_algorithmic_incl_mean_var=(2 * fs + ws) * 1024 / with_mv)
            out.append("   PMC all %d passes of k_sweep_mid together: %.1f us, MFMA pipe busy %.1f%% of all SIMD cycles, "
                       "%.1f MB (1e6 B) HBM traffic per sweep%s"
                       % (len(mids), dur / 1e3, 100 * busy / (gui * 1024) if gui else 0, (2 * fs + ws) * 1024 / 1e6,
                          (" = %.2fx of SURVEY 8d's %.1f MB" % (extra["executed_over_algorithmic_bytes"], ab / 1e6)) if ab else ""))
            traffic["config%d" % c] = dict(
                kernel=" + ".join(mids), FETCH_SIZE_KB=fs, WRITE_SIZE_KB=ws,
                hbm_bytes_per_launch=(2 * fs + ws) * 1024, **extra,
                mfma_pipe_busy=busy / (gui * 1024) if gui else 0,
                clock_ghz_under_profiler=gui / dur if dur else 0,
                note="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, summed over the sweep's "
                     "launches; FETCH_SIZE doubled (gfx950 reports 1/2 of wide coalesced reads, "
                     "MI355X_MICROARCH.md section HBM)")
        out.append("")
    for c in (3, 2):
        for sub in ("pmc_insts_cfg%d" % c, "pmc_issue_cfg%d" % c, "pmc_l2_cfg%d" % c):
            for k, v in pmc(d, sub).items():
                if "k_sweep" in k:
                    cm = {n: mean(vals) for n, vals in sorted(v.items())}
                    out.append("PMC %s %s: %s" % (sub, k, {n: "%.4g" % x for n, x in cm.items()}))
                    for n, val in cm.items():
                        rows[c].append((sub, k, n, val))
    for name in ("ab_kernels.txt", "ablation.txt", "ablation_cfg2.txt", "stamps.txt",
                 "stamps_cfg2.txt", "reconcile.txt", "ab_tables.txt", "probes.txt"):
        p = os.path.join(d, name)
        if os.path.exists(p):
            out.append("\n== %s\n%s" % (name, open(p).read()))
    return "\n".join(out), traffic, rows


def main(argv):
    d = argv[1]
    text, traffic, rows = digest(d)
    print(text)
    if len(argv) > 3 and argv[2] == "--publish":
        tag = argv[3]
        dst = os.path.join(ROOT, "profiles", tag)
        os.makedirs(dst, exist_ok=True)
        open(os.path.join(dst, "SUMMARY.txt"), "w").write(text + "\n")
        for c in (2, 3, 4, 5):
            for name in ("bench_cfg%d.json" % c,):
                if os.path.exists(os.path.join(d, name)):
                    shutil.copy(os.path.join(d, name), os.path.join(dst, name))
            f = newest(os.path.join(d, "stats_cfg%d" % c, "*", "*_kernel_stats.csv"))
            if f:
                shutil.copy(f, os.path.join(dst, "kernel_stats_cfg%d.csv" % c))
            if rows[c]:
                with open(os.path.join(dst, "pmc_cfg%d.csv" % c), "w") as fh:
                    w = csv.writer(fh)
                    w.writerow(["pass", "kernel", "counter", "avg_value_per_dispatch"])
                    w.writerows(rows[c])
        for name in ("ab_kernels.txt", "ablation.txt", "ablation_cfg2.txt", "stamps.txt",
                     "stamps_cfg2.txt", "reconcile.txt", "probes.txt", "bo_loop.json",
                     "swarm_small.txt", "product_kernels.txt", "multirank_path_cost.txt",
                     "bench_default.json", "small_n.txt", "small_n_few.txt", "high_d.txt", "ab_tables.txt",
                     "small_step.txt", "mid_kernel.txt"):
            if os.path.exists(os.path.join(d, name)):
                shutil.copy(os.path.join(d, name), os.path.join(dst, name))
        tj = os.path.join(ROOT, "profiles", "traffic.json")
        cur = {}
        if os.path.exists(tj):
            cur = json.load(open(tj))
        # where the bytes beyond the algorithmic ones come from (scripts/dev/probe_mall.hip,
        # profiles/<tag>/served_by.txt): average outstanding time of the L2's read requests
        sb = os.path.join(dst, "served_by.json")
        served = json.load(open(sb)) if os.path.exists(sb) else None
        for k, v in traffic.items():
            v["note"] += "; source profiles/%s/pmc_%s.csv" % (tag, k.replace("config", "cfg"))
            c = k.replace("config", "cfg")
            if served and c in served["sweeps"]:
                lat = served["sweeps"][c]
                v["served_by"] = dict(
                    level="infinity cache" if lat < 0.5 * (served["hbm_reference_cycles"] +
                                                          served["infinity_cache_reference_cycles"])
                    else "hbm",
                    fabric_read_latency_cycles=lat,
                    hbm_reference_cycles=served["hbm_reference_cycles"],
                    infinity_cache_reference_cycles=served["infinity_cache_reference_cycles"],
                    source="profiles/%s/served_by.txt (TCC_EA0_RDREQ_LEVEL_sum / TCC_EA0_RDREQ_sum "
                           "against one-load-in-flight reference kernels)" % tag)
            cur[k] = v
        json.dump(cur, open(tj, "w"), indent=1)
        print("published", sorted(os.listdir(dst)), file=sys.stderr)


if __name__ == "__main__":
    main(sys.argv)
