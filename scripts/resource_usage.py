#!/usr/bin/env python
"""Register / scratch / LDS use of every kernel instance of the library
(hipcc -Rpass-analysis=kernel-resource-usage, no GPU needed).

    python scripts/resource_usage.py [file.hip ...] > profiles/rNN/resource_usage.txt

A spill shows up in the `scratch` column (bytes per lane).
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "safeopt_amd", "csrc")


def demangle(names):
    r = subprocess.run(["c++filt"], input="\n".join(names),
                       capture_output=True, text=True)
    return r.stdout.split("\n")


def usage(src):
    # the flags of the product build (safeopt_amd/build.py)
    sys.path.insert(0, ROOT)
    from safeopt_amd import build as _b
    extra = list(_b.EXTRA.get(os.path.basename(src), []))
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17",
           "-I", os.path.join(ROOT, "include"), "-I", CSRC, "--cuda-device-only", "-c",
           "-Rpass-analysis=kernel-resource-usage", "-o", "/dev/null", src] + extra
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    rows, cur = [], None
    for line in err.split("\n"):
        m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:"):
            cur = {"name": t.split(":", 1)[1].strip()}
            rows.append(cur)
        elif cur is not None and ":" in t:
            k, v = t.split(":", 1)
            cur[k.strip()] = v.strip()
    return rows


def main():
    files = sys.argv[1:] or [os.path.join(CSRC, f) for f in
                             ("sweep.hip", "sweep_pair.hip", "sweep_tiny.hip", "sets.hip", "swarm.hip",
                              "factor.hip")]
    print("%-88s %5s %5s %5s %8s %5s %8s" % ("kernel", "VGPR", "AGPR", "SGPR", "scratch",
                                            "occ", "LDS"))
    for f in files:
        rows = usage(f)
        names = demangle([r["name"] for r in rows])
        for r, n in zip(rows, names):
            n = re.sub(r"\(anonymous namespace\)::", "", n)
            n = re.sub(r"\(.*", "", n).replace("void ", "")
            print("%-88s %5s %5s %5s %8s %5s %8s" % (
                n[:88], r.get("VGPRs", "?"), r.get("AGPRs", "?"), r.get("TotalSGPRs", "?"),
                r.get("ScratchSize [bytes/lane]", "?"), r.get("Occupancy [waves/SIMD]", "?"),
                r.get("LDS Size [bytes/block]", "?")))


if __name__ == "__main__":
    main()
