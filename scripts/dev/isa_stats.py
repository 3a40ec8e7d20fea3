#!/usr/bin/env python
"""Per-kernel instruction statistics of a gfx950 assembly listing (hipcc -S --cuda-device-only):
MFMAs, SGPR spills to lanes (v_writelane / v_readlane), scratch traffic, s_waitcnt, branches.

    python scripts/dev/isa_stats.py file.s [name-filter]
"""
import re
import subprocess
import sys


def main(path, flt=None):
    names, bodies, cur = [], [], None
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = []
            names.append(m.group(1))
            bodies.append(cur)
        elif cur is not None:
            if ".end_amdhsa_kernel" in line or line.startswith("\t.section"):
                cur = None
            else:
                cur.append(line)
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True,
                         text=True).stdout.split("\n")
    for name, body in zip(dem, bodies):
        name = name.replace("(anonymous namespace)::", "").split("(")[0]
        if flt and flt not in name:
            continue
        txt = "".join(body)
        c = lambda pat: len(re.findall(pat, txt))
        print("%-44s insts %6d mfma %5d writelane %3d readlane %3d scratch %3d waitcnt %4d "
              "branch %4d ds %4d v_mov %4d" % (
                  name[:44], c(r"\n\t[a-z]"), c(r"v_mfma"), c(r"v_writelane_b32"),
                  c(r"v_readlane_b32"), c(r"scratch_(?:load|store)"), c(r"s_waitcnt"),
                  c(r"s_c?branch"), c(r"\tds_"), c(r"\tv_mov_b")))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
