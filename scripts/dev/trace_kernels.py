"""Kernel timeline of a command's last N launches from a rocprofv3 kernel trace CSV:
    python scripts/dev/trace_kernels.py trace.csv [anchor-substring] [before] [after]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
anchor = sys.argv[2] if len(sys.argv) > 2 else None
before = int(sys.argv[3]) if len(sys.argv) > 3 else 10
after = int(sys.argv[4]) if len(sys.argv) > 4 else 40
i0 = 0
if anchor:
    hits = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
    i0 = hits[0] if hits else 0
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[max(0, i0 - before):i0 + after]:
    name = r["Kernel_Name"]
    name = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:34]
    print("%-36s grid %-16s start %10.1f us  dur %9.1f us" % (
        name, "%sx%sx%s" % (r.get("Grid_Size_X", ""), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", "")),
        (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
