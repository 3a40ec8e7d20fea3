# kernel timeline of ONE optimize() of a converged state (big passes); ARGS = arguments of no_expander.py
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tr
mkdir -p $GRAFT_REPO_ROOT/gpurun_out
ONLY_BIG=1 rocprofv3 --kernel-trace -d /tmp/tr -o tr --output-format csv -- python $GRAFT_REPO_ROOT/scripts/dev/no_expander.py ${ARGS:-1000 margin=0.05 ls=0.7 rings=5 dring=0.3 dmid=0.8 dtop=0.4 r0=2.0 dout=1.4 plateau=0.6} > /tmp/tr.log 2>&1
tail -3 /tmp/tr.log | cut -c1-200
F=$(find /tmp/tr -name "*kernel_trace.csv" | head -1)
python - "$F" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
hits = [i for i, r in enumerate(rows) if "k_sweep" in r["Kernel_Name"]]
i0 = hits[-1]
t0 = int(rows[i0]["Start_Timestamp"])
prev_end = t0
for r in rows[i0:]:
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:40]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-42s grid %-10s start %9.1f us  dur %8.1f us  gap %7.1f" % (name, r.get("Grid_Size_X", ""), (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3))
    prev_end = e
PY
