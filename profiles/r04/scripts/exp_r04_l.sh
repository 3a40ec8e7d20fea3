#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r04l; mkdir -p $OUT; cd $R
python bench.py --no-cpu-baseline --no-check-chosen > $OUT/default.json 2> $OUT/default.err; tail -3 $OUT/default.err
python bench.py --config 2 --no-cpu-baseline --no-check-chosen > $OUT/cfg2.json 2> $OUT/cfg2.err; tail -3 $OUT/cfg2.err
python bench.py --config 5 --no-cpu-baseline > $OUT/cfg5.json 2> $OUT/cfg5.err; tail -3 $OUT/cfg5.err
python - <<'PY'
import json
for f in ("default","cfg2","cfg5"):
    try:
        j=json.load(open("gpurun_out/r04l/%s.json"%f))
    except Exception as e:
        print(f, "ERR", e); continue
    print(f, j["value"], j["ms_per_step"], j["roofline"]["frac"], j["roofline"]["kernel_ms_avg"])
    for k in ("sets_roofline","bo_iteration","config4_strong","three_call_step","shared_factor","extras_error"):
        if k in j: print("   ", k, json.dumps(j[k])[:400])
PY
