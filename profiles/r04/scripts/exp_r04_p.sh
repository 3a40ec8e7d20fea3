#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r04p; mkdir -p $OUT; cd $R
timeout 1500 python -m pytest tests -x -q -m gpu ${PYTEST_K} > $OUT/pytest.txt 2>&1
tail -5 $OUT/pytest.txt
python bench.py --config 2 --no-cpu-baseline --no-check-chosen > $OUT/bench2.json 2>$OUT/bench2.err
python - <<'PY'
import json
j=json.loads(open("gpurun_out/r04p/bench2.json").read().strip().splitlines()[-1])
print("cfg2 ms/step", j["ms_per_step"], "value %.4g"%j["value"], "roofline", j["roofline"]["frac"])
for k in ("bo_iteration","sets_roofline","rank1_roofline"):
    print(k, j.get(k))
PY
