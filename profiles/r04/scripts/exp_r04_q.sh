#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r04q; mkdir -p $OUT; cd $R
for nb in 64 128 256 512 1024; do
SGP_SETS_BLOCKS=$nb python bench.py --config 2 --no-cpu-baseline --no-check-chosen --no-shared-pass > $OUT/bench2_$nb.json 2>$OUT/bench2.err
python - $nb <<'PY'
import json,sys
j=json.loads(open("gpurun_out/r04q/bench2_%s.json"%sys.argv[1]).read().strip().splitlines()[-1])
print("blocks", sys.argv[1], "cfg2 ms/step %.4f"%j["ms_per_step"], "value %.4g"%j["value"], "sweep frac", round(j["roofline"]["frac"],3), "sets ms %.4f"%j["sets_roofline"]["ms"], "bo_iter ms %.4f"%j["bo_iteration"]["ms"])
PY
done
