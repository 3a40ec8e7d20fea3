#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r04h; mkdir -p $OUT; cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tensor_grid or grid_sweep or reduced_configs" > $OUT/pytest.txt 2>&1
tail -25 $OUT/pytest.txt
for rep in 1 2; do
  AB_SEP=0 AB_ONLY=classic AB_TAG="generic" python scripts/dev/ab_sweep.py 2 2>&1 | tail -1
  AB_ONLY=classic AB_TAG="tables" python scripts/dev/ab_sweep.py 2 2>&1 | tail -1
done | tee $OUT/ab.txt
