#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r04o; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tensor_grid or product or grid_sweep or split_remainder or reduced_configs" > $OUT/pytest.txt 2>&1
tail -8 $OUT/pytest.txt
for rep in 1 2; do
  AB_ONLY=pair AB_TAG="tables" timeout 300 python scripts/dev/ab_sweep.py 4 2>&1 | grep "^cfg"
  AB_SEP=0 AB_ONLY=pair AB_TAG="evaluated" timeout 300 python scripts/dev/ab_sweep.py 4 2>&1 | grep "^cfg"
done | tee $OUT/ab.txt
python scripts/dev/high_d.py 2>&1 | tee $OUT/high_d.txt
