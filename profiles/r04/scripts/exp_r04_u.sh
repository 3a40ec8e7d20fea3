#!/bin/bash
# the VALU kernel for GPs with <= 32 observations (sweep_tiny.hip)
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r04u; mkdir -p $OUT; cd $R
timeout 1500 python -m pytest tests -x -q -m gpu > $OUT/pytest.txt 2>&1
tail -6 $OUT/pytest.txt
python scripts/dev/small_n.py 4 8 16 20 32 48 64 2>&1 | tee $OUT/small_n.txt
