#!/bin/bash
# small factors resident in LDS for the whole launch (4-wave kernel) against streamed
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r04t; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "resident or grid_sweep or tensor_grid or shared_factor or golden or product" > $OUT/pytest.txt 2>&1
tail -6 $OUT/pytest.txt
python scripts/dev/small_n.py 8 20 48 64 96 112 2>&1 | tee $OUT/small_n_resident.txt
SGP_NO_RESIDENT=1 python scripts/dev/small_n.py 8 20 48 64 96 112 2>&1 | tee $OUT/small_n_streamed.txt
