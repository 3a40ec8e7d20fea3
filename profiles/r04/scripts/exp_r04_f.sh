#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r04f; mkdir -p $OUT; cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "grid_sweep or predict_noiseless or shared_factor or swarm_fitness_both or reduced_configs or full_size_config2 or replay or product" > $OUT/pytest.txt 2>&1
tail -15 $OUT/pytest.txt
for rep in 1 2; do
  SGP_NO_NARROW=1 AB_ONLY=classic AB_TAG="asm no_narrow" python scripts/dev/ab_sweep.py 2 2>&1 | tail -1
  AB_ONLY=classic AB_TAG="asm narrow" python scripts/dev/ab_sweep.py 2 2>&1 | tail -1
done | tee $OUT/ab.txt
