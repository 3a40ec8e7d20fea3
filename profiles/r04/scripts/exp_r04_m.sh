#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r04m; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "grid_sweep or split_remainder or shared_factor or swarm_fitness_both or reduced_configs or product or tensor_grid" > $OUT/pytest.txt 2>&1
tail -6 $OUT/pytest.txt
for rep in 1 2; do
  for v in cur oldpair; do
    lib=scripts/dev/ab/$v.so; [ $v = cur ] && lib=safeopt_amd/libsafeopt_hip.so
    SAFEOPT_HIP_LIB=$lib AB_SEP=0 AB_ONLY=pair AB_TAG=$v timeout 300 python scripts/dev/ab_sweep.py 3 4 5 2>&1 | grep "^cfg"
  done
done | tee $OUT/ab.txt
AB_ONLY=classic AB_TAG="k_sweep tables" python scripts/dev/ab_sweep.py 2 2>&1 | tail -1
AB_SEP=0 AB_ONLY=classic AB_TAG="k_sweep generic" python scripts/dev/ab_sweep.py 2 2>&1 | tail -1
