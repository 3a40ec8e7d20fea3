cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06a
(timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "grid_sweep_both or predict_noiseless or product_kernels or tensor_grid or split_remainder or shared_factor or swarm_fitness_both or full_size or reduced_configs" 2>&1 | tail -15) > gpurun_out/r06a/tests.txt
for rep in 1 2; do
for v in pair pair-unmerged; do
  AB_ONLY=$v AB_TAG=$v timeout 300 python scripts/dev/ab_sweep.py 3 4 5 2>&1 | grep "^cfg"
done; done > gpurun_out/r06a/ab.txt 2>&1
AB_SHARE=1 AB_ONLY=pair AB_TAG=shared timeout 300 python scripts/dev/ab_sweep.py 3 2>&1 | grep "^cfg" >> gpurun_out/r06a/ab.txt
AB_SHARE=1 AB_ONLY=pair-unmerged AB_TAG=shared-unmerged timeout 300 python scripts/dev/ab_sweep.py 3 2>&1 | grep "^cfg" >> gpurun_out/r06a/ab.txt
cat gpurun_out/r06a/tests.txt gpurun_out/r06a/ab.txt
