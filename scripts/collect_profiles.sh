#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 evidence for profiles/<tag>/.
#   scripts/collect_profiles.sh r03 [configs...]      (default configs: 3 2 4 5)
# Per config: the bench JSON line, kernel stats of the same command, and three
# separate PMC passes (MFMA busy + clock; FETCH_SIZE; WRITE_SIZE) -- never
# combined with trace domains other than --kernel-trace, as the MI355X guide
# prescribes.  Config 3 (the north-star config) also gets the instruction-mix /
# issue / L2 counters.  Then: A/B of the two sweep kernels, ablation and phase
# stamps of the paired kernel (variant builds under scripts/dev/ab/, made by
# scripts/dev/build_variant.sh before the call), hardware probes, BO-loop and
# small-swarm timings.
TAG=${1:-r03}; shift
CFGS=${@:-3 2 4 5}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles_$TAG
mkdir -p $OUT
cd $R
[ -z "$ONLY_PMC" ] && for c in $CFGS; do
  extra=""; [ $c = 4 ] && extra="--warmup 2 --profile-steps 3"
  python bench.py --config $c $extra > $OUT/bench_cfg$c.json 2> $OUT/bench_cfg$c.err
done
cd /tmp && export TMPDIR=/tmp
SHORT="--steps 3 --warmup 1 --profile-steps 1 --no-cpu-baseline --no-check-chosen --no-shared-pass"
for c in $CFGS; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_cfg$c -- \
    python $R/bench.py --config $c --steps 5 --warmup 2 --profile-steps 2 --no-cpu-baseline --no-check-chosen --no-shared-pass > $OUT/stats_cfg$c.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES \
    --output-format csv -d $OUT/pmc_mfma_cfg$c -- python $R/bench.py --config $c $SHORT > $OUT/pmc_mfma_cfg$c.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_cfg$c -- \
    python $R/bench.py --config $c $SHORT > $OUT/pmc_fetch_cfg$c.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_cfg$c -- \
    python $R/bench.py --config $c $SHORT > $OUT/pmc_write_cfg$c.log 2>&1
done
if echo " $CFGS " | grep -q " 3 "; then
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM \
    --output-format csv -d $OUT/pmc_insts_cfg3 -- python $R/bench.py --config 3 $SHORT > $OUT/pmc_insts_cfg3.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH \
    --output-format csv -d $OUT/pmc_issue_cfg3 -- python $R/bench.py --config 3 $SHORT > $OUT/pmc_issue_cfg3.log 2>&1
  rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2_cfg3 -- \
    python $R/bench.py --config 3 $SHORT > $OUT/pmc_l2_cfg3.log 2>&1
fi
cd $R
if [ -n "$ONLY_PMC" ]; then python scripts/profiles_digest.py $OUT > $OUT/SUMMARY.txt 2>&1; cat $OUT/SUMMARY.txt; exit 0; fi
# the two sweep kernels side by side (same process, same box), un-shared path
python scripts/dev/ab_sweep.py 3 2 4 5 > $OUT/ab_kernels.txt 2>&1
# ablation ("what does the paired sweep cost without X") and per-phase cycle stamps
if [ -f scripts/dev/ab/instr.so ]; then
  for c in 3 4; do for m in 0 1 2 4 8 16 32 6 7 15; do
    SAFEOPT_HIP_LIB=scripts/dev/ab/instr.so SGP_ABLATE=$m AB_ONLY=pair AB_TAG="ablate $m" timeout 200 python scripts/dev/ab_sweep.py $c 2>&1 | tail -1
  done; done > $OUT/ablation.txt
fi
if [ -f scripts/dev/ab/stamps.so ]; then
  for c in 3 4 5; do
    SAFEOPT_HIP_LIB=scripts/dev/ab/stamps.so AB_ONLY=pair AB_TAG=stamps timeout 200 python scripts/dev/ab_sweep.py $c 2>&1 | tail -3
  done > $OUT/stamps.txt
fi
# hardware probes behind the design decisions (standalone HIP programs)
{
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o /tmp/probe1 scripts/dev/probe_r02.hip && /tmp/probe1 | grep -v "^[ABD][0-9]*:"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o /tmp/probe2 scripts/dev/probe_coexec.hip && /tmp/probe2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w -o /tmp/probe3 scripts/dev/probe_interleave.hip && /tmp/probe3
} > $OUT/probes.txt 2>&1
# product kernels (the reference's context example) on both sweep kernels; cost of the
# N-rank control flow on one GPU (one-rank RCCL communicator posing as world 2)
python scripts/dev/ab_product.py 64 200 256 > $OUT/product_kernels.txt 2>&1
python scripts/dev/multirank_path_cost.py 2>&1 | tail -6 > $OUT/multirank_path_cost.txt
# what a user of the drop-in sees per BO iteration (incremental path) and per
# SafeOptSwarm.optimize() with the default swarm
{ python scripts/bench_bo_loop.py --config 2; python scripts/bench_bo_loop.py --config 3; } > $OUT/bo_loop.json 2>$OUT/bo_loop.err
python scripts/dev/swarm_small.py > $OUT/swarm_small.txt 2>&1
python scripts/profiles_digest.py $OUT > $OUT/SUMMARY.txt 2>&1
cat $OUT/SUMMARY.txt
