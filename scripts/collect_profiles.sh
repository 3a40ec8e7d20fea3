#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 evidence for profiles/<tag>/.
#   scripts/collect_profiles.sh r04 [configs...]      (default configs: 3 2 4 5; QUICK=1: without the
#   reconcile / probes / product-kernel / swarm blocks, whose code did not change since r04)
# Per config: the bench JSON line, kernel stats of the same command, and three
# separate PMC passes (MFMA busy + clock; FETCH_SIZE; WRITE_SIZE) -- never
# combined with trace domains other than --kernel-trace, as the MI355X guide
# prescribes.  Config 3 (the north-star config) also gets the instruction-mix /
# issue / L2 counters.  Then: A/B of the two sweep kernels, ablation and phase
# stamps of the paired kernel (variant builds under scripts/dev/ab/, made by
# scripts/dev/build_variant.sh before the call), hardware probes, BO-loop and
# small-swarm timings.
TAG=${1:-r04}; shift
CFGS=${@:-3 2 4 5}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles_$TAG
mkdir -p $OUT
cd $R
if [ -z "$ONLY_PMC" ]; then
  # the driver's command (config 3 + bo_iteration / sets_roofline / rank1_roofline /
  # config4_strong / shared_factor keys), then every config on its own
  python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
  for c in $CFGS; do
    extra=""; [ $c = 4 ] && extra="--warmup 2 --profile-steps 3"
    python bench.py --config $c $extra > $OUT/bench_cfg$c.json 2> $OUT/bench_cfg$c.err
  done
fi
cd /tmp && export TMPDIR=/tmp
SHORT="--steps 3 --warmup 1 --profile-steps 1 --no-extras --no-cpu-baseline --no-check-chosen --no-shared-pass"
# (kernel stats: enough steps that the launches of the clock ramp -- the first ~30 ms of
# load run ~10 % slower, profiles/r04/clock_ramp.txt -- are a small part of the average)
declare -A STEPS=([2]="--steps 300 --warmup 20 --profile-steps 100" [3]="--steps 40 --warmup 5 --profile-steps 20" [4]="--steps 5 --warmup 2 --profile-steps 3" [5]="--steps 30 --warmup 5 --profile-steps 20")
for c in $CFGS; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_cfg$c -- \
    python $R/bench.py --config $c ${STEPS[$c]} --no-extras --no-cpu-baseline --no-check-chosen --no-shared-pass > $OUT/stats_cfg$c.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES \
    --output-format csv -d $OUT/pmc_mfma_cfg$c -- python $R/bench.py --config $c $SHORT > $OUT/pmc_mfma_cfg$c.log 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_cfg$c -- \
    python $R/bench.py --config $c $SHORT > $OUT/pmc_fetch_cfg$c.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_cfg$c -- \
    python $R/bench.py --config $c $SHORT > $OUT/pmc_write_cfg$c.log 2>&1
done
for c in 3 2; do echo " $CFGS " | grep -q " $c " || continue
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM \
    --output-format csv -d $OUT/pmc_insts_cfg$c -- python $R/bench.py --config $c $SHORT > $OUT/pmc_insts_cfg$c.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH \
    --output-format csv -d $OUT/pmc_issue_cfg$c -- python $R/bench.py --config $c $SHORT > $OUT/pmc_issue_cfg$c.log 2>&1
  rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_l2_cfg$c -- \
    python $R/bench.py --config $c $SHORT > $OUT/pmc_l2_cfg$c.log 2>&1
done
cd $R
if [ -n "$ONLY_PMC" ]; then python scripts/profiles_digest.py $OUT > $OUT/SUMMARY.txt 2>&1; cat $OUT/SUMMARY.txt; exit 0; fi
# the two sweep kernels side by side (same process, same box), un-shared path
python scripts/dev/ab_sweep.py 3 2 4 5 > $OUT/ab_kernels.txt 2>&1
# factor tables (tensor grids, RBF) against evaluated covariances: config 2 (4-wave kernel:
# tables by default) and config 4 (paired kernel: tables only while they fit half an L2 --
# the default evaluates; SGP_SEP_PAIR=1 forces them)
{ AB_ONLY=auto AB_TAG="  [config 2, automatic choice since round 5: k_sweep_mid in passes of row blocks, factor tables]" python scripts/dev/ab_sweep.py 2 2>&1 | tail -1
  AB_ONLY=classic AB_TAG="  [config 2, 4-wave kernel, factor tables (the choice until round 5)]" python scripts/dev/ab_sweep.py 2 2>&1 | tail -1
  AB_ONLY=classic AB_SEP=0 AB_TAG="  [config 2, evaluated]" python scripts/dev/ab_sweep.py 2 2>&1 | tail -1
  AB_ONLY=pair AB_TAG="  [config 4, default: evaluated (tables 4.8 MB > half an L2)]" python scripts/dev/ab_sweep.py 4 2>&1 | tail -1
  SGP_SEP_PAIR=1 AB_ONLY=pair AB_TAG="  [config 4, factor tables forced (SGP_SEP_PAIR=1)]" python scripts/dev/ab_sweep.py 4 2>&1 | tail -1
  AB_ONLY=pair AB_SEP=0 AB_TAG="  [config 4, no axes declared]" python scripts/dev/ab_sweep.py 4 2>&1 | tail -1
} > $OUT/ab_tables.txt 2>&1
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
  # the 4-wave kernel at config 2: factor tables / evaluated covariances
  { SAFEOPT_HIP_LIB=scripts/dev/ab/stamps.so AB_ONLY=classic AB_TAG="stamps, factor tables" timeout 200 python scripts/dev/ab_sweep.py 2 2>&1 | tail -3
    SAFEOPT_HIP_LIB=scripts/dev/ab/stamps.so AB_SEP=0 AB_ONLY=classic AB_TAG="stamps, evaluated" timeout 200 python scripts/dev/ab_sweep.py 2 2>&1 | tail -3
    SAFEOPT_HIP_LIB=scripts/dev/ab/stamps.so timeout 300 python scripts/dev/small_n.py 8 20 64 2>&1
  } > $OUT/stamps_cfg2.txt
fi
if [ -f scripts/dev/ab/instr.so ]; then
  for sep in 1 0; do for m in 0 1 2 4 8 16 32; do
    SAFEOPT_HIP_LIB=scripts/dev/ab/instr.so SGP_ABLATE=$m AB_SEP=$sep AB_ONLY=classic AB_TAG="tables=$sep ablate $m" timeout 200 python scripts/dev/ab_sweep.py 2 2>&1 | tail -1
  done; done > $OUT/ablation_cfg2.txt
fi
# the reference's own regime (n <= 256) and SafeOptSwarm's input dimensions
python scripts/dev/small_n.py > $OUT/small_n.txt 2>&1
python scripts/dev/small_n.py 4 8 16 20 32 48 64 > $OUT/small_n_few.txt 2>&1
python scripts/dev/high_d.py > $OUT/high_d.txt 2>&1
# the reference's own problem sizes: the one-launch step against the large-grid path
python scripts/dev/small_step_time.py > $OUT/small_step.txt 2>&1
# 49 .. 128 observations: the resident-factor kernel against the 4-wave kernel and the oracle,
# its counters at n = 64 and 128 (separate PMC passes, kernel trace only)
{ python scripts/dev/mid_check.py; bash scripts/dev/mid_pmc.sh 2>&1 | grep dispatches; } > $OUT/mid_kernel.txt 2>&1
if [ -n "$QUICK" ]; then
  { python scripts/bench_bo_loop.py --config 2; python scripts/bench_bo_loop.py --config 3; } > $OUT/bo_loop.json 2>$OUT/bo_loop.err
  python scripts/profiles_digest.py $OUT > $OUT/SUMMARY.txt 2>&1; cat $OUT/SUMMARY.txt; exit 0
fi
# hipEvent vs rocprof on identical launches, and the clock ramp
{
  for mult in 1 8; do python scripts/dev/clock_reconcile.py $mult 40; done
  for mult in 1 8; do
    ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rp_x$mult -- python $R/scripts/dev/clock_reconcile.py $mult 40 2>&1 | tail -1 )
    f=$(find $OUT/rp_x$mult -name "*kernel_stats.csv" | head -1); echo "  rocprofv3 kernel stats of the same process:"; grep -i "k_sweep" $f | head -2
  done
} > $OUT/reconcile.txt 2>&1
# (hardware probes behind the design decisions: profiles/r02/probes.txt, profiles/r03/probes.txt)
# product kernels (the reference's context example) on both sweep kernels; cost of the
# N-rank control flow on one GPU (one-rank RCCL communicator posing as world 2)
python scripts/dev/ab_product.py 64 200 256 > $OUT/product_kernels.txt 2>&1
python scripts/dev/multirank_path_cost.py 2>&1 | tail -6 > $OUT/multirank_path_cost.txt
# what a user of the drop-in sees per BO iteration (incremental path) and per
# SafeOptSwarm.optimize() with the default swarm
{ python scripts/bench_bo_loop.py --config 2; python scripts/bench_bo_loop.py --config 3; } > $OUT/bo_loop.json 2>$OUT/bo_loop.err
python scripts/dev/swarm_small.py > $OUT/swarm_small.txt 2>&1
# where the L2 misses are served (fabric read latency against reference kernels)
cp profiles/r04/scripts/exp_r04_r.sh /tmp/served_by.sh
bash /tmp/served_by.sh > $OUT/served_by.log 2>&1
python scripts/profiles_digest.py $OUT > $OUT/SUMMARY.txt 2>&1
cat $OUT/SUMMARY.txt
