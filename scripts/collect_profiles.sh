#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 evidence for profiles/.
#   scripts/collect_profiles.sh r01
# Writes gpurun_out/profiles_<tag>/ : kernel stats (csv) of the bench command,
# PMC passes (MFMA busy / FETCH_SIZE / WRITE_SIZE, separate runs as the
# MI355X guide prescribes), the bench JSON lines and the issue-rate probes.
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles_$TAG
mkdir -p $OUT
cd $R
python bench.py --steps 20 --warmup 3 > $OUT/bench_cfg2.json 2> $OUT/bench_cfg2.err
python bench.py --config 3 --steps 5 --warmup 2 > $OUT/bench_cfg3.json 2> $OUT/bench_cfg3.err
python bench.py --config 4 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_cfg4_1gpu.json 2> $OUT/bench_cfg4.err
python bench.py --config 5 --steps 5 --warmup 1 > $OUT/bench_cfg5.json 2> $OUT/bench_cfg5.err
python scripts/bench_bo_loop.py --config 2 > $OUT/bo_loop.json 2> /dev/null
python scripts/bench_bo_loop.py --config 3 >> $OUT/bo_loop.json 2> /dev/null
python scripts/microbench.py > $OUT/microbench.txt 2>&1
python scripts/stagebench.py > $OUT/stagebench.txt 2>&1
bash scripts/ablate.sh 2 3 4 5 > $OUT/ablation.txt 2>&1
cd /tmp && export TMPDIR=/tmp
for c in 2 3; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_cfg$c -- \
    python $R/bench.py --config $c --steps 5 --warmup 2 --no-cpu-baseline > $OUT/stats_cfg$c.log 2>&1
done
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES \
  --output-format csv -d $OUT/pmc_mfma_cfg2 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_mfma.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_cfg2 -- \
  python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_cfg2 -- \
  python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA \
  --output-format csv -d $OUT/pmc_lds_cfg2 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_lds.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SALU \
  --output-format csv -d $OUT/pmc_issue_cfg2 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_issue.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL \
  --output-format csv -d $OUT/pmc_coexec_cfg2 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_coexec.log 2>&1
cd $R
python scripts/summarize_profiles.py $OUT > $OUT/SUMMARY.txt 2>&1
cat $OUT/SUMMARY.txt
