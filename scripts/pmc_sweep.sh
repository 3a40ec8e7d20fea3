#!/bin/bash
# PMC breakdown of k_sweep (run on the GPU box): where do the wave cycles go?
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/pmc_sweep_$1; CFG=${2:-2}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/a -- python $R/bench.py --config $CFG --steps 3 --warmup 1 --no-cpu-baseline > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d $OUT/b -- python $R/bench.py --config $CFG --steps 3 --warmup 1 --no-cpu-baseline > $OUT/b.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/c -- python $R/bench.py --config $CFG --steps 3 --warmup 1 --no-cpu-baseline > $OUT/c.log 2>&1
cd $R
python - <<PY
import csv, glob, collections
for sub in "abc":
    fs = glob.glob("$OUT/%s/*/*_counter_collection.csv" % sub)
    if not fs: print("no data", sub); continue
    agg = collections.defaultdict(list); dur=[]
    for x in csv.DictReader(open(fs[0])):
        if "k_sweep" in x["Kernel_Name"]:
            agg[x["Counter_Name"]].append(float(x["Counter_Value"]))
    for k, v in sorted(agg.items()):
        print("%-32s %.4g" % (k, sum(v)/len(v)))
PY
