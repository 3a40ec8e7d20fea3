#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r04d; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_list.txt 2>&1
P="python $R/scripts/dev/clock_reconcile.py 1 40"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $OUT/p1 -- $P > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH --output-format csv -d $OUT/p2 -- $P > $OUT/p2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_MOPS_F64 --output-format csv -d $OUT/p3 -- $P > $OUT/p3.log 2>&1
cd $R
python - <<'PY' > $OUT/summary.txt
import csv, glob, collections
for p in ("p1","p2","p3"):
    for f in glob.glob("gpurun_out/r04d/%s/*/*counter_collection.csv" % p):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "k_sweep" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            v = v[len(v)//2:]      # warm dispatches
            print(p, k, "%.4g" % (sum(v)/len(v)), "n=%d" % len(v))
PY
cat $OUT/summary.txt; tail -2 $OUT/p3.log
