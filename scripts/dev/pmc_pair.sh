#!/bin/bash
# instruction mix / issue counters of the paired sweep at config 3 (separate PMC passes, kernel trace only)
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/pmc_pair; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export AB_ONLY=${AB_ONLY:-pair}
CFG=${1:-3}
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
  --output-format csv -d $OUT/insts -- python $R/scripts/dev/ab_sweep.py $CFG > $OUT/insts.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA \
  --output-format csv -d $OUT/issue -- python $R/scripts/dev/ab_sweep.py $CFG > $OUT/issue.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH \
  --output-format csv -d $OUT/mfma -- python $R/scripts/dev/ab_sweep.py $CFG > $OUT/mfma.log 2>&1
cd $R
python - <<'PY'
import csv, glob, collections, os
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "pmc_pair")
for name in ("insts", "issue", "mfma"):
    fs = glob.glob(os.path.join(out, name, "*", "*_counter_collection.csv"))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in fs:
        for x in csv.DictReader(open(f)):
            if "k_sweep" in x["Kernel_Name"]:
                k = "pair" if "pair" in x["Kernel_Name"] else "classic"
                agg[k][x["Counter_Name"]].append(float(x["Counter_Value"]))
    for k, d in agg.items():
        print(name, k, {c: "%.4g" % (sum(v) / len(v)) for c, v in sorted(d.items())})
PY
