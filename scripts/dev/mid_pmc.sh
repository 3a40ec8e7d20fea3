cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for n in 64 128; do
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES --output-format csv -d $R/gpurun_out/midpmc/a$n -- python $R/scripts/dev/mid_run.py $n RBF 12 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_INSTS_VMEM_RD --output-format csv -d $R/gpurun_out/midpmc/b$n -- python $R/scripts/dev/mid_run.py $n RBF 12 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH --output-format csv -d $R/gpurun_out/midpmc/c$n -- python $R/scripts/dev/mid_run.py $n RBF 12 > /dev/null 2>&1
done
python3 - <<'PY'
import csv,glob,os,collections
R=os.environ['GRAFT_REPO_ROOT']
for d in sorted(glob.glob(R+'/gpurun_out/midpmc/*')):
    for f in glob.glob(d+'/*/*counter_collection.csv'):
        acc=collections.defaultdict(float); n=0
        for r in csv.DictReader(open(f)):
            if 'k_sweep_mid' not in r['Kernel_Name']: continue
            acc[r['Counter_Name']]+=float(r['Counter_Value'])
        disp=len(set(r['Dispatch_Id'] for r in csv.DictReader(open(f)) if 'k_sweep_mid' in r['Kernel_Name']))
        print(os.path.basename(d), 'dispatches', disp, {k: round(v/max(disp,1)) for k,v in acc.items()})
PY
