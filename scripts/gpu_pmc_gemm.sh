#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $R/gpurun_out/pmc; export TMPDIR=/tmp; cd /tmp
for v in -1 -2; do
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
     --output-format csv -d $R/gpurun_out/pmc/v$v -o pmc -- python3 $R/scripts/run_one_gemm.py $v 10240 6144 2048 > $R/gpurun_out/pmc/v$v.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_SALU \
     --output-format csv -d $R/gpurun_out/pmc/w$v -o pmc -- python3 $R/scripts/run_one_gemm.py $v 10240 6144 2048 > $R/gpurun_out/pmc/w$v.log 2>&1
done
cd $R; python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/pmc/*/')):
    for f in glob.glob(d + '**/*counter_collection.csv', recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            kn = r['Kernel_Name'][:60]
            agg[kn][r['Counter_Name']] += float(r['Counter_Value']); cnt[(kn, r['Counter_Name'])] += 1
        for kn, c in agg.items():
            if 'gemm' in kn.lower() or 'Cijk' in kn or 'cijk' in kn.lower():
                print(d, kn, {k: round(v / cnt[(kn, k)]) for k, v in c.items()})
PY
