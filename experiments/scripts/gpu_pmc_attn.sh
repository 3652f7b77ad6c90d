#!/bin/bash
# PMC counters for the attention kernels (separate passes; no tracing domains combined with --pmc)
R=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $R/gpurun_out/pmca; export TMPDIR=/tmp; cd /tmp
timeout 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
   --output-format csv -d $R/gpurun_out/pmca/p1 -o pmc -- python3 $R/scripts/bench_attn.py > $R/gpurun_out/pmca/p1.log 2>&1
timeout 150 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE \
   --output-format csv -d $R/gpurun_out/pmca/p2 -o pmc -- python3 $R/scripts/bench_attn.py > $R/gpurun_out/pmca/p2.log 2>&1
timeout 150 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD SQ_WAVES SQ_INSTS_MFMA SQ_ACTIVE_INST_FLAT \
   --output-format csv -d $R/gpurun_out/pmca/p3 -o pmc -- python3 $R/scripts/bench_attn.py > $R/gpurun_out/pmca/p3.log 2>&1
cd $R; python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/pmca/*/')):
    for f in glob.glob(d + '**/*counter_collection.csv', recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            kn = r['Kernel_Name'].replace('(anonymous namespace)::', '')[:40]
            agg[kn][r['Counter_Name']] += float(r['Counter_Value']); cnt[(kn, r['Counter_Name'])] += 1
        for kn, c in agg.items():
            if 'attn' in kn: print(kn, {k: round(v / cnt[(kn, k)]) for k, v in c.items()})
PY
