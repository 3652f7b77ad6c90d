#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $R/gpurun_out/pmctn; export TMPDIR=/tmp; cd /tmp
for shape in "10240 6144 2048" "10240 8192 2048"; do
  tag=$(echo $shape | tr ' ' '_')
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE \
     --output-format csv -d $R/gpurun_out/pmctn/$tag -o pmc -- python3 $R/scripts/run_one_tn.py $shape > $R/gpurun_out/pmctn/$tag.log 2>&1
done
cd $R; python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/pmctn/*/')):
    for f in glob.glob(d + '**/*counter_collection.csv', recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            kn = r['Kernel_Name'][:70]
            agg[kn][r['Counter_Name']] += float(r['Counter_Value']); cnt[(kn, r['Counter_Name'])] += 1
        for kn, c in agg.items():
            if 'gemm' in kn.lower(): print(d[-18:], kn[30:], {k: round(v / cnt[(kn, k)]) for k, v in c.items()})
PY
