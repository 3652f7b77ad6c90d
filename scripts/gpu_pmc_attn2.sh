#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $R/gpurun_out/pmca2; export TMPDIR=/tmp; cd /tmp
rocprofv3 --pmc LdsLatency SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_IDX_ACTIVE \
   --output-format csv -d $R/gpurun_out/pmca2/p1 -o pmc -- python3 $R/scripts/bench_attn.py > $R/gpurun_out/pmca2/p1.log 2>&1
rocprofv3 --pmc SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_STALL SQ_LDS_IDX_ACTIVE \
   --output-format csv -d $R/gpurun_out/pmca2/p2 -o pmc -- python3 $R/scripts/bench_attn.py > $R/gpurun_out/pmca2/p2.log 2>&1
cd $R; tail -3 gpurun_out/pmca2/p2.log | cut -c1-300; python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/pmca2/*/')):
    for f in glob.glob(d + '**/*counter_collection.csv', recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            kn = r['Kernel_Name'].replace('(anonymous namespace)::', '')[:40]
            agg[kn][r['Counter_Name']] += float(r['Counter_Value']); cnt[(kn, r['Counter_Name'])] += 1
        for kn, c in agg.items():
            if 'attn' in kn: print(kn, {k: round(v / cnt[(kn, k)], 1) for k, v in c.items()})
PY
