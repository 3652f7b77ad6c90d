#!/bin/bash
# Memory-side PMC comparison of one GEMM shape: production (-1), hipBLASLt via torch (-2), ring variant (50).  usage: gpu_pmc_mem.sh M N K
R=${GRAFT_REPO_ROOT:-$(pwd)}; mkdir -p $R/gpurun_out/pmcmem; export TMPDIR=/tmp; cd /tmp
M=${1:-8192}; N=${2:-2048}; K=${3:-10240}
for v in ${VARIANTS:--2 50}; do
  i=0
  for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" \
             ; do
    i=$((i+1))
    timeout 150 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmcmem/v${v}_$i -o pmc -- python3 $R/scripts/run_one_gemm.py $v $M $N $K > $R/gpurun_out/pmcmem/v${v}_$i.log 2>&1
  done
done
cd $R; python3 - <<'PY'
import csv, glob, collections
res = collections.defaultdict(dict)
for d in sorted(glob.glob('gpurun_out/pmcmem/*/')):
    for f in glob.glob(d + '**/*counter_collection.csv', recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            kn = r['Kernel_Name'][:40]
            agg[kn][r['Counter_Name']] += float(r['Counter_Value']); cnt[(kn, r['Counter_Name'])] += 1
        for kn, c in agg.items():
            if 'gemm' in kn.lower() or 'cijk' in kn.lower():
                for k, v in c.items(): res[kn][k] = round(v / cnt[(kn, k)])
for kn, c in res.items(): print(kn, c)
PY
