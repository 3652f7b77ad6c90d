#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 600 -p no:cacheprovider -k "gemm" 2>&1 | tail -8 > gpurun_out/pair2_tests.log
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_fullwidth_oracle.py -m gpu -q --timeout 900 -p no:cacheprovider -k "unidisc_s or golden" 2>&1 | tail -6 >> gpurun_out/pair2_tests.log
B="timeout 600 python bench.py --steps 16 --warmup 3 --no-cpu-baseline --table-steps 0"
for i in 1 2; do
  $B --workload unidisc-s-l384 > gpurun_out/p2_s_pair_$i.json 2> /dev/null
  UDM_PAIR_WGRADS=0 $B --workload unidisc-s-l384 > gpurun_out/p2_s_nopair_$i.json 2> /dev/null
done
cat gpurun_out/pair2_tests.log
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/p2_*.json')):
    try:
        s=open(f).read(); j=json.loads(s[s.index('{'):])
        print(f, round(j['ms_per_step'],2), round(j['ms_per_step_median'],2), round(j['step_mfu'],4), round(j['roofline']['frac'],4), j['loss'])
    except Exception as e: print(f,'FAILED',e)
PY
