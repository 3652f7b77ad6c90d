#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 600 -p no:cacheprovider -k "subs_ce or sampler or ddpm" 2>&1 | tail -6 > gpurun_out/ce_tests.log
timeout 1500 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_fullwidth_oracle.py tests/test_interleaved.py -m gpu -q --timeout 900 -p no:cacheprovider -k "not 24blocks" 2>&1 | tail -6 >> gpurun_out/ce_tests.log
B="timeout 600 python bench.py --steps 16 --warmup 3 --no-cpu-baseline --table-steps 2"
$B > gpurun_out/ce_head.json 2> /dev/null
$B --workload unidisc-s-l384 > gpurun_out/ce_s.json 2> /dev/null
cat gpurun_out/ce_tests.log
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/ce_*.json')):
    try:
        s=open(f).read(); j=json.loads(s[s.index('{'):])
        print(f, round(j['ms_per_step'],2), round(j['ms_per_step_median'],2), round(j['step_mfu'],4))
        for r in j['roofline_table']:
            if 'subs_ce' in r['entry_point']: print('   ', r['entry_point'], round(r['ms_per_step'],3), r.get('achieved'), r.get('frac'))
    except Exception as e: print(f,'FAILED',e)
PY
