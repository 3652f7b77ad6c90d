#!/bin/bash
# modality-split head: e2e / full-width parity, then A/B of the step
mkdir -p gpurun_out; export TMPDIR=/tmp
UDM_LEDGER=gpurun_out/ledger_head.json timeout 2400 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_fullwidth_oracle.py tests/test_gpu_fullsize.py tests/test_interleaved.py tests/test_attn_dropout.py tests/test_optimizer.py -m gpu -q --timeout 1500 -p no:cacheprovider 2>&1 | tail -15 > gpurun_out/head_tests.log
B="timeout 600 python bench.py --steps 16 --warmup 3 --no-cpu-baseline --table-steps 0"
for i in 1 2; do
  $B > gpurun_out/hd_split_$i.json 2> gpurun_out/hd_split_$i.err
  UDM_SPLIT_HEAD=0 $B > gpurun_out/hd_nosplit_$i.json 2> /dev/null
done
$B --workload unidisc-s-l384 > gpurun_out/hd_s_split.json 2> /dev/null
UDM_SPLIT_HEAD=0 $B --workload unidisc-s-l384 > gpurun_out/hd_s_nosplit.json 2> /dev/null
$B --workload unidisc-1.4b-interleaved-l4608 > gpurun_out/hd_e_split.json 2> /dev/null
UDM_SPLIT_HEAD=0 $B --workload unidisc-1.4b-interleaved-l4608 > gpurun_out/hd_e_nosplit.json 2> /dev/null
cat gpurun_out/head_tests.log
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/hd_*.json')):
    try:
        s=open(f).read(); j=json.loads(s[s.index('{'):])
        print(f, round(j['ms_per_step'],2), round(j['ms_per_step_median'],2), round(j['step_mfu'],4), round(j['roofline']['frac'],4), j['loss'])
    except Exception as e: print(f,'FAILED',e)
PY
tail -n 3 gpurun_out/hd_split_1.err
