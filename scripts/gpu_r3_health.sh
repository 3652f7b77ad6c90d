#!/bin/bash
# round 3: tree health after the stream-K revert - the whole GPU suite, then benches (headline; 16 CUs held by a spinning kernel; config E bf16 / fp8; UniDisc-S)
mkdir -p gpurun_out; export TMPDIR=/tmp
UDM_LEDGER=gpurun_out/ledger_full.json timeout 2700 python -m pytest tests -m gpu -q --timeout 1500 -p no:cacheprovider -x 2>&1 | tail -25 > gpurun_out/gpu_suite.log
B="timeout 600 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --table-steps 0"
$B > gpurun_out/h_head.json 2> gpurun_out/h_head.err
$B --hog-cus 16 > gpurun_out/h_hog16_uncapped.json 2> /dev/null
UDM_GEMM_CUS=240 $B --hog-cus 16 > gpurun_out/h_hog16_cap240.json 2> /dev/null
$B --workload unidisc-1.4b-interleaved-l4608 > gpurun_out/h_e_bf16.json 2> /dev/null
$B --workload unidisc-1.4b-interleaved-l4608 --fp8-attention > gpurun_out/h_e_fp8.json 2> gpurun_out/h_e_fp8.err
$B --workload unidisc-s-l384 > gpurun_out/h_s.json 2> /dev/null
cat gpurun_out/gpu_suite.log
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/h_*.json')):
    try:
        s=open(f).read(); j=json.loads(s[s.index('{'):])
        print(f, round(j['ms_per_step'],2), round(j['ms_per_step_median'],2), round(j['step_mfu'],4), round(j['roofline']['frac'],4))
    except Exception as e: print(f,'FAILED',e)
PY
tail -n 3 gpurun_out/h_head.err gpurun_out/h_e_fp8.err
