#!/bin/bash
# round 3, call 4: stream-K GEMM tests + the quad kernels' bit-identity tests (regression), fp8 two-group forward, config E oracle test, benches at 256 / 248 / 240 / 224 CUs, config E and UniDisc-S
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 600 -p no:cacheprovider -k "gemm or fp8" 2>&1 | tail -25 > gpurun_out/sk_gemm_tests.log
timeout 600 python scripts/bench_attn_fp8.py > gpurun_out/bench_attn_fp8_stag.log 2>&1
UDM_ATTN_FP8_STAG=0 timeout 600 python scripts/bench_attn_fp8.py > gpurun_out/bench_attn_fp8_nostag.log 2>&1
UDM_LEDGER=gpurun_out/ledger_e.json timeout 1500 python -m pytest tests/test_gpu_fullwidth_oracle.py -m gpu -q --timeout 1200 -p no:cacheprovider -k "config_e" 2>&1 | tail -15 > gpurun_out/config_e_oracle.log
for cus in 0 248 240 224; do
  UDM_GEMM_CUS=$cus timeout 600 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --table-steps 0 > gpurun_out/bench_cus_$cus.json 2> gpurun_out/bench_cus_$cus.err
done
UDM_GEMM_STREAMK=0 timeout 600 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --table-steps 0 > gpurun_out/bench_nosk.json 2> /dev/null
UDM_GEMM_STREAMK=0 UDM_GEMM_CUS=240 timeout 600 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --table-steps 0 > gpurun_out/bench_nosk_240.json 2> /dev/null
for w in unidisc-1.4b-interleaved-l4608 unidisc-s-l384; do
  timeout 600 python bench.py --workload $w --steps 12 --warmup 3 --no-cpu-baseline --table-steps 2 > gpurun_out/bench_sk_$w.json 2> /dev/null
  UDM_GEMM_STREAMK=0 timeout 600 python bench.py --workload $w --steps 12 --warmup 3 --no-cpu-baseline --table-steps 0 > gpurun_out/bench_nosk_$w.json 2> /dev/null
done
timeout 600 python bench.py --workload unidisc-1.4b-interleaved-l4608 --fp8-attention --steps 12 --warmup 3 --no-cpu-baseline --table-steps 2 > gpurun_out/bench_fp8_e.json 2> gpurun_out/bench_fp8_e.err
cat gpurun_out/sk_gemm_tests.log; cat gpurun_out/bench_attn_fp8_stag.log gpurun_out/bench_attn_fp8_nostag.log; cat gpurun_out/config_e_oracle.log
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/bench_cus_*.json')+glob.glob('gpurun_out/bench_nosk*.json')+glob.glob('gpurun_out/bench_sk_*.json')+['gpurun_out/bench_fp8_e.json']):
    try:
        s=open(f).read(); j=json.loads(s[s.index('{'):])
        print(f, round(j['ms_per_step'],2), round(j['ms_per_step_median'],2), round(j['step_mfu'],4), round(j['roofline']['frac'],4))
    except Exception as e: print(f,'FAILED',e)
PY
tail -3 gpurun_out/bench_cus_240.err gpurun_out/bench_fp8_e.err
