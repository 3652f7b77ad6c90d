#!/bin/bash
# round 3, call 5: remainder-split GEMM tests, fp8 kernel tests + micro-bench, benches: headline; 16 CUs held by a spinning kernel with / without the split; config E, UniDisc-S with / without; config E fp8
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 600 -p no:cacheprovider -k "gemm or fp8" 2>&1 | tail -15 > gpurun_out/split_gemm_tests.log
timeout 600 python scripts/bench_attn_fp8.py > gpurun_out/bench_attn_fp8_lazy.log 2>&1
UDM_LEDGER=gpurun_out/ledger_e.json timeout 1500 python -m pytest tests/test_gpu_fullwidth_oracle.py -m gpu -q --timeout 1200 -p no:cacheprovider -k "config_e" 2>&1 | tail -8 > gpurun_out/config_e_oracle.log
B="timeout 600 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --table-steps 0"
$B > gpurun_out/b_head.json 2> gpurun_out/b_head.err
UDM_GEMM_STREAMK=0 $B > gpurun_out/b_head_nosplit.json 2> /dev/null
UDM_GEMM_CUS=240 $B --hog-cus 16 > gpurun_out/b_hog16_split.json 2> gpurun_out/b_hog16_split.err
UDM_GEMM_STREAMK=0 UDM_GEMM_CUS=240 $B --hog-cus 16 > gpurun_out/b_hog16_nosplit.json 2> /dev/null
UDM_GEMM_STREAMK=0 $B --hog-cus 16 > gpurun_out/b_hog16_uncapped.json 2> /dev/null
UDM_GEMM_CUS=248 $B --hog-cus 8 > gpurun_out/b_hog8_split.json 2> /dev/null
UDM_GEMM_CUS=224 $B --hog-cus 32 > gpurun_out/b_hog32_split.json 2> /dev/null
for w in unidisc-1.4b-interleaved-l4608 unidisc-s-l384; do
  $B --workload $w > gpurun_out/b_split_$w.json 2> gpurun_out/b_split_$w.err
  UDM_GEMM_STREAMK=0 $B --workload $w > gpurun_out/b_nosplit_$w.json 2> /dev/null
done
$B --workload unidisc-1.4b-interleaved-l4608 --fp8-attention > gpurun_out/b_fp8_e.json 2> gpurun_out/b_fp8_e.err
cat gpurun_out/split_gemm_tests.log; cat gpurun_out/bench_attn_fp8_lazy.log; cat gpurun_out/config_e_oracle.log
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/b_*.json')):
    try:
        s=open(f).read(); j=json.loads(s[s.index('{'):])
        print(f, round(j['ms_per_step'],2), round(j['ms_per_step_median'],2), round(j['step_mfu'],4), round(j['roofline']['frac'],4))
    except Exception as e: print(f,'FAILED',e)
PY
tail -n 3 gpurun_out/b_head.err gpurun_out/b_hog16_split.err gpurun_out/b_fp8_e.err
