#!/bin/bash
# paired wgrad launch + fp8 bit-identity fix: kernel tests, then A/B of the step with and without the pairing
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_kernels.py -m gpu -q --timeout 600 -p no:cacheprovider -k "gemm or fp8" 2>&1 | tail -12 > gpurun_out/pair_tests.log
timeout 900 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_fullwidth_oracle.py -m gpu -q --timeout 900 -p no:cacheprovider -k "golden or 1block or 2blocks or checkpoint or key_padding" 2>&1 | tail -8 >> gpurun_out/pair_tests.log
B="timeout 600 python bench.py --steps 16 --warmup 3 --no-cpu-baseline --table-steps 0"
for i in 1 2; do
  $B > gpurun_out/p_pair_$i.json 2> gpurun_out/p_pair_$i.err
  UDM_PAIR_WGRADS=0 $B > gpurun_out/p_nopair_$i.json 2> /dev/null
done
$B --workload unidisc-1.4b-interleaved-l4608 > gpurun_out/p_e_pair.json 2> /dev/null
UDM_PAIR_WGRADS=0 $B --workload unidisc-1.4b-interleaved-l4608 > gpurun_out/p_e_nopair.json 2> /dev/null
cat gpurun_out/pair_tests.log
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/p_*.json')):
    try:
        s=open(f).read(); j=json.loads(s[s.index('{'):])
        print(f, round(j['ms_per_step'],2), round(j['ms_per_step_median'],2), round(j['step_mfu'],4), round(j['roofline']['frac'],4))
    except Exception as e: print(f,'FAILED',e)
PY
tail -n 3 gpurun_out/p_pair_1.err
