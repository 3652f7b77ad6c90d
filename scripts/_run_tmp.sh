cd $GRAFT_REPO_ROOT
timeout 300 ./experiments/ubench/mfma_power.bin 2>&1 | tail -12
timeout 1200 python -m pytest tests/test_gpu_kernels.py -x -q -k "norm or resid or modul or adaln" 2>&1 | tail -4
timeout 1200 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_fullwidth_oracle.py -x -q -k "cond or adaln or time" 2>&1 | tail -4
timeout 900 python bench.py --workload unidisc-1.4b-l1280-adaln --steps 8 --warmup 3 --no-cpu-baseline > gpurun_out/adaln_fix.json 2>gpurun_out/adaln_fix.err; tail -2 gpurun_out/adaln_fix.err
python3 - <<'PY'
import json
s=open('gpurun_out/adaln_fix.json').read(); j=json.loads(s[s.index('{'):])
print(j['ms_per_step'])
for r in sorted(j.get('roofline_table',[]), key=lambda r:-r['ms_per_step'])[:14]:
    print(r['entry_point'], r['launches_per_step'], round(r['ms_per_step'],2), r.get('frac'))
PY
