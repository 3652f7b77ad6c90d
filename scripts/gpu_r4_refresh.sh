TAG=r04; R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
UDM_DUMP_GRAD_ERRS=gpurun_out/graderrs_$TAG UDM_LEDGER=gpurun_out/parity_ledger_$TAG.json timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 1500 2>&1 | tail -15 > gpurun_out/gputests_$TAG.log
timeout 600 python bench.py --workload unidisc-1.4b-interleaved-l4608 --steps 12 --warmup 3 --no-cpu-baseline > gpurun_out/bench_1.4b_interleaved_l4608_b2_$TAG.json 2>/dev/null
timeout 600 python bench.py --workload unidisc-1.4b-interleaved-l4608 --fp8-attention --steps 12 --warmup 3 --no-cpu-baseline > gpurun_out/bench_1.4b_interleaved_l4608_b2_fp8_$TAG.json 2>/dev/null
timeout 600 python bench.py --workload unidisc-s-l384 --steps 12 --warmup 3 --no-cpu-baseline > gpurun_out/bench_unidisc_s_b64_$TAG.json 2>/dev/null
EXTRA="--workload unidisc-1.4b-interleaved-l4608" bash scripts/gpu_prof.sh ${TAG}e > gpurun_out/prof_summary_${TAG}e.log 2>&1
EXTRA="--workload unidisc-1.4b-interleaved-l4608 --fp8-attention" bash scripts/gpu_prof.sh ${TAG}e8 > gpurun_out/prof_summary_${TAG}e8.log 2>&1
EXTRA="--workload unidisc-s-l384" bash scripts/gpu_prof.sh ${TAG}s > gpurun_out/prof_summary_${TAG}s.log 2>&1
rm -rf gpurun_out/prof_${TAG}e gpurun_out/prof_${TAG}e8 gpurun_out/prof_${TAG}s
tail -4 gpurun_out/gputests_$TAG.log
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/bench_*_r04.json')):
    try:
        s=open(f).read(); j=json.loads(s[s.index('{'):])
        print(f, round(j['ms_per_step'],2), round(j['ms_per_step_median'],2), round(j['step_mfu'],4), round(j['roofline']['frac'],4))
    except Exception as e: print(f,'FAILED',e)
PY
