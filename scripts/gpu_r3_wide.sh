#!/bin/bash
# wide (16-byte, whole-line) epilogue of the persistent 8-wave NT GEMM: kernel tests, epilogue micro-bench and the step, against a build without it (same box)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out/wide.log; : > $O
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -p no:cacheprovider -k "gemm" 2>&1 | tail -3 >> $O
echo "== wide" >> $O
python scripts/bench_gemm_epi.py >> $O 2>&1
for i in 1 2; do python bench.py --steps 20 --warmup 4 --no-cpu-baseline --table-steps 0 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('step wide', j['ms_per_step'], j['ms_per_step_median'])" >> $O; done
cp unidisc_amd/libunidisc_hip.so /tmp/lib_wide.so
(cd unidisc_amd/csrc && rm -f gemm.o && make CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result -Wno-inline-asm -DUDM_EPI_WIDE=0" > /tmp/make.log 2>&1; tail -2 /tmp/make.log >> $R/$O)
echo "== narrow" >> $O
python scripts/bench_gemm_epi.py >> $O 2>&1
for i in 1 2; do python bench.py --steps 20 --warmup 4 --no-cpu-baseline --table-steps 0 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('step narrow', j['ms_per_step'], j['ms_per_step_median'])" >> $O; done
cp /tmp/lib_wide.so unidisc_amd/libunidisc_hip.so
python bench.py --steps 20 --warmup 4 --no-cpu-baseline --table-steps 0 2>/dev/null | python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('step wide', j['ms_per_step'], j['ms_per_step_median'])" >> $O
cat $O
