#!/bin/bash
# Round-6 diagnostics of the generated dK / dV program: ablation builds and the cycle timeline, made on the box; the tree ends on the shipped build whatever happens.
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
trap "make -C unidisc_amd/csrc regen all > /dev/null 2>&1" EXIT
make -C unidisc_amd/csrc regen all UDM_DKV64_ABL="1 2 4 8 16" UDM_DQ64_ABL="1 2 4 8 16" > gpurun_out/r06_dkv64_diag_build.log 2>&1
for a in 0 1 2 4 8; do UDM_ATTN_DKV64_ABL=$a timeout 200 python scripts/bench_attn_dkv64_abl.py 2>/dev/null | tail -1; done > gpurun_out/r06_attn_dkv64_ablations.log
UDM_ATTN_DKV64=0 timeout 200 python scripts/bench_attn_dkv64_abl.py 2>/dev/null | tail -1 | sed 's/"abl": 0/"abl": "8-wave kernel"/' >> gpurun_out/r06_attn_dkv64_ablations.log
timeout 300 python scripts/attn_dkv64_timeline.py 2>/dev/null | grep -v amdgpu.ids > gpurun_out/r06_attn_dkv64_timeline.log
for a in 0 1 2 4 8; do UDM_ATTN_DQ64_ABL=$a timeout 200 python scripts/bench_attn_dkv64_abl.py 2>/dev/null | tail -1 | sed "s/\"abl\"/\"dq64_abl\"/"; done > gpurun_out/r06_attn_dq64_ablations.log
UDM_TL=dq64 timeout 300 python scripts/attn_dkv64_timeline.py 2>/dev/null | grep -v amdgpu.ids > gpurun_out/r06_attn_dq64_timeline.log
cat gpurun_out/r06_attn_dq64_ablations.log; head -8 gpurun_out/r06_attn_dq64_timeline.log | cut -c1-420
cat gpurun_out/r06_attn_dkv64_ablations.log; head -16 gpurun_out/r06_attn_dkv64_timeline.log | cut -c1-400
