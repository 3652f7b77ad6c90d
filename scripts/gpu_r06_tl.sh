#!/bin/bash
# timeline-only diagnostics of the generated backward programs (timeline build on the box; the tree ends on the shipped build)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
trap "make -C unidisc_amd/csrc regen all > /dev/null 2>&1" EXIT
make -C unidisc_amd/csrc regen all UDM_DKV64_ABL="16" UDM_DQ64_ABL="16" > gpurun_out/r06_tl_build.log 2>&1
timeout 300 python scripts/attn_dkv64_timeline.py 2>/dev/null | grep -v amdgpu.ids > gpurun_out/r06_attn_dkv64_timeline.log
UDM_TL=dq64 timeout 300 python scripts/attn_dkv64_timeline.py 2>/dev/null | grep -v amdgpu.ids > gpurun_out/r06_attn_dq64_timeline.log
head -2 gpurun_out/r06_attn_dkv64_timeline.log | cut -c1-700; head -2 gpurun_out/r06_attn_dq64_timeline.log | cut -c1-700
