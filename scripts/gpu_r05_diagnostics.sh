#!/bin/bash
# Round-5 diagnostics behind RESULTS.md "Round 5", in one GPU call: the MFMA power micro-benchmark, the GEMM asm K loop's cycle stamps and its A/B against the C++ loop,
# the attention forward's timeline and A/B, the attention backward and row-kernel micro-benchmarks.  Diagnostic builds are made on the box and rebuilt to the shipped
# form at the end.  Outputs: gpurun_out/r05_*.log|json (copied to profiles/ by hand).
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 experiments/ubench/mfma_power.hip -o /tmp/mfma_power && timeout 300 /tmp/mfma_power > gpurun_out/r05_mfma_power.log 2>&1
timeout 600 python scripts/bench_gemm_asm_loop.py 2>/dev/null | tail -1 > gpurun_out/r05_gemm_asm_loop_ab.json
timeout 600 python scripts/bench_attn_fwd64.py 2>/dev/null | tail -1 > gpurun_out/r05_attn_fwd64_ab.json
timeout 600 python scripts/bench_attn_bwd.py 2>/dev/null | tail -1 > gpurun_out/r05_attn_bwd.json
timeout 600 python scripts/bench_residual_dropout.py 2>/dev/null | tail -1 > gpurun_out/r05_residual_dropout.json
timeout 600 python scripts/bench_qknorm_rope.py 2>/dev/null | tail -1 > gpurun_out/r05_qknorm_rope.json
trap "make -C unidisc_amd/csrc regen all > /dev/null 2>&1" EXIT   # whatever happens below, the tree ends on the shipped build
make -C unidisc_amd/csrc regen all UDM_QUADLOOP=timeline > /dev/null 2>&1
timeout 600 python scripts/gemm_asm_loop_timeline.py 2>/dev/null | grep "^{" > gpurun_out/r05_gemm_asm_loop_timeline.json
make -C unidisc_amd/csrc regen all UDM_FWD64_ABL=16 > /dev/null 2>&1
timeout 600 python scripts/attn_fwd64_timeline.py 2>/dev/null | grep -v amdgpu.ids > gpurun_out/r05_attn_fwd64_timeline.log
make -C unidisc_amd/csrc regen all > /dev/null 2>&1
head -20 gpurun_out/r05_mfma_power.log; cat gpurun_out/r05_gemm_asm_loop_timeline.json; cut -c1-600 gpurun_out/r05_attn_fwd64_ab.json; cat gpurun_out/r05_attn_bwd.json
