"""The copy-engine gradient exchange (unidisc_amd/ddp_copy.py, opt-in `UDM_DDP_MODE=copy_engine`) with two ranks on the GPU box's one GPU, in child processes
(scripts/ddp_copy_engine_check.py): peer buffers across processes through CUDA IPC, reduce-scatter + all-gather as copies on a copy stream, the helper
thread's host fences, the join at the end of the HIP backward.  Values: the reference BF16 hook's (main.py:641-656), identical on both ranks."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_copy_engine_exchange_two_ranks_on_one_gpu():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "ddp_copy_engine_check.py")], capture_output=True, text=True, env=env, timeout=900)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert lines, (out.returncode, out.stdout[-2000:], out.stderr[-2000:])
    res = json.loads(lines[-1])
    assert out.returncode == 0 and res["ok"], (res, out.stderr[-2000:])
    for side in ("rank0", "rank1"):
        for k, v in res[side].items():
            assert v["ranks_identical"] and v["worst_rel_vs_bf16_mean"] <= 2e-2 and v["bytes_pushed_to_peers"] > 0, (side, k, v)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "ddp_copy_engine_2ranks_1gpu_check.json"), "w") as f:
        json.dump(res, f, indent=1)
