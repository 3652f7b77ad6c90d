"""The data-parallel gradient path under REAL RCCL on the GPU box (pytest -m gpu): backend "nccl", world_size 1, in a child process
(scripts/ddp_rccl_check.py).  World 1 cannot measure scaling; it proves that process-group init, the comm-stream cast kernels, RCCL's all_reduce,
the stream hand-offs, accumulate-then-sync and the `UDM_GEMM_CUS` reservation run and give bf16(local gradients) exactly as the reference's BF16
compress hook would (main.py:641-656).  The world-2 semantics are covered on CPU by tests/test_ddp_gloo.py."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rccl_world1_training_step_through_ddp_wrap():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29531", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "ddp_rccl_check.py")], capture_output=True, text=True, env=env, timeout=600)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert lines, (out.returncode, out.stdout[-2000:], out.stderr[-2000:])
    res = json.loads(lines[-1])
    assert out.returncode == 0 and res["ok"], (res, out.stderr[-2000:])
    for k, v in res.items():
        if k.startswith("bucket"):
            assert v["bf16_representable"] and v["bytes_on_wire"] > 0
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "ddp_rccl_world1_check.json"), "w") as f:
        json.dump(res, f, indent=1)
