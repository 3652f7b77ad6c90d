"""What HBM sustains by read / write mix, as a yardstick for the row kernels' `roofline_table` fractions (algorithmic bytes over time against 8 TB/s):
a device-to-device copy (1 read : 1 write - the mix of residual_fwd and qknorm_rope), an fp32 -> bf16 cast (2 : 1), a read-only reduction, a write-only fill;
buffers rotate over 3 sets of 168 MB (not cache resident)."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unidisc_amd import kernels as K

n = 10240 * 4096
src = [torch.randn(n, device="cuda") for _ in range(3)]
dst = [torch.empty(n, device="cuda") for _ in range(3)]
dst16 = [torch.empty(n, device="cuda", dtype=torch.bfloat16) for _ in range(3)]


def timed(fn, reps=15):
    ts = []
    for it in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(it % 3); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e-3)
    ts.sort()
    return ts[len(ts) // 2]


res = {}
t = timed(lambda i: dst[i].copy_(src[i])); res["copy_fp32 (1 read : 1 write)"] = round(8 * n / t / 1e12, 2)
t = timed(lambda i: dst16[i].copy_(src[i])); res["cast_fp32_to_bf16 torch (2 : 1)"] = round(6 * n / t / 1e12, 2)
t = timed(lambda i: K.cast_f32_bf16(src[i], dst16[i]) if hasattr(K, "cast_f32_bf16") else None); res["cast_fp32_to_bf16 udm (2 : 1)"] = round(6 * n / t / 1e12, 2)
t = timed(lambda i: src[i].sum()); res["sum_fp32 (read only)"] = round(4 * n / t / 1e12, 2)
t = timed(lambda i: dst[i].fill_(1.0)); res["fill_fp32 (write only)"] = round(4 * n / t / 1e12, 2)
print(json.dumps(res))
