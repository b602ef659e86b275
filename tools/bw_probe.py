"""Streaming rates of the device by direction (tools): torch fill_ (write only), copy_ (read + write), sum (read only)."""
import torch, json
dev = "cuda"
n = 256 << 20
x = torch.empty(n // 8, dtype=torch.float64, device=dev); y = torch.empty_like(x)
def t(fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3
out = {}
out["fill_GBs"] = n / t(lambda: x.fill_(1.5)) / 1e9
out["copy_GBs_read_plus_write"] = 2 * n / t(lambda: y.copy_(x)) / 1e9
out["sum_GBs"] = n / t(lambda: x.sum()) / 1e9
for mb in (32, 64):
    m = mb << 20
    xs = x[: m // 8]; ys = y[: m // 8]
    out[f"fill_{mb}MB_us"] = t(lambda: xs.fill_(1.5), 200) * 1e6
    out[f"copy_{mb}MB_us"] = t(lambda: ys.copy_(xs), 200) * 1e6
print(json.dumps(out))
