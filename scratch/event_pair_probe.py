"""What an empty hipEvent pair reads on an MI355X stream (idle stream / behind a kernel)."""
import torch
s = torch.cuda.Stream()
x = torch.zeros(1 << 20, device="cuda")
res = {}
with torch.cuda.stream(s):
    for name, pre in (("idle", False), ("behind_kernel", True)):
        v = []
        for _ in range(200):
            if pre:
                x.add_(1.0)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(s); b.record(s)
            s.synchronize()
            v.append(a.elapsed_time(b) * 1e3)
        v.sort()
        res[name] = dict(median_us=round(v[100], 2), p10=round(v[20], 2), p90=round(v[180], 2))
    # a pair around a tiny kernel
    v = []
    for _ in range(200):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s); x.add_(1.0); b.record(s)
        s.synchronize()
        v.append(a.elapsed_time(b) * 1e3)
    v.sort()
    res["around_4MB_add"] = dict(median_us=round(v[100], 2), p10=round(v[20], 2), p90=round(v[180], 2))
print(res)
