"""How much would overlapping consecutive C2 buffers buy?  Two independent demodulators on
two streams, fed alternately, against one demodulator on one stream (same total work)."""
import sys, time
sys.path.insert(0, ".")
import torch
import bench

dev = torch.device("cuda:0")
wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "c2"]
K = 400
a = bench.build_workload(wl, dev, 1)
b = bench.build_workload(wl, dev, 2)
outs = [torch.empty_like(a[2]) for _ in range(2)]
sa, sb = torch.cuda.Stream(dev), torch.cuda.Stream(dev)


def run(two):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(K):
        if two and (k & 1):
            b[0].process_device(b[1][k % 8], outs[1], sb)
        else:
            a[0].process_device(a[1][k % 8], outs[0], sa)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e6


for rep in range(3):
    print("one stream  %.2f us/buffer    two streams %.2f us/buffer" % (run(False), run(True)), flush=True)
