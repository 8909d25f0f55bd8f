"""In-order (process_device on one stream) against pipelined (submit_device / wait) throughput."""
import sys
sys.path.insert(0, ".")
import torch
import bench

dev = torch.device("cuda:0")
for key in sys.argv[1:] or ["c2", "c3", "pfb", "c4"]:
    wl = bench.WORKLOADS[key]
    for rep in range(2):
        a = bench.time_workload(wl, dev, 1, steps=400, warmup=20, profile=False)
        line = "%-4s in-order %7.2f us/buffer" % (key, a["elapsed"] / 400 * 1e6)
        for depth in (2, 3, 4):
            b = bench.time_pipelined(wl, dev, 1, steps=400, warmup=20, depth=depth)
            line += "   depth %d: %7.2f" % (depth, b["elapsed"] / 400 * 1e6)
        print(line, flush=True)
