"""65 536 tones, decim 1000: Msamples/s of the pipelined entry against stream count and queue kind."""
import os, sys
sys.path.insert(0, ".")
import torch
import bench
dev = torch.device("cuda:0")
wl = dict(bench.WORKLOADS["c3"])
for q in ("1", "0"):
    for st in ("1", "2", "3"):
        os.environ["GSDR_PIPE_QUEUES"] = q
        os.environ["GSDR_PIPE_STREAMS"] = st
        r = bench.time_workload(wl, dev, 1, steps=120, warmup=5, n_tones=65536, profile=False, api="pipelined")
        print("queues=%s streams=%s  %.1f Msamples/s" % (q, st, 120 * bench.L / r["elapsed"] / 1e6), flush=True)
r = bench.time_workload(wl, dev, 1, steps=120, warmup=5, n_tones=65536, profile=False, api="inorder")
print("in-order  %.1f Msamples/s" % (120 * bench.L / r["elapsed"] / 1e6), flush=True)
