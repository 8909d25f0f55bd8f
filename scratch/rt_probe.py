"""Row tiles per workgroup (GSDR_MFMA_RT) against tone count at decim 100 (13-block windows)."""
import os, sys
sys.path.insert(0, ".")
import torch
import bench
dev = torch.device("cuda:0")
wl = dict(bench.WORKLOADS["c2"])
for n in (128, 256, 512, 1024, 2048):
    line = "tones %5d:" % n
    for api in ("inorder", "pipelined"):
        for rt in ("1", "2"):
            os.environ["GSDR_MFMA_RT"] = rt
            r = bench.time_workload(wl, dev, 1, steps=600, warmup=30, n_tones=n, profile=False, api=api)
            line += "  %s rt=%s %7.2f us" % (api, rt, r["elapsed"] / 600 * 1e6)
    print(line, flush=True)
