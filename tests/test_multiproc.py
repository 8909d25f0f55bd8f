"""N > 1 path of bench.py on CPU: world_size-2 gloo group exercising the same
barrier / max-over-ranks / per-rank stream assignment the GPU run uses.  The
data path has no collective (front-end streams are independent), so this is
all the distributed logic there is."""
import os
import socket
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import bench
    assert bench.dist_env() == (rank, rank, world)
    dist = bench.init_group("gloo")
    bench.barrier(dist)
    # rank r pretends its K steps took (1 + r) seconds over L samples each
    elapsed = bench.max_over_ranks(dist, 1.0 + rank)
    total = bench.sum_over_ranks(dist, 10.0)
    from gpu_sdr_amd.source import tone_comb
    f, _, _ = tone_comb(8, 1000, bench.stream_seed(rank))
    q.put((rank, elapsed, total, bench.stream_seed(rank), [int(v) for v in f]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gloo_timing_and_stream_assignment():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [2.0, 2.0]      # MAX over ranks on every rank
    assert [r[2] for r in res] == [20.0, 20.0]
    assert res[0][3] != res[1][3]                 # one independent stream per rank
    assert res[0][4] != res[1][4]


def test_workload_table_matches_baseline_json():
    sys.path.insert(0, ROOT)
    import json
    import bench
    cfgs = json.load(open(os.path.join(ROOT, "BASELINE.json")))["configs"]
    assert "256-tone" in cfgs[1] and bench.WORKLOADS["c2"]["n_tones"] == 256 and bench.WORKLOADS["c2"]["decim"] == 100
    assert "2048-tone" in cfgs[2] and bench.WORKLOADS["c3"]["n_tones"] == 2048 and bench.WORKLOADS["c3"]["decim"] == 1000
    assert "Chirp" in cfgs[3] and bench.WORKLOADS["c4"]["swipe_s"] == 1_000_000
    b, f = bench.algorithmic(bench.WORKLOADS["c2"], 256)
    assert abs(b - 28.48) < 1e-9 and f == 5632
    b, f = bench.algorithmic(bench.WORKLOADS["c3"], 2048)
    assert abs(b - 24.384) < 1e-9 and f == 45056
