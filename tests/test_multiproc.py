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


# ---------------------------------------------------------------------------
# bench.main() end to end on CPU: spawn -> time_workload (barrier, repeats agreed over the
# ranks, max-over-ranks) -> ONE line from rank 0 whose value is the aggregate of all ranks.
# The demodulator is a stub that sleeps (no GPU here); everything else is bench.py's own code.
# ---------------------------------------------------------------------------
class _StubDemod:
    kernel_name = "stub_kernel"

    def __init__(self, step_s):
        self.step_s, self.pending, self.calls = step_s, 0, 0

    def process_device(self, buf, out, stream):
        import time
        time.sleep(self.step_s)
        self.calls += 1
        return 1

    def submit_device(self, buf, out):
        import time
        time.sleep(self.step_s)
        self.calls += 1
        self.pending += 1

    def wait(self):
        assert self.pending > 0
        self.pending -= 1
        return 1

    def profile_enable(self, every):
        self.every = every

    def profile_read(self):
        return self.calls // max(self.every, 1), 1e3 * self.step_s * (self.calls // max(self.every, 1))

    def describe(self):
        return {"kernel": self.kernel_name, "env": {}}

    def close(self):
        assert self.pending == 0


class _StubEngine:
    """rank r's steps take (1 + r) ms: the slower rank sets the time of the job"""

    def __init__(self, rank):
        self.rank = rank

    def sync(self, device):
        pass

    def stream(self, device):
        return None

    def build(self, wl, device, seed, ring=8, n_tones=None):
        import bench
        assert seed == bench.stream_seed(self.rank)
        return _StubDemod(1e-3 * (1 + self.rank)), [None] * ring, [None] * bench.PIPE_DEPTH, n_tones or wl.get("n_tones", 1)


def _bench_worker(rank, world, port, q):
    import contextlib
    import io
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), GSDR_BENCH_BACKEND="gloo")
    import bench
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        line = bench.main(["--gpus", str(world), "--steps", "20", "--warmup", "2", "--min-seconds", "0.2",
                           "--workload", "c3"], engine=_StubEngine(rank))
    q.put((rank, buf.getvalue(), line))


def test_bench_main_two_ranks_stub_engine():
    import json
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=180) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    out0, out1 = res[0][1].strip(), res[1][1].strip()
    assert out1 == ""                                   # only rank 0 prints
    assert len(out0.splitlines()) == 1                  # ... ONE line
    line = json.loads(out0)
    assert line["n_gpus"] == 2 and line["steps"] == 20 and line["warmup"] == 2
    assert line["config"]["key"] == "c3" and line["config"]["streams"] == 2
    assert line["config"]["control_plane"] == "gloo"
    assert line["scaling"] == "weak" and line["vs_baseline"] is None
    # both ranks agreed on the same repeat count (MAX of the per-rank estimates -> from the slow rank)
    assert res[0][2]["repeats"] == res[1][2]["repeats"] >= 1
    R = line["repeats"]
    assert line["timed_region_s"] >= 0.2 * 0.8
    # the slow rank (2 ms per step) sets the elapsed time; value = samples of BOTH ranks / that
    ms = line["ms_per_step"]
    assert 1.9 <= ms <= 4.0, ms
    want = 2 * 20 * R * 1_000_000 / (ms * 1e-3 * 20 * R) / 1e6
    assert abs(line["value"] - want) / want < 1e-3
    assert line["roofline"] is not None and line["roofline"]["kernel"] == "stub_kernel"
    # every rank's own clock is in the line: the fast rank (1 ms per step) is visibly twice as fast as the slow one
    pr = line["per_rank"]
    assert [p["rank"] for p in pr] == [0, 1]
    assert 0.9 <= pr[0]["ms_per_step"] <= 1.9 and 1.9 <= pr[1]["ms_per_step"] <= 4.0, pr
    assert pr[1]["ms_per_step"] <= ms * 1.001


def test_bench_refuses_timing_switches():
    import subprocess
    env = dict(os.environ, GSDR_MFMA_TIMING="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu", "--no-extras"],
                       env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode == 2 and "refusing" in p.stderr
    env = dict(os.environ, GSDR_LIB="/tmp/some_ablation_build.so")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu", "--no-extras"],
                       env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode == 2 and "refusing" in p.stderr


def test_backend_choice():
    sys.path.insert(0, ROOT)
    import bench
    old = os.environ.pop("GSDR_BENCH_BACKEND", None)
    try:
        assert bench.pick_backend(8, 8) == "nccl"      # one GPU per rank: RCCL
        assert bench.pick_backend(2, 1) == "gloo"      # ranks share a device: RCCL would refuse ("Duplicate GPU")
        assert bench.pick_backend(1, 1) == "nccl"
    finally:
        if old is not None:
            os.environ["GSDR_BENCH_BACKEND"] = old


def test_rooflines_of_the_hbm_bound_paths():
    """bench.rooflines(): TONES through the frame-per-workgroup kernel is priced against HBM (algorithmic
    bytes: one read of the samples + the selected bins), not against a matrix or vector pipe; a chirp
    launch shorter than an event pair is priced with the step period of the event-free region."""
    import bench
    wl = bench.WORKLOADS["pfb"]
    r = dict(kernel="pfb_lds_kernel", kernel_ms=20.0e-3 * 100, kernel_launches=100, n_tones=1024, api="inorder")
    roof, roof_hbm = bench.rooflines(wl, r, "pfb")
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and roof["peak"] == bench.HBM_PEAK_GBS
    ab = 8.0 * (1.0 + 1024 / 1230) * bench.L
    assert abs(roof["algorithmic_bytes_per_launch"] - ab) < 1.0
    assert abs(roof["achieved"] - ab / 20.0e-6 / 1e9) < 0.5 and abs(roof["frac"] - roof["achieved"] / 8000.0) < 1e-4
    # the same workload on the DDC kernels stays an MFMA roofline with both flop counts
    r2 = dict(r, kernel="ddc_mfma_ring16p_kernel")
    roof2, _ = bench.rooflines(wl, r2, "pfb")
    assert roof2["bound"] == "mfma" and roof2["executed_mfma_flops_per_launch"] == round(24.0 * 4 * 1024 * bench.L)
    assert roof2["algorithmic_flops_per_launch"] == round(1024 * (6 + 4 * 4) * bench.L)
    # chirp: events say 7 us, back-to-back launches 4 us per step
    c4 = bench.WORKLOADS["c4"]
    rc = dict(kernel="chirp_lockin_kernel", kernel_ms=7.0e-3 * 50, kernel_launches=50, n_tones=1, api="inorder")
    roof3, _ = bench.rooflines(c4, rc, "c4", period_s=4.0e-6)
    assert roof3["kernel_us"] == 4.0 and roof3["kernel_us_between_events"] == 7.0 and "step period" in roof3["measured_with"]
    roof4, _ = bench.rooflines(c4, rc, "c4", period_s=9.0e-6)      # a period longer than the events say: keep the events
    assert roof4["kernel_us"] == 7.0 and "kernel_us_between_events" not in roof4
