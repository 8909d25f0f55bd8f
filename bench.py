#!/usr/bin/env python3
"""bench.py -- throughput of the RX demodulation hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload c3|c2|c4|pfb|c1]

A "step" is one pass of the hot path through the C ABI over one 1 M-sample buffer of
synthetic IQ that is already resident in HBM (a ring of distinct buffers made by the HIP
source kernel before the timed region).  The headline workload is BASELINE.json's largest
single-GPU configuration, C3 (2048-tone DDC, decim 1000, 200 Msps); C2, the PFB, the chirp
path and the real-time tone count are reported under `extras`.

Timed region: W warm-up steps, then R repetitions of the K steps, R chosen so that the
region lasts at least --min-seconds (default 1 s: a few hundred microseconds of kernels
say nothing about the power-capped steady state the chip settles into), bracketed by a
barrier + device synchronise on both sides; `ms_per_step` = elapsed / (K*R) (max over
ranks), `repeats` = R.

One independent synthetic 200 Msps stream per GPU, no inter-GPU data traffic (front-end
streams are independent: SURVEY.md section 8e), so scaling is "weak" and `value` is the
aggregate Msamples/s of all ranks.  For N > 1 the driver launches this file under
torch.distributed.run; the process group is used ONLY for the barrier and the
max-over-ranks of the elapsed time (RCCL when every rank has a GPU of its own, gloo when
ranks share a device -- the one-GPU rehearsal).

Rank 0 prints ONE JSON line (contract in the task statement) with
  roofline      dominant kernel: ALGORITHMIC flops per launch, N(6+4f)L,
                over its average launch duration with the chip to itself (in-order entry,
                hipEvents on the launch stream) against the dense peak of the pipe it runs
                on; the flops the matrix cores actually execute (24fNL, hi/lo split) are
                reported separately as executed_mfma_*; `roofline_hbm` is the HBM view,
  cpu_baseline  the reference's own offline CPU recipe (scripts/raw_data_analisys.py:55-68)
                on one buffer of the same workload over the box's CPU share, N = 1 only;
                the OpenMP C oracle and the numpy restatement are reported beside it.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

RATE = 200_000_000
L = 1_000_000
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8 TB/s
FP32_PEAK_TFLOPS = 157.3        # MI355X_MICROARCH.md: FP32 vector == FP32 matrix peak
F16_MFMA_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense f16/bf16 MFMA
PROFILE_EVERY = int(os.environ.get("GSDR_BENCH_PROFILE_EVERY", "8"))   # timed region: every n-th launch
PIPE_DEPTH = int(os.environ.get("GSDR_BENCH_DEPTH", "3"))   # buffers outstanding, <= GSDR_PIPELINE_DEPTH (4)
CPU_PROCS_CAP = int(os.environ.get("GSDR_BENCH_CPU_PROCS", "16"))      # one GPU's share of the box's cores

WORKLOADS = {
    # BASELINE.json configs[1]
    "c2": dict(kind="direct", n_tones=256, decim=100, pf_average=4,
               name="256-tone DDC + polyphase FIR, decim=100, 200 Msps synthetic stream"),
    # configs[2] / configs[4]: the largest single-GPU configuration -- the headline
    "c3": dict(kind="direct", n_tones=2048, decim=1000, pf_average=4,
               name="2048-tone DDC readout, decim=1000, 200 Msps"),
    # configs[3]
    "c4": dict(kind="chirp", decim=1, chirp_t=1.0, swipe_s=1_000_000,
               name="Chirp VNA demod (USRP_VNA), 1e6-point sweep over 200 MHz, lock-in ppt=200"),
    # TONES mode (PFB channelizer + tone select) at array scale: not a BASELINE config
    # line of its own, but the same DDC kernel with M = nfft, F = pf_average
    "pfb": dict(kind="pfb", n_tones=1024, fft_tones=1230, pf_average=4, decim=0,
                name="1024-tone PFB channelizer (TONES), nfft=1230, pf_average=4, 200 Msps"),
    # configs[0] shape on the GPU (the CPU-runnable plumbing case)
    "c1": dict(kind="direct", n_tones=16, decim=100, pf_average=4, rate=100_000_000,
               name="16-tone DDC, decim=100, 100 Msps"),
}


# --------------------------------------------------------------------------
# multi-process plumbing (covered by tests/test_multiproc.py with gloo)
# --------------------------------------------------------------------------
def dist_env():
    """(rank, local_rank, world_size) from the torchrun environment."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def stream_seed(rank: int) -> int:
    """Seed of the synthetic stream owned by `rank`."""
    return 20251004 + rank


def init_group(backend: str):
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    if not dist.is_initialized():
        # gloo announces its connections on stdout; stdout carries ONE line, the result: park fd 1 on fd 2 meanwhile
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group(backend=backend)
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)
    return dist


def _ctl(device):
    return device if (device is not None and getattr(device, "type", "cpu") == "cuda") else None


def barrier(dist, device=None):
    if dist is None:
        return
    import torch
    dev = _ctl(device)
    t = torch.zeros(1, device=dev) if dev is not None else torch.zeros(1)
    dist.all_reduce(t)
    if dev is not None:
        torch.cuda.synchronize(dev)


def max_over_ranks(dist, seconds: float, device=None) -> float:
    if dist is None:
        return seconds
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=_ctl(device) or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(dist, v: float, device=None) -> float:
    if dist is None:
        return v
    import torch
    t = torch.tensor([v], dtype=torch.float64, device=_ctl(device) or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_over_ranks(dist, v: float, device=None):
    """[v of rank 0, v of rank 1, ...] on every rank."""
    if dist is None:
        return [v]
    import torch
    t = torch.tensor([v], dtype=torch.float64, device=_ctl(device) or "cpu")
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [float(o.item()) for o in out]


def pick_backend(world: int, n_devices: int) -> str:
    """RCCL needs one device per rank ("Duplicate GPU detected" otherwise: that, and nothing
    else, is what stopped round 1's two-rank attempt on a one-GPU box); ranks that share a
    device talk over gloo.  GSDR_BENCH_BACKEND overrides."""
    forced = os.environ.get("GSDR_BENCH_BACKEND")
    if forced:
        return forced
    return "nccl" if n_devices >= world else "gloo"


# --------------------------------------------------------------------------
# workload construction
# --------------------------------------------------------------------------
def algorithmic(wl, n_tones):
    """(bytes, flops) per input sample, SURVEY.md section 8d."""
    if wl["kind"] == "direct":
        M, f = wl["decim"], wl["pf_average"]
        return 8.0 * (1.0 + n_tones / M), float(n_tones * (6 + 4 * f))
    if wl["kind"] == "pfb":
        M, f = wl["fft_tones"], wl["pf_average"]
        return 8.0 * (1.0 + n_tones / M), float(n_tones * (6 + 4 * f))
    ppt = 200 * wl["decim"]
    return 8.0 + 8.0 / ppt, 8.0 + 30.0


class HipEngine:
    """The product path: libgsdr.so through the Python mirror of the reference's class."""
    name = "hip"

    def check(self, allow_ablation=False):
        from gpu_sdr_amd import _lib
        info = _lib.lib().gsdr_build_info().decode()
        if "timing_build 0" not in info and not allow_ablation:
            raise SystemExit(f"bench.py: {_lib.LIB_PATH} is a timing-only ablation build ({info}); refusing to benchmark it")
        return info

    def sync(self, device):
        import torch
        torch.cuda.synchronize(device)

    def stream(self, device):
        import torch
        return torch.cuda.Stream(device)   # the hot path runs on its own (non-null) stream

    def build(self, wl, device, seed, ring=8, n_tones=None):
        import torch
        import gpu_sdr_amd as g
        from gpu_sdr_amd.source import device_chirp, device_tones, tone_comb
        rate = wl.get("rate", RATE)
        bufs = [torch.empty(L, dtype=torch.complex64, device=device) for _ in range(ring)]
        if wl["kind"] in ("direct", "pfb"):
            N = n_tones or wl["n_tones"]
            freq, ampl, phase = tone_comb(N, rate, seed)
            if wl["kind"] == "direct":
                p = g.param(mode="RX", rate=rate, buffer_len=L, decim=wl["decim"], pf_average=wl["pf_average"],
                            freq=[int(f) for f in freq], wave_type=[g.w_type.DIRECT] * N)
            else:
                p = g.param(mode="RX", rate=rate, buffer_len=L, decim=0, pf_average=wl["pf_average"],
                            fft_tones=wl["fft_tones"], freq=[int(f) for f in freq],
                            wave_type=[g.w_type.TONES] * N)
            for i, b in enumerate(bufs):
                device_tones(b, i * L, rate, freq, ampl, phase, sigma=1e-3, seed=seed * 1000 + i)
        else:
            N = 1
            p = g.param(mode="RX", rate=rate, buffer_len=L, decim=wl["decim"], freq=[-rate // 2],
                        chirp_f=[rate // 2], swipe_s=[wl["swipe_s"]], chirp_t=[wl["chirp_t"]],
                        wave_type=[g.w_type.CHIRP])
            cp = g.chirp_derive(rate, -rate // 2, rate // 2, wl["swipe_s"], wl["chirp_t"])
            gen = torch.Generator(device=device).manual_seed(seed)
            for i, b in enumerate(bufs):
                device_chirp(b, i * L, cp, scale=0.5)
                b += 1e-3 * torch.view_as_complex(torch.randn(L, 2, device=device, generator=gen))
        dem = g.RX_buffer_demodulator(p, device_index=device.index)
        outs = [torch.empty(dem.out_capacity, dtype=torch.complex64, device=device) for _ in range(PIPE_DEPTH)]
        torch.cuda.synchronize(device)
        return dem, bufs, outs, N


def run_steps(dem, bufs, out, steps, stream=None, k0=0):
    n = 0
    for k in range(k0, k0 + steps):
        n = dem.process_device(bufs[k % len(bufs)], out, stream)
    return n


def run_steps_pipelined(dem, bufs, outs, steps, k0=0):
    """gsdr_demod_submit_device / gsdr_demod_wait: up to len(outs) buffers outstanding, one
    output buffer each; every submitted buffer is waited for before this returns."""
    depth, pending = len(outs), 0
    for k in range(k0, k0 + steps):
        if pending == depth:
            dem.wait()
            pending -= 1
        dem.submit_device(bufs[k % len(bufs)], outs[k % depth])
        pending += 1
    while pending:
        dem.wait()
        pending -= 1


def time_workload(engine, wl, device, seed, steps, warmup, dist=None, n_tones=None, api="inorder",
                  min_seconds=1.0, profile_every=None, max_seconds=30.0, fixed_repeats=None):
    """W warm-up steps, then R x K timed steps through one entry of the C ABI.

    api "inorder": gsdr_demod_process_device on one stream, step after step.
    api "pipelined": gsdr_demod_submit_device / gsdr_demod_wait, PIPE_DEPTH outstanding: the
    kernels of consecutive buffers overlap on the GPU (include/gsdr.h).
    R: repetitions of the K steps so that the timed region lasts >= min_seconds (estimated from
    one untimed pass of K steps, maximum over ranks so that every rank runs the same count)."""
    dem, bufs, outs, N = engine.build(wl, device, seed, n_tones=n_tones)
    stream = engine.stream(device)

    def run(count, k0):
        if api == "pipelined":
            run_steps_pipelined(dem, bufs, outs, count, k0)
        else:
            run_steps(dem, bufs, outs[0], count, stream, k0)

    engine.sync(device)
    run(warmup, 0)
    engine.sync(device)
    # how long do K steps take?  (untimed pass; also part of the warm-up)
    t0 = time.perf_counter()
    run(steps, warmup)
    engine.sync(device)
    est = max_over_ranks(dist, time.perf_counter() - t0, device)
    if fixed_repeats:
        repeats = int(fixed_repeats)
    else:
        # (the untimed pass runs warmer caches and a cooler chip than the steady state: aim 15 % over)
        repeats = max(1, int(math.ceil(1.15 * min_seconds / max(est, 1e-9))))
        repeats = max(1, min(repeats, int(max_seconds / max(est, 1e-9)) or 1))
    total = steps * repeats
    every = PROFILE_EVERY if profile_every is None else profile_every
    if every > 0:
        # hipEvents around every n-th launch of the dominant kernel, inside the timed region, on
        # the stream it is launched on (around every launch they cost ~6 us of stream time per step)
        dem.profile_enable(every)
    barrier(dist, device)
    engine.sync(device)
    t0 = time.perf_counter()
    run(total, warmup + steps)
    engine.sync(device)
    t1 = time.perf_counter()
    barrier(dist, device)
    elapsed = max_over_ranks(dist, t1 - t0, device)
    kn, kms = dem.profile_read() if every > 0 else (0, 0.0)
    kname = dem.kernel_name
    desc = dem.describe() if hasattr(dem, "describe") else {}
    dem.close()
    return dict(elapsed=elapsed, local_elapsed=t1 - t0, kernel_launches=kn, kernel_ms=kms, steps=steps,
                repeats=repeats, total_steps=total, kernel=kname, n_tones=N, api=api, engine=desc)


def host_api_rates(wl, device, seed, seconds=0.4):
    """The reference's own entry, RX_buffer_demodulator::process(float2** host, float2** host)
    (gsdr_demod_process: H2D + kernels + D2H, synchronous), and the overlapped host-pointer
    entry gsdr_demod_submit/_wait, on pinned host buffers.  PCIe-inclusive: never `value`."""
    import numpy as np
    import torch
    import gpu_sdr_amd as g
    from gpu_sdr_amd.source import host_tones, tone_comb
    if wl["kind"] != "direct":
        return None
    rate, N = wl.get("rate", RATE), wl["n_tones"]
    freq, ampl, phase = tone_comb(N, rate, seed)
    p = g.param(mode="RX", rate=rate, buffer_len=L, decim=wl["decim"], pf_average=wl["pf_average"],
                freq=[int(f) for f in freq], wave_type=[g.w_type.DIRECT] * N)
    dem = g.RX_buffer_demodulator(p, device_index=device.index)
    k = min(N, 16)
    xin = [torch.from_numpy(host_tones(L, i * L, rate, freq[:k], ampl[:k], phase[:k], sigma=1e-3, seed=seed + i)).pin_memory()
           for i in range(2)]
    outs = [torch.empty(dem.out_capacity, dtype=torch.complex64).pin_memory() for _ in range(PIPE_DEPTH)]
    xn, on = [t.numpy() for t in xin], [t.numpy() for t in outs]
    res = {}
    dem.prepare()          # what the C++ class does in its constructor (like the reference's): no lazy allocation in the loop
    for _ in range(3):
        dem.process(xn[0], on[0])
    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < seconds or n < 20:
        dem.process(xn[n % 2], on[0])
        n += 1
    dt = time.perf_counter() - t0
    res["process"] = dict(api="gsdr_demod_process (host pointers, synchronous, the reference's process())",
                          msamples_per_s=round(n * L / dt / 1e6, 1), ms_per_buffer=round(dt / n * 1e3, 4), buffers=n)
    t0, n, pending, worst = time.perf_counter(), 0, 0, 0.0
    while time.perf_counter() - t0 < seconds or n < 20:
        ts = time.perf_counter()
        if pending == PIPE_DEPTH:
            dem.wait()
            pending -= 1
        dem.submit(xn[n % 2], on[n % PIPE_DEPTH])
        pending += 1
        n += 1
        if n > PIPE_DEPTH + 2:
            worst = max(worst, time.perf_counter() - ts)
    while pending:
        dem.wait()
        pending -= 1
    dt = time.perf_counter() - t0
    res["submit_wait"] = dict(api="gsdr_demod_submit/_wait (host pointers, H2D | kernels | D2H overlapped)",
                              msamples_per_s=round(n * L / dt / 1e6, 1), ms_per_buffer=round(dt / n * 1e3, 4),
                              worst_call_ms=round(worst * 1e3, 3), buffers=n)
    dem.close()
    return res


def max_realtime_tones(engine, device, seed, budget_s=75.0, api="pipelined"):
    """Largest N (multiple of 1024) sustaining >= 200 Msps over 200 consecutive
    1 M-sample buffers with decim=1000, f=4."""
    wl = dict(WORKLOADS["c3"])
    t_start = time.perf_counter()
    best, probes = 0, []
    lo, hi = 2048, None
    n = 2048
    while time.perf_counter() - t_start < budget_s:
        r = time_workload(engine, wl, device, seed, steps=200, warmup=5, n_tones=n, api=api,
                          fixed_repeats=1, profile_every=0)
        msps = 200 * L / r["elapsed"] / 1e6
        probes.append((n, round(msps, 1)))
        if msps >= 200.0:
            best, lo = n, n
            n = n * 2 if hi is None else (lo + hi) // 2 // 1024 * 1024
        else:
            hi = n
            n = (lo + hi) // 2 // 1024 * 1024
        if hi is not None and (hi - lo <= 1024 or n <= lo):
            break
        if n > 131072:
            break
    return best, probes


# --------------------------------------------------------------------------
# CPU baselines (rank 0, N = 1 only; they run BEFORE this process touches the GPU:
# the process pool fork()s)
# --------------------------------------------------------------------------
def _cpu_input(wl, seed):
    """One buffer shaped like the workload's stream: a few of its tones + noise (the CPU
    recipes do the same work whatever the samples are; 2048 host-side tones would take a
    minute to synthesise)."""
    from gpu_sdr_amd.source import host_tones, tone_comb
    rate = wl.get("rate", RATE)
    freq, ampl, phase = tone_comb(wl["n_tones"], rate, seed)
    k = min(16, len(freq))
    return host_tones(L, 0, rate, freq[:k], ampl[:k], phase[:k], sigma=1e-3, seed=seed), freq


def cpu_recipe_a(wl, seed, procs, what):
    """Recipe A (oracle/recipe_a.py: scripts/raw_data_analisys.py:55-68), every tone of one
    1 M-sample buffer of `wl`, tones spread over `procs` processes."""
    from oracle import recipe_a
    Z, freq = _cpu_input(wl, seed)
    r = recipe_a.run(Z, freq, wl.get("rate", RATE), wl["decim"], procs)
    return dict(value=round(r["msamples_per_s"], 5), unit="Msamples/s", cores=r["procs"], kind="port",
                tone_msamples_per_s=round(r["tone_msamples_per_s"], 3),
                sample=f"{what}: all {r['tones']} tones x one 1M-sample buffer, the reference's offline recipe "
                       f"(scripts/raw_data_analisys.py:55-68: numpy mix + scipy.signal.decimate ftype='fir') restated "
                       f"for Python 3, tones spread over {r['procs']} processes "
                       f"(pyUSRP's own default is N_CORES = 10), {r['seconds']:.1f} s")


def cpu_oracle(wl, seed, threads, min_seconds=4.0, max_buffers=32):
    """The C oracle (OpenMP restatement of the reference algorithm) on the same workload."""
    import numpy as np
    import oracle
    from gpu_sdr_amd.source import tone_comb
    oracle.build()
    oracle.set_num_threads(threads)
    rate = wl.get("rate", RATE)
    rng = np.random.default_rng(seed)
    if wl["kind"] == "direct":
        N = wl["n_tones"]
        freq, _, _ = tone_comb(N, rate, seed)
        dem = oracle.Direct(freq, rate, wl["decim"], wl["pf_average"], L)
        what = f"{N} tones"
    elif wl["kind"] == "pfb":
        N = wl["n_tones"]
        freq, _, _ = tone_comb(N, rate, seed)
        dem = oracle.Pfb(freq, rate, wl["fft_tones"], wl["pf_average"], L)
        what = f"{N} PFB tones"
    else:
        dem = oracle.Chirp(rate, -rate // 2, rate // 2, wl["swipe_s"], wl["chirp_t"], wl["decim"], L)
        what = "chirp lock-in"
    x = (rng.standard_normal(L) + 1j * rng.standard_normal(L)).astype(np.complex64)
    t0 = time.perf_counter()
    nb = 0
    while nb < max_buffers and (nb == 0 or time.perf_counter() - t0 < min_seconds):
        dem.process(x)
        nb += 1
    dt = time.perf_counter() - t0
    return dict(value=round(nb * L / dt / 1e6, 4), unit="Msamples/s", cores=oracle.num_threads(), kind="port",
                sample=f"{nb} x 1M-sample buffers, {what}, oracle/gsdr_oracle.c (OpenMP), {dt:.1f} s")


def cpu_recipe_b(seed):
    """Recipe B (oracle/recipe_b.py, the reference-exact numpy restatement) on BASELINE config 1
    in full: 16 tones, 1 Msample, 100 Msps, decim 100; single process, numpy's BLAS threads."""
    from oracle import recipe_b
    wl = WORKLOADS["c1"]
    Z, freq = _cpu_input(wl, seed)
    dem = recipe_b.Direct(freq, wl["rate"], wl["decim"], wl["pf_average"], L)
    t0 = time.perf_counter()
    dem.process(Z)
    dt = time.perf_counter() - t0
    return dict(value=round(L / dt / 1e6, 4), unit="Msamples/s", cores=1, kind="port",
                sample=f"config 1 in full (16 tones, 1 Msample, 100 Msps, decim 100), oracle/recipe_b.py "
                       f"(integer-phase NCO, cgemm + shifted axpy as cpp/fir.cu:48-61), one process, {dt:.1f} s")


def cpu_baselines(key, seed):
    from oracle import recipe_a
    cores = recipe_a.host_cores()
    procs = max(1, min(cores, CPU_PROCS_CAP))
    wl = WORKLOADS[key]
    out = {}
    note = f"host cores visible {os.cpu_count()}, usable {cores}, used {procs} (one GPU's share of the box)"
    if wl["kind"] == "direct":
        out["cpu_baseline"] = cpu_recipe_a(wl, seed, procs, key.upper())
        out["cpu_baseline"]["host"] = note
        for other in ("c1", "c2"):
            if other != key:
                out["cpu_baseline_" + other] = cpu_recipe_a(WORKLOADS[other], seed, procs,
                                                            "config 1 in full" if other == "c1" else other.upper())
        out["cpu_baseline_oracle"] = cpu_oracle(wl, seed, procs)
    else:
        out["cpu_baseline"] = cpu_oracle(wl, seed, procs, min_seconds=10.0, max_buffers=64)
        out["cpu_baseline"]["host"] = note
    out["cpu_baseline_recipe_b"] = cpu_recipe_b(seed)
    return out


# --------------------------------------------------------------------------
# rooflines
# --------------------------------------------------------------------------
def recorded_traffic(workload: str):
    """HBM bytes per launch of the dominant kernel from the rocprofv3 --pmc runs
    committed under profiles/ (FETCH_SIZE corrected x2 per MI355X_MICROARCH.md)."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        return json.load(open(path)).get(workload, {}).get("hbm_bytes_per_launch")
    except Exception:
        return None


def kernel_seconds(r):
    return (r["kernel_ms"] / r["kernel_launches"] * 1e-3) if r["kernel_launches"] else None


def rooflines(wl, r, key, period_s=None):
    """(roofline of the bounding pipe, HBM roofline) of the dominant kernel from the hipEvent
    durations `r` (a time_workload() result) carries; (None, None) without them.

    achieved = ALGORITHMIC flops (bytes) per launch / average launch duration."""
    ab, af = algorithmic(wl, r["n_tones"])
    kt = kernel_seconds(r)
    if not kt:
        return None, None
    how = f"hipEvents on the launch stream, {r['api']} entry"
    extra = {}
    if period_s is not None and period_s < kt:
        extra = dict(kernel_us_between_events=round(kt * 1e6, 2))
        how = (f"step period of back-to-back launches on one stream (one kernel per step, {r['api']} entry, no events in "
               "the region): an upper bound of the launch duration; a hipEvent pair around a launch this short adds "
               "~2.5 us (kernel_us_between_events); rocprofv3 --kernel-trace: profiles/")
        kt = period_s
    gbs = ab * L / kt / 1e9
    tfl = af * L / kt / 1e12
    traffic = recorded_traffic(key)
    common = dict(traffic=traffic, kernel=r["kernel"], kernel_us=round(kt * 1e6, 2), launches_timed=r["kernel_launches"],
                  measured_with=how, **extra)
    roof_hbm = dict(bound="hbm", achieved=round(gbs, 2), peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=round(gbs / HBM_PEAK_GBS, 5), algorithmic_bytes_per_launch=round(ab * L), **common)
    if wl["kind"] in ("direct", "pfb") and r["kernel"].startswith("ddc_mfma"):
        # Matrix-core DDC.  Algorithmic work: N(6+4f) flop per input sample.  The
        # kernel runs it on the f16 MFMA pipe, so that pipe's dense peak is the roof.  What the
        # pipe executes is more: every (tone, sample, tap phase) is one complex MAC done as three
        # fp16 x fp16 -> fp32 products of a hi/lo split, 3 * 8 = 24 flop, i.e. 24 f N per sample
        # (8f instead of 6+4f from folding the taps into the A operand, x 3 for the split).
        mf = 24.0 * wl["pf_average"] * r["n_tones"]
        mtfl = mf * L / kt / 1e12
        roof = dict(bound="mfma", pipe="f16 MFMA (dense peak 2.5 PFLOP/s), fp32 accumulate",
                    achieved=round(tfl, 2), peak=F16_MFMA_PEAK_TFLOPS, unit="TFLOP/s",
                    frac=round(tfl / F16_MFMA_PEAK_TFLOPS, 4),
                    algorithmic_flops_per_launch=round(af * L),
                    executed_mfma_flops_per_launch=round(mf * L),
                    executed_mfma_tflops=round(mtfl, 1),
                    executed_mfma_frac=round(mtfl / F16_MFMA_PEAK_TFLOPS, 4),
                    frac_of_fp32_vector_peak=round(tfl / FP32_PEAK_TFLOPS, 4), **common)
    elif r["kernel"] in ("pfb_lds_kernel", "pfb_cu_kernel"):
        # TONES as the reference does it: polyphase filter + FFT + bin selection, a frame per workgroup
        # inside the LDS.  A few flops per byte (4 f + 5 log2 nfft per sample): HBM-bound -- one read of
        # the window, one write of the selected bins.
        roof = dict(roof_hbm, note="polyphase filter + in-LDS FFT + bin selection (the reference's own algorithm): "
                                   "HBM-bound; algorithmic bytes = 8 B read per sample + 8 B per selected bin and frame")
    elif wl["kind"] in ("direct", "pfb"):
        # packed-FP32 DDC (GSDR_DDC_MFMA=0): FP32-compute bound. The FP32
        # vector peak equals the FP32 (f32-input) MFMA peak on gfx950: 157.3 TF.
        roof = dict(bound="mfma", pipe="fp32 valu (no MFMA used; same 157.3 TF peak)",
                    achieved=round(tfl, 3), peak=FP32_PEAK_TFLOPS, unit="TFLOP/s",
                    frac=round(tfl / FP32_PEAK_TFLOPS, 4), algorithmic_flops_per_launch=round(af * L), **common)
    else:
        roof = roof_hbm
    return roof, roof_hbm


def measure(engine, key, device, seed, steps, warmup, dist, api, min_seconds, with_inorder=True):
    """The timed region of one workload through `api`, plus (for the overlapped entry) the
    in-order pass the roofline is taken from: a launch of the overlapped entry shares the chip
    with its neighbours, so its duration is not the kernel's."""
    wl = WORKLOADS[key]
    # chirp: a step is ONE launch of ~4 us; an event pair around it costs ~2.5 us of stream time, so the
    # timed region of `value` carries no events and the event-bracketed durations come from a pass of their own
    # (chirp: 4 - 5 us per launch; TONES / NOISE inside the LDS: one launch of 10 - 15 us per step when run in order)
    short_step = wl["kind"] == "chirp" or (wl["kind"] == "pfb" and api == "inorder")
    r = time_workload(engine, wl, device, seed, steps, warmup, dist, api=api, min_seconds=min_seconds,
                      profile_every=0 if short_step else None)
    world = dist.get_world_size() if dist is not None else 1
    res = dict(r=r, value=r["total_steps"] * L * world / r["elapsed"] / 1e6,
               ms_per_step=r["elapsed"] / r["total_steps"] * 1e3)
    ri = r
    if short_step:
        ri = time_workload(engine, wl, device, seed, steps, warmup, dist, api=api, min_seconds=min(min_seconds, 0.3))
    if api == "pipelined" and with_inorder:
        ri = time_workload(engine, wl, device, seed, steps, warmup, dist, api="inorder",
                           min_seconds=min_seconds, profile_every=4)
        res["inorder"] = dict(api="gsdr_demod_process_device, one stream",
                              value=round(ri["total_steps"] * L / ri["local_elapsed"] / 1e6, 2),
                              unit="Msamples/s per GPU", repeats=ri["repeats"],
                              ms_per_step=round(ri["local_elapsed"] / ri["total_steps"] * 1e3, 5))
    # one kernel per step, steps back to back on one stream: the step period of the event-free region
    # bounds the launch duration from above
    roof, roof_hbm = rooflines(wl, ri, key, period_s=r["local_elapsed"] / r["total_steps"] if short_step else None)
    if roof and ri is not r and not short_step:
        kt = kernel_seconds(r)
        if kt:
            for ro in {id(roof): roof, id(roof_hbm): roof_hbm}.values():
                ro["overlapped_entry"] = dict(
                    kernel_us=round(kt * 1e6, 2),
                    launches_in_flight=round(kt / (r["local_elapsed"] / r["total_steps"]), 2),
                    note="average launch duration inside the timed region of `value`, where up to "
                         f"{PIPE_DEPTH} launches share the chip")
    if roof and ri is not r and not short_step and roof.get("bound") == "mfma":
        # the entry `value` is timed through: what ran there, and the same algorithmic work over ITS time per buffer
        _, af = algorithmic(wl, r["n_tones"])
        step_s = r["local_elapsed"] / r["total_steps"]
        conv = r["kernel"] == "ddc_mfma_ring16p_kernel"
        roof["timed_entry"] = dict(
            api=r["api"], kernels=["absmax_kernel"] + (["ddc_convert_kernel"] if conv else []) + [r["kernel"]],
            gpu_us_per_buffer=round(step_s * 1e6, 2),
            achieved=round(af * L / step_s / 1e12, 2), unit=roof["unit"], peak=roof["peak"],
            frac=round(af * L / step_s / 1e12 / roof["peak"], 4),
            note="algorithmic flops per buffer / time per buffer of the timed region (launches of consecutive buffers "
                 "overlap there, so this is chip throughput, not one launch's duration) / the same peak")
    if roof and ri is not r and not short_step and ri["kernel"] != r["kernel"]:
        roof["note"] = (f"kernel / kernel_us are those of the in-order pass ({ri['kernel']}: the library picks the kernel per "
                        f"launch, DESIGN.md 4.1a/4.1b); the overlapped entry that `value` is timed through ran {r['kernel']} "
                        f"(+ ddc_convert_kernel when that is the pre-converted path); same tables, bit-identical results")
    res["roofline"], res["roofline_hbm"] = roof, roof_hbm
    return res


def fp32_valu_extras(engine, device, seed):
    """C3 through the engine that follows north_star's arithmetic to the letter -- fp32 multiplies on the VALU, no
    matrix cores (GSDR_DDC_MFMA=0: ddc_flat_kernel, packed FP32; the reference multiplies in fp32 too: cuBLAS Cgemm,
    cpp/fir.cu:48-54) -- as a figure at the reference's own arithmetic beside the headline."""
    saved = os.environ.get("GSDR_DDC_MFMA")
    os.environ["GSDR_DDC_MFMA"] = "0"       # read when a demodulator is created
    try:
        e = measure(engine, "c3", device, seed, steps=100, warmup=10, dist=None, api="inorder", min_seconds=0.5)
        best, probes = max_realtime_tones(engine, device, seed, budget_s=40.0, api="inorder")
    finally:
        if saved is None:
            del os.environ["GSDR_DDC_MFMA"]
        else:
            os.environ["GSDR_DDC_MFMA"] = saved
    ro = e["roofline"] or {}
    return dict(workload=WORKLOADS["c3"]["name"], engine=e["r"]["engine"], dtype="f32 (packed FP32 VALU multiplies and adds)",
                msamples_per_s=round(e["value"], 2), api=e["r"]["api"], us_per_buffer=round(e["ms_per_step"] * 1e3, 2),
                kernel=ro.get("kernel"), kernel_us=ro.get("kernel_us"), frac=ro.get("frac"), peak=ro.get("peak"),
                achieved=ro.get("achieved"), unit=ro.get("unit"), roofline=ro,
                max_realtime_tones_200Msps=dict(value=best, decim=1000, pf_average=4, buffers=200, probes=probes))


# --------------------------------------------------------------------------
def main(argv=None, engine=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--api", default="auto", choices=["auto", "inorder", "pipelined"],
                    help="entry the timed steps go through; auto = pipelined for the DDC (DIRECT, TONES) workloads")
    ap.add_argument("--min-seconds", type=float, default=1.0,
                    help="the K steps are repeated until the timed region is at least this long")
    ap.add_argument("--tones", type=int, default=0,
                    help="override the workload's tone count (experiments; the line says so in config.tones_per_stream)")
    ap.add_argument("--no-extras", action="store_true", help="skip the C2 / PFB / chirp / max-tone extras")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baselines")
    ap.add_argument("--no-host-api", action="store_true",
                    help="skip the host-pointer (PCIe-inclusive) rates: profiling runs want the traced kernels to be "
                         "those of the timed entry only")
    ap.add_argument("--ablation", action="store_true",
                    help="allow a timing-only ablation build (GSDR_LIB / -DGSDR_TIMING_BUILD); the line is marked invalid")
    args = ap.parse_args(argv)

    rank, local_rank, world = dist_env()
    stub = engine is not None
    if args.gpus > 1 and world == 1 and not stub:
        # convenience: re-launch under torchrun as a child process (never exec
        # after the GPU may have been touched)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
               f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1", "--master-port", "29517",
               os.path.abspath(__file__)] + (sys.argv[1:] if argv is None else list(argv))
        sys.exit(subprocess.call(cmd))

    # a bench whose kernels can be told to skip work by the environment is not a bench
    timing_env = {k: v for k, v in os.environ.items()
                  if (k == "GSDR_LIB" and v) or (k == "GSDR_MFMA_TIMING" and v not in ("", "0"))}
    if timing_env and not args.ablation:
        print(f"bench.py: refusing to run with {timing_env} set (timing-only ablation switches); "
              f"pass --ablation to get a line marked invalid", file=sys.stderr)
        sys.exit(2)

    wl = WORKLOADS[args.workload]
    if args.tones > 0 and "n_tones" in wl:
        wl = WORKLOADS[args.workload] = dict(wl, n_tones=args.tones, name=wl["name"] + f" [tones overridden: {args.tones}]")
    seed = stream_seed(rank)
    cpu = {}
    if not stub and world == 1 and rank == 0 and not args.no_cpu:
        cpu = cpu_baselines(args.workload, seed)      # before the GPU is touched (fork)

    import torch
    if stub:
        device = torch.device("cpu")
        n_dev = 0
        build_info = "stub"
    else:
        engine = HipEngine()
        if not torch.cuda.is_available():
            print("bench.py needs a GPU (the product path has no CPU fallback)", file=sys.stderr)
            sys.exit(2)
        n_dev = torch.cuda.device_count()
        # one GPU per rank (the reference is one GPU per server process,
        # cpp/USRP_hardware_manager.cpp:68); with fewer devices than ranks -- the one-GPU
        # rehearsal -- ranks share devices and the control plane moves to gloo
        device = torch.device("cuda", local_rank % max(n_dev, 1))
        torch.cuda.set_device(device)
        build_info = engine.check(args.ablation)
    backend = pick_backend(world, n_dev) if not stub else os.environ.get("GSDR_BENCH_BACKEND", "gloo")
    dist = init_group(backend) if world > 1 else None
    # where the tensors of barrier()/max_over_ranks() live: on the GPU for RCCL, on the CPU for gloo
    ctl_device = device if (dist is None or backend == "nccl") else None

    # the matrix-core DDC overlaps consecutive buffers on rotating streams; TONES / NOISE / chirp launches are 4 - 15 us
    # and keep one stream (an event between calls costs 2 - 3 us of stream time: profiles/r03_pfb_api_ab.log)
    api = args.api if args.api != "auto" else ("pipelined" if wl["kind"] == "direct" else "inorder")

    m = _measure_with_ctl(engine, args.workload, device, ctl_device, seed, args.steps, args.warmup, dist, api,
                          args.min_seconds)
    r = m["r"]
    line = {
        "metric": "IQ Msamples/s ingested (one synthetic 200 Msps stream per GPU)",
        "value": round(m["value"], 2), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "repeats": r["repeats"], "timed_region_s": round(r["elapsed"], 4),
        "ms_per_step": round(m["ms_per_step"], 5), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 (f16 hi/lo-split MFMA, f32 accumulate)" if r["kernel"].startswith("ddc_mfma") else "f32",
        "dtype_note": ("complex64 in and out, fp32 accumulate; the DDC multiplies on the f16 MFMA as three products "
                       "of hi/lo-split fp32 operands under one power-of-two scale per output row (error per tone vs the "
                       "fp64 oracle: profiles/r03_parity_margins.json, bar 1e-5; the same workload with fp32 multiplies on "
                       "the VALU is extras.fp32_valu); chirp: fp32 VALU with an exact integer phase"),
        "data": "synthetic",
        "config": {"workload": wl["name"], "key": args.workload, "buffer_len": L,
                   "rate": wl.get("rate", RATE), "tones_per_stream": r["n_tones"],
                   "streams": world, "parallelism": f"{world} independent stream(s), one per GPU, no collective",
                   "control_plane": backend if world > 1 else "none",
                   "api": ("device-resident extension of the C ABI: gsdr_demod_submit_device/gsdr_demod_wait, "
                           "%d buffers outstanding (inputs and outputs stay in HBM; the reference's own "
                           "process(host, host) rate is under host_api)" % PIPE_DEPTH)
                          if api == "pipelined" else
                          "device-resident extension of the C ABI: gsdr_demod_process_device, one stream "
                          "(inputs and outputs stay in HBM)"},
        "realtime_factor": round(m["value"] / world / (wl.get("rate", RATE) / 1e6), 3),
        "roofline": m["roofline"], "roofline_hbm": m["roofline_hbm"],
        "engine": dict(r["engine"], build=build_info),
    }
    if "inorder" in m:
        line["inorder"] = m["inorder"]
    if world > 1:
        # every rank's own clock around its own timed region: a straggler shows here, not only in the maximum
        local = gather_over_ranks(dist, r["local_elapsed"], ctl_device)
        line["per_rank"] = [dict(rank=i, msamples_per_s=round(r["total_steps"] * L / t / 1e6, 2),
                                 ms_per_step=round(t / r["total_steps"] * 1e3, 5)) for i, t in enumerate(local)]
    if timing_env or "timing_build 0" not in build_info and not stub:
        line["INVALID"] = f"timing-only ablation build or switches ({timing_env or build_info}): not a benchmark"

    if world == 1 and rank == 0 and not stub:
        ha = None if args.no_host_api else host_api_rates(wl, device, seed)
        if ha:
            line["host_api"] = ha
        if not args.no_extras:
            extras = {}
            for key in ("c2", "pfb", "c4"):
                if key == args.workload:
                    continue
                ek = WORKLOADS[key]
                eapi = "pipelined" if ek["kind"] == "direct" else "inorder"   # the DDC overlaps buffers; the others keep one stream
                e = measure(engine, key, device, seed, steps=500, warmup=50, dist=None, api=eapi, min_seconds=0.5)
                er = e["r"]
                extras[key] = dict(workload=ek["name"], msamples_per_s=round(e["value"], 2), api=er["api"],
                                   us_per_buffer=round(e["ms_per_step"] * 1e3, 2), repeats=er["repeats"],
                                   roofline=e["roofline"], roofline_hbm=e["roofline_hbm"])
                if "inorder" in e:
                    extras[key]["inorder"] = e["inorder"]
            best, probes = max_realtime_tones(engine, device, seed)
            extras["max_realtime_tones_200Msps"] = dict(value=best, decim=1000, pf_average=4,
                                                        buffers=200, probes=probes)
            if wl["kind"] == "direct":
                extras["fp32_valu"] = fp32_valu_extras(engine, device, seed)
            line["extras"] = extras
        line.update(cpu)
    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return line


def _measure_with_ctl(engine, key, device, ctl_device, seed, steps, warmup, dist, api, min_seconds):
    """measure() with the reduction tensors of barrier()/max_over_ranks() on `ctl_device`
    (None = CPU, for a gloo control plane beside GPU work)."""
    if dist is None or ctl_device is device:
        return measure(engine, key, device, seed, steps, warmup, dist, api, min_seconds)

    class Wrapped:
        """engine whose sync() still drains the GPU while the collectives stay on the CPU"""
        def __init__(self, e):
            self.e = e
        def sync(self, _dev):
            self.e.sync(device)
        def stream(self, _dev):
            return self.e.stream(device)
        def build(self, wl, _dev, seed, ring=8, n_tones=None):
            return self.e.build(wl, device, seed, ring=ring, n_tones=n_tones)
    return measure(Wrapped(engine), key, ctl_device, seed, steps, warmup, dist, api, min_seconds)


if __name__ == "__main__":
    main()
