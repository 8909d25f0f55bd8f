#!/usr/bin/env python3
"""bench.py -- throughput of the RX demodulation hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload c2|c3|c4|c1]

A "step" is one pass of the hot path (gsdr_demod_process_device through the
C ABI) over one 1 M-sample buffer of synthetic IQ that is already resident in
HBM (ring of distinct buffers made by the HIP source kernel before the timed
region).  One independent synthetic 200 Msps stream per GPU, no inter-GPU data
traffic (front-end streams are independent: SURVEY.md section 8e), so scaling
is "weak" and `value` is the aggregate Msamples/s of all ranks.

For N > 1 the driver launches this file under torch.distributed.run; the
process group (RCCL) is used ONLY for the barrier and the max-over-ranks of the
elapsed time.

Rank 0 prints ONE JSON line (contract in the task statement), with
  roofline      dominant kernel vs the roof that bounds it (FP32 compute for the
                fused DDC, HBM for the chirp path); `roofline_hbm` always
                carries the HBM view (north_star asks for HBM GB/s),
  cpu_baseline  the CPU oracle (OpenMP, all host cores) on a bounded sample of
                the same workload, N = 1 only.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

RATE = 200_000_000
L = 1_000_000
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8 TB/s
FP32_PEAK_TFLOPS = 157.3    # MI355X_MICROARCH.md: FP32 vector == FP32 matrix peak
PROFILE_EVERY = int(os.environ.get("GSDR_BENCH_PROFILE_EVERY", "8"))   # 0: no kernel timing
F16_MFMA_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense f16/bf16 MFMA

WORKLOADS = {
    # BASELINE.json configs[1]
    "c2": dict(kind="direct", n_tones=256, decim=100, pf_average=4,
               name="256-tone DDC + polyphase FIR, decim=100, 200 Msps synthetic stream"),
    # configs[2] / configs[4]
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
        dist.init_process_group(backend=backend)
    return dist


def barrier(dist, device=None):
    if dist is None:
        return
    import torch
    t = torch.zeros(1, device=device) if device is not None else torch.zeros(1)
    dist.all_reduce(t)
    if device is not None and device.type == "cuda":
        torch.cuda.synchronize(device)


def max_over_ranks(dist, seconds: float, device=None) -> float:
    if dist is None:
        return seconds
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(dist, v: float, device=None) -> float:
    if dist is None:
        return v
    import torch
    t = torch.tensor([v], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


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


def build_workload(wl, device, seed, ring=8, n_tones=None):
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
    out = torch.empty(dem.out_capacity, dtype=torch.complex64, device=device)
    torch.cuda.synchronize(device)
    return dem, bufs, out, N, p


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


PIPE_DEPTH = int(os.environ.get("GSDR_BENCH_DEPTH", "3"))   # buffers outstanding, <= GSDR_PIPELINE_DEPTH (4)


def time_workload(wl, device, seed, steps, warmup, dist=None, n_tones=None, profile=True,
                  ctl_device="same", api="inorder"):
    """api "inorder": gsdr_demod_process_device on one stream, step after step.
    api "pipelined": gsdr_demod_submit_device / gsdr_demod_wait, PIPE_DEPTH outstanding: the
    kernels of consecutive buffers overlap on the GPU (include/gsdr.h)."""
    import torch
    if ctl_device == "same":
        ctl_device = device
    dem, bufs, out, N, _ = build_workload(wl, device, seed, n_tones=n_tones)
    stream = torch.cuda.Stream(device)  # the hot path runs on its own (non-null) stream
    outs = [out] + [torch.empty_like(out) for _ in range(PIPE_DEPTH - 1)] if api == "pipelined" else None

    def run(count, k0):
        if api == "pipelined":
            run_steps_pipelined(dem, bufs, outs, count, k0)
        else:
            run_steps(dem, bufs, out, count, stream, k0)

    torch.cuda.synchronize(device)
    run(warmup, 0)
    torch.cuda.synchronize(device)
    profile = profile and PROFILE_EVERY > 0
    if profile:
        # hipEvents around every 8th launch of the dominant kernel, inside the timed region
        # (around every launch they cost ~6 us of stream time per step)
        dem.profile_enable(PROFILE_EVERY)
    barrier(dist, ctl_device)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    run(steps, warmup)
    torch.cuda.synchronize(device)
    t1 = time.perf_counter()
    barrier(dist, ctl_device)
    elapsed = max_over_ranks(dist, t1 - t0, ctl_device)
    kn, kms = dem.profile_read() if profile else (0, 0.0)
    kname = dem.kernel_name
    dem.close()
    return dict(elapsed=elapsed, local_elapsed=t1 - t0, kernel_launches=kn, kernel_ms=kms,
                kernel=kname, n_tones=N, api=api)


def max_realtime_tones(device, seed, budget_s=60.0):
    """Largest N (multiple of 1024) sustaining >= 200 Msps over 200 consecutive
    1 M-sample buffers with decim=1000, f=4."""
    wl = dict(WORKLOADS["c3"])
    t_start = time.perf_counter()
    best, probes = 0, []
    lo, hi = 2048, None
    n = 2048
    while time.perf_counter() - t_start < budget_s:
        r = time_workload(wl, device, seed, steps=200, warmup=5, n_tones=n, profile=False, api="pipelined")
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
# CPU baselines (rank 0, N = 1 only)
# --------------------------------------------------------------------------
def cpu_baseline_oracle(wl, seed, min_seconds=10.0, max_buffers=64):
    """The CPU oracle (kind "port": OpenMP C restatement of the reference
    algorithm) on a bounded sample of the same workload."""
    import numpy as np
    import oracle
    from gpu_sdr_amd.source import tone_comb
    oracle.build()
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
    return dict(value=round(nb * L / dt / 1e6, 4), unit="Msamples/s", cores=oracle.num_threads(),
                kind="port", sample=f"{nb} x 1M-sample buffers, {what}, oracle/gsdr_oracle.c (OpenMP), {dt:.1f} s")


def cpu_baseline_numpy(wl, seed, tones=16):
    """pyUSRP/numpy offline demod recipe (ref: scripts/raw_data_analisys.py:55-68):
    per tone conj(exp(2 pi i f n / rate)) * Z then scipy.signal.decimate(ftype='fir')."""
    import numpy as np
    from scipy import signal
    from gpu_sdr_amd.source import tone_comb
    if wl["kind"] != "direct":
        return None
    rate = wl.get("rate", RATE)
    freq, _, _ = tone_comb(wl["n_tones"], rate, seed)
    rng = np.random.default_rng(seed)
    Z = (rng.standard_normal(L) + 1j * rng.standard_normal(L)).astype(np.complex64)
    n = np.arange(L)
    t0 = time.perf_counter()
    for f in freq[:tones]:
        res = np.conj(np.exp(1.j * (np.pi * 2. * f / rate * n))) * Z
        signal.decimate(res, wl["decim"], ftype="fir")
    dt = time.perf_counter() - t0
    tone_msps = tones * L / dt / 1e6
    return dict(value=round(tone_msps / wl["n_tones"], 5), unit="Msamples/s", cores=1, kind="port",
                tone_msamples_per_s=round(tone_msps, 3),
                sample=f"{tones} of {wl['n_tones']} tones x 1 buffer, numpy/scipy recipe of "
                       f"scripts/raw_data_analisys.py:55-68, single process, {dt:.1f} s; value scaled to all tones")


def recorded_traffic(workload: str):
    """HBM bytes per launch of the dominant kernel from the rocprofv3 --pmc runs
    committed under profiles/ (FETCH_SIZE corrected x2 per MI355X_MICROARCH.md)."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        return json.load(open(path)).get(workload, {}).get("hbm_bytes_per_launch")
    except Exception:
        return None


def rooflines(wl, r, key):
    """(roofline of the bounding pipe, HBM roofline) of the dominant kernel from the hipEvent
    durations time_workload() collected; (None, None) without them."""
    ab, af = algorithmic(wl, r["n_tones"])
    kt = (r["kernel_ms"] / r["kernel_launches"] * 1e-3) if r["kernel_launches"] else None
    if not kt:
        return None, None
    gbs = ab * L / kt / 1e9
    tfl = af * L / kt / 1e12
    traffic = recorded_traffic(key)
    roof_hbm = dict(bound="hbm", achieved=round(gbs, 2), peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=round(gbs / HBM_PEAK_GBS, 5), traffic=traffic,
                    kernel=r["kernel"], kernel_us=round(kt * 1e6, 2))
    if wl["kind"] in ("direct", "pfb") and r["kernel"].startswith("ddc_mfma"):
        # matrix-core DDC: every (tone, sample, tap phase) is one complex MAC done as
        # three fp16 x fp16 -> fp32 products of a hi/lo split: 3 * 4 real MACs = 24 flop
        # on the f16 MFMA pipe; peak = dense f16 MFMA.
        mf = 24.0 * wl["pf_average"] * r["n_tones"]
        mtfl = mf * L / kt / 1e12
        roof = dict(bound="mfma", pipe="f16 MFMA, fp32 accumulate, 3-product hi/lo split of fp32 operands",
                    achieved=round(mtfl, 1), peak=F16_MFMA_PEAK_TFLOPS, unit="TFLOP/s",
                    frac=round(mtfl / F16_MFMA_PEAK_TFLOPS, 4), traffic=traffic,
                    fp32_equivalent_tflops=round(tfl, 1),
                    kernel=r["kernel"], kernel_us=round(kt * 1e6, 2))
    elif wl["kind"] in ("direct", "pfb"):
        # packed-FP32 DDC (GSDR_DDC_MFMA=0): FP32-compute bound. The FP32
        # vector peak equals the FP32 (f32-input) MFMA peak on gfx950: 157.3 TF.
        roof = dict(bound="mfma", pipe="fp32 valu (no MFMA used; same 157.3 TF peak)",
                    achieved=round(tfl, 3), peak=FP32_PEAK_TFLOPS, unit="TFLOP/s",
                    frac=round(tfl / FP32_PEAK_TFLOPS, 4), traffic=traffic,
                    kernel=r["kernel"], kernel_us=round(kt * 1e6, 2))
    else:
        roof = roof_hbm
    return roof, roof_hbm


# --------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--api", default="auto", choices=["auto", "inorder", "pipelined"],
                    help="entry the timed steps go through; auto = pipelined for the DDC (DIRECT, TONES) workloads")
    ap.add_argument("--no-extras", action="store_true", help="skip c3/c4/max-tone extras")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline")
    args = ap.parse_args()

    rank, local_rank, world = dist_env()
    if args.gpus > 1 and world == 1:
        # convenience: re-launch under torchrun as a child process (never exec
        # after the GPU may have been touched)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
               f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1", "--master-port", "29517",
               os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    import torch
    if not torch.cuda.is_available():
        print("bench.py needs a GPU (the product path has no CPU fallback)", file=sys.stderr)
        sys.exit(2)
    # GSDR_BENCH_ONE_DEVICE=1 / GSDR_BENCH_BACKEND=gloo: rehearsal of the N > 1 path on a
    # one-GPU box (all ranks on cuda:0, control-plane collectives over gloo on the CPU)
    one_dev = os.environ.get("GSDR_BENCH_ONE_DEVICE") == "1"
    backend = os.environ.get("GSDR_BENCH_BACKEND", "nccl")
    device = torch.device("cuda", 0 if one_dev else local_rank)
    torch.cuda.set_device(device)
    dist = init_group(backend) if world > 1 else None
    ctl_device = device if backend == "nccl" else None   # where barrier/reduce tensors live

    wl = WORKLOADS[args.workload]
    seed = stream_seed(rank)
    api = args.api if args.api != "auto" else ("pipelined" if wl["kind"] in ("direct", "pfb") else "inorder")
    r = time_workload(wl, device, seed, args.steps, args.warmup, dist, ctl_device=ctl_device, api=api)
    samples_total = args.steps * L * world
    value = samples_total / r["elapsed"] / 1e6
    ms_per_step = r["elapsed"] / args.steps * 1e3

    roof, roof_hbm = rooflines(wl, r, args.workload)
    inorder = None
    if api == "pipelined" and roof:
        # Launches of consecutive buffers overlap in this entry: a launch does not have the chip
        # to itself for its duration, so flops-per-launch / launch-duration against the whole-chip
        # peak would count the chip several times over.  What the chip did over the timed region
        # is flops-per-launch x launches / elapsed; that is `achieved`.  The per-launch figures
        # stay beside it (`per_launch`: kernel_us is what rocprofv3 --stats averages too).
        for ro in {id(roof): roof, id(roof_hbm): roof_hbm}.values():
            per_launch = {k: ro[k] for k in ("achieved", "frac", "kernel_us")}
            scale = ro["kernel_us"] * 1e-3 / (r["local_elapsed"] / args.steps * 1e3)
            ro["achieved"] = round(ro["achieved"] * scale, 1)
            ro["frac"] = round(ro["achieved"] / ro["peak"], 4)
            if "fp32_equivalent_tflops" in ro:
                ro["fp32_equivalent_tflops"] = round(ro["fp32_equivalent_tflops"] * scale, 1)
            ro["per_launch"] = per_launch
            ro["launches_in_flight"] = round(scale, 2)
        roof["note"] = ("achieved = algorithmic flops per launch x launches / timed region (launches of consecutive "
                        "buffers overlap); per_launch = flops per launch / average launch duration (hipEvents); "
                        "alone = the same kernel with the chip to itself (in-order pass of this run)")
    if api == "pipelined" and not args.no_extras:
        # the same kernel with the GPU to itself: the in-order entry, one stream, same K steps
        ra = time_workload(wl, device, seed, args.steps, args.warmup, None, api="inorder")
        alone, _ = rooflines(wl, ra, args.workload)
        inorder = dict(api="gsdr_demod_process_device, one stream",
                       value=round(args.steps * L / ra["local_elapsed"] / 1e6, 2), unit="Msamples/s per GPU",
                       ms_per_step=round(ra["local_elapsed"] / args.steps * 1e3, 5))
        if roof and alone:
            roof["alone"] = {k: alone[k] for k in ("achieved", "frac", "kernel_us")}

    line = {
        "metric": "IQ Msamples/s ingested (one synthetic 200 Msps stream per GPU)",
        "value": round(value, 2), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 5), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "dtype_note": ("complex64 in and out, fp32 accumulate; the DDC multiplies on the f16 MFMA as three products "
                       "of hi/lo-split fp32 operands (error per tone vs the fp64 oracle 1e-7..6e-6, bar 1e-5); "
                       "chirp: fp32 VALU with an exact integer phase"),
        "data": "synthetic",
        "config": {"workload": wl["name"], "key": args.workload, "buffer_len": L,
                   "rate": wl.get("rate", RATE), "tones_per_stream": r["n_tones"],
                   "streams": world, "parallelism": f"{world} independent stream(s), one per GPU, no collective",
                   "api": ("gsdr_demod_submit_device/gsdr_demod_wait, %d buffers outstanding" % PIPE_DEPTH)
                          if api == "pipelined" else "gsdr_demod_process_device, one stream"},
        "realtime_factor": round(value / world / (wl.get("rate", RATE) / 1e6), 3),
        "roofline": roof, "roofline_hbm": roof_hbm,
    }
    if inorder:
        line["inorder"] = inorder

    if world == 1 and rank == 0:
        if not args.no_extras:
            extras = {}
            for key in ("c3", "c4"):
                if key == args.workload:
                    continue
                e = time_workload(WORKLOADS[key], device, seed, steps=300, warmup=20,
                                  api="pipelined" if WORKLOADS[key]["kind"] == "direct" else "inorder")
                eb, ef = algorithmic(WORKLOADS[key], e["n_tones"])
                ekt = e["kernel_ms"] / max(e["kernel_launches"], 1) * 1e-3
                # rates over the timed region (launches overlap in the pipelined entry)
                per = e["elapsed"] / 300
                extras[key] = dict(msamples_per_s=round(300 * L / e["elapsed"] / 1e6, 2), api=e["api"],
                                   kernel=e["kernel"], kernel_us=round(ekt * 1e6, 2),
                                   us_per_buffer=round(per * 1e6, 2),
                                   hbm_gbs=round(eb * L / per / 1e9, 2),
                                   fp32_equivalent_tflops=round(ef * L / per / 1e12, 3))
                if e["kernel"].startswith("ddc_mfma"):
                    emf = 24.0 * WORKLOADS[key]["pf_average"] * e["n_tones"] * L / per / 1e12
                    extras[key]["f16_mfma_tflops"] = round(emf, 1)
                    extras[key]["f16_mfma_frac"] = round(emf / F16_MFMA_PEAK_TFLOPS, 4)
            best, probes = max_realtime_tones(device, seed)
            extras["max_realtime_tones_200Msps"] = dict(value=best, decim=1000, pf_average=4,
                                                        buffers=200, probes=probes)
            line["extras"] = extras
        if not args.no_cpu:
            line["cpu_baseline"] = cpu_baseline_oracle(wl, seed)
            nb = cpu_baseline_numpy(wl, seed)
            if nb:
                line["cpu_baseline_numpy"] = nb
    if rank == 0:
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
