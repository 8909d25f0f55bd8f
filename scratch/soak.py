"""Soak: the overlapped entry against the in-order entry, bit for bit, over thousands of buffers of
the bench workloads.  With the default per-launch kernel choice this also compares kernels with
each other: C3 in order runs the 8-wave kernel, overlapped the pre-converted path; 8192 tones run
the pre-converted path in both.  Timing-dependent hazards (rules R1-R3) show up here or nowhere."""
import sys, time
sys.path.insert(0, ".")
import torch
import bench

dev = torch.device("cuda:0")
eng = bench.HipEngine()
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
total_bad = 0
for key, steps, tones in (("c2", 6000, None), ("c3", 2500, None), ("pfb", 1500, None), ("c3", 600, 8192)):
    steps = int(steps * scale)
    wl = bench.WORKLOADS[key]
    da, bufs, outs_a, _ = eng.build(wl, dev, 11, n_tones=tones)
    db, _, outs, _ = eng.build(wl, dev, 11, n_tones=tones)
    out_a = outs_a[0]
    for i, x in enumerate(bufs):          # very different loudness from buffer to buffer
        x *= (1e-3, 1.0, 40.0, 1.0, 1e-2, 5.0, 1.0, 0.3)[i % 8]
    torch.cuda.synchronize()
    bad = 0
    t0 = time.time()
    pend = []
    for k in range(steps):
        n = da.process_device(bufs[k % 8], out_a)
        full = out_a[:n].clone() if k % 50 == 0 else None
        ref_fp = torch.view_as_real(out_a[:n]).view(torch.int32).sum(dtype=torch.int64)   # fingerprint of every buffer
        if len(pend) == 3:
            kk, fp, fl = pend.pop(0)
            m = db.wait()
            o = outs[kk % 3][:m]
            got = torch.view_as_real(o).view(torch.int32).sum(dtype=torch.int64)
            if got.item() != fp.item() or (fl is not None and not torch.equal(o, fl)):
                bad += 1
                print(key, tones, "MISMATCH at buffer", kk, flush=True)
        db.submit_device(bufs[k % 8], outs[k % 3])
        pend.append((k, ref_fp, full))
    while pend:
        kk, fp, fl = pend.pop(0)
        m = db.wait()
        o = outs[kk % 3][:m]
        got = torch.view_as_real(o).view(torch.int32).sum(dtype=torch.int64)
        if got.item() != fp.item() or (fl is not None and not torch.equal(o, fl)):
            bad += 1
            print(key, tones, "MISMATCH at buffer", kk, flush=True)
    torch.cuda.synchronize()
    print("%s%s: %d buffers, %d mismatches, %.1f s; in-order kernel %s, overlapped %s" % (
        key, "" if tones is None else " (%d tones)" % tones, steps, bad, time.time() - t0, da.kernel_name, db.kernel_name), flush=True)
    total_bad += bad
    da.close(); db.close()
sys.exit(1 if total_bad else 0)
