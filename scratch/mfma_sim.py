"""Numerics rehearsal of the split-fp16 MFMA formulation of the DDC (CPU, numpy).
y[n,G] = rot[n,G] * sum_hi P_n[hi] * sum_lo (S h[t] x[(G-F+1)M+t]) * B_n[lo],  t = hi*PK+lo
with both MFMA operands split into fp16 hi+lo and three products kept."""
import sys
import numpy as np

def split16(v):
    hi = v.astype(np.float16)
    lo = (v - hi.astype(np.float32)).astype(np.float16)
    return hi.astype(np.float32), lo.astype(np.float32)

def run(N, M, F, nblk, rate, PK=32, amp=1.0, sigma=1e-3, seed=0, weak=None):
    rng = np.random.default_rng(seed)
    L = nblk * M
    f = rng.integers(-rate // 2, rate // 2, size=N)
    fm = np.mod(f, rate).astype(np.int64)
    n = np.arange(-(F - 1) * M, L)
    a = np.full(N, amp / N)
    if weak is not None:
        a[0] *= weak
    x = np.zeros(len(n), np.complex128)
    for k in range(N):
        x += a[k] * np.exp(2j * np.pi * ((fm[k] * (n % rate)) % rate) / rate + 1j * k)
    x += sigma * (rng.standard_normal(len(n)) + 1j * rng.standard_normal(len(n)))
    x = x.astype(np.complex64)
    MF = M * F
    h = (np.sinc((np.arange(MF) - MF // 2) * 0.75 / M) * np.hanning(MF)).astype(np.float32)
    h /= h.sum()
    # exact reference in fp64
    s_abs = n.astype(np.int64)  # idx0 = 0 at n = 0
    ref = np.zeros((nblk, N), np.complex128)
    for k in range(N):
        z = x.astype(np.complex128) * np.exp(-2j * np.pi * ((fm[k] * np.mod(s_abs, rate)) % rate) / rate)
        for G in range(nblk):
            ref[G, k] = np.dot(h.astype(np.float64), z[G * M: G * M + MF])
    # fp32 direct evaluation, for scale (what a straightforward fp32 kernel gives)
    # split-fp16 emulation
    nk = (MF + 7) // 8 * 8
    hp = np.zeros(nk, np.float32); hp[:MF] = h
    xmax = np.abs(np.concatenate([x.real, x.imag])).max()
    S = np.float32(2.0 ** (13 - np.ceil(np.log2(np.abs(h).max() * xmax))))
    out = np.zeros((nblk, N), np.complex64)
    nhi = (nk + PK - 1) // PK
    lo_idx = np.arange(PK)
    Bt = np.exp(-2j * np.pi * ((fm[None, :] * lo_idx[:, None]) % rate) / rate)          # [PK, N]
    Br_h, Br_l = split16(Bt.real.astype(np.float32)); Bi_h, Bi_l = split16(Bt.imag.astype(np.float32))
    P = np.exp(-2j * np.pi * ((fm[None, :] * (np.arange(nhi) * PK)[:, None]) % rate) / rate).astype(np.complex64)
    xp = np.concatenate([x, np.zeros(nk, np.complex64)])
    Gs = np.arange(nblk)
    acc = np.zeros((nblk, N), np.complex64)
    for hi in range(nhi):
        t = hi * PK + lo_idx
        t = t[t < nk]
        idx = Gs[:, None] * M + t[None, :]
        b = (xp[idx] * (hp[t] * S)[None, :]).astype(np.complex64)          # [G, lo]
        br_h, br_l = split16(b.real.copy()); bi_h, bi_l = split16(b.imag.copy())
        k = len(t)
        def mm(u, v):
            return (u @ v[:k]).astype(np.float32)
        Cr = mm(br_h, Br_h) + mm(br_h, Br_l) + mm(br_l, Br_h) - (mm(bi_h, Bi_h) + mm(bi_h, Bi_l) + mm(bi_l, Bi_h))
        Ci = mm(br_h, Bi_h) + mm(br_h, Bi_l) + mm(br_l, Bi_h) + (mm(bi_h, Br_h) + mm(bi_h, Br_l) + mm(bi_l, Br_h))
        C = (Cr + 1j * Ci).astype(np.complex64)
        acc = (acc + P[hi][None, :] * C).astype(np.complex64)
    rot = np.exp(-2j * np.pi * ((fm[None, :] * np.mod((Gs[:, None] - (F - 1)) * M, rate)) % rate) / rate)
    out = (acc * rot.astype(np.complex64) / S).astype(np.complex64)
    err = np.linalg.norm(out - ref, axis=0) / np.linalg.norm(ref, axis=0)
    return err

if __name__ == "__main__":
    for cfg in [dict(N=8, M=50, F=4, nblk=64, rate=1_000_000),
                dict(N=64, M=100, F=4, nblk=64, rate=200_000_000),
                dict(N=64, M=100, F=4, nblk=64, rate=200_000_000, amp=1e-4, sigma=1e-7),
                dict(N=64, M=100, F=4, nblk=64, rate=200_000_000, weak=1e-2),
                dict(N=256, M=1000, F=4, nblk=32, rate=200_000_000),
                dict(N=64, M=10, F=1, nblk=128, rate=1_000_000),
                dict(N=64, M=1000, F=4, nblk=32, rate=200_000_000, sigma=0.2)]:
        e = run(**cfg)
        print(cfg, "max rel err %.2e  median %.2e" % (e.max(), np.median(e)), flush=True)
