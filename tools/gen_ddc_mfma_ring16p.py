#!/usr/bin/env python3
"""Generates gpu_sdr_amd/csrc/ddc_mfma_ring16p_gen.h: main loop of ddc_mfma_ring16p_kernel
(gfx950) -- the 16x16x32 ring loop of tools/gen_ddc_mfma_ring16.py fed with PRE-CONVERTED operands.

For launches of many rounds (thousands of tones) every workgroup of a row tile repeats the same
work on the same input: 3 scattered loads (64 cache lines each) and 28 conversion instructions per
wave and block, to produce the tone-independent A operand -- the timing-only ablations price the
loads at 11 % and the conversion at 3 % of such a launch.  Here a small pass (ddc_convert_kernel)
converts every (row tile, block) ONCE per buffer into the 8-KiB slot image the ring holds, and the
loop only copies images into its ring: two global_load_lds_dwordx4 per wave and block (1 KiB
contiguous each, no VGPR, no VALU), three blocks ahead of their use in a ring of four slots.

    python3 tools/gen_ddc_mfma_ring16p.py > gpu_sdr_amd/csrc/ddc_mfma_ring16p_gen.h
"""
import os
import sys

# timing-only builds (WRONG results): GEN_ABLATE=rot,prod,lds,gload,bar,bimg drops the rotation
# FMAs / the conversion arithmetic / the operand reads of the ring / the input loads / the barrier
# from the loop, the phasor-image loads from the prologue
ABLATE = set(filter(None, os.environ.get("GEN_ABLATE", "").split(",")))
KS = 4                     # k-steps per block (PK = 32)
SLOT = KS * 2 * 1024       # bytes of one ring slot
NSLOT = 4                  # ring slots: images arrive three blocks ahead

# ---- register map (TT = 1) -------------------------------------------------
VB = 12                    # v0..v11 stay with the compiler
ACC = (VB + 0, VB + 16)    # accumulators re, im
CA = (VB + 32, VB + 48)    # C set A: re, im
CB = (VB + 64, VB + 80)    # C set B
F0 = VB + 96               # operand buffers: fragment (k2, rh, sp) at v[F0 + 16*k2 + 8*rh + 4*sp : +3]
XA, XB, HV = VB + 128, VB + 132, VB + 136
HI4, LO4 = VB + 140, VB + 144
PA = VB + 152              # (Pr, Pi) of tone half 0, (Pr, Pi) of tone half 1: C set A
PB = VB + 156
# ring addresses, one set per iteration parity (R2)
ADDR = {"A": (VB + 161, VB + 162, VB + 163), "B": (VB + 164, VB + 165, VB + 166)}
FALT = VB + 168            # second buffer of the fragments (k-step 0, row half 1): hi 4, lo 4 (even: 64-bit aligned tuples)
V_LAST = VB + 175
NVGPR_CLOBBER = list(range(VB, V_LAST + 1))
NAGPR = 64

# private SGPRs
# scalar bases of the global loads, one set per iteration parity (same reason as
# the ring address registers: never rewrite what a queued memory instruction reads)
SB = {"A": dict(x=36, t=38, p=40), "B": dict(x=60, t=62, p=64), "C": dict(x=76, t=78, p=0)}   # C: prologue only
S_NLEFT, S_K, S_NHI1 = 42, 43, 44
S_RD, S_RDN, S_WR = 45, 46, 47
S_RD2 = 72     # slot of block b+2 (between RDN and WR)
S_M0 = 73      # M0 on entry
S_WRS = 74     # wave-uniform LDS base of this wave's two image pieces
S_SC = 48      # s[48:49] = (S, S)
S_T0, S_T1 = 50, 51
S_PH = 59      # wave slot parity (HW_ID wave_id & 1): GEN_PRIO=hw
S_XB = 52      # s[52:53] x base, block 0
S_BF = 56      # s[56:57] phasor-table images
S_PSTRIDE = 58
SGPR_CLOBBER = list(range(36, 80))


def vr(base, n=1):
    return f"v{base}" if n == 1 else f"v[{base}:{base + n - 1}]"


def ar(base, n=4):
    return f"a[{base}:{base + n - 1}]"


def frag(k2, rh, sp):
    return F0 + 16 * k2 + 8 * rh + 4 * sp


def bfrag(k2, th, c, sp):
    return ar(((((k2 * 2 + th) * 2 + c) * 2) + sp) * 4)


class Counters:
    """Outstanding LDS (lgkmcnt) and vector-memory (vmcnt) operations in issue order."""

    def __init__(self, out):
        self.out = out
        self.lgkm = []
        self.vm = []

    def issue_lgkm(self, tag):
        self.lgkm.append(tag)

    def issue_vm(self, tag):
        self.vm.append(tag)

    def _need(self, lst, tag, name):
        if tag not in lst:
            return
        i = len(lst) - 1 - lst[::-1].index(tag)
        n = len(lst) - 1 - i
        self.out.append(f"s_waitcnt {name}({n})")
        del lst[: i + 1]

    def need_lgkm(self, tag):
        self._need(self.lgkm, tag, "lgkmcnt")

    def need_vm(self, tag):
        self._need(self.vm, tag, "vmcnt")

    def drain_lgkm(self):
        self.out.append("s_waitcnt lgkmcnt(0)")
        self.lgkm = []


def rotate_ops(cset, p):
    """acc += P * C: 64 v_fma_f32 in four sweeps (an accumulator is read again 16
    instructions after it was written).  p = (Pr, Pi)."""
    cr, ci = cset
    ops = []
    for term in range(4):
        for i in range(16):
            th = (i >> 2) & 1            # register i belongs to tile (rh, th) = (i >> 3, (i >> 2) & 1)
            pr, pi = vr(p + 2 * th), vr(p + 2 * th + 1)
            a_r, a_i = vr(ACC[0] + i), vr(ACC[1] + i)
            c_r, c_i = vr(cr + i), vr(ci + i)
            if term == 0:
                ops.append(f"v_fma_f32 {a_r}, {pr}, {c_r}, {a_r}")
            elif term == 1:
                ops.append(f"v_fma_f32 {a_i}, {pr}, {c_i}, {a_i}")
            elif term == 2:
                ops.append(f"v_fma_f32 {a_r}, -{pi}, {c_i}, {a_r}")
            else:
                ops.append(f"v_fma_f32 {a_i}, {pi}, {c_r}, {a_i}")
    return ops


# MFMA order inside k-step k2: (row half, hi|lo of the A fragment) major.  Entries:
# (rh, sp_a, th, c, sp_b); the first 8 of a row half use its hi fragment (products hi*hi, hi*lo),
# the next 4 its lo fragment (lo*hi).
def kstep_order():
    seq = []
    for rh in range(2):
        for sp_b in range(2):
            for th in range(2):
                for c in range(2):
                    seq.append((rh, 0, th, c, sp_b))
        for th in range(2):
            for c in range(2):
                seq.append((rh, 1, th, c, 0))
    return seq


ORDER = kstep_order()
assert len(ORDER) == 24


def first_use(k2, rh, sp):
    """gap (0..47) of the first MFMA that reads fragment (k2, rh, sp)"""
    for m, e in enumerate(ORDER):
        if e[0] == rh and e[1] == sp:
            return 24 * k2 + m
    raise AssertionError


def last_use(k2, rh, sp):
    return max(24 * k2 + m for m, e in enumerate(ORDER) if e[0] == rh and e[1] == sp)


def frag_for(label, k2, rh, sp):
    """Operand buffer of fragment (k2, rh, sp) in an iteration of parity `label`.  The two
    fragments of (k-step 0, row half 1) are used last in their k-step (gaps 12..23) and would
    have to be re-read for the next block at the very end of this one; they alternate between
    two buffers instead (no register is rewritten under R1's distance, and every read of a
    block has been issued ten MFMAs before its end)."""
    if k2 == 0 and rh == 1 and label == "B":
        return FALT + 4 * sp
    return frag(k2, rh, sp)


def image_pointer(par):
    """SALU: s[SB[par].x] = image base + 8192 * min(S_K, nhi-1), then S_K += 1 (parity set `par`:
    a scalar base is never rewritten under a queued load that reads it, rule R2)."""
    S_X = SB[par]["x"]
    return [
        f"s_min_u32 s{S_T0}, s{S_K}, s{S_NHI1}",
        f"s_lshl_b32 s{S_T1}, s{S_T0}, 13",
        f"s_add_u32 s{S_X}, s{S_XB}, s{S_T1}",
        f"s_addc_u32 s{S_X + 1}, s{S_XB + 1}, 0",
        f"s_add_u32 s{S_K}, s{S_K}, 1",
    ]


def dma_ops(par, slot_sreg, which):
    """One LDS-DMA piece of the image s[SB[par].x] into ring slot `slot_sreg`: which = 0 the hi
    image of this wave's k-step (1 KiB), 1 the lo image.  M0 carries the wave-uniform LDS address;
    it is written right in front of its only reader and not again for many MFMAs."""
    S_X = SB[par]["x"]
    return [
        f"s_add_u32 m0, s{slot_sreg}, s{S_WRS}" if which == 0 else f"s_add_u32 m0, m0, 1024",
        "s_nop 0",
        f"global_load_lds_dwordx4 {'%[io_hi]' if which == 0 else '%[io_lo]'}, s[{S_X}:{S_X + 1}]",
    ]


def iteration(cnt, out, cur, prev, p_cur, p_prev, label):
    """One block of 32 samples: 48 MFMAs.  cur/prev: C sets; p_cur/p_prev: P registers (4 each).
    Block b computes from ring slot RD, prefetches block b+1's first fragments from RDN and
    starts the copy of block b+3's image into slot WR (free since the barrier that ended b-1)."""
    out.append(f"; ---- block iteration, C set {label}")
    rot = rotate_ops(prev, p_prev)
    other = "B" if label == "A" else "A"
    S_P, N_P = SB[label]["p"], SB[other]["p"]
    NG = 48
    gaps = {g: [] for g in range(NG)}
    V_RD, V_RDN, _ = ADDR[label]
    N_RD, N_RDN, _ = ADDR[other]

    def read_frag(g, slot_reg, k2, rh, sp, into):
        off = k2 * 4096 + sp * 1024 + rh * 256
        gaps[g].append(("lds", f"ds_read_b128 {vr(into, 4)}, {vr(slot_reg)} offset:{off}", f"f{k2}{rh}{sp}"))

    for (rh, sp, g) in ((0, 0, 8), (0, 1, 12), (1, 0, 20), (1, 1, 24)):
        assert g + NG - last_use(1, rh, sp) >= 24 and g < first_use(1, rh, sp) - 8
        read_frag(g, V_RD, 1, rh, sp, frag_for(label, 1, rh, sp))
    for (rh, sp, g) in ((0, 0, 32), (1, 0, 34), (0, 1, 36), (1, 1, 38)):
        if rh == 0:
            assert g - last_use(0, rh, sp) >= 24
        read_frag(g, V_RDN, 0, rh, sp, frag_for(other, 0, rh, sp))
    # P of this block (used by the next iteration's rotation): tone halves 0 and 1
    gaps[0].append(("vm", f"global_load_dwordx2 {vr(p_cur, 2)}, %[po], s[{S_P}:{S_P + 1}]", "pr" + label))
    gaps[0].append(("vm", f"global_load_dwordx2 {vr(p_cur + 2, 2)}, %[po], s[{S_P}:{S_P + 1}] offset:128", "p" + label))
    gaps[1].append(("salu", f"s_add_u32 s{N_P}, s{S_P}, s{S_PSTRIDE}", None))
    gaps[1].append(("salu", f"s_addc_u32 s{N_P + 1}, s{S_P + 1}, 0", None))
    # image of block b+3 -> slot WR: pointer (this parity's set) in gaps 2..4, the two pieces at
    # gaps 6 and 18 (M0 is rewritten 12 MFMAs after its first reader was issued)
    for i, sx in enumerate(image_pointer(label)):
        gaps[2 + i // 2].append(("salu", sx, None))
    for i, tx in enumerate(dma_ops(label, S_WR, 0)):
        gaps[6].append(("dma" if tx.startswith("global") else "salu", tx, "dh" + label))
    for i, tx in enumerate(dma_ops(label, S_WR, 1)):
        gaps[18].append(("dma" if tx.startswith("global") else "salu", tx, "dl" + label))
    # P*C of the previous block: 64 plain FMAs, gaps 4..47
    for k, op in enumerate(rot):
        gaps[4 + (k * 44) // len(rot)].append(("rot", op, None))
    # ring slot rotation (four slots) and the read addresses of the next iteration, once every
    # ring access of this iteration has been issued (gap 38)
    gaps[40].append(("salu", f"s_mov_b32 s{S_T0}, s{S_RD}", None))
    gaps[40].append(("salu", f"s_mov_b32 s{S_RD}, s{S_RDN}", None))
    gaps[41].append(("salu", f"s_mov_b32 s{S_RDN}, s{S_RD2}", None))
    gaps[41].append(("salu", f"s_mov_b32 s{S_RD2}, s{S_WR}", None))
    gaps[42].append(("salu", f"s_mov_b32 s{S_WR}, s{S_T0}", None))
    gaps[43].append(("addr", f"v_add_u32 {vr(N_RD)}, s{S_RD}, %[lane16]", None))
    gaps[44].append(("addr", f"v_add_u32 {vr(N_RDN)}, s{S_RDN}, %[lane16]", None))

    first_rot = True
    for g in range(NG):
        k2, m = divmod(g, 24)
        rh, sp_a, th, c, sp_b = ORDER[m]
        if first_use(k2, rh, sp_a) == g:
            cnt.need_lgkm(f"f{k2}{rh}{sp_a}")
        dst = (cur[0] if c == 0 else cur[1]) + 4 * (2 * rh + th)
        first = k2 == 0 and sp_a == 0 and sp_b == 0
        src_c = "0" if first else vr(dst, 4)
        if "mfma" not in ABLATE:
            out.append(f"v_mfma_f32_16x16x32_f16 {vr(dst, 4)}, {vr(frag_for(label, k2, rh, sp_a), 4)}, "
                       f"{bfrag(k2, th, c, sp_b)}, {src_c}")
        for kind, text, tag in gaps[g]:
            if kind == "lds":
                if "lds" not in ABLATE:
                    out.append(text)
                    cnt.issue_lgkm(tag)
            elif kind == "vm":
                out.append(text)
                cnt.issue_vm(tag)
            elif kind == "dma":
                if "gload" not in ABLATE:
                    out.append(text)
                    cnt.issue_vm(tag)
            elif kind == "rot":
                if first_rot:
                    cnt.need_vm("pB" if label == "A" else "pA")
                    first_rot = False
                if "rot" not in ABLATE:
                    out.append(text)
            else:
                out.append(text)
    # the image this wave started one iteration ago (block b+2's) must have landed before the
    # barrier publishes it: block b+1 prefetches from it
    cnt.need_vm("dl" + other)
    cnt.drain_lgkm()
    if "bar" not in ABLATE:
        out.append("s_barrier")


def generate():
    out = []
    cnt = Counters(out)
    o = out.append
    o("; ===== prologue =====")
    o(f"s_mov_b32 s{S_M0}, m0")
    o(f"s_mov_b32 s{S_XB}, %[ib_lo]")          # image base of this row tile, block 0
    o(f"s_mov_b32 s{S_XB + 1}, %[ib_hi]")
    o(f"s_mov_b32 s{S_WRS}, %[wrs]")
    o(f"s_mov_b32 s{SB['A']['p']}, %[pp_lo]")
    o(f"s_mov_b32 s{SB['A']['p'] + 1}, %[pp_hi]")
    o(f"s_mov_b32 s{S_BF}, %[bf_lo]")
    o(f"s_mov_b32 s{S_BF + 1}, %[bf_hi]")
    o(f"s_mov_b32 s{S_PSTRIDE}, %[pstride]")
    o(f"s_mov_b32 s{S_NLEFT}, %[nhi]")
    o(f"s_add_u32 s{S_NHI1}, %[nhi], -1")
    o(f"s_mov_b32 s{S_K}, 0")
    o(f"s_mov_b32 s{S_RD}, 0")
    o(f"s_mov_b32 s{S_RDN}, {SLOT}")
    o(f"s_mov_b32 s{S_RD2}, {2 * SLOT}")
    o(f"s_mov_b32 s{S_WR}, {3 * SLOT}")
    o("s_nop 4")
    BF = [S_BF, 66, 68, 70]
    for j in range(1, 4):
        o(f"s_add_u32 s{BF[j]}, s{S_BF}, {4096 * j}")
        o(f"s_addc_u32 s{BF[j] + 1}, s{S_BF + 1}, 0")
    o("s_nop 4")
    o("s_cmp_eq_u32 %[first], 0")
    o("s_cbranch_scc1 4f")
    for f in range(16):
        b = BF[f // 4]
        if "bimg" not in ABLATE:
            o(f"global_load_dwordx4 {ar(4 * f)}, %[bo], s[{b}:{b + 1}] offset:{(f % 4) * 1024}")
    o("4:")
    for base in (CB[0], CB[1], ACC[0], ACC[1]):
        for i in range(16):
            o(f"v_mov_b32 {vr(base + i)}, 0")
    for i in range(4):
        o(f"v_mov_b32 {vr(PB + i)}, 0")
    # images of blocks 0, 1, 2 into slots 0, 1, 2 (pointer sets A, B, C: one per load pair)
    for par, slot in (("A", S_RD), ("B", S_RDN), ("C", S_RD2)):
        out.extend(image_pointer(par))
        o("s_nop 4")
        out.extend(dma_ops(par, slot, 0))
        o("s_nop 4")
        out.extend(dma_ops(par, slot, 1))
        o("s_nop 4")
    V_RD, V_RDN, _ = ADDR["A"]
    o(f"v_add_u32 {vr(V_RD)}, s{S_RD}, %[lane16]")
    o(f"v_add_u32 {vr(V_RDN)}, s{S_RDN}, %[lane16]")
    o("s_waitcnt vmcnt(0)")          # the phasor images and the three slot images
    o("s_barrier")
    for rh in range(2):
        for sp in range(2):
            o(f"ds_read_b128 {vr(frag_for('A', 0, rh, sp), 4)}, {vr(V_RD)} offset:{sp * 1024 + rh * 256}")
    o("s_waitcnt lgkmcnt(0)")
    cnt.lgkm = []

    def trip(out_, cnt_):
        iteration(cnt_, out_, CA, CB, PA, PB, "A")
        out_.append(f"s_sub_u32 s{S_NLEFT}, s{S_NLEFT}, 1")
        out_.append(f"s_cmp_eq_u32 s{S_NLEFT}, 0")
        out_.append("s_cbranch_scc1 2f")
        iteration(cnt_, out_, CB, CA, PB, PA, "B")
        out_.append(f"s_sub_u32 s{S_NLEFT}, s{S_NLEFT}, 1")
        out_.append(f"s_cmp_lg_u32 s{S_NLEFT}, 0")
        out_.append("s_cbranch_scc1 1b")

    state = []
    for _ in range(4):
        probe = Counters([])
        probe.vm = list(state)
        trip(probe.out, probe)
        assert probe.lgkm == []
        if probe.vm == state:
            break
        state = list(probe.vm)
    else:
        raise AssertionError("no steady state")
    cnt.vm = list(state)
    o(f"; ===== main loop, two blocks per trip; vm at the top: {state}")
    o("1:")
    trip(out, cnt)
    assert cnt.lgkm == [] and cnt.vm == state, (cnt.lgkm, cnt.vm, state)
    o("s_waitcnt vmcnt(0)")
    o("s_nop 15")
    o("s_nop 15")
    out.extend(rotate_ops(CB, PB))
    o("s_branch 3f")
    o("2:")
    o("s_waitcnt vmcnt(0)")
    o("s_nop 15")
    o("s_nop 15")
    out.extend(rotate_ops(CA, PA))
    o("3:")
    # every image copy has landed (vmcnt(0) above) and every wave is past the last barrier: the
    # ring is idle, the accumulators go to the C++ epilogue through it
    o("s_barrier")
    for q in range(8):
        base = (ACC[0] if q < 4 else ACC[1]) + 4 * (q & 3)
        o(f"ds_write_b128 %[accaddr], {vr(base, 4)} offset:{q * 1024}")
    o("s_waitcnt lgkmcnt(0)")
    o(f"s_mov_b32 m0, s{S_M0}")
    return out


def main():
    lines = generate()
    PFX = "GSDR_MFMA_RING16P"
    print("// GENERATED by tools/gen_ddc_mfma_ring16p.py -- do not edit.")
    print("// Main loop of ddc_mfma_ring16p_kernel (pre-converted operands by LDS-DMA, v_mfma_f32_16x16x32_f16): see the generator for the schedule and register map.")
    print("#pragma once")
    print(f"#define {PFX}_VB {VB}")
    print(f"#define {PFX}_BYTES {NSLOT * SLOT}")
    print(f"#define {PFX}_TEXT \\")
    for ln in lines:
        if ln.startswith(";"):
            continue
        print(f'    "{ln}\\n\\t" \\')
    print('    ""')
    clob = [f'"v{i}"' for i in NVGPR_CLOBBER] + [f'"a{i}"' for i in range(NAGPR)] + \
           [f'"s{i}"' for i in SGPR_CLOBBER] + ['"vcc"', '"scc"', '"memory"']
    print(f"#define {PFX}_CLOBBERS \\")
    for i in range(0, len(clob), 12):
        tail = ", \\" if i + 12 < len(clob) else ""
        print("    " + ", ".join(clob[i:i + 12]) + tail)
    n_mfma = sum(1 for l in lines if l.startswith("v_mfma"))
    print(f"// {len(lines)} lines, {n_mfma} MFMAs, VGPRs v{VB}..v{V_LAST}, AGPRs a0..a{NAGPR - 1}")


if __name__ == "__main__":
    main()
