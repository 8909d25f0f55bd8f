#!/usr/bin/env python3
"""Generates gpu_sdr_amd/csrc/ddc_mfma_ring_gen.h: the main loop of ddc_mfma_ring_kernel
(gfx950) as one inline-asm block with a fixed register map; the four waves of a
workgroup SHARE the converted A operand through an LDS ring.

Why assembly: the loop's correctness depends on register reuse distances and
operand forms the compiler does not know about (see "Rules" below, all measured
on MI355X with scratch/mfma_probe.py), and its speed on an exact MFMA/VALU interleave.

Work of one wave: 32 output rows x 32 tones; every phasor block of 32 samples is
4 k-steps x 6 MFMAs (v_mfma_f32_32x32x16_f16: Cr/Ci x {hi*Bhi, hi*Blo, lo*Bhi}).
Each wave converts one k-step of every block (x * taps * S -> fp16 hi/lo, 28 plain
VALU instructions) into ring slot (b+2)%3 and reads all four k-steps of block b back
as MFMA operands: a quarter of the global loads and of the conversion work of a loop
in which every wave converts for itself (round 1 measured that one: slower), at the
price of one s_barrier per block and ten LDS instructions per wave and block.  In the
MFMAs' shadow the wave also applies P*C of the previous block on the VALU (C alternates
between two register sets, hence the 2x unroll).  All of that is plain (non-packed)
FP32: tools/ubench_mfma.hip measures that v_pk_fma_f32 / v_pk_mul_f32 do not overlap
with the f16 MFMA at all (+10 cycles each), while v_fma_f32, v_mul_f32,
v_cvt_pk_f16_f32 and v_fma_mix_f32 nearly vanish in its shadow at two waves per SIMD
(6 per MFMA: +7 %).

Rules this file and its descendants (gen_ddc_mfma_ring16*.py) keep -- each one cost a
debugging session; tools/check_asm_rules.py re-checks R1..R3 on the emitted text:
  R1  A register that an MFMA reads as A/B operand is not rewritten before twelve
      further MFMAs have been issued (four operand buffers).  With less distance
      rows 16..31 of the MFMA came out computed from the new contents.
  R2  Address, data and scalar-base registers of a memory instruction stay unchanged
      for two k-steps after it was issued: with two workgroups on a CU they are read
      late (registers rewritten ~4 MFMAs later gave half of the lanes the new address).
      The ring address registers and the scalar load bases exist once per iteration parity.
  R3  No v_pk_*_f32 that broadcasts the HIGH half of a pair written by the packed
      instruction in front of it (op_sel:[0,1] / [1,..]): it returned a stale value in
      lanes 48..63 now and then while another wave of the SIMD ran this loop
      (isolated reproducer: tools/ubench_pk_hazard.hip).  The loop has no packed FP32
      at all (see above); the kernel around it is compiled without packed FP32.
  R4  s_waitcnt values come from a model of the counters (class Counters).

    python3 tools/gen_ddc_mfma_ring.py > gpu_sdr_amd/csrc/ddc_mfma_ring_gen.h
"""
import os
import sys

# timing-only builds (WRONG results): GEN_ABLATE=rot,prod,lds,gload,bar,bimg drops the rotation
# FMAs / the conversion arithmetic / the operand reads of the ring / the input loads / the barrier
# from the loop, the phasor-image loads from the prologue
ABLATE = set(filter(None, os.environ.get("GEN_ABLATE", "").split(",")))
KS = 4                     # k-steps per block (PK = 32)
SLOT = KS * 2 * 1024       # bytes of one ring slot

# ---- register map (TT = 1) -------------------------------------------------
VB = 12                    # v0..v11 stay with the compiler
ACC = (VB + 0, VB + 16)    # accumulators re, im
CA = (VB + 32, VB + 48)    # C set A: re, im
CB = (VB + 64, VB + 80)    # C set B
F0 = VB + 96               # operand buffers: step s hi v[F0+8s:+3], lo v[F0+8s+4:+3]
XA, XB, HV = VB + 128, VB + 132, VB + 136
HI4, LO4 = VB + 140, VB + 144
HS = VB + 148              # scaled taps of the k-step being converted (4)
PA = (VB + 152, VB + 153)  # (Pr, Pi) of C set A
PB = (VB + 154, VB + 155)
V_SC = VB + 156            # S
# ring addresses, one set per iteration parity (R2)
ADDR = {"A": (VB + 157, VB + 158, VB + 159), "B": (VB + 160, VB + 161, VB + 162)}
V_LAST = VB + 162
NVGPR_CLOBBER = list(range(VB, V_LAST + 1))
NAGPR = 64

# private SGPRs
# scalar bases of the global loads, one set per iteration parity (same reason as
# the ring address registers: never rewrite what a queued memory instruction reads)
SB = {"A": dict(x=36, t=38, p=40), "B": dict(x=60, t=62, p=64), "C": dict(x=76, t=78, p=0)}   # C: prologue only
S_NLEFT, S_K, S_NHI1 = 42, 43, 44
S_RD, S_RDN, S_WR = 45, 46, 47
S_SC = 48      # s[48:49] = (S, S)
S_T0, S_T1 = 50, 51
S_XB = 52      # s[52:53] x base, block 0
S_TB = 54      # s[54:55] taps base, block 0
S_BF = 56      # s[56:57] phasor-table images
S_PSTRIDE = 58
SGPR_CLOBBER = list(range(36, 80))


def vr(base, n=1):
    return f"v{base}" if n == 1 else f"v[{base}:{base + n - 1}]"


def ar(base, n=4):
    return f"a[{base}:{base + n - 1}]"


def bfrag(ks, c, sp):
    return ar((((ks * 2 + c) * 2) + sp) * 4)


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
    pr, pi = vr(p[0]), vr(p[1])
    ops = []
    for term in range(4):
        for i in range(16):
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


def produce_ops(xa=None, xb=None, hv=None):
    """x (4 complex samples in XA, XB) * taps (HV) * S -> fp16 hi (HI4) and lo (LO4):
    28 plain VALU instructions (no packed FP32, see the rules above)."""
    xa, xb, hv = (XA if xa is None else xa), (XB if xb is None else xb), (HV if hv is None else hv)
    ops = []
    for j in range(4):
        ops.append(f"v_mul_f32 {vr(HS + j)}, {vr(hv + j)}, {vr(V_SC)}")
    xs = [xa, xa + 2, xb, xb + 2]
    for j in range(4):
        ops.append(f"v_mul_legacy_f32 {vr(xs[j])}, {vr(xs[j])}, {vr(HS + j)}")   # 0 * anything = 0: zero-padded taps meet samples outside the window
        ops.append(f"v_mul_legacy_f32 {vr(xs[j] + 1)}, {vr(xs[j] + 1)}, {vr(HS + j)}")
    for j in range(4):
        ops.append(f"v_cvt_pk_f16_f32 {vr(HI4 + j)}, {vr(xs[j])}, {vr(xs[j] + 1)}")
    for j in range(4):
        ops.append(f"v_fma_mix_f32 {vr(xs[j])}, {vr(xs[j])}, 1.0, -{vr(HI4 + j)} op_sel_hi:[0,0,1]")
        ops.append(f"v_fma_mix_f32 {vr(xs[j] + 1)}, {vr(xs[j] + 1)}, 1.0, -{vr(HI4 + j)} op_sel:[0,0,1] op_sel_hi:[0,0,1]")
    for j in range(4):
        ops.append(f"v_cvt_pk_f16_f32 {vr(LO4 + j)}, {vr(xs[j])}, {vr(xs[j] + 1)}")
    return ops


def gload_ops(cnt, out, par, xa=None, xb=None, hv=None, off=None):
    S_X, S_T = SB[par]["x"], SB[par]["t"]
    xa, xb, hv = (XA if xa is None else xa), (XB if xb is None else xb), (HV if hv is None else hv)
    out.append(f"global_load_dwordx4 {vr(hv, 4)}, %[to], s[{S_T}:{S_T + 1}]")
    cnt.issue_vm("hv")
    out.append(f"global_load_dwordx4 {vr(xa, 4)}, %[xo], s[{S_X}:{S_X + 1}]")
    cnt.issue_vm("xa")
    out.append(f"global_load_dwordx4 {vr(xb, 4)}, %[xo], s[{S_X}:{S_X + 1}] offset:16")
    cnt.issue_vm("xb")


def advance_load_pointers(par):
    """SALU: pointers (parity set `par`) of block min(S_K, nhi-1), then S_K += 1."""
    S_X, S_T = SB[par]["x"], SB[par]["t"]
    return [
        f"s_min_u32 s{S_T0}, s{S_K}, s{S_NHI1}",
        f"s_lshl_b32 s{S_T1}, s{S_T0}, 8",
        f"s_add_u32 s{S_X}, s{S_XB}, s{S_T1}",
        f"s_addc_u32 s{S_X + 1}, s{S_XB + 1}, 0",
        f"s_lshl_b32 s{S_T1}, s{S_T0}, 7",
        f"s_add_u32 s{S_T}, s{S_TB}, s{S_T1}",
        f"s_addc_u32 s{S_T + 1}, s{S_TB + 1}, 0",
        f"s_add_u32 s{S_K}, s{S_K}, 1",
    ]


def mfma(cset, ks, m):
    cr, ci = cset
    fh, fl = F0 + 8 * ks, F0 + 8 * ks + 4
    a = fh if m < 4 else fl
    c = m & 1
    sp = 1 if m in (2, 3) else 0
    dst = cr if c == 0 else ci
    src_c = "0" if (ks == 0 and m < 2) else vr(dst, 16)
    return f"v_mfma_f32_32x32x16_f16 {vr(dst, 16)}, {vr(a, 4)}, {bfrag(ks, c, sp)}, {src_c}"


def iteration(cnt, out, cur, prev, p_cur, p_prev, label):
    """One block.  cur/prev: C sets; p_cur/p_prev: P registers."""
    out.append(f"; ---- block iteration, C set {label}")
    rot = rotate_ops(prev, p_prev)
    prod = produce_ops()
    other = "B" if label == "A" else "A"
    salu = advance_load_pointers(other)          # for the next iteration's loads
    S_P, N_P = SB[label]["p"], SB[other]["p"]
    gaps = {g: [] for g in range(24)}

    def read_step(g, slot_reg, ks_src, buf):
        fh, fl = F0 + 8 * buf, F0 + 8 * buf + 4
        gaps[g].append(("lds", f"ds_read_b128 {vr(fh, 4)}, {vr(slot_reg)} offset:{ks_src * 2048}", f"f{buf}h"))
        gaps[g].append(("lds", f"ds_read_b128 {vr(fl, 4)}, {vr(slot_reg)} offset:{ks_src * 2048 + 1024}", f"f{buf}l"))

    V_RD, V_RDN, V_WR = ADDR[label]
    N_RD, N_RDN, N_WR = ADDR["B" if label == "A" else "A"]
    # operand reads: step ks+1 at the first MFMA of step ks; next block's step 0 at step 3
    read_step(0, V_RD, 1, 1)
    read_step(6, V_RD, 2, 2)
    read_step(12, V_RD, 3, 3)
    read_step(18, V_RDN, 0, 0)
    # P of this block (used next iteration), pointer advance afterwards
    gaps[0].append(("vm", f"global_load_dword {vr(p_cur[0])}, %[po], s[{S_P}:{S_P + 1}]", "pr" + label))
    gaps[0].append(("vm", f"global_load_dword {vr(p_cur[1])}, %[po], s[{S_P}:{S_P + 1}] offset:4", "p" + label))
    gaps[1].append(("salu", f"s_add_u32 s{N_P}, s{S_P}, s{S_PSTRIDE}", None))
    gaps[1].append(("salu", f"s_addc_u32 s{N_P + 1}, s{S_P + 1}, 0", None))
    for i, s in enumerate(salu):
        gaps[1 + i // 3].append(("salu", s, None))
    # P*C of the previous block: gaps 3..9 (3 each) and 18..23 (2 each)
    ri = 0
    for g in list(range(3, 10)):
        for _ in range(5):
            gaps[g].append(("rot", rot[ri], None))
            ri += 1
    per = -(-(len(rot) - ri) // 6)
    for g in range(18, 24):
        for _ in range(per):
            if ri < len(rot):
                gaps[g].append(("rot", rot[ri], None))
                ri += 1
    assert ri == len(rot), ri
    # conversion of block b+2: gaps 10..17
    pi = 0
    for g in range(10, 18):
        for _ in range(4):
            if pi < len(prod):
                gaps[g].append(("prod", prod[pi], None))
                pi += 1
    assert pi == len(prod), (pi, len(prod))

    gaps[17].append(("ldsw", f"ds_write_b128 {vr(V_WR)}, {vr(HI4, 4)}", "wh"))
    gaps[17].append(("ldsw", f"ds_write_b128 {vr(V_WR)}, {vr(LO4, 4)} offset:1024", "wl"))
    # loads of block b+3 once the conversion has read XA/XB/HV
    gaps[19].append(("gload", None, None))
    # ring slot rotation and addresses of the next iteration (all ring accesses issued by gap 18)
    gaps[20].append(("salu", f"s_mov_b32 s{S_T0}, s{S_RD}", None))
    gaps[20].append(("salu", f"s_mov_b32 s{S_RD}, s{S_RDN}", None))
    gaps[20].append(("salu", f"s_mov_b32 s{S_RDN}, s{S_WR}", None))
    gaps[20].append(("salu", f"s_mov_b32 s{S_WR}, s{S_T0}", None))
    gaps[21].append(("addr", f"v_add_u32 {vr(N_RD)}, s{S_RD}, %[lane16]", None))
    gaps[22].append(("addr", f"v_add_u32 {vr(N_RDN)}, s{S_RDN}, %[lane16]", None))
    gaps[22].append(("addr", f"v_add_u32 {vr(N_WR)}, s{S_WR}, %[wr16]", None))

    first_rot = True
    first_prod = True
    for g in range(24):
        ks, m = divmod(g, 6)
        if m == 0:
            cnt.need_lgkm(f"f{ks}h")
        if m == 4:
            cnt.need_lgkm(f"f{ks}l")
        out.append(mfma(cur, ks, m))
        for kind, text, tag in gaps[g]:
            if kind == "lds" and "lds" in ABLATE:
                pass
            elif kind == "lds" or kind == "ldsw":
                out.append(text)
                cnt.issue_lgkm(tag)
            elif kind == "vm":
                out.append(text)
                cnt.issue_vm(tag)
            elif kind == "rot":
                if first_rot:
                    cnt.need_vm("pB" if label == "A" else "pA")
                    first_rot = False
                if "rot" not in ABLATE:
                    out.append(text)
            elif kind == "prod":
                if first_prod:
                    cnt.need_vm("xb")
                    first_prod = False
                if "prod" not in ABLATE:
                    out.append(text)
            elif kind == "gload":
                if "gload" not in ABLATE:
                    gload_ops(cnt, out, label)
            else:
                out.append(text)
    cnt.drain_lgkm()
    if "bar" not in ABLATE:
        out.append("s_barrier")


def generate():
    out = []
    cnt = Counters(out)
    o = out.append
    o("; ===== prologue =====")
    o(f"s_mov_b32 s{S_XB}, %[xb_lo]")
    o(f"s_mov_b32 s{S_XB + 1}, %[xb_hi]")
    o(f"s_mov_b32 s{S_TB}, %[tp_lo]")
    o(f"s_mov_b32 s{S_TB + 1}, %[tp_hi]")
    o(f"s_mov_b32 s{SB['A']['p']}, %[pp_lo]")
    o(f"s_mov_b32 s{SB['A']['p'] + 1}, %[pp_hi]")
    o(f"s_mov_b32 s{S_BF}, %[bf_lo]")
    o(f"s_mov_b32 s{S_BF + 1}, %[bf_hi]")
    o(f"s_mov_b32 s{S_PSTRIDE}, %[pstride]")
    o(f"s_mov_b32 s{S_NLEFT}, %[nhi]")
    o(f"s_add_u32 s{S_NHI1}, %[nhi], -1")
    o(f"s_mov_b32 s{S_K}, 0")
    o(f"v_mov_b32 {vr(V_SC)}, %[scale]")
    o(f"s_mov_b32 s{S_RD}, 0")
    o(f"s_mov_b32 s{S_RDN}, {SLOT}")
    o(f"s_mov_b32 s{S_WR}, {2 * SLOT}")
    o("s_nop 4")
    # phasor-table operand images -> AGPRs (16 x 16 bytes per lane, 1 KiB apart).  Four
    # bases, all computed before the first load: a base is never rewritten under a load.
    BF = [S_BF, 66, 68, 70]
    for j in range(1, 4):
        o(f"s_add_u32 s{BF[j]}, s{S_BF}, {4096 * j}")
        o(f"s_addc_u32 s{BF[j] + 1}, s{S_BF + 1}, 0")
    o("s_nop 4")
    # (a workgroup that runs the loop for a second row tile keeps them: %[first] == 0)
    o("s_cmp_eq_u32 %[first], 0")
    o("s_cbranch_scc1 4f")
    for f in range(16):
        b = BF[f // 4]
        if "bimg" not in ABLATE:
            o(f"global_load_dwordx4 {ar(4 * f)}, %[bo], s[{b}:{b + 1}] offset:{(f % 4) * 1024}")
    o("4:")
    # zero: C set B, accumulators, P_B
    for base in (CB[0], CB[1], ACC[0], ACC[1]):
        for i in range(16):
            o(f"v_mov_b32 {vr(base + i)}, 0")
    o(f"v_mov_b32 {vr(PB[0])}, 0")
    o(f"v_mov_b32 {vr(PB[1])}, 0")
    # blocks 0, 1 and 2 are loaded at once (one round trip): block 0 into the input
    # registers, 1 and 2 into the still idle operand buffers; 0 and 1 are converted
    # into ring slots 0 and 1, block 2 is moved to the input registers for trip 0
    T1 = (F0, F0 + 4, F0 + 8)          # xa, xb, hv of block 1
    T2 = (F0 + 12, F0 + 16, F0 + 20)   # of block 2
    OFF_C = (CA[0], CA[0] + 1)          # offsets of block 2: set A is written by the first MFMA only
    out.extend(advance_load_pointers("A"))
    out.extend(advance_load_pointers("B"))
    out.extend(advance_load_pointers("C"))
    o("s_nop 4")
    gload_ops(cnt, out, "A")
    gload_ops(cnt, out, "B", *T1)
    gload_ops(cnt, out, "C", *T2, off=OFF_C)
    o("s_waitcnt vmcnt(0)")          # the phasor images as well
    cnt.vm = []
    for blk, src in ((0, (None, None, None)), (1, T1)):
        out.extend(produce_ops(*src))
        o(f"v_add_u32 {vr(ADDR['B'][blk])}, {blk * SLOT}, %[wr16]")
        o(f"ds_write_b128 {vr(ADDR['B'][blk])}, {vr(HI4, 4)}")
        o(f"ds_write_b128 {vr(ADDR['B'][blk])}, {vr(LO4, 4)} offset:1024")
        o("s_waitcnt lgkmcnt(0)")
    for i in range(4):
        o(f"v_mov_b32 {vr(XA + i)}, {vr(T2[0] + i)}")
        o(f"v_mov_b32 {vr(XB + i)}, {vr(T2[1] + i)}")
        o(f"v_mov_b32 {vr(HV + i)}, {vr(T2[2] + i)}")
    out.extend(advance_load_pointers("A"))   # block 3: iteration 0 ("A") loads it
    V_RD, V_RDN, V_WR = ADDR["A"]
    o(f"v_add_u32 {vr(V_RD)}, s{S_RD}, %[lane16]")
    o(f"v_add_u32 {vr(V_RDN)}, s{S_RDN}, %[lane16]")
    o(f"v_add_u32 {vr(V_WR)}, s{S_WR}, %[wr16]")
    o("s_waitcnt lgkmcnt(0)")
    o("s_barrier")
    o(f"ds_read_b128 {vr(F0, 4)}, {vr(V_RD)}")
    o(f"ds_read_b128 {vr(F0 + 4, 4)}, {vr(V_RD)} offset:1024")
    o("s_waitcnt lgkmcnt(0)")
    cnt.lgkm = []
    # steady state entry: vm = [hv, xa, xb]; the loop expects [p_prev, hv, xa, xb]
    XL = ["hv", "xa", "xb"]
    cnt.vm = ["prB", "pB"] + XL
    o("; ===== main loop, two blocks per trip =====")
    o("1:")
    iteration(cnt, out, CA, CB, PA, PB, "A")
    o(f"s_sub_u32 s{S_NLEFT}, s{S_NLEFT}, 1")
    o(f"s_cmp_eq_u32 s{S_NLEFT}, 0")
    o("s_cbranch_scc1 2f")
    state_a = (list(cnt.lgkm), list(cnt.vm))
    iteration(cnt, out, CB, CA, PB, PA, "B")
    o(f"s_sub_u32 s{S_NLEFT}, s{S_NLEFT}, 1")
    o(f"s_cmp_lg_u32 s{S_NLEFT}, 0")
    o("s_cbranch_scc1 1b")
    if "gload" not in ABLATE:
        assert cnt.lgkm == [] and cnt.vm == ["prB", "pB"] + XL, (cnt.lgkm, cnt.vm)
        assert state_a == ([], ["prA", "pA"] + XL), state_a
    # exits: P*C of the last block
    o("; last block was in set B")
    o("s_waitcnt vmcnt(0)")
    o("s_nop 15")
    o("s_nop 15")
    out.extend(rotate_ops(CB, PB))
    o("s_branch 3f")
    o("2:")
    o("; last block was in set A")


    o("s_waitcnt vmcnt(0)")
    o("s_nop 15")
    o("s_nop 15")
    out.extend(rotate_ops(CA, PA))
    o("3:")
    # hand the accumulators to the C++ epilogue through LDS (the ring is idle: every
    # wave passed the barrier that ended the last iteration)
    for q in range(8):
        base = (ACC[0] if q < 4 else ACC[1]) + 4 * (q & 3)
        o(f"ds_write_b128 %[accaddr], {vr(base, 4)} offset:{q * 1024}")
    o("s_waitcnt lgkmcnt(0)")
    return out


def main():
    lines = generate()
    PFX = "GSDR_MFMA_RING"
    print("// GENERATED by tools/gen_ddc_mfma_ring.py -- do not edit.")
    print("// Main loop of ddc_mfma_ring_kernel: see the generator for the schedule and register map.")
    print("#pragma once")
    print(f"#define {PFX}_VB {VB}")
    print(f"#define {PFX}_BYTES {3 * SLOT}")
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
