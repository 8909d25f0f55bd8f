#!/usr/bin/env python3
"""Generates gpu_sdr_amd/csrc/ddc_mfma_ring16_gen.h: main loop of ddc_mfma_ring16_kernel
(gfx950) -- the LDS-ring loop of tools/gen_ddc_mfma_ring.py re-tiled for
v_mfma_f32_16x16x32_f16.

Why: every DDC workload runs at the package power cap, and under the
cap the 16x16x32 shape delivers 12-15 % more FLOP/s than 32x32x16 at equal cycles per FLOP
(MI355X_MICROARCH.md, DVFS give-back item 7).  Same arithmetic, same ring, same writers:

  * one wave = 32 rows x 32 tones = 2 x 2 tiles of 16 x 16, x (re, im): 8 accumulators of 4
    registers (the 32 result registers of the 32x32 form, regrouped);
  * a block of 32 samples = 2 k-steps of 16 samples (K = 32 reals): 48 MFMAs of 16 cycles
    instead of 24 of 32.  A fragment (k-step k2, row half rh, hi|lo) is read from the ring
    UNCHANGED in layout -- the writers still convert 8-sample k-steps -- with a per-lane base:
    lane l takes old k-step 2*k2 + (l >> 5), old lane (16*rh + (l & 15)) + 32*((l >> 4) & 1);
    conflict-free for ds_read_b128 (the 16 lanes of a service group differ in l & 15 only);
  * MFMA order inside a k-step: (row half, hi|lo) major, so that an operand buffer is dead 8
    or 4 MFMAs after its first use and is re-read one block later at least 24 MFMAs (384
    cycles, the distance rule R1 was measured at) after its last use;
  * the block phasor P exists per tone half: two dwordx2 loads per block instead of two dword
    loads, 64 v_fma_f32 as before.

Rules R1..R4 of tools/gen_ddc_mfma_ring.py apply unchanged.

    python3 tools/gen_ddc_mfma_ring16.py > gpu_sdr_amd/csrc/ddc_mfma_ring16_gen.h
"""
import os
import sys

# timing-only builds (WRONG results): GEN_ABLATE=rot,prod,lds,gload,bar,bimg drops the rotation
# FMAs / the conversion arithmetic / the operand reads of the ring / the input loads / the barrier
# from the loop, the phasor-image loads from the prologue
ABLATE = set(filter(None, os.environ.get("GEN_ABLATE", "").split(",")))
# GEN_PRIO: how the two waves that share a SIMD (one of each of the CU's two workgroups) take
# turns on the matrix pipe.  With equal priorities the OLDER wave wins every arbitration
# (MI355X_MICROARCH.md, two waves per SIMD, item 2): measured with in-kernel stamps
# (scratch/stamp_probe.py), the older workgroup of a CU ends after 105 us of a C3 launch, the
# younger after 145 us, the last 40 us alone on its SIMDs at a third of the pipe rate.
#   "ab" : priority 1 in the even block of a trip, 0 in the odd one
PRIO = os.environ.get("GEN_PRIO", "none")
KS = 4                     # k-steps per block (PK = 32)
SLOT = KS * 2 * 1024       # bytes of one ring slot

# ---- register map (TT = 1) -------------------------------------------------
VB = 12                    # v0..v11 stay with the compiler
ACC = (VB + 0, VB + 16)    # accumulators re, im
CA = (VB + 32, VB + 48)    # C set A: re, im
CB = (VB + 64, VB + 80)    # C set B
F0 = VB + 96               # operand buffers: fragment (k2, rh, sp) at v[F0 + 16*k2 + 8*rh + 4*sp : +3]
XA, XB, HV = VB + 128, VB + 132, VB + 136
HI4, LO4 = VB + 140, VB + 144
HS = VB + 148              # scaled taps of the k-step being converted (4)
PA = VB + 152              # (Pr, Pi) of tone half 0, (Pr, Pi) of tone half 1: C set A
PB = VB + 156
V_SC = VB + 160            # S
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
S_SC = 48      # s[48:49] = (S, S)
S_T0, S_T1 = 50, 51
S_PH = 59      # wave slot parity (HW_ID wave_id & 1): GEN_PRIO=hw
S_XB = 52      # s[52:53] x base, block 0
S_TB = 54      # s[54:55] taps base, block 0
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


def produce_ops(xa=None, xb=None, hv=None):
    """x (4 complex samples in XA, XB) * taps (HV) * S -> fp16 hi (HI4) and lo (LO4): 24 plain
    VALU instructions (no packed FP32, see tools/gen_ddc_mfma_ring.py).  The product is never formed
    on its own: hi = f16(x*hs) by v_fma_mixlo/mixhi_f16 (an f32 fma, then one rounding to f16),
    lo = f16(fma(x, hs, -hi)), the residual of the exact product."""
    xa, xb, hv = (XA if xa is None else xa), (XB if xb is None else xb), (HV if hv is None else hv)
    ops = []
    hs_base = HS
    if "noscale" in ABLATE:        # timing-only (WRONG results): what pre-scaled taps would save
        hs_base = hv
    else:
        for j in range(4):
            ops.append(f"v_mul_f32 {vr(HS + j)}, {vr(hv + j)}, {vr(V_SC)}")
    xs = [xa, xa + 2, xb, xb + 2]
    if os.environ.get("GEN_CONV", "mul") == "mul":     # default; GEN_CONV=mix: 24 instructions through v_fma_mixlo/hi_f16, measured 1.2 % SLOWER on C3 (same box, scratch/ab.sh)
        for j in range(4):
            ops.append(f"v_mul_legacy_f32 {vr(xs[j])}, {vr(xs[j])}, {vr(hs_base + j)}")   # 0 * anything = 0: zero-padded taps meet samples outside the window
            ops.append(f"v_mul_legacy_f32 {vr(xs[j] + 1)}, {vr(xs[j] + 1)}, {vr(hs_base + j)}")
        for j in range(4):
            ops.append(f"v_cvt_pk_f16_f32 {vr(HI4 + j)}, {vr(xs[j])}, {vr(xs[j] + 1)}")
        for j in range(4):
            ops.append(f"v_fma_mix_f32 {vr(xs[j])}, {vr(xs[j])}, 1.0, -{vr(HI4 + j)} op_sel_hi:[0,0,1]")
            ops.append(f"v_fma_mix_f32 {vr(xs[j] + 1)}, {vr(xs[j] + 1)}, 1.0, -{vr(HI4 + j)} op_sel:[0,0,1] op_sel_hi:[0,0,1]")
        for j in range(4):
            ops.append(f"v_cvt_pk_f16_f32 {vr(LO4 + j)}, {vr(xs[j])}, {vr(xs[j] + 1)}")
        return ops
    for j in range(4):
        ops.append(f"v_fma_mixlo_f16 {vr(HI4 + j)}, {vr(xs[j])}, {vr(HS + j)}, 0")
        ops.append(f"v_fma_mixhi_f16 {vr(HI4 + j)}, {vr(xs[j] + 1)}, {vr(HS + j)}, 0")
    for j in range(4):
        ops.append(f"v_fma_mix_f32 {vr(xs[j])}, {vr(xs[j])}, {vr(HS + j)}, -{vr(HI4 + j)} op_sel_hi:[0,0,1]")
        ops.append(f"v_fma_mix_f32 {vr(xs[j] + 1)}, {vr(xs[j] + 1)}, {vr(HS + j)}, -{vr(HI4 + j)} op_sel:[0,0,1] op_sel_hi:[0,0,1]")
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


def iteration(cnt, out, cur, prev, p_cur, p_prev, label):
    """One block of 32 samples: 48 MFMAs.  cur/prev: C sets; p_cur/p_prev: P registers (4 each)."""
    out.append(f"; ---- block iteration, C set {label}")
    if PRIO == "ab":
        out.append("s_setprio 1" if label == "A" else "s_setprio 0")
    elif PRIO == "ba":
        out.append("s_setprio 0" if label == "A" else "s_setprio 1")
    elif PRIO == "hw":
        # opposite phases for the two wave slots of a SIMD: slot parity p has priority in its
        # blocks of parity p
        want = 0 if label == "A" else 1
        out.append(f"s_cmp_eq_u32 s{S_PH}, {want}")
        out.append(f"s_cbranch_scc1 {7 if label == 'A' else 8}f")
        out.append("s_setprio 0")
        out.append(f"s_branch {5 if label == 'A' else 6}f")
        out.append(f"{7 if label == 'A' else 8}:")
        out.append("s_setprio 1")
        out.append(f"{5 if label == 'A' else 6}:")
    elif PRIO == "hwstatic":
        pass
    rot = rotate_ops(prev, p_prev)
    prod = produce_ops()
    other = "B" if label == "A" else "A"
    salu = advance_load_pointers(other)          # for the next iteration's loads
    S_P, N_P = SB[label]["p"], SB[other]["p"]
    NG = 48
    gaps = {g: [] for g in range(NG)}
    V_RD, V_RDN, V_WR = ADDR[label]
    N_RD, N_RDN, N_WR = ADDR[other]

    def read_frag(g, slot_reg, k2, rh, sp, into):
        # ring slot layout (unchanged): old k-step ks at ks*2048, hi at +0, lo at +1024, old lane
        # (row, hh) at 16*(row + 32*hh).  %[lane16] carries the per-lane part (see the kernel).
        off = k2 * 4096 + sp * 1024 + rh * 256
        gaps[g].append(("lds", f"ds_read_b128 {vr(into, 4)}, {vr(slot_reg)} offset:{off}", f"f{k2}{rh}{sp}"))

    # this block's k-step 1 (ring slot RD); each buffer at least 24 MFMAs after its last use
    for (rh, sp, g) in ((0, 0, 8), (0, 1, 12), (1, 0, 20), (1, 1, 24)):
        assert g + NG - last_use(1, rh, sp) >= 24 and g < first_use(1, rh, sp) - 8
        read_frag(g, V_RD, 1, rh, sp, frag_for(label, 1, rh, sp))
    # the next block's k-step 0 (ring slot RDN): row half 0 into its only buffer once that is
    # free, row half 1 into the buffer the next iteration uses
    for (rh, sp, g) in ((0, 0, 32), (1, 0, 34), (0, 1, 36), (1, 1, 38)):
        if rh == 0:
            assert g - last_use(0, rh, sp) >= 24
        read_frag(g, V_RDN, 0, rh, sp, frag_for(other, 0, rh, sp))
    # P of this block (used by the next iteration's rotation): tone halves 0 and 1
    gaps[0].append(("vm", f"global_load_dwordx2 {vr(p_cur, 2)}, %[po], s[{S_P}:{S_P + 1}]", "pr" + label))
    gaps[0].append(("vm", f"global_load_dwordx2 {vr(p_cur + 2, 2)}, %[po], s[{S_P}:{S_P + 1}] offset:128", "p" + label))
    gaps[1].append(("salu", f"s_add_u32 s{N_P}, s{S_P}, s{S_PSTRIDE}", None))
    gaps[1].append(("salu", f"s_addc_u32 s{N_P + 1}, s{S_P + 1}, 0", None))
    for i, sx in enumerate(salu):
        gaps[2 + i // 2].append(("salu", sx, None))
    # P*C of the previous block: two per gap in gaps 4..19, the rest behind the conversion
    ri = 0
    for g in range(4, 20):
        for _ in range(2):
            gaps[g].append(("rot", rot[ri], None))
            ri += 1
    # conversion of block b+2: gaps 20..33
    pi = 0
    for g in range(20, 34):
        for _ in range(2):
            if pi < len(prod):
                gaps[g].append(("prod", prod[pi], None))
                pi += 1
    assert pi == len(prod), (pi, len(prod))
    left = len(rot) - ri
    for k in range(left):
        g = 34 + (k * 14) // left
        gaps[g].append(("rot", rot[ri], None))
        ri += 1
    assert ri == len(rot), ri
    gaps[34].append(("ldsw", f"ds_write_b128 {vr(V_WR)}, {vr(HI4, 4)}", "wh"))
    gaps[34].append(("ldsw", f"ds_write_b128 {vr(V_WR)}, {vr(LO4, 4)} offset:1024", "wl"))
    # loads of block b+3 once the conversion has read XA/XB/HV
    gaps[36].append(("gload", None, None))
    # ring slot rotation and addresses of the next iteration (all ring accesses issued by gap 38)
    if "noaddr" in ABLATE:         # timing-only (WRONG results): ring addresses never advance
        pass
    else:
      gaps[40].append(("salu", f"s_mov_b32 s{S_T0}, s{S_RD}", None))
      gaps[40].append(("salu", f"s_mov_b32 s{S_RD}, s{S_RDN}", None))
      gaps[41].append(("salu", f"s_mov_b32 s{S_RDN}, s{S_WR}", None))
      gaps[41].append(("salu", f"s_mov_b32 s{S_WR}, s{S_T0}", None))
      gaps[43].append(("addr", f"v_add_u32 {vr(N_RD)}, s{S_RD}, %[lane16]", None))
      gaps[44].append(("addr", f"v_add_u32 {vr(N_RDN)}, s{S_RDN}, %[lane16]", None))
      gaps[45].append(("addr", f"v_add_u32 {vr(N_WR)}, s{S_WR}, %[wr16]", None))

    first_rot = True
    first_prod = True
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


# ---- the scale of the lane's output row, inside the prologue -------------------------------------
# S = 2^se with se = 140 - (biased exponent of the largest finite |x| of the row's window), clamped to
# +-100 (row_scale_exp in csrc/ddc_mfma.hip, bit for bit).  The window's segment maxima -- up to eight
# entries of the table absmax_kernel left, seg[q0 .. q0 + span] -- are loaded HERE, with the phasor
# images and the first input blocks, and reduced behind the prologue's one s_waitcnt vmcnt(0): computed in
# C++ in front of this block they cost every workgroup a memory round trip of their own (2 us of a 20 us
# workgroup on C2).  %[sgo]: byte offset of seg[q0]; %[sgn]: span; s[S_SC:S_SC+1]: the table.
SG_T = CA[1]               # 8 offsets + 8 maxima: C set A (im) is not written before the first MFMA


def scale_loads():
    ops = []
    for i in range(8):
        if i == 0:
            ops.append(f"v_mov_b32 {vr(SG_T)}, %[sgo]")
        else:
            ops.append(f"v_min_u32 {vr(SG_T + i)}, {i}, %[sgn]")
            ops.append(f"v_lshl_add_u32 {vr(SG_T + i)}, {vr(SG_T + i)}, 2, %[sgo]")
    ops.append("s_nop 2")
    for i in range(8):
        ops.append(f"global_load_dword {vr(SG_T + 8 + i)}, {vr(SG_T + i)}, s[{S_SC}:{S_SC + 1}]")
    return ops


def scale_compute():
    m = SG_T + 8
    return [
        f"v_max3_u32 {vr(m)}, {vr(m)}, {vr(m + 1)}, {vr(m + 2)}",
        f"v_max3_u32 {vr(m + 3)}, {vr(m + 3)}, {vr(m + 4)}, {vr(m + 5)}",
        f"v_max3_u32 {vr(m)}, {vr(m)}, {vr(m + 6)}, {vr(m + 7)}",
        f"v_max_u32 {vr(m)}, {vr(m)}, {vr(m + 3)}",
        f"v_bfe_u32 {vr(m)}, {vr(m)}, 23, 8",
        f"v_sub_u32 {vr(m)}, 140, {vr(m)}",
        f"v_max_i32 {vr(m)}, 0xffffff9c, {vr(m)}",        # -100
        f"v_min_i32 {vr(m)}, 100, {vr(m)}",
        f"v_add_u32 {vr(m)}, 127, {vr(m)}",
        f"v_lshlrev_b32 {vr(V_SC)}, 23, {vr(m)}",
    ]


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
    if PRIO in ("hw", "hwstatic"):
        o(f"s_getreg_b32 s{S_PH}, hwreg(HW_REG_HW_ID, 0, 4)")
        o(f"s_and_b32 s{S_PH}, s{S_PH}, 1")
    if PRIO == "hwstatic":
        # the wave in the odd slot of its SIMD (the later arrival) takes priority for good
        o(f"s_cmp_eq_u32 s{S_PH}, 1")
        o("s_cbranch_scc0 9f")
        o("s_setprio 1")
        o("9:")
    o(f"s_mov_b32 s{S_SC}, %[sg_lo]")
    o(f"s_mov_b32 s{S_SC + 1}, %[sg_hi]")
    o("s_nop 2")
    out.extend(scale_loads())
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
    for i in range(4):
        o(f"v_mov_b32 {vr(PB + i)}, 0")
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
    o("s_waitcnt vmcnt(0)")          # the phasor images and the segment maxima as well
    cnt.vm = []
    out.extend(scale_compute())
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
    for rh in range(2):
        for sp in range(2):
            o(f"ds_read_b128 {vr(frag_for('A', 0, rh, sp), 4)}, {vr(V_RD)} offset:{sp * 1024 + rh * 256}")
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
    # ... and the lane's scale (the bits of S = 2^se): the epilogue needs 1 / S of every row again, and loading the
    # segment maxima a second time cost it a microsecond (C2, same-box A/B)
    o(f"ds_write_b32 %[seaddr], {vr(V_SC)}")
    o("s_waitcnt lgkmcnt(0)")
    return out


def main():
    lines = generate()
    PFX = "GSDR_MFMA_RING16"
    print("// GENERATED by tools/gen_ddc_mfma_ring16.py -- do not edit.")
    print("// Main loop of ddc_mfma_ring16_kernel (v_mfma_f32_16x16x32_f16): see the generator for the schedule and register map.")
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
