#!/usr/bin/env python3
"""Generates gpu_sdr_amd/csrc/ddc_mfma_gen.h: the main loop of ddc_mfma_asm_kernel
(gfx950) as one inline-asm block with a fixed register map.

Why assembly: the loop's correctness depends on register reuse distances and
operand forms the compiler does not know about (see "Rules" below, all measured
on MI355X with scratch/mfma_probe.py and scratch/mfma_diag*.py), and its speed
on an exact MFMA/VALU interleave.

Work of one wave: 32 output rows x 32 tones; every phasor block of 32 samples is
4 k-steps x 6 MFMAs (v_mfma_f32_32x32x16_f16: Cr/Ci x {hi*Bhi, hi*Blo, lo*Bhi}).
No LDS ring and no barrier in the loop: each wave converts its own A operand (the
four waves of a workgroup redo the same 20 VALU instructions per k-step, which fit
in the MFMAs' shadow, and share the input through the L1).  In the shadow of
k-step t the wave
  * converts k-step t+1: x * (taps*S) -> fp16 hi/lo, registers only,
  * issues the loads of k-step t+5 (x, global) and t+3 (taps*S, the workgroup's
    LDS table),
  * applies a quarter of P*C of the previous block on the VALU (C alternates
    between two register sets, hence the 2x unroll).
All of that is plain (non-packed) FP32: tools/ubench_mfma.hip measures that
v_pk_fma_f32 / v_pk_mul_f32 do not overlap with the f16 MFMA at all (+10 cycles
each), while v_fma_f32, v_mul_f32, v_cvt_pk_f16_f32 and v_fma_mix_f32 nearly
vanish in its shadow at two waves per SIMD (6 per MFMA: +7 %).

Rules this file keeps (each one cost a debugging session):
  R1  A register that an MFMA reads as A/B operand is not rewritten before twelve
      further MFMAs have been issued (four operand buffers).  With less distance
      rows 16..31 of the MFMA came out computed from the new contents.
  R2  Address, data and scalar-base registers of a memory instruction stay unchanged
      for two k-steps after it was issued: with two workgroups on a CU they are read
      late (registers rewritten ~4 MFMAs later gave half of the lanes the new address).
  R3  No v_pk_*_f32 that broadcasts the HIGH half of a pair written by the packed
      instruction in front of it (op_sel:[0,1] / [1,..]): it returned a stale value in
      lanes 48..63 now and then while another wave of the SIMD ran this loop
      (isolated reproducer: tools/ubench_pk_hazard.hip).  The loop has no packed FP32
      at all (see above); the kernel around it is compiled without packed FP32.
  R4  s_waitcnt values come from a model of the counters (class Counters).

    python3 tools/gen_ddc_mfma.py > gpu_sdr_amd/csrc/ddc_mfma_gen.h
"""

import os

# timing-only builds (wrong results): GEN_ABLATE=rot,conv,loads,mfma drops that work from the loop
ABLATE = set(filter(None, os.environ.get("GEN_ABLATE", "").split(",")))

KS = 4                     # k-steps per block (PK = 32)

# ---- register map ------------------------------------------------------------
VB = 12                    # v0..v11 stay with the compiler
CA = (VB + 0, VB + 16)     # C set A: re, im (16 registers each)
CB = (VB + 32, VB + 48)    # C set B
ACC = (VB + 64, VB + 80)   # accumulators re, im
F0 = VB + 96               # operand buffers: k-step s hi v[F0+8s:+3], lo v[F0+8s+4:+3]
X0 = VB + 128              # input buffers: k-step t in v[X0+8(t%4):+7]
H0 = VB + 160              # scaled taps of a k-step: parity p: v[H0+4p:+3]
PA = (VB + 168, VB + 169)  # (Pr, Pi) of C set A
PB = (VB + 170, VB + 171)
V_TB = [VB + 172, VB + 173, VB + 174, VB + 175]   # taps table address of k-step position q
V_LAST = VB + 175
NVGPR_CLOBBER = list(range(VB, V_LAST + 1))
NAGPR = 64                 # phasor-table images

# private SGPRs
S_XB = 36                  # s[36:37] x base, k-step 0
S_XS = [38, 40, 42, 44]    # x base of the next load of k-step position q (R2: one each)
S_P = {"A": 46, "B": 48}   # P row pointer, one per parity (R2)
S_NLEFT, S_B, S_KMAX = 50, 51, 52
S_T0 = 53
S_PSTRIDE = 55
S_BF = [56, 58, 60, 62]    # phasor-table image bases
SGPR_CLOBBER = list(range(36, 64))


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


def convert_ops(xb, hp, buf):
    """x (4 complex samples in X buffer xb) * scaled taps (H[hp]) -> fp16 hi/lo in operand
    buffer `buf`.  24 VALU instructions, x is consumed in place."""
    xa = X0 + 8 * xb
    xs = [xa, xa + 2, xa + 4, xa + 6]
    h = H0 + 4 * hp
    fh, fl = F0 + 8 * buf, F0 + 8 * buf + 4
    ops = []
    for j in range(4):
        ops.append(f"v_mul_f32 {vr(xs[j])}, {vr(xs[j])}, {vr(h + j)}")
        ops.append(f"v_mul_f32 {vr(xs[j] + 1)}, {vr(xs[j] + 1)}, {vr(h + j)}")
    for j in range(4):
        ops.append(f"v_cvt_pk_f16_f32 {vr(fh + j)}, {vr(xs[j])}, {vr(xs[j] + 1)}")
    for j in range(4):                       # residual v - float(hi), in place
        ops.append(f"v_fma_mix_f32 {vr(xs[j])}, {vr(xs[j])}, 1.0, -{vr(fh + j)} op_sel_hi:[0,0,1]")
        ops.append(f"v_fma_mix_f32 {vr(xs[j] + 1)}, {vr(xs[j] + 1)}, 1.0, -{vr(fh + j)} op_sel:[0,0,1] op_sel_hi:[0,0,1]")
    for j in range(4):
        ops.append(f"v_cvt_pk_f16_f32 {vr(fl + j)}, {vr(xs[j])}, {vr(xs[j] + 1)}")
    return ops


def load_x(cnt, out, q, xb):
    """x of the next k-step at position q of its block, into X buffer xb."""
    xa = X0 + 8 * xb
    out.append(f"global_load_dwordx4 {vr(xa, 4)}, %[xo], s[{S_XS[q]}:{S_XS[q] + 1}]")
    cnt.issue_vm(f"xa{xb}")
    out.append(f"global_load_dwordx4 {vr(xa + 4, 4)}, %[xo], s[{S_XS[q]}:{S_XS[q] + 1}] offset:16")
    cnt.issue_vm(f"xb{xb}")


def load_h(cnt, out, q, hp):
    """scaled taps of the next k-step at position q, into H[hp]."""
    out.append(f"ds_read_b128 {vr(H0 + 4 * hp, 4)}, {vr(V_TB[q])}")
    cnt.issue_lgkm(f"h{hp}")


def advance_x(q, add):
    """x base of position q for block S_B + add; the k-step index is clamped to the last
    one (loads past the window: valid addresses, zero taps)."""
    return [
        f"s_add_u32 s{S_T0}, s{S_B}, {add}",
        f"s_lshl_b32 s{S_T0}, s{S_T0}, 2",
        f"s_add_u32 s{S_T0}, s{S_T0}, {q}",
        f"s_min_u32 s{S_T0}, s{S_T0}, s{S_KMAX}",
        f"s_lshl_b32 s{S_T0}, s{S_T0}, 6",
        f"s_add_u32 s{S_XS[q]}, s{S_XB}, s{S_T0}",
        f"s_addc_u32 s{S_XS[q] + 1}, s{S_XB + 1}, 0",
    ]


def advance_h(q):
    return [f"v_add_u32 {vr(V_TB[q])}, 128, {vr(V_TB[q])}"]


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
    """One block = 4 k-steps.  In k-step s (global index t):
       convert k-step t+1 (X[(s+1)%4], H[(s+1)%2] -> operand buffer (s+1)%4);
       x of k-step t+5 -> X[(s+1)%4] (position (s+1)%4 of the next block, or of the one
       after for s == 3); taps of k-step t+3 -> H[(s+1)%2] (position (s+3)%4);
       a quarter of the previous block's P*C;
       bases: the x base used two k-steps ago and the taps address used two k-steps
       ago move one block on (R2)."""
    other = "B" if label == "A" else "A"
    rot = rotate_ops(prev, p_prev)
    first_rot = True
    for s in range(4):
        xb, hp = (s + 1) & 3, (s + 1) & 1
        conv = convert_ops(xb, hp, (s + 1) & 3)
        fill = [("conv", i) for i in range(len(conv))]
        fill.append(("loadx", (s + 1) & 3, xb))
        fill.append(("loadh", (s + 3) & 3, hp))
        fill += [("rot", r) for r in rot[16 * s: 16 * s + 16]]
        # x base of position q is used at k-step (q-1)%4 and moved at (q+1)%4;
        # taps address of position q is used at k-step (q+1)%4 and moved at (q+3)%4
        qx = (s - 1) & 3
        salu = advance_x(qx, 2 if qx in (1, 2, 0) else 1) + advance_h((s + 1) & 3)
        if s == 0:
            salu += [f"s_add_u32 s{S_P[other]}, s{S_P[label]}, s{S_PSTRIDE}",
                     f"s_addc_u32 s{S_P[other] + 1}, s{S_P[label] + 1}, 0"]
        per_gap = -(-len(fill) // 6)
        pos = 0
        for m in range(6):
            if "mfma" not in ABLATE:
                out.append(mfma(cur, s, m))
            if m == 0 and s == 0:
                out.append(f"global_load_dword {vr(p_cur[0])}, %[po], s[{S_P[label]}:{S_P[label] + 1}]")
                cnt.issue_vm("pr" + label)
                out.append(f"global_load_dword {vr(p_cur[1])}, %[po], s[{S_P[label]}:{S_P[label] + 1}] offset:4")
                cnt.issue_vm("p" + label)
            if m == 1:
                out.extend(salu)
            for item in fill[pos: pos + per_gap]:
                if item[0] == "conv":
                    if item[1] == 0 and "loads" not in ABLATE:
                        cnt.need_lgkm(f"h{hp}")
                        cnt.need_vm(f"xb{xb}")
                    if "conv" not in ABLATE:
                        out.append(conv[item[1]])
                elif item[0] == "loadx":
                    if "loads" not in ABLATE:
                        load_x(cnt, out, item[1], item[2])
                elif item[0] == "loadh":
                    if "loads" not in ABLATE:
                        load_h(cnt, out, item[1], item[2])
                else:
                    if first_rot:
                        cnt.need_vm("p" + other)
                        first_rot = False
                    if "rot" not in ABLATE:
                        out.append(item[1])
            pos += per_gap
        assert pos >= len(fill)
    out.append(f"s_add_u32 s{S_B}, s{S_B}, 1")


def body(cnt, out):
    o = out.append
    o("1:")
    iteration(cnt, out, CA, CB, PA, PB, "A")
    o(f"s_sub_u32 s{S_NLEFT}, s{S_NLEFT}, 1")
    o(f"s_cmp_eq_u32 s{S_NLEFT}, 0")
    o("s_cbranch_scc1 2f")
    mid = (list(cnt.lgkm), list(cnt.vm))
    iteration(cnt, out, CB, CA, PB, PA, "B")
    o(f"s_sub_u32 s{S_NLEFT}, s{S_NLEFT}, 1")
    o(f"s_cmp_lg_u32 s{S_NLEFT}, 0")
    o("s_cbranch_scc1 1b")
    return mid


def generate():
    out = []
    cnt = Counters(out)
    o = out.append
    o(f"s_mov_b32 s{S_XB}, %[xb_lo]")
    o(f"s_mov_b32 s{S_XB + 1}, %[xb_hi]")
    o(f"s_mov_b32 s{S_P['A']}, %[pp_lo]")
    o(f"s_mov_b32 s{S_P['A'] + 1}, %[pp_hi]")
    o(f"s_mov_b32 s{S_PSTRIDE}, %[pstride]")
    o(f"s_mov_b32 s{S_NLEFT}, %[nhi]")
    o(f"s_lshl_b32 s{S_KMAX}, %[nhi], 2")
    o(f"s_sub_u32 s{S_KMAX}, s{S_KMAX}, 1")          # last k-step index
    o(f"s_mov_b32 s{S_BF[0]}, %[bf_lo]")
    o(f"s_mov_b32 s{S_BF[0] + 1}, %[bf_hi]")
    for j in range(1, 4):
        o(f"s_add_u32 s{S_BF[j]}, %[bf_lo], {4096 * j}")
        o(f"s_addc_u32 s{S_BF[j] + 1}, %[bf_hi], 0")
    # bases of block 0
    o(f"s_mov_b32 s{S_B}, 0")
    for q in range(4):
        o(f"v_add_u32 {vr(V_TB[q])}, {32 * q}, %[tb]")
        out.extend(advance_x(q, 0))
    o("s_nop 4")
    for f in range(16):
        b = S_BF[f // 4]
        o(f"global_load_dwordx4 {ar(4 * f)}, %[bo], s[{b}:{b + 1}] offset:{(f % 4) * 1024}")
    for base in (CB[0], CB[1], ACC[0], ACC[1]):
        for i in range(16):
            o(f"v_mov_b32 {vr(base + i)}, 0")
    o(f"v_mov_b32 {vr(PB[0])}, 0")
    o(f"v_mov_b32 {vr(PB[1])}, 0")
    # x of k-steps 0..3, taps of k-steps 0 and 1; k-step 0 converted; then x of k-step 4,
    # taps of k-step 2, and the bases the first trip expects already moved on
    for q in range(4):
        load_x(cnt, out, q, q)
    load_h(cnt, out, 0, 0)
    load_h(cnt, out, 1, 1)
    o("s_waitcnt vmcnt(0)")
    o("s_waitcnt lgkmcnt(0)")
    cnt.vm, cnt.lgkm = [], []
    out.extend(convert_ops(0, 0, 0))
    out.extend(advance_x(0, 1))
    out.extend(advance_x(1, 1))
    out.extend(advance_x(2, 1))
    out.extend(advance_h(0))
    o("s_nop 4")
    load_x(cnt, out, 0, 0)
    load_h(cnt, out, 2, 0)
    # the loop's first conversions expect k-steps 1..3 (x) and 1 (taps) pending in this order
    cnt.vm = ["xa1", "xb1", "xa2", "xb2", "xa3", "xb3"] + cnt.vm
    cnt.lgkm = ["h1"] + cnt.lgkm
    entry = (list(cnt.lgkm), list(cnt.vm))
    # steady-state counter state: simulate one trip, then emit from that state
    sim = Counters([])
    sim.lgkm, sim.vm = list(cnt.lgkm), list(cnt.vm)
    body(sim, [])
    steady = (list(sim.lgkm), list(sim.vm))
    cnt.lgkm, cnt.vm = list(steady[0]), list(steady[1])
    body(cnt, out)
    assert (cnt.lgkm, cnt.vm) == steady, ((cnt.lgkm, cnt.vm), steady)
    # the waits were computed for the steady state; at first entry the same loads are
    # pending in the same order, minus the P loads of a previous block
    if not ABLATE:
        assert [t for t in steady[1] if not t.startswith("p")] == entry[1], (steady, entry)
        assert steady[0] == entry[0], (steady, entry)
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
    # the accumulators go to the C++ epilogue through LDS, over the taps table: every
    # wave of the workgroup must be done reading it
    o("s_waitcnt lgkmcnt(0)")
    o("s_barrier")
    for qd in range(8):
        base = (ACC[0] if qd < 4 else ACC[1]) + 4 * (qd & 3)
        o(f"ds_write_b128 %[accaddr], {vr(base, 4)} offset:{qd * 1024}")
    o("s_waitcnt lgkmcnt(0)")
    return out


def main():
    lines = generate()
    print("// GENERATED by tools/gen_ddc_mfma.py -- do not edit.")
    print("// Main loop of ddc_mfma_asm_kernel: see the generator for the schedule, the register map and the rules.")
    print("#pragma once")
    print(f"#define GSDR_MFMA_ASM_VB {VB}")
    print("#define GSDR_MFMA_ASM_TEXT \\")
    for ln in lines:
        print(f'    "{ln}\\n\\t" \\')
    print('    ""')
    clob = [f'"v{i}"' for i in NVGPR_CLOBBER] + [f'"a{i}"' for i in range(NAGPR)] + \
           [f'"s{i}"' for i in SGPR_CLOBBER] + ['"vcc"', '"scc"', '"memory"']
    print("#define GSDR_MFMA_ASM_CLOBBERS \\")
    for i in range(0, len(clob), 12):
        tail = ", \\" if i + 12 < len(clob) else ""
        print("    " + ", ".join(clob[i:i + 12]) + tail)
    n_mfma = sum(1 for l in lines if l.startswith("v_mfma"))
    print(f"// {len(lines)} lines, {n_mfma} MFMAs, VGPRs v{VB}..v{V_LAST}, AGPRs a0..a{NAGPR - 1}")


if __name__ == "__main__":
    main()
