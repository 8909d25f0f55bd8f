#!/usr/bin/env python3
"""Generates gpu_sdr_amd/csrc/ddc_steps_gen.h: the inline-asm body of one DDC
sub-block (PK samples) for every (F, PK) the flat kernel is instantiated for.

A sub-block is a sequence of steps; step k
    s_waitcnt lgkmcnt(0)                       (operands of group k have landed)
    s_load ... -> the OTHER SGPR set           (group k+1, or the look-ahead)
    per sample of group k:  u = x*B[lo]  (v_pk_mul + v_pk_fma)
                            S[j] += h[j]*u     (v_pk_fma, j < F)
Groups alternate between SGPR set A and set B, which live in s[52:99] -- outside
what hipcc may allocate (the kernel is built with amdgpu_num_sgpr(52)).
Group sizes are 4 samples, plus two 2-sample groups where PK is not a multiple
of 8, so that the number of steps is even and every sub-block starts in set A.

Run:  python tools/gen_ddc_steps.py   (the output is committed)
"""
import os

BASE = 52                      # first private SGPR
XA, TA = BASE, BASE + 8        # set A: x s[52:59], taps s[60:75]
XB, TB = BASE + 24, BASE + 32  # set B: x s[76:83], taps s[84:99]
SIZES = {12: [4, 4, 2, 2], 16: [4, 4, 4, 4], 20: [4, 4, 4, 4, 2, 2], 40: [4] * 10}
LOADOP = {2: "s_load_dwordx2", 4: "s_load_dwordx4", 8: "s_load_dwordx8", 16: "s_load_dwordx16"}


def sreg(lo, n):
    return f"s[{lo}:{lo + n - 1}]"


def clobbers(x0, t0):
    return ", ".join(f'"s{r}"' for r in list(range(x0, x0 + 8)) + list(range(t0, t0 + 16)))


def step(F, FP, cur_is_a, n, lo0, nxt_n, nxt_from_next, nxt_off, first=False, masked=False):
    """asm text + operand lists of one step.  first: S is write-only (the very
    first MAC of the sub-block is a multiply, so S needs no zeroing).  masked: the last
    sub-block of a block whose length is not a multiple of PK -- samples lo >= R belong to the
    NEXT block and meet zero taps here; they are replaced by zeros on the scalar side (three SALU
    instructions per sample), because 0 * Inf is NaN and a sample of the next block must not
    reach this block's outputs whatever it is (ref: cpp/fir.cu:48-61 sums one window)."""
    cx, ct = (XA, TA) if cur_is_a else (XB, TB)
    nx, nt = (XB, TB) if cur_is_a else (XA, TA)
    lines = ["@W"]
    xp, tp = ("%[xn]", "%[tn]") if nxt_from_next else ("%[xp]", "%[tp]")
    xo = 0 if nxt_from_next else nxt_off * 8
    to = 0 if nxt_from_next else nxt_off * FP * 4
    lines.append("@L" + f"{LOADOP[2 * nxt_n]} {sreg(nx, 2 * nxt_n)}, {xp}, {xo}")
    lines.append("@L" + f"{LOADOP[nxt_n * FP]} {sreg(nt, nxt_n * FP)}, {tp}, {to}")
    if masked:
        for s in range(n):
            lines.append(f"s_cmp_gt_u32 %[R], {lo0 + s}")
            lines.append(f"s_cselect_b32 s{cx + 2 * s}, s{cx + 2 * s}, 0")
            lines.append(f"s_cselect_b32 s{cx + 2 * s + 1}, s{cx + 2 * s + 1}, 0")
    for s in range(n):
        x = sreg(cx + 2 * s, 2)
        b = f"%[b{s}]"
        lines.append(f"v_pk_mul_f32 %[t], {x}, {b} op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[1,0]")
        lines.append(f"v_pk_fma_f32 %[u], {x}, {b}, %[t] op_sel_hi:[0,1,1]")
        for j in range(F):
            flat = s * FP + j               # index of h[j] of sample s in the tap tuple
            pair = sreg(ct + (flat // 2) * 2, 2)
            if first and s == 0:
                if flat % 2 == 0:
                    lines.append(f"v_pk_mul_f32 %[s{j}], {pair}, %[u] op_sel_hi:[0,1]")
                else:
                    lines.append(f"v_pk_mul_f32 %[s{j}], {pair}, %[u] op_sel:[1,0] op_sel_hi:[1,1]")
            elif flat % 2 == 0:
                lines.append(f"v_pk_fma_f32 %[s{j}], {pair}, %[u], %[s{j}] op_sel_hi:[0,1,1]")
            else:
                lines.append(f"v_pk_fma_f32 %[s{j}], {pair}, %[u], %[s{j}] op_sel:[1,0,0] op_sel_hi:[1,1,1]")
    sc = '"=&v"' if first else '"+v"'
    outs = ", ".join([f'[s{j}] {sc}(S[{j}])' for j in range(F)] + ['[t] "=&v"(t)', '[u] "=&v"(u)'])
    ins = ", ".join([f'[b{s}] "v"(B[{lo0 + s}])' for s in range(n)] +
                    (['[xn] "s"(xnext)', '[tn] "s"(tnext)'] if nxt_from_next else ['[xp] "s"(xg)', '[tp] "s"(tg)']) +
                    (['[R] "s"(R)'] if masked else []))
    def emit(ln):
        if ln == "@W":
            return "        GSDR_ASM_WAIT"
        if ln.startswith("@L"):
            return f'        GSDR_ASM_LOAD("{ln[2:]}\\n\\t")'
        return f'        "{ln}\\n\\t"'
    text = "\n".join(emit(ln) for ln in lines)
    clob = clobbers(nx, nt)
    if masked:
        clob += ", " + ", ".join(f'"s{r}"' for r in range(cx, cx + 2 * n)) + ', "scc"'
    return f"    asm volatile(\n{text}\n        : {outs}\n        : {ins}\n        : {clob});\n"


def subblock(F, PK):
    FP = 4 if F == 3 else F
    sizes = SIZES[PK]
    assert sum(sizes) == PK and len(sizes) % 2 == 0 and sizes[0] == 4
    out = [f"template <> struct SubBlock<{F}, {PK}> {{",
           "    // S is an OUTPUT: the sub-block sums start from zero",
           f"    static __device__ __forceinline__ void run(f2v (&S)[{F}], const f2v (&B)[{PK}], const float2 *xg,",
           "                                               const float *tg, const float2 *xnext,",
           "                                               const float *tnext) {",
           "    f2v t, u;"]
    lo = 0
    for k, n in enumerate(sizes):
        last = (k == len(sizes) - 1)
        nxt_n = sizes[0] if last else sizes[k + 1]
        out.append(step(F, FP, k % 2 == 0, n, lo, nxt_n, last, lo + n, first=(k == 0)))
        lo += n
    out.append("    }")
    # the last sub-block of a block with M % PK != 0: only its first R samples are the block's
    out += ["    // samples lo >= R are zeroed on the scalar side (see tools/gen_ddc_steps.py, step(masked=True))",
            f"    static __device__ __forceinline__ void run_masked(f2v (&S)[{F}], const f2v (&B)[{PK}], const float2 *xg,",
            "                                                      const float *tg, const float2 *xnext,",
            "                                                      const float *tnext, int R) {",
            "    f2v t, u;"]
    lo = 0
    for k, n in enumerate(sizes):
        last = (k == len(sizes) - 1)
        nxt_n = sizes[0] if last else sizes[k + 1]
        out.append(step(F, FP, k % 2 == 0, n, lo, nxt_n, last, lo + n, first=(k == 0), masked=True))
        lo += n
    out.append("    }\n};\n")
    return "\n".join(out)


def prime(F):
    FP = 4 if F == 3 else F
    return (f"template <> __device__ __forceinline__ void prime_set_a<{F}>(const void *xp, const void *tp) {{\n"
            f'    asm volatile("{LOADOP[8]} {sreg(XA, 8)}, %0, 0\\n\\t{LOADOP[4 * FP]} {sreg(TA, 4 * FP)}, %1, 0"\n'
            f'                 :: "s"(xp), "s"(tp) : {clobbers(XA, TA)});\n}}\n')


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    dst = os.path.join(here, "..", "gpu_sdr_amd", "csrc", "ddc_steps_gen.h")
    parts = ["// GENERATED by tools/gen_ddc_steps.py -- do not edit.\n"
             "// Inline-asm sub-block bodies of ddc_flat_kernel (see ddc_pipe.hip).\n"
             "#pragma once\n\n"
             f"#define GSDR_PRIVATE_SGPR_BASE {BASE}\n\n"
             "// timing-only ablation builds (results are WRONG): -DGSDR_ABLATE=1 drops the\n"
             "// scalar loads and their waits, =2 keeps the loads but never waits for them\n"
             "#if defined(GSDR_ABLATE) && GSDR_ABLATE == 1\n"
             "#define GSDR_ASM_WAIT\n#define GSDR_ASM_LOAD(s)\n"
             "#elif defined(GSDR_ABLATE) && GSDR_ABLATE == 2\n"
             "#define GSDR_ASM_WAIT\n#define GSDR_ASM_LOAD(s) s\n"
             "#else\n"
             "#define GSDR_ASM_WAIT \"s_waitcnt lgkmcnt(0)\\n\\t\"\n#define GSDR_ASM_LOAD(s) s\n"
             "#endif\n\n"
             "template <int F, int PK> struct SubBlock;\n"
             "template <int F> __device__ __forceinline__ void prime_set_a(const void *xp, const void *tp);\n"]
    for F in (1, 2, 3, 4):
        parts.append(prime(F))
        for PK in (12, 16, 20, 40):
            parts.append(subblock(F, PK))
    with open(dst, "w") as f:
        f.write("\n".join(parts))
    print("wrote", os.path.normpath(dst))


if __name__ == "__main__":
    main()
