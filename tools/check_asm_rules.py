#!/usr/bin/env python3
"""Static check of the generated assembly loops against the hazard rules of DESIGN.md section 4.1.

The rules were found empirically on MI355X (two waves per SIMD, one of them in an MFMA loop) and
the generators encode them in their schedules; nothing but a soak test would notice an edit
that breaks one.  This script re-derives them from the emitted instruction stream of the MAIN
LOOP (the text between the loop label and the backward branch, walked twice so that distances
across the back edge are seen) and fails loudly:

  R1  a VGPR that an MFMA reads as A or B operand is not rewritten by an LDS return (ds_read)
      or a global load before R1_MIN further matrix-pipe cycles of this wave have been issued
      (32 cycles per 32x32x16 MFMA, 16 per 16x16x32: the rule was measured as ~12 MFMAs of the
      32x32 shape = 384 cycles; the generators keep >= 384);
  R2  the address VGPRs, the scalar base pair and M0 of a queued LDS / global / LDS-DMA
      instruction are not rewritten before R2_MIN MFMAs have been issued behind it (measured: a
      rewrite 4 MFMAs later was seen by half of the lanes; the generators keep per-parity sets);
  R3  no packed FP32 arithmetic (v_pk_*_f32) anywhere in the loop.

    python3 tools/check_asm_rules.py gpu_sdr_amd/csrc/ddc_mfma_ring16_gen.h [...]
"""
import re
import sys

R1_MIN_CYCLES = 352      # 11 MFMAs of 32 cycles: what the 32x32 ring loop of round 1 keeps (its own schedule)
R2_MIN_MFMAS = 5


def parse(path):
    text = open(path).read()
    m = re.search(r"#define \w+_TEXT \\\n(.*?)\n    \"\"", text, re.S)
    if not m:
        return None          # not a generated MFMA loop (ddc_steps_gen.h has no _TEXT block)
    body = m.group(1)
    lines = re.findall(r'"(.*?)\\n\\t"', body)
    return lines


def reg_range(tok):
    """'v[12:15]' -> ('v', 12..15); 'v7' -> ('v', [7]); 'a[0:3]', 's[36:37]', 's45', 'm0'"""
    tok = tok.strip().lstrip("-")
    m = re.fullmatch(r"([vas])\[(\d+):(\d+)\]", tok)
    if m:
        return m.group(1), list(range(int(m.group(2)), int(m.group(3)) + 1))
    m = re.fullmatch(r"([vas])(\d+)", tok)
    if m:
        return m.group(1), [int(m.group(2))]
    if tok == "m0":
        return "m", [0]
    return None, []


def operands(line):
    op, _, rest = line.partition(" ")
    rest = re.sub(r"\b(offset|op_sel|op_sel_hi):\S+", "", rest)
    return op, [t for t in (x.strip() for x in rest.split(",")) if t]


def loops(lines):
    """[(label, start, end)] of every backward branch 's_cbranch_scc1 <n>b'"""
    out = []
    for i, ln in enumerate(lines):
        m = re.fullmatch(r"s_cbranch_scc1 (\d+)b", ln)
        if m:
            lab = m.group(1) + ":"
            start = max(j for j in range(i) if lines[j] == lab)
            out.append((lab, start, i))
    return out


def check(path):
    lines = parse(path)
    if lines is None:
        return None
    errs = []
    found = loops(lines)
    if not found:
        return [f"{path}: no loop found"]
    for lab, s, e in found:
        body = [ln for ln in lines[s + 1:e + 1] if not re.fullmatch(r"\d+:", ln)]
        stream = body + body                      # walk the loop twice: back-edge distances
        mfma_cycles = 0                           # matrix-pipe cycles issued so far
        mfma_count = 0
        last_operand_use = {}                     # ('v', n) -> cycles at the MFMA that read it as A/B
        queued = []                               # (mfma_count at issue, set of address regs, text)
        for idx, ln in enumerate(stream):
            op, ops = operands(ln)
            if op.startswith("v_pk_") and op.endswith("_f32"):
                errs.append(f"{path} loop {lab} R3: packed FP32 in the loop: {ln}")
            if op.startswith("v_mfma"):
                cyc = 16 if "16x16x32" in op else 32
                for t in ops[1:3]:                # A and B operands
                    f, regs = reg_range(t)
                    if f == "v":
                        for r in regs:
                            last_operand_use[("v", r)] = mfma_cycles
                mfma_cycles += cyc
                mfma_count += 1
                continue
            written, addr = [], []
            if op.startswith("ds_read") or (op.startswith("global_load") and "lds" not in op):
                f, regs = reg_range(ops[0])
                written = [(f, r) for r in regs]
                for t in ops[1:]:
                    f2, r2 = reg_range(t)
                    if f2 in ("v", "s"):
                        addr += [(f2, r) for r in r2]
            elif op.startswith("ds_write"):
                f2, r2 = reg_range(ops[0])
                addr += [(f2, r) for r in r2]
            elif op.startswith("global_load_lds"):
                for t in ops:
                    f2, r2 = reg_range(t)
                    if f2 in ("v", "s"):
                        addr += [(f2, r) for r in r2]
                addr.append(("m", 0))
            elif op.startswith(("v_", "s_")) and ops:
                f, regs = reg_range(ops[0])
                if f:
                    written = [(f, r) for r in regs]
                if op in ("s_add_u32", "s_addc_u32", "s_mov_b32", "s_lshl_b32", "s_min_u32") and ops[0].strip() == "m0":
                    written = [("m", 0)]
            # R1: LDS / global returns into a recent MFMA operand
            if op.startswith("ds_read") or (op.startswith("global_load") and "lds" not in op):
                for w in written:
                    # the MFMA that read it was issued `d` pipe cycles ago (its own cycles included)
                    d = mfma_cycles - last_operand_use[w] if w in last_operand_use else R1_MIN_CYCLES
                    if d < R1_MIN_CYCLES:
                        errs.append(f"{path} loop {lab} R1: {ln!r} rewrites {w[0]}{w[1]} {d} matrix-pipe cycles "
                                    f"after an MFMA read it (< {R1_MIN_CYCLES})")
                        break
            # R2: rewriting an address register of a queued memory instruction
            for (cnt0, regs0, txt) in queued:
                if mfma_count - cnt0 < R2_MIN_MFMAS and any(w in regs0 for w in written):
                    errs.append(f"{path} loop {lab} R2: {ln!r} rewrites an address register of {txt!r} "
                                f"{mfma_count - cnt0} MFMAs after it was issued (< {R2_MIN_MFMAS})")
            if addr:
                queued.append((mfma_count, set(addr), ln))
                queued = [q for q in queued if mfma_count - q[0] < 64]
    return errs


def main():
    bad, checked = [], 0
    for p in sys.argv[1:]:
        r = check(p)
        if r is None:
            print(f"{p}: skipped (no *_TEXT assembly block: not a generated MFMA loop)")
            continue
        checked += 1
        bad += r
    for b in bad:
        print(b)
    print(f"{checked} file(s) checked, {len(bad)} violation(s)")
    if checked == 0:
        print("nothing was checked: pass the generated headers, e.g. gpu_sdr_amd/csrc/ddc_mfma_ring*_gen.h")
        sys.exit(2)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
