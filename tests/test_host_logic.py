"""CPU tests of libgsdr.so's host-side logic against the oracle, and of the
C-ABI surface (every symbol include/gsdr.h declares must be exported).
No GPU compute here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(gsdr_lib):
    hdr = open(os.path.join(ROOT, "include", "gsdr.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(gsdr_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 25
    from gpu_sdr_amd import _lib
    bound = {name for name, _, _ in _lib.SIGNATURES}
    assert declared == bound, (declared - bound, bound - declared)
    for name in declared:
        assert hasattr(gsdr_lib, name), name
    assert gsdr_lib.gsdr_abi_version() == 1


def test_cpp_shim_header_compiles_with_gxx(tmp_path):
    """include/USRP_demodulator.hpp is what the reference's server code would
    include instead of its CUDA class: it must compile with plain g++."""
    src = tmp_path / "t.cpp"
    src.write_text('#include "USRP_demodulator.hpp"\n'
                   "int main(){ param p; p.buffer_len = 100; p.rate = 1000; p.decim = 0;\n"
                   " p.wave_type.push_back(NODSP); RX_wrapper w; (void)w;\n"
                   " return sizeof(RX_buffer_demodulator) > 0 ? 0 : 1; }\n")
    import subprocess
    subprocess.check_call(["g++", "-std=c++11", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), str(src)])
    # and the TX side: include/USRP_buffer_generator.hpp (class TX_buffer_generator)
    src2 = tmp_path / "t2.cpp"
    src2.write_text('#include "USRP_buffer_generator.hpp"\n'
                    "int main(){ param p; p.buffer_len = 100; p.rate = 1000; p.wave_type.push_back(TONES);\n"
                    " p.freq.push_back(10); p.ampl.push_back(0.5f);\n"
                    " return sizeof(TX_buffer_generator) > 0 ? 0 : 1; }\n")
    subprocess.check_call(["g++", "-std=c++11", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), str(src2)])


def test_c_header_is_plain_c(tmp_path):
    """include/gsdr.h is the FFI boundary: it must compile as C (C99, pedantic),
    which is what a cgo/JNI/ctypes binding generator would feed on."""
    src = tmp_path / "t.c"
    src.write_text('#include "gsdr.h"\n'
                   "int main(void){ gsdr_param_c p; gsdr_buffer_helper b; gsdr_vna_helper v; gsdr_chirp_param c;\n"
                   " (void)p; (void)b; (void)v; (void)c; return GSDR_ABI_VERSION == 1 && GSDR_PIPELINE_DEPTH >= 2 ? 0 : 1; }\n")
    import subprocess
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-fsyntax-only",
                           "-I", os.path.join(ROOT, "include"), str(src)])


@pytest.mark.parametrize("length,fc", [(40, 0.0375), (400, 0.75 / 200), (4000, 0.75 / 2000),
                                        (41, 0.0375), (30, 1. / 20), (12, 1. / 6), (1, 0.1), (2, 0.25)])
def test_sinc_window_bit_exact(gsdr_lib, oracle_mod, length, fc):
    import gpu_sdr_amd as g
    np.testing.assert_array_equal(g.make_sinc_window(length, np.float32(fc)),
                                  oracle_mod.make_sinc_window(length, np.float32(fc)))


@pytest.mark.parametrize("length,side", [(20, 2), (200, 20), (7, 0), (300, 30), (10, 1)])
def test_flat_window_bit_exact(gsdr_lib, oracle_mod, length, side):
    import gpu_sdr_amd as g
    np.testing.assert_array_equal(g.make_flat_window(length, side), oracle_mod.make_flat_window(length, side))


def test_buffer_helper_exhaustive_small(gsdr_lib, oracle_mod):
    import gpu_sdr_amd as g
    for nfft in (1, 2, 3, 7, 10, 16, 33):
        for avg in (1, 2, 4, 5):
            for L in (1, 5, 50, 103, 128, 1000):
                a = g.buffer_helper(nfft, L, avg, 3)
                b = oracle_mod.BufferHelper(nfft, L, avg, 3)
                for _ in range(12):
                    assert a.state() == b.state(), (nfft, avg, L)
                    a.update()
                    b.update()


def test_buffer_helper_baseline_shapes(gsdr_lib, oracle_mod):
    import gpu_sdr_amd as g
    for nfft, avg, L in [(1230, 4, 1_000_000), (1000, 4, 1_000_000), (100, 8, 1_000_000), (65536, 2, 1_000_000)]:
        a = g.buffer_helper(nfft, L, avg, 256)
        b = oracle_mod.BufferHelper(nfft, L, avg, 256)
        consumed = 0
        for _ in range(40):
            assert a.state() == b.state()
            consumed += a.current_batch
            a.update()
            b.update()
        # no frame is lost or duplicated: frames emitted == floor((40 L - avg*nfft)/nfft)+1 or one less
        assert abs(consumed - ((40 * L - avg * nfft - 1) // nfft + 1)) <= 1


def test_vna_helper_exhaustive_small(gsdr_lib, oracle_mod):
    import gpu_sdr_amd as g
    for ppt in (1, 2, 7, 14, 200, 300, 999):
        for L in (1000, 1001, 5000):
            a = g.VNA_decimator_helper(ppt, L)
            b = oracle_mod.VnaHelper(ppt, L)
            total = 0
            for c in range(15):
                assert a.state() == b.state()
                total += a.valid_size
                a.update()
                b.update()
            assert total == (15 * L) // ppt


def test_tone_bins_match_literal_scan(gsdr_lib, oracle_mod):
    import gpu_sdr_amd as g
    rng = np.random.default_rng(7)
    for rate, nfft in [(1000, 10), (1200, 12), (200_000_000, 1230), (100_000_000, 1000),
                       (200_000_000, 65536), (1000, 7), (999, 9), (200_000_000, 3)]:
        f = rng.integers(-rate // 2 - rate // nfft, rate // 2 + rate // nfft, size=300).astype(np.int32)
        bs = rate / nfft
        centres = (np.arange(nfft) * bs - bs * (nfft // 2))
        extra = np.concatenate([centres[:50], centres[-50:], centres[:20] + 1, centres[:20] - 1])
        f = np.concatenate([f, np.round(extra).astype(np.int32)])
        np.testing.assert_array_equal(g.pfb_tone_bins(rate, nfft, f), oracle_mod.pfb_tone_bins(rate, nfft, f))
    assert g.pfb_batching(1_000_000, 1230, 4) == oracle_mod.pfb_batching(1_000_000, 1230, 4) == 823


def test_chirp_derive_matches_oracle(gsdr_lib, oracle_mod):
    import gpu_sdr_amd as g
    cases = [(200_000_000, -100_000_000, 100_000_000, 1_000_000, 1.0),
             (200_000_000, -100_000_000, 100_000_000, 1_000_000, 1.5),
             (200_000_000, -90_000_000, 90_000_000, 1000, 3.5e-5),
             (100_000_000, 10_000_000, -40_000_000, 5000, 0.01),   # downward sweep: wraps
             (1_000_000, -100_000, 100_000, 0, 0.001),             # swipe_s < 1 -> chirp_t*rate steps
             (1_000_000, 0, 0, 1, 0.001),                          # single step: division by zero slope
             (200_000_000, 200_000_000, 100_000_000, 100, 1e-9)]   # f0 overflow, length clamp
    for c in cases:
        a, b = g.chirp_derive(*c), oracle_mod.chirp_params(*c)
        assert (a.num_steps, a.length, a.chirpness, a.f0) == (b.num_steps, b.length, b.chirpness, b.f0), c


def test_chirp_derive_tx_matches_oracle_and_recipe_b(gsdr_lib, oracle_mod):
    """The TX generator's own derivation (ref: cpp/USRP_buffer_generator.cpp:107-129): three statements of it
    agree -- the library, the C oracle, the numpy restatement -- including the corner the RX derivation does
    not have: swipe_s > chirp_t * rate resets num_steps BEFORE the slope is computed (:118-125)."""
    import gpu_sdr_amd as g
    from gpu_sdr_amd.demodulator import chirp_derive_tx
    from oracle import recipe_b
    cases = [(200_000_000, -100_000_000, 100_000_000, 1_000_000, 1.0),
             (200_000_000, -80_000_000, 80_000_000, 10_000, 2e-5),      # 4000 samples for 10000 steps: reset
             (1_000_000, -100_000, 100_000, 5000, 0.001),               # reset
             (1_000_000, -100_000, 100_000, 0, 0.001),                  # swipe_s < 1
             (100_000_000, 10_000_000, -40_000_000, 5000, 0.01),
             (200_000_000, 200_000_000, 100_000_000, 100, 1e-9)]
    differ = 0
    for c in cases:
        a, b, r = chirp_derive_tx(*c), oracle_mod.chirp_params_tx(*c), recipe_b.chirp_params_tx(*c)
        assert (a.num_steps, a.length, a.chirpness, a.f0) == (b.num_steps, b.length, b.chirpness, b.f0), c
        if b.f0 != -2**31:       # (recipe_b states f0 for in-range values only)
            assert (b.num_steps, b.length, b.chirpness, b.f0) == r, c
        rx = g.chirp_derive(*c)
        differ += (rx.num_steps, rx.chirpness) != (a.num_steps, a.chirpness)
    assert differ >= 2     # the reset cases really are different from the RX derivation


def _param_c(**kw):
    from gpu_sdr_amd import _lib
    keep = []
    pc = _lib.ParamC()
    for k in ("rate", "decim", "fft_tones", "pf_average", "buffer_len"):
        setattr(pc, k, kw.get(k, 0))
    for name, ct in (("wave_type", C.c_int), ("freq", C.c_int), ("chirp_t", C.c_float),
                     ("chirp_f", C.c_int), ("swipe_s", C.c_int)):
        vals = kw.get(name, [])
        arr = (ct * max(len(vals), 1))(*vals)
        keep.append(arr)
        setattr(pc, name, C.cast(arr, C.POINTER(ct)))
        setattr(pc, "n_" + name, len(vals))
    pc.device_index = -1
    return pc, keep


def test_create_rejects_what_the_reference_exits_on(gsdr_lib):
    """Mode checks happen before any device call (ref: USRP_demodulator.cpp:27-39)."""
    pc, keep = _param_c(rate=1000, buffer_len=100, wave_type=[6, 0], freq=[1, 2])
    assert not gsdr_lib.gsdr_demod_create(C.byref(pc))
    assert b"Mixed RX buffer demodulation" in gsdr_lib.gsdr_last_error(None)
    pc, keep = _param_c(rate=1000, buffer_len=100, wave_type=[1, 1], freq=[1, 2])
    assert not gsdr_lib.gsdr_demod_create(C.byref(pc))
    assert b"Multiple chirp" in gsdr_lib.gsdr_last_error(None)
    assert not gsdr_lib.gsdr_demod_create(None)


def test_python_mirror_raises_with_reference_message(gsdr_lib):
    import gpu_sdr_amd as g
    p = g.param(rate=1000, buffer_len=100, wave_type=[g.w_type.DIRECT, g.w_type.TONES], freq=[1, 2])
    with pytest.raises(g.GsdrError, match="Mixed RX buffer demodulation"):
        g.RX_buffer_demodulator(p)
    assert g.string_to_w_type("DIRECT") == g.w_type.DIRECT
    assert g.string_to_w_type("RAMP") == g.w_type.NODSP  # not parsed by the reference
    assert g.string_to_w_type("bogus") == g.w_type.NODSP
    assert g.w_type_to_str(g.w_type.CHIRP) == "CHIRP"


def test_no_cpu_fallback_in_product():
    """The product package must never import the oracle."""
    pkg = os.path.join(ROOT, "gpu_sdr_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp")):
                txt = open(os.path.join(dp, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, f
                assert "gsdr_oracle" not in txt and "liboracle" not in txt, f


@pytest.mark.parametrize("gen,header", [("gen_ddc_mfma_ring.py", "ddc_mfma_ring_gen.h"),
                                        ("gen_ddc_mfma_ring16.py", "ddc_mfma_ring16_gen.h"),
                                        ("gen_ddc_mfma_ring16w8.py", "ddc_mfma_ring16w8_gen.h"),
                                        ("gen_ddc_mfma_ring16p.py", "ddc_mfma_ring16p_gen.h"),
                                        ("gen_ddc_steps.py", "ddc_steps_gen.h")])
def test_generated_headers_are_current(gen, header):
    """The committed assembly headers are exactly what their generators produce
    (the two MFMA generators print, gen_ddc_steps.py rewrites its file)."""
    import subprocess
    import sys
    path = os.path.join(ROOT, "gpu_sdr_amd", "csrc", header)
    committed = open(path).read()
    env = {k: v for k, v in os.environ.items() if not k.startswith("GEN_")}
    gen, *flags = gen.split()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", gen)] + flags, capture_output=True, text=True,
                         check=True, env=env).stdout
    produced = open(path).read() if gen == "gen_ddc_steps.py" else out
    if produced != committed:
        open(path, "w").write(committed)      # leave the tree as it was
    assert produced == committed


RING_HEADERS = ["ddc_mfma_ring_gen.h", "ddc_mfma_ring16_gen.h", "ddc_mfma_ring16w8_gen.h",
                "ddc_mfma_ring16p_gen.h"]


def test_generated_loops_obey_the_hazard_rules(tmp_path):
    """tools/check_asm_rules.py re-derives rules R1-R3 of DESIGN.md section 4.1 (operand registers
    not rewritten under an MFMA, address registers not rewritten under a queued memory
    instruction, no packed FP32) from the emitted instruction stream of every ring loop -- and
    notices when one is broken (three seeded violations)."""
    import importlib.util
    import re
    spec = importlib.util.spec_from_file_location("check_asm_rules", os.path.join(ROOT, "tools", "check_asm_rules.py"))
    chk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(chk)
    for h in RING_HEADERS:
        assert chk.check(os.path.join(ROOT, "gpu_sdr_amd", "csrc", h)) == [], h
    src = open(os.path.join(ROOT, "gpu_sdr_amd", "csrc", "ddc_mfma_ring16_gen.h")).read()
    lines = src.split("\n")
    loop_at = next(i for i, ln in enumerate(lines) if ln.strip().startswith('"1:'))
    mf = next(i for i in range(loop_at, len(lines)) if "v_mfma" in lines[i])
    a_op = re.search(r"v_mfma\S+ v\[\d+:\d+\], (v\[\d+:\d+\])", lines[mf]).group(1)
    rd = next(i for i in range(loop_at, len(lines)) if "ds_read_b128" in lines[i])
    addr = re.search(r"ds_read_b128 v\[\d+:\d+\], (v\d+)", lines[rd]).group(1)
    seeded = {
        "R1": (mf + 1, f'    "ds_read_b128 {a_op}, {addr}\\n\\t" \\'),
        "R2": (rd + 1, f'    "v_add_u32 {addr}, s45, v0\\n\\t" \\'),
        "R3": (mf + 1, '    "v_pk_mul_f32 v[20:21], v[20:21], v[22:23]\\n\\t" \\'),
    }
    for rule, (at, text) in seeded.items():
        bad = tmp_path / f"bad_{rule}.h"
        bad.write_text("\n".join(lines[:at] + [text] + lines[at:]))
        errs = chk.check(str(bad))
        assert any(f" {rule}:" in e for e in errs), (rule, errs[:3])


def test_assembly_kernels_keep_two_waves_per_simd(gsdr_lib, tmp_path):
    """The matrix-core kernels are scheduled for two waves per SIMD (512 registers per lane: at most
    256 VGPRs + AGPRs per wave).  C++ around the assembly block that needs a few registers too many
    makes the compiler park them in extra AGPRs, the kernel still runs -- at one wave per SIMD and
    10 % slower (it happened with a wider absmax slot).  Read the register counts out of the code
    object of the library as built."""
    import shutil
    import subprocess
    from gpu_sdr_amd import _lib
    llvm = "/opt/rocm/lib/llvm/bin"
    if not os.path.exists(os.path.join(llvm, "llvm-objdump")):
        pytest.skip("no ROCm llvm tools")
    so = tmp_path / "libgsdr.so"
    shutil.copy(_lib.LIB_PATH, so)
    subprocess.run([os.path.join(llvm, "llvm-objdump"), "--offloading", str(so)], check=True, capture_output=True, cwd=tmp_path)
    seen = {}
    for f in tmp_path.iterdir():
        if "amdgcn" not in f.name:
            continue
        notes = subprocess.run([os.path.join(llvm, "llvm-readelf"), "--notes", str(f)], check=True, capture_output=True, text=True).stdout
        for blk in notes.split("- .agpr_count:")[1:]:
            name = re.search(r"\.name:\s+(\S+)", blk)
            vg = re.search(r"\.vgpr_count:\s+(\d+)", blk)
            if name and vg:
                seen[name.group(1)] = int(vg.group(1))       # .vgpr_count is VGPRs + AGPRs on gfx90a and later
    asm_kernels = [k for k in seen if re.search(r"ddc_mfma_ring(16|16w8|16p)?_kernel", k)]
    assert len(asm_kernels) == 4, sorted(seen)
    for k in asm_kernels:
        assert seen[k] <= 256, (k, seen[k])
    # and no kernel of the library spills to scratch or reads through the flat aperture: an LDS pointer
    # that is advanced in a loop silently turns ds_read into flat_load (it did, in the in-LDS FFT: +50 %)
    for f in tmp_path.iterdir():
        if "amdgcn" not in f.name:
            continue
        dis = subprocess.run([os.path.join(llvm, "llvm-objdump"), "-d", str(f)], check=True, capture_output=True, text=True).stdout
        bad = [ln.strip() for ln in dis.splitlines() if re.search(r"\b(flat_load|flat_store|scratch_load|scratch_store)", ln)]
        assert not bad, (f.name, bad[:4])
        # and packed FP32 only where it is meant to be: a wave executing v_pk_*_f32 returns stale values now
        # and then while a wave of the matrix-core loop shares its SIMD (rule 3 --
        # across kernels of concurrently used handles too), so every kernel that can meet that loop is built
        # with no-packed-fp32-ops -- since round 3 the synthetic sources / TX chirp generator too (TX and RX share
        # the GPU).  Exempt: the packed-FP32 DDC itself (engine of GSDR_DDC_MFMA=0, where no matrix-core loop
        # runs) and the compiler-scheduled matrix-core kernel (A/B runs only)
        sym, packed = "?", {}
        for ln in dis.splitlines():
            m = re.match(r"^[0-9a-f]+ <(\S+)>:", ln)
            if m:
                sym = m.group(1)
            elif re.search(r"\bv_pk_[a-z]+_f32\b", ln):
                packed[sym] = packed.get(sym, 0) + 1
        allowed = ("ddc_flat_kernel", "ddc_mfma_kernelILi")
        offenders = {k: v for k, v in packed.items() if not any(a in k for a in allowed)}
        assert not offenders, (f.name, offenders)


def test_pfb_lds_stage_plan(gsdr_lib):
    """gsdr_pfb_lds_stages: the radices of the in-LDS transform multiply to the frame length, prime
    factors above 13 come first (the largest in front: its stage needs no twiddles), then 16s (long frames), 8s, 4, 6 / 10 (a 2 joined
    with a 3 / 5), 2, and the small odd primes; lengths above 8192 points or with a prime factor above 127 are refused (they take
    the other paths)."""
    def stages(n):
        r = (C.c_int * 16)()
        k = gsdr_lib.gsdr_pfb_lds_stages(n, r)
        return None if k < 0 else [r[i] for i in range(k)]
    assert stages(1) == []
    assert stages(1024) == [8, 8, 8, 2]          # (round 3) 8 = 4 x 2 in registers: four stages instead of five
    assert stages(2048) == [8, 8, 8, 4]
    assert stages(4096) == [16, 16, 16]          # (round 3) two radix-4 levels per LDS round trip from 4096 points on
    assert stages(1230) == [41, 6, 5]            # the single 2 joins the 3
    assert stages(17 * 19 * 4) == [19, 17, 4]
    assert stages(127 * 8) == [127, 8]
    assert stages(8192) == [16, 16, 16, 2]
    assert stages(131 * 4) is None and stages(8193) is None and stages(16384) is None and stages(0) is None
    assert stages(4099) is None                      # prime above 127
    for n in range(1, 8193):
        st = stages(n)
        if st is None:
            m, q = n, 2
            while q * q <= m:
                while m % q == 0:
                    m //= q
                q += 1
            assert m > 127, n                       # only a large prime factor keeps a length out
            continue
        assert int(np.prod(st, dtype=np.int64)) == n if st else n == 1
        big = [r for r in st if r > 13]
        assert st[:len(big)] == sorted(big, reverse=True)
        assert all(r in (2, 3, 4, 5, 6, 7, 8, 10, 11, 13, 16) for r in st[len(big):])
