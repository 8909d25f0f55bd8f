"""Host-only sources under AddressSanitizer + UndefinedBehaviorSanitizer (CPU build: the GPU pool runs no
sanitizers).  The command parser is the network-facing piece of row f2 -- it reads what a client sends over
TCP -- so it gets every golden command, truncated at every length, byte-flipped and nested deeply; the
host-side helpers of the path get their edge values."""
import json
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

DRIVER = r'''
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "gsdr.h"

static int parses = 0, accepted = 0;

static void try_parse(const std::string &s) {
    gsdr_command *c = gsdr_command_parse(s.data(), (int)s.size());
    ++parses;
    if (c) {
        ++accepted;
        for (int a = 0; a < 4; ++a) {
            gsdr_param_c p;
            gsdr_antenna_info info;
            (void)gsdr_command_antenna(c, a, &p, &info);
        }
        (void)gsdr_command_device(c);
        gsdr_command_free(c);
    } else {
        (void)gsdr_command_error();
    }
}

int main(int argc, char **argv) {
    // 1. the command surface: every golden command whole, truncated, with flipped bytes, wrapped deeply
    for (int i = 1; i < argc; ++i) {
        FILE *f = std::fopen(argv[i], "rb");
        if (!f) return 2;
        std::string s;
        char buf[4096];
        size_t n;
        while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) s.append(buf, n);
        std::fclose(f);
        try_parse(s);
        for (size_t cut = 0; cut < s.size(); cut += (s.size() > 600 ? 7 : 1)) try_parse(s.substr(0, cut));
        unsigned seed = 12345u + (unsigned)i;
        for (int k = 0; k < 400; ++k) {
            std::string t = s;
            for (int j = 0; j < 1 + k % 3; ++j) {
                seed = seed * 1664525u + 1013904223u;
                t[(seed >> 8) % t.size()] = (char)(seed >> 24);
            }
            try_parse(t);
        }
        try_parse(std::string(5000, '[') + s + std::string(5000, ']'));
        try_parse("{\"A_TXRX\": " + std::string(3000, '{'));
    }
    try_parse("");
    try_parse("null");
    try_parse("{\"a\": 1e999999, \"b\": -1e-999999, \"c\": 123456789012345678901234567890}");
    try_parse(std::string("{\"x\": \"") + std::string(100000, 'a') + "\"}");
    // 2. host-side helpers at their edges
    std::vector<float> w(4096);
    for (int len : {1, 2, 3, 64, 4096}) {
        gsdr_make_sinc_window(len, 0.01f, w.data());
        gsdr_make_flat_window(len, len / 10, w.data());
        gsdr_make_flat_window(len, len, w.data());
    }
    for (int n : {1, 7, 64, 8192})
        for (int L : {1, 100, 1000000})
            for (int avg : {1, 4, 8}) {
                gsdr_buffer_helper b;
                gsdr_buffer_helper_init(&b, n, L, avg, n);
                for (int k = 0; k < 50; ++k) gsdr_buffer_helper_update(&b);
                (void)gsdr_pfb_batching(L, n, avg);
            }
    for (int ppt : {1, 7, 200, 1000000}) {
        gsdr_vna_helper v;
        gsdr_vna_helper_init(&v, ppt, 1000000);
        for (int k = 0; k < 50; ++k) gsdr_vna_helper_update(&v);
    }
    {
        int freq[5] = {0, -2147483647, 2147483647, 100, 100}, bins[5], of[5];
        float ampl[5] = {1, 2, 3, 4, 5}, oa[5];
        gsdr_pfb_tone_bins(200000000, 1230, freq, 5, bins);
        gsdr_pfb_tone_bins(1, 1, freq, 5, bins);
        (void)gsdr_tx_tone_bins(1000, freq, ampl, 5, of, oa);
        (void)gsdr_tx_tone_bins(0, freq, ampl, 5, of, oa);
        gsdr_chirp_param cp;
        gsdr_chirp_derive(200000000, -100000000, 100000000, 1000000, 1.0f, &cp);
        gsdr_chirp_derive(1, 0, 0, 0, 0.0f, &cp);
        gsdr_chirp_derive(200000000, 100000000, -100000000, 1, 1e-9f, &cp);
    }
    std::printf("parses %d accepted %d\n", parses, accepted);
    return 0;
}
'''


def test_host_sources_under_asan_and_ubsan(tmp_path):
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("no g++")
    files = [os.path.join(ROOT, "tests", "golden", f) for f in sorted(os.listdir(os.path.join(ROOT, "tests", "golden")))
             if f.startswith("cmd_") and f.endswith(".json")]
    assert files
    for f in files:
        json.load(open(f))                           # they are commands, not data files
    (tmp_path / "driver.cpp").write_text(DRIVER)
    exe = tmp_path / "driver"
    src = [os.path.join(ROOT, "gpu_sdr_amd", "csrc", f) for f in ("host_logic.cpp", "command.cpp")]
    build = subprocess.run([gxx, "-std=c++17", "-g", "-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                            "-fno-omit-frame-pointer", "-I", os.path.join(ROOT, "include"), str(tmp_path / "driver.cpp"), *src,
                            "-o", str(exe)], capture_output=True, text=True)
    if build.returncode != 0 and "sanitize" in build.stderr and "cannot find" in build.stderr:
        pytest.skip("no sanitizer runtime")
    assert build.returncode == 0, build.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    run = subprocess.run([str(exe), *files], capture_output=True, text=True, env=env, timeout=600)
    assert run.returncode == 0, (run.stdout[-500:], run.stderr[-4000:])
    assert "ERROR: AddressSanitizer" not in run.stderr and "runtime error" not in run.stderr, run.stderr[-4000:]
    assert run.stdout.startswith("parses "), run.stdout
    n_parses, n_ok = int(run.stdout.split()[1]), int(run.stdout.split()[3])
    assert n_parses > 1000 and n_ok >= len(files)           # the whole commands parse; most of the damaged ones must not
