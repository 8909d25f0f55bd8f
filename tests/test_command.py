"""CPU tests of the pyUSRP command surface in libgsdr.so (row f2): JSON command ->
usrp_param + chk_param, ack/nack text, wire headers.  Fixtures are command
dictionaries transcribed from the client's own rules (tests/golden/make_commands.py)."""
import ctypes as C
import json
import os
import struct

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def parse(lib, text):
    raw = text.encode()
    return lib.gsdr_command_parse(raw, len(raw))


def antenna(lib, h, i):
    from gpu_sdr_amd import _lib
    p, info = _lib.ParamC(), _lib.AntennaInfoC()
    assert lib.gsdr_command_antenna(h, i, C.byref(p), C.byref(info)) == 0
    return p, info


def test_get_noise_direct_command(gsdr_lib):
    cmd = json.load(open(os.path.join(HERE, "golden", "cmd_get_noise_direct.json")))
    h = parse(gsdr_lib, json.dumps(cmd))
    assert h, gsdr_lib.gsdr_command_error()
    assert gsdr_lib.gsdr_command_device(h) == 0
    tx, txi = antenna(gsdr_lib, h, 0)
    rx, rxi = antenna(gsdr_lib, h, 2)
    off, offi = antenna(gsdr_lib, h, 1)
    assert (txi.mode, rxi.mode, offi.mode) == (0, 1, 2)
    assert rx.rate == 100000000 and rx.decim == 100 and rx.pf_average == 4 and rx.buffer_len == 1000000
    assert rx.n_wave_type == 16 and all(rx.wave_type[k] == 6 for k in range(16))      # DIRECT
    assert [rx.freq[k] for k in range(16)] == cmd["A_RX2"]["freq"]
    assert all(tx.wave_type[k] == 0 for k in range(16))                               # TONES
    np.testing.assert_allclose([txi.ampl[k] for k in range(txi.n_ampl)], 1 / 16)
    assert rxi.samples == 100000000 and rxi.data_mem_mult == 1 and rxi.rf == 300e6
    assert off.buffer_len == 0                      # OFF antennas are not normalised by chk_param
    gsdr_lib.gsdr_command_free(h)


def test_single_vna_command(gsdr_lib):
    cmd = json.load(open(os.path.join(HERE, "golden", "cmd_single_vna.json")))
    h = parse(gsdr_lib, json.dumps(cmd))
    assert h, gsdr_lib.gsdr_command_error()
    rx, rxi = antenna(gsdr_lib, h, 2)
    assert rx.n_wave_type == 1 and rx.wave_type[0] == 1 and rx.decim == 1             # CHIRP
    assert (rx.freq[0], rx.chirp_f[0], rx.swipe_s[0]) == (-100000000, 100000000, 1000000)
    assert rx.chirp_t[0] == 1.0 and rx.buffer_len == 1000000                           # 1e6 came as a float
    gsdr_lib.gsdr_command_free(h)


def test_chk_param_normalisation_and_nacks(gsdr_lib):
    from tests.golden.make_commands import command, get_noise_direct
    c = get_noise_direct([1000, 2000], 1000000, 1, 10, 0)
    c["A_RX2"]["buffer_len"] = 10                    # outside [50000, 6000000] -> default
    c["A_TXRX"]["buffer_len"] = 0
    c["A_TXRX"]["fft_tones"] = 0                     # PFB mode: fft_tones <= 0 -> 2
    c["A_TXRX"]["pf_average"] = 0                    # PFB mode: pf_average <= 0 -> 1
    h = parse(gsdr_lib, json.dumps(c))
    assert h
    tx, _ = antenna(gsdr_lib, h, 0)
    rx, _ = antenna(gsdr_lib, h, 2)
    assert (tx.buffer_len, tx.fft_tones, tx.pf_average, rx.buffer_len) == (1000000, 2, 1, 1000000)
    gsdr_lib.gsdr_command_free(h)
    # every key is mandatory, even for OFF antennas
    for key in ("rf", "pf_average", "data_mem_mult", "wave_type", "swipe_s", "mode"):
        c = command()
        del c["B_RX2"][key]
        assert not parse(gsdr_lib, json.dumps(c)), key
        assert key.encode() in gsdr_lib.gsdr_command_error()
    c = command()
    del c["device"]
    assert not parse(gsdr_lib, json.dumps(c))
    assert b"missing device ID" in gsdr_lib.gsdr_command_error()
    assert not parse(gsdr_lib, "{not json")
    # Nyquist checks (TONES / CHIRP only)
    c = get_noise_direct([1000, 2000], 1000000, 1, 10, 0)
    c["A_TXRX"]["freq"][1] = 1000001
    assert not parse(gsdr_lib, json.dumps(c))
    assert b"out of Nyquist range" in gsdr_lib.gsdr_command_error()
    c["A_TXRX"]["freq"] = [1000]                    # fewer descriptors than wave_type entries
    assert not parse(gsdr_lib, json.dumps(c))
    assert b"does not match" in gsdr_lib.gsdr_command_error()
    c = get_noise_direct([1000], 1000000, 1, 10, 0)
    c["A_RX2"]["freq"] = [5000000]                  # DIRECT is not range-checked by the reference
    assert parse(gsdr_lib, json.dumps(c))
    c["A_RX2"]["freq"] = [1000.5]                   # a non-integer where get<int> is used
    assert not parse(gsdr_lib, json.dumps(c))
    # unknown wave types (and "RAMP") fall back to NODSP like string_to_w_type
    c = get_noise_direct([1000], 1000000, 1, 10, 0)
    c["A_RX2"]["wave_type"] = ["RAMP"]
    h = parse(gsdr_lib, json.dumps(c))
    rx, _ = antenna(gsdr_lib, h, 2)
    assert rx.wave_type[0] == 4
    gsdr_lib.gsdr_command_free(h)


def test_replies_and_headers(gsdr_lib):
    from gpu_sdr_amd import _lib
    buf = C.create_string_buffer(256)
    n = gsdr_lib.gsdr_server_reply(1, b"Message received", buf, 256)
    assert buf.value.decode() == '{\n    "type": "ack",\n    "payload": "Message received"\n}\n' and n == len(buf.value)
    gsdr_lib.gsdr_server_reply(0, b"Cannot convert JSON to params", buf, 256)
    assert json.loads(buf.value) == {"type": "nack", "payload": "Cannot convert JSON to params"}
    gsdr_lib.gsdr_server_reply(1, b"EOM: end of measurement", buf, 256)
    assert "EOM" in json.loads(buf.value)["payload"]         # the client keys on this substring
    hdr8 = (C.c_ubyte * 8)()
    gsdr_lib.gsdr_format_async_header(1234, hdr8)
    assert bytes(hdr8) == struct.pack("I", 0) + struct.pack("I", 1234)   # Encode_async_message
    h = _lib.RxHeaderC(usrp_number=3, front_end_code=b"B", packet_number=77, length=2560000, errors=1, channels=256)
    hdr21 = (C.c_ubyte * 21)()
    gsdr_lib.gsdr_format_rx_header(C.byref(h), hdr21)
    header_type = np.dtype([("usrp_number", np.int32), ("front_end_code", "|S1"), ("packet_number", np.int32),
                            ("length", np.int32), ("errors", np.int32), ("channels", np.int32)])  # USRP_low_level.py:63-70
    assert header_type.itemsize == 21
    rec = np.frombuffer(bytes(hdr21), dtype=header_type)[0]
    assert (rec["usrp_number"], rec["front_end_code"], rec["packet_number"], rec["length"], rec["errors"],
            rec["channels"]) == (3, b"B", 77, 2560000, 1, 256)


def test_out_of_range_and_non_finite_values_are_nacked(gsdr_lib):
    """The command socket is network facing: values the target type cannot hold are a type
    error (the reference narrows a get<double>() with undefined behaviour there), text that is
    not plain decimal is refused like boost's stream extraction refuses it."""
    from tests.golden.make_commands import get_noise_direct

    def cmd():
        return get_noise_direct([1000, 2000], 1000000, 1, 10, 0)

    def nacked(c, key, raw=None):
        text = json.dumps(c) if raw is None else raw
        h = parse(gsdr_lib, text)
        if h:
            gsdr_lib.gsdr_command_free(h)
            return False
        return key.encode() in gsdr_lib.gsdr_command_error()

    for key, bad in [("decim", -1), ("decim", 1e300), ("rate", 1e300), ("rate", -3e9), ("fft_tones", 2.0 ** 40),
                     ("pf_average", -4), ("buffer_len", -1), ("buffer_len", 1e30), ("bw", 1e12), ("gain", -1e12),
                     ("data_mem_mult", -1), ("rf", -1.0), ("rf", 1e300), ("samples", -5), ("tuning_mode", 2 ** 40)]:
        c = cmd()
        c["A_RX2"][key] = bad
        assert nacked(c, key), (key, bad)
    # numeric text: nan / inf / hex are not numbers for a stream extraction
    for key, bad in [("rate", "nan"), ("decim", "inf"), ("buffer_len", "0x10"), ("gain", "-inf"), ("delay", "NaN")]:
        c = cmd()
        c["A_RX2"][key] = bad
        assert nacked(c, key), (key, bad)
    # bare nan / Infinity tokens are not JSON at all
    for tok in ("NaN", "Infinity", "-Infinity", "0x20"):
        raw = json.dumps(cmd()).replace('"decim": 10', '"decim": ' + tok)
        assert '"decim": ' + tok in raw
        assert nacked(None, "JSON", raw), tok
    # list elements are narrowed to int: range-checked first (2^32+5 must not become 5)
    for key, bad in [("freq", 2 ** 32 + 5), ("freq", -2 ** 31), ("chirp_f", 2 ** 31), ("swipe_s", -2 ** 40)]:
        c = cmd()
        c["A_RX2"][key] = [bad]
        assert nacked(c, key), (key, bad)
    # INT_MIN + 1 is representable and its magnitude is too: accepted for DIRECT (not Nyquist-checked)
    c = cmd()
    c["A_RX2"]["freq"] = [-(2 ** 31) + 1, 5]
    h = parse(gsdr_lib, json.dumps(c))
    assert h
    gsdr_lib.gsdr_command_free(h)
    # a TX tone comb with fewer amplitudes (or frequencies) than wave_type entries
    c = cmd()
    assert c["A_TXRX"]["mode"] == "TX" and c["A_TXRX"]["wave_type"] == ["TONES", "TONES"]
    c["A_TXRX"]["ampl"] = [0.5]
    assert nacked(c, "does not match")
    c = cmd()
    c["A_TXRX"]["ampl"] = []
    assert nacked(c, "does not match")
    # the untouched command still parses
    h = parse(gsdr_lib, json.dumps(cmd()))
    assert h
    gsdr_lib.gsdr_command_free(h)
