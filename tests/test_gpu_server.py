"""End to end over TCP (row f2): a client that frames its command exactly like
pyUSRP (Encode_async_message, USRP_connections.py:484-498; packets decoded with
header_type, USRP_low_level.py:63-70) talks to tools/gsdr_server.cpp running the
software loop-back on the GPU."""
import json
import os
import socket
import struct
import subprocess
import sys
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

HEADER = np.dtype([("usrp_number", np.int32), ("front_end_code", "|S1"), ("packet_number", np.int32),
                   ("length", np.int32), ("errors", np.int32), ("channels", np.int32)])


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def recv_all(sock, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(n - len(buf))
        if not chunk:
            raise ConnectionError("socket closed")
        buf += chunk
    return bytes(buf)


def recv_async(sock):
    zero, n = struct.unpack("II", recv_all(sock, 8))
    assert zero == 0
    return json.loads(recv_all(sock, n))


def connect(port, tries=100):
    for _ in range(tries):
        try:
            return socket.create_connection(("127.0.0.1", port), timeout=180)    # (generous: a shared box has stalled a reply for a minute once)
        except OSError:
            time.sleep(0.1)
    raise ConnectionError(port)


@pytest.fixture()
def server(cuda_device, gsdr_lib):
    exe = os.path.join(ROOT, "gpu_sdr_amd", "gsdr_server")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "gpu_sdr_amd", "csrc")])
    pa, pd = free_port(), free_port()
    proc = subprocess.Popen([exe, "--async", str(pa), "--data", str(pd), "--device", "0", "--sw_loop"])
    data = connect(pd)
    asyn = connect(pa)
    yield asyn, data
    asyn.close()
    data.close()
    try:
        proc.wait(timeout=20)
    except subprocess.TimeoutExpired:
        proc.kill()


def send_command(asyn, cmd):
    payload = json.dumps(cmd).encode()
    asyn.sendall(struct.pack("I", 0) + struct.pack("I", len(payload)) + payload)


def test_get_noise_direct_over_tcp(server):
    from make_commands import get_noise_direct
    asyn, data = server
    tones = [-40000000 + 5000000 * k + 1234 for k in range(16)]
    cmd = get_noise_direct(tones, 100000000, 0.03, 100, 300e6)          # 3 buffers of 1 M samples
    send_command(asyn, cmd)
    assert recv_async(asyn) == {"type": "ack", "payload": "Message received"}
    rows = []
    for k in range(3):
        h = np.frombuffer(recv_all(data, 21), dtype=HEADER)[0]
        assert (h["usrp_number"], h["front_end_code"], h["packet_number"], h["errors"], h["channels"]) == \
            (0, b"B", k, 0, 16)
        assert h["length"] == 16 * 10000
        z = np.frombuffer(recv_all(data, int(h["length"]) * 8), dtype=np.complex64)
        rows.append(z.reshape(-1, 16))                                    # Packets_to_file: (samples, channels)
    reply = recv_async(asyn)
    assert reply["type"] == "ack" and "EOM" in reply["payload"]
    y = np.concatenate(rows)
    # TX comb with ampl 1/16 per tone, looped back and demodulated: every channel sits at 1/16 ...
    assert np.abs(y[8:] - 1.0 / 16).max() < 2e-4
    # ... and, packet for packet, on what the oracle computes for the same chain: the reference's TX tone
    # comb (tone_gen: kernels.cu:589-684) into the DIRECT demodulator (process_direct: USRP_demodulator.cpp:400-464)
    import oracle
    oracle.build()
    rate, L = 100000000, 1000000
    ref = oracle.Direct(tones, rate, 100, cmd["A_RX2"]["pf_average"], L)
    want = np.concatenate([ref.process(oracle.tone_gen(tones, [1.0 / 16] * 16, rate, k * L, L)) for k in range(3)])
    err = np.linalg.norm(y - want, axis=0) / np.linalg.norm(want, axis=0)
    assert err.max() <= 1e-5, err.max()
    # a malformed command is nack'ed and the server keeps serving
    bad = dict(cmd)
    bad = {k: v for k, v in cmd.items() if k != "B_RX2"}
    send_command(asyn, bad)
    assert recv_async(asyn) == {"type": "nack", "payload": "Cannot convert JSON to params"}


def test_single_vna_over_tcp(server):
    from make_commands import single_vna
    asyn, data = server
    cmd = single_vna(-80000000, 80000000, 0.01, 10000, 200000000, 300e6, amplitude=0.5)   # 2 buffers, ppt 200
    send_command(asyn, cmd)
    assert recv_async(asyn)["type"] == "ack"
    got = []
    for k in range(2):
        h = np.frombuffer(recv_all(data, 21), dtype=HEADER)[0]
        assert h["channels"] == 1 and h["front_end_code"] == b"B" and h["packet_number"] == k
        assert h["length"] == 5000
        got.append(np.frombuffer(recv_all(data, int(h["length"]) * 8), dtype=np.complex64))
    assert "EOM" in recv_async(asyn)["payload"]
    np.testing.assert_allclose(np.concatenate(got), 0.5 + 0j, rtol=0, atol=3e-6)     # flat S21 of the loop-back
    # the same chain through the oracle: TX chirp law (kernels.cu:335-372) into the chirp demodulator + lock-in
    import oracle
    oracle.build()
    rate, L = 200000000, 1000000
    rx = cmd["A_RX2"]
    cp = oracle.chirp_params(rate, rx["freq"][0], rx["chirp_f"][0], rx["swipe_s"][0], rx["chirp_t"][0])
    ref = oracle.Chirp(rate, rx["freq"][0], rx["chirp_f"][0], rx["swipe_s"][0], rx["chirp_t"][0], rx["decim"], L)
    want = np.concatenate([ref.process(oracle.chirp_gen(cp, k * L, L, 0.5)) for k in range(2)])
    got = np.concatenate(got)
    assert got.shape == want.shape
    assert np.linalg.norm(got - want) / np.linalg.norm(want) <= 1e-5


def test_two_front_ends_in_one_command_and_burst_buffer_length(server):
    """A_RX2 (DIRECT, fed by the A_TXRX tone comb) and B_RX2 (CHIRP, fed by the B_TXRX chirp) active in ONE command:
    the reference builds a demodulator per front-end and runs a link thread each (ref:
    cpp/USRP_server_link_threads.cpp:121,136,325-336); their packets share the data socket -- tagged 'B' for
    front-end A and 'D' for B (ref: cpp/USRP_hardware_manager.cpp:1413-1418), each with its own packet counter --
    and the EOM comes after both are done.  Front-end B is in burst mode: its buffer is one burst long,
    buffer_len = burst_on * rate (ref: link_threads.cpp:99-102), whatever the command's buffer_len says.
    Payloads against the oracle chains, <= 1e-5 per channel."""
    from make_commands import get_noise_direct, single_vna
    asyn, data = server
    tones = [-40000000 + 5000000 * k + 1234 for k in range(16)]
    cmd = get_noise_direct(tones, 100000000, 0.03, 100, 300e6)                    # A: 3 buffers of 1 M samples
    vna = single_vna(-80000000, 80000000, 0.01, 10000, 200000000, 300e6, amplitude=0.5)   # B: 2 M samples
    for key_b, key_a in (("B_TXRX", "A_TXRX"), ("B_RX2", "A_RX2")):
        cmd[key_b] = dict(vna[key_a], burst_on=0.0025, burst_off=0.001)           # bursts of 500 000 samples
    send_command(asyn, cmd)
    assert recv_async(asyn) == {"type": "ack", "payload": "Message received"}
    got = {b"B": [], b"D": []}
    want_packets = {b"B": 3, b"D": 4}
    while any(len(got[c]) < want_packets[c] for c in got):
        h = np.frombuffer(recv_all(data, 21), dtype=HEADER)[0]
        code = bytes(h["front_end_code"])
        assert code in got and h["usrp_number"] == 0 and h["errors"] == 0
        assert h["packet_number"] == len(got[code])                                # a counter per front-end
        assert h["channels"] == (16 if code == b"B" else 1)
        assert h["length"] == (16 * 10000 if code == b"B" else 2500)               # B: 500 000 samples / ppt 200
        got[code].append(np.frombuffer(recv_all(data, int(h["length"]) * 8), dtype=np.complex64))
    reply = recv_async(asyn)
    assert reply["type"] == "ack" and "EOM" in reply["payload"]
    import oracle
    oracle.build()
    # front-end A: the TX tone comb into the DIRECT demodulator
    rate, L = 100000000, 1000000
    ref = oracle.Direct(tones, rate, 100, cmd["A_RX2"]["pf_average"], L)
    want = np.concatenate([ref.process(oracle.tone_gen(tones, [1.0 / 16] * 16, rate, k * L, L)) for k in range(3)])
    y = np.concatenate(got[b"B"]).reshape(-1, 16)
    assert (np.linalg.norm(y - want, axis=0) / np.linalg.norm(want, axis=0)).max() <= 1e-5
    # front-end B: the TX chirp (the TX side's own parameter derivation) into the chirp demodulator + lock-in,
    # in buffers of one burst
    rate, L = 200000000, 500000
    rx = cmd["B_RX2"]
    cp = oracle.chirp_params_tx(rate, rx["freq"][0], rx["chirp_f"][0], rx["swipe_s"][0], rx["chirp_t"][0])
    refc = oracle.Chirp(rate, rx["freq"][0], rx["chirp_f"][0], rx["swipe_s"][0], rx["chirp_t"][0], rx["decim"], L)
    wantc = np.concatenate([refc.process(oracle.chirp_gen(cp, k * L, L, 0.5)) for k in range(4)])
    gotc = np.concatenate(got[b"D"])
    assert gotc.shape == wantc.shape
    assert np.linalg.norm(gotc - wantc) / np.linalg.norm(wantc) <= 1e-5
