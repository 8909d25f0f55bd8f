"""Writes the command fixtures tests/golden/cmd_*.json BY HAND-TRANSCRIBED RULES
from the client source (data, not code of the reference):
  * blank antenna spec: pyUSRP/USRP_files.py:449-478 (every key, its default)
  * Get_noise DIRECT branch: pyUSRP/USRP_noise.py:573-625
  * Single_VNA: pyUSRP/USRP_VNA.py:384-417
  * data_mem_mult for DIRECT: pyUSRP/USRP_files.py:683-684
"""
import json
import math
import os

HERE = os.path.dirname(os.path.abspath(__file__))


def blank():
    return dict(mode="OFF", rate=0, rf=0, gain=0, bw=0, samples=0, delay=1, burst_on=0, burst_off=0,
                buffer_len=0, freq=[0], wave_type=[0], ampl=[0], decim=0, chirp_f=[0], swipe_s=[0],
                chirp_t=[0], fft_tones=0, pf_average=4, data_mem_mult=1, tuning_mode=1)


def command():
    return {"A_TXRX": blank(), "B_TXRX": blank(), "A_RX2": blank(), "B_RX2": blank(), "device": 0}


def get_noise_direct(tones, rate, measure_t, decimation, rf, pf_average=4, tx_gain=0, delay=0):
    c = command()
    n = rate * measure_t
    tx, rx = c["A_TXRX"], c["A_RX2"]
    tx.update(mode="TX", buffer_len=1000000, gain=tx_gain, delay=1, samples=n, rate=rate, bw=2 * rate,
              wave_type=["TONES"] * len(tones), ampl=[1. / len(tones)] * len(tones), freq=list(tones),
              rf=rf, fft_tones=100)
    rx.update(mode="RX", buffer_len=1000000, gain=0, delay=1 + delay, samples=n, rate=rate, bw=2 * rate,
              wave_type=["DIRECT"] * len(tones), freq=list(tones), rf=rf, fft_tones=0,
              pf_average=pf_average, decim=decimation,
              data_mem_mult=max(math.ceil(len(tones) / max(float(decimation), 1)), 1))
    return c


def single_vna(start_f, last_f, measure_t, n_points, rate, rf, amplitude=1.0, decimation=1, delay=0):
    c = command()
    n = rate * measure_t
    for key, mode in (("A_TXRX", "TX"), ("A_RX2", "RX")):
        c[key].update(mode=mode, buffer_len=1e6, gain=0, delay=1 if mode == "TX" else 1 + delay, samples=n,
                      rate=rate, bw=2 * rate, wave_type=["CHIRP"], ampl=[amplitude], freq=[start_f],
                      chirp_f=[last_f], swipe_s=[n_points], chirp_t=[measure_t], rf=rf)
    c["A_RX2"]["decim"] = decimation
    return c


if __name__ == "__main__":
    json.dump(get_noise_direct([-40000000 + 5000000 * k + 1234 for k in range(16)], 100000000, 1, 100, 300e6),
              open(os.path.join(HERE, "cmd_get_noise_direct.json"), "w"), indent=1)
    json.dump(single_vna(-100000000, 100000000, 1.0, 1000000, 200000000, 300e6),
              open(os.path.join(HERE, "cmd_single_vna.json"), "w"), indent=1)
    print("command fixtures written")
