"""Parity margins: worst per-tone relative error per test id (the id carries the engine /
kernel variant), written when the session ends.  On the GPU box only gpurun_out/ travels
back, so the record goes there; the copy under profiles/ is the committed one."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MARGINS = {}
INFO = {}        # figures recorded beside the errors (durations, ratios): never part of "worst"


def record_margin(value, label=None):
    test = os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0]
    key = test if not label else f"{test} :: {label}"
    MARGINS[key] = max(MARGINS.get(key, 0.0), float(value))


def record_info(value, label):
    test = os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0]
    key = f"{test} :: {label}"
    INFO[key] = max(INFO.get(key, 0.0), float(value))


def write_margins(exitstatus=0):
    if not MARGINS:
        return
    summary = {}
    for k, v in MARGINS.items():
        name = k.split("::")[1].split("[")[0] if "::" in k else k
        summary[name] = max(summary.get(name, 0.0), v)
    doc = {"bar": 1e-5, "metric": "max over tones of ||y - y_oracle||_2 / ||y_oracle||_2 (oracle: fp64 accumulate)",
           "worst": max(MARGINS.values()), "worst_at": max(MARGINS, key=MARGINS.get), "tests": len(MARGINS),
           "exit_status": int(exitstatus),
           "worst_per_test_function": dict(sorted(summary.items(), key=lambda t: -t[1])),
           "per_test": dict(sorted(MARGINS.items(), key=lambda t: -t[1])),
           "other_figures": dict(sorted(INFO.items()))}
    out_dir = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, "parity_margins.json"), "w") as fh:
            json.dump(doc, fh, indent=1)
    except OSError:
        pass
