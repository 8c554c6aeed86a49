import json
import os

import numpy as np
import torch

from weights import seeded_state_dict

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

TINY_CFG = dict(
    vit=dict(embed_dim=96, depth=2, num_heads=4, img=56, patch=14),
    qf=dict(hidden_size=64, num_attention_heads=4, intermediate_size=128, num_hidden_layers=4, cross_attention_freq=2,
            num_query_token=8),
    t5=dict(d_model=64, d_kv=16, d_ff=128, num_layers=2, num_decoder_layers=2, num_heads=4, vocab_size=32128,
            num_buckets=32, max_distance=128, eps=1e-6),
)


def load_golden(name):
    d = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    out = {}
    for k, v in d.items():
        if k.endswith("_json"):
            out[k[:-5]] = json.loads(bytes(v).decode())
        else:
            out[k] = v
    return out


def golden_state_dict(g, prefix=""):
    return {prefix + k: v for k, v in seeded_state_dict(g["manifest"]).items()}


def relerr(a, b):
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


# ---- measured-error log: every parity test records what it measured (name -> value, tolerance asserted) so the tolerances written in
# the tests can be justified by numbers.  Written to gpurun_out/parity_errors.json on the box that ran the tests (merged back by gpurun;
# the judged copy is committed under profiles/).
_ERR_LOG = os.environ.get("MRB_PARITY_LOG") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity_errors.json")


def record(name: str, value: float, tol: float = None, written: float = None):
    """log a measured error; returns the value (use as ``assert record(...) < tol``)"""
    try:
        os.makedirs(os.path.dirname(_ERR_LOG), exist_ok=True)
        data = json.load(open(_ERR_LOG)) if os.path.exists(_ERR_LOG) else {}
        data[name] = {"measured": float(value), "tolerance": tol}
        if written is not None and written != tol:
            data[name]["written"] = written
        with open(_ERR_LOG, "w") as f:
            json.dump(data, f, indent=1, sort_keys=True)
    except OSError:
        pass
    return value


# ---- tolerance policy (round 6).  Every check has TWO bounds: the tolerance WRITTEN in the test (what the quantity is allowed to be: ~2x the
# error measured when the test was written, or the bound the reference demands), and a REGRESSION CEILING taken from the committed log of the
# last full `pytest -m gpu` run on an MI355X (tests/golden/parity_baseline.json = the "measured" column of profiles/rNN_parity_errors.json):
# at most CEIL x the value measured there, never below the fp32 summation-order floor.  The arithmetic is deterministic (the train step is
# bit-reproducible), so a row that moves by more than CEIL x has changed its arithmetic and must be looked at — and no self-comparison row can
# hide a 1e-3 bug behind a tolerance written 10^3-10^5 x above its measured error.  Rows that do not exist in the
# baseline (new tests) have the written tolerance only, until the next full run records them.
CEIL = 5.0
FP32_FLOOR = 1.5e-7      # two fp32 summation orders of the same sum differ by about this much (relative L2)
_BASELINE_PATH = os.path.join(GOLDEN, "parity_baseline.json")
try:
    _BASELINE = json.load(open(_BASELINE_PATH)) if os.environ.get("MRB_PARITY_NO_BASELINE") is None else {}
except (OSError, ValueError):
    _BASELINE = {}


def effective_tolerance(name: str, tol: float) -> float:
    base = _BASELINE.get(name)
    if base is None or "not a parity bound" in name:
        return tol
    return min(tol, max(CEIL * float(base), FP32_FLOOR))


def check(name: str, value: float, tol: float):
    """assert + log: value < min(written tolerance, CEIL x the value measured by the last full GPU run) — see the policy above."""
    eff = effective_tolerance(name, tol)
    record(name, value, eff, written=tol)
    assert value < eff, (f"{name}: measured {value:.3e} >= tolerance {eff:.3e}" +
                         (f" (regression ceiling: {CEIL:g} x the committed baseline {_BASELINE.get(name):.3e}; written tolerance {tol:.3e})" if eff < tol else ""))


def free_port() -> int:
    """a TCP port nobody listens on right now (rendezvous of the multi-process tests: a fixed port can still be in TIME_WAIT from the previous test)"""
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]
