"""Audit trail of every parity band the GPU tests apply (VERDICT round 4 "weak" #1 / "do this" #6).

A band-checked comparison calls record(): the PLAIN band the contract states (gradients 1e-4 of a tensor's
max-abs, loss traces 1e-4), the band the test actually USED (plain, or widened by the oracle's own f32 <-> f64 distance where the
reference arithmetic itself is that far from the exact value), that distance, and what the HIP path measured.  The session writes
all of them to gpurun_out/r05/parity_bands.json (tests/conftest.py), tools/parity_bands_summary.py turns the file into
profiles/r05_parity_bands.md, and tests/test_zz_band_audit.py fails the run when too many cases widen or a gradient band leaves 1e-3."""
import json
import os

CASES = []


def record(kind, case, plain, used, own, hip):
    """kind: 'grad' | 'trace' | 'forward'; case: what was compared (shape, ...); plain / used: the stated and the applied band;
    own: the oracle's f32 <-> f64 distance (None where it was not computed); hip: the HIP path's distance from the f32 oracle"""
    CASES.append({"test": os.environ.get("PYTEST_CURRENT_TEST", "?").split(" ")[0], "kind": kind, "case": case,
                  "plain": float(plain), "used": float(used), "own": None if own is None else float(own), "hip": float(hip),
                  "widened": bool(used > plain * (1 + 1e-12))})


def dump(path):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        json.dump({"cases": CASES}, f, indent=1)


# what the audit asserts (tests/test_zz_band_audit.py)
MAX_WIDENED_FRACTION = {"grad": 0.15, "forward": 0.15}      # (loss-trace steps: no fraction — each widened step is held to 3x the oracle's own distance instead)
MAX_BAND = {"grad": 1e-3, "forward": 2e-4, "trace": 5e-2}
