"""Runs last (file name): the audit of every parity band this session applied (tests/_bands.py).  Gradient and forward bands may be
widened — only by the oracle's own f32 <-> f64 distance — in at most the stated fraction of the band-checked cases, and never
beyond the stated cap; loss-trace bands of the chaotic wide-net configurations (tests/test_gpu_wide.py) are capped at 5 %."""
import pytest

from . import _bands

pytestmark = pytest.mark.gpu


def test_parity_bands_are_mostly_the_plain_ones_and_never_wider_than_stated():
    if not _bands.CASES:
        pytest.skip("no band-checked comparison ran in this session")
    lines = []
    for kind in ("grad", "forward", "trace"):
        cases = [c for c in _bands.CASES if c["kind"] == kind]
        if not cases:
            continue
        wide = [c for c in cases if c["widened"]]
        worst = max(c["used"] for c in cases)
        lines.append("%s: %d cases, %d widened (%.1f %%), widest band %.2e" % (kind, len(cases), len(wide), 100.0 * len(wide) / len(cases), worst))
        if len(cases) >= 20:      # (a partial run — one file, -k — has too few cases for a fraction to mean anything)
            assert len(wide) <= _bands.MAX_WIDENED_FRACTION[kind] * len(cases), (kind, len(wide), len(cases), [c["case"] for c in wide][:8])
        assert worst <= _bands.MAX_BAND[kind] * (1 + 1e-9), (kind, worst, [c for c in cases if c["used"] == worst][:2])
        for c in wide:
            # a widened band is justified by the oracle's own distance from its f64 instantiation, nothing else
            assert c["own"] is not None and c["used"] <= max(c["plain"], 30.0 * c["own"]) * (1 + 1e-9) and c["used"] <= 0.05, c
    print("parity bands: " + " | ".join(lines))
