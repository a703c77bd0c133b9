"""Runs last (file name): the audit of every parity band this session applied (tests/_bands.py).  Gradient and forward bands may be
widened — only by the oracle's own f32 <-> f64 distance — in at most the stated fraction of the band-checked cases, and never
beyond the stated cap; the loss-trace steps of the chaotic wide-net configurations (tests/test_gpu_wide.py: most of them are past the point where
the oracle's own f32 and f64 instantiations separate) are held to 3x the oracle's own f32 <-> f64 distance, step by step."""
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
        if kind != "trace" and len(cases) >= 20:      # (a partial run — one file, -k — has too few cases for a fraction to mean anything)
            assert len(wide) <= _bands.MAX_WIDENED_FRACTION[kind] * len(cases), (kind, len(wide), len(cases), [c["case"] for c in wide][:8])
        assert worst <= _bands.MAX_BAND[kind] * (1 + 1e-9), (kind, worst, [c for c in cases if c["used"] == worst][:2])
        for c in wide:
            # a widened band is justified by the oracle's own distance from its f64 instantiation, nothing else
            assert c["own"] is not None and c["used"] <= max(c["plain"], 30.0 * c["own"]) * (1 + 1e-9) and c["used"] <= 0.05, c
            if kind == "trace":
                # the loss traces of the chaotic wide-net fits (tests/test_gpu_wide.py: every entry of this kind is one step of those): the
                # band applied is generous (30x), the MEASURED distance is not allowed to be — the HIP path must stay as close to the f32
                # oracle as the f64 oracle does (3x its distance one step later), step by step
                assert c["hip"] <= max(c["plain"], 3.0 * c["own"]), c
    print("parity bands: " + " | ".join(lines))
