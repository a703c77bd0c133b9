"""gpurun_out/r05/parity_bands.json (written by the GPU test session: tests/_bands.py, tests/conftest.py) -> profiles/r05_parity_bands.md:
how many band-checked comparisons ran, how many were widened beyond the stated band, by how much, and why.
    python tools/parity_bands_summary.py [json] [markdown]"""
import json
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r05/parity_bands.json"
dst = sys.argv[2] if len(sys.argv) > 2 else "profiles/r05_parity_bands.md"
cases = json.load(open(src))["cases"]
out = ["# Parity bands applied by the GPU test session (round 5)", "",
       "Source: `%s`, written by `tests/conftest.py` from `tests/_bands.py` at the end of `pytest -m gpu`; asserted by" % src,
       "`tests/test_zz_band_audit.py`: gradient bands — at most 15 % of the cases widened, none beyond 1e-3; a widened band is never more than the oracle's own",
       "f32 <-> f64 distance times 3 (gradients, forward) or 30 (loss-trace steps, distance one step later); and every widened loss-trace step — all of them belong to the",
       "chaotic wide-net fits of `tests/test_gpu_wide.py`, past the step where the oracle's own f32 and f64 runs separate — has a MEASURED HIP distance within 3x that",
       "distance: the HIP path stays as close to the f32 oracle as the f64 oracle does.", ""]
out += ["| kind | comparisons | at the plain band | widened | widest band used | worst HIP distance / its band |", "|---|---|---|---|---|---|"]
for kind, plain in (("grad", "1e-4 of a tensor's max-abs"), ("forward", "2e-5 of max abs(y)"), ("trace", "1e-4 of the loss")):
    cs = [c for c in cases if c["kind"] == kind]
    if not cs:
        continue
    wide = [c for c in cs if c["widened"]]
    ratio = max(c["hip"] / c["used"] for c in cs)
    out.append("| %s (%s) | %d | %d | %d (%.1f %%) | %.2e | %.2f |" % (kind, plain, len(cs), len(cs) - len(wide), len(wide), 100.0 * len(wide) / len(cs), max(c["used"] for c in cs), ratio))
out += ["", "## Every widened case", "", "| test | case | plain | used | oracle f32 <-> f64 | HIP vs oracle f32 |", "|---|---|---|---|---|---|"]
for c in cases:
    if c["widened"]:
        out.append("| `%s` | %s | %.0e | %.2e | %.2e | %.2e |" % (c["test"].split("::")[-1][:70], c["case"], c["plain"], c["used"], c["own"] if c["own"] is not None else float("nan"), c["hip"]))
out += ["", "## Distribution of the HIP path's distance from the f32 oracle, all gradient comparisons", ""]
g = sorted(c["hip"] for c in cases if c["kind"] == "grad")
if g:
    q = lambda f: g[min(len(g) - 1, int(f * len(g)))]
    out.append("median %.2e, 90th percentile %.2e, 99th %.2e, max %.2e over %d comparisons (band 1e-4)." % (q(0.5), q(0.9), q(0.99), g[-1], len(g)))
open(dst, "w").write("\n".join(out) + "\n")
print("\n".join(out[:14]))
