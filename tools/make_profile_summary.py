"""Turn the output of tools/profile_round.sh into profiles/<tag>_final_summary.md, <tag>_final_kernel_stats.csv,
<tag>_traffic.json and <tag>_bench_line.json:  python tools/make_profile_summary.py r01"""
import collections
import csv
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = "gpurun_out/prof_%s" % tag


def pmc(sub):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(os.path.join(src, sub, "p_counter_collection.csv"))):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}


stats = list(csv.DictReader(open(os.path.join(src, "trace", "p_kernel_stats.csv"))))
fetch, write, sq = pmc("fetch"), pmc("write"), pmc("sq")
shutil.copy(os.path.join(src, "trace", "p_kernel_stats.csv"), "profiles/%s_final_kernel_stats.csv" % tag)
line = open(os.path.join(src, "bench_default.json")).read().strip().splitlines()[-1]
open("profiles/%s_bench_line.json" % tag, "w").write(line + "\n")
trace_line = open(os.path.join(src, "bench_trace_run.json")).read().strip().splitlines()[-1]
rows, traffic = [], {}
for s in stats:
    k = s["Name"]
    if not any(t in k for t in ("k_fused", "k_wgrad", "k_reduce", "k_repack", "k_small", "k16")):
        continue
    f = fetch.get(k, {}).get("FETCH_SIZE")
    w = write.get(k, {}).get("WRITE_SIZE")
    q = sq.get(k, {})
    hbm = (2 * f + w) * 1024 if f is not None and w is not None else None      # guide: FETCH_SIZE under-counts by 2 on gfx950; units KB
    rows.append("| %s | %s | %.1f | %s | %s | %s | %s | %s | %s | %s |" % (
        k[:44], s["Calls"], float(s["AverageNs"]) / 1e3, "%.0f" % f if f is not None else "-", "%.0f" % w if w is not None else "-",
        "%.4g" % hbm if hbm else "-", "%.4g" % q.get("SQ_VALU_MFMA_BUSY_CYCLES", 0), "%.4g" % q.get("SQ_INSTS_VALU", 0),
        "%.4g" % q.get("SQ_INSTS_MFMA", 0), "%.4g" % q.get("SQ_VALU_MFMA_COEXEC_CYCLES", 0)))
    short = "k_fused" if "k_fused" in k else ("k_wgrad" if "k_wgrad" in k else None)
    if short and hbm:
        traffic[short] = {"avg_us": float(s["AverageNs"]) / 1e3, "fetch_kb": f, "write_kb": w, "hbm_bytes": hbm}
json.dump(traffic, open("profiles/%s_traffic.json" % tag, "w"), indent=1)
kf = next((k for k in sq if "k_fused" in k), None)
busy = ""
if kf:
    q = sq[kf]
    frac = q["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * q["GRBM_GUI_ACTIVE"] / 8.0)
    busy = ("k_fused: MFMA pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE/8) = %.1f %%; SQ_VALU_MFMA_COEXEC_CYCLES = %d "
            "(the f32 MFMA never co-executes with VALU work on gfx950)." % (100 * frac, q.get("SQ_VALU_MFMA_COEXEC_CYCLES", 0)))
md = """# round %s — final profile of the bench workload (4x256 SIREN, 100 000 samples/step, 512^3 volume)
Commands (MI355X, ROCm 7.2; `tools/profile_round.sh`, summarised by `tools/make_profile_summary.py`):
* `rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-psnr --no-extras`
* `rocprofv3 --kernel-trace --pmc FETCH_SIZE ...`, `--pmc WRITE_SIZE ...`, `--pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE ...` (separate passes, `--steps 10 --warmup 2 --preroll 20`)
* `python3 bench.py --steps 20 --warmup 5` (the driver's flags; configs, sweeps, wall clocks and the CPU baseline included): `profiles/%s_bench_line.json`

bench line of the kernel-trace run: %s

| kernel | calls | avg us | FETCH_SIZE KB | WRITE_SIZE KB | HBM bytes/launch (2*FETCH+WRITE, guide's gfx950 correction) | MFMA busy cyc | VALU insts (incl. MFMA) | MFMA insts | COEXEC cyc |
|---|---|---|---|---|---|---|---|---|---|
%s

%s
""" % (tag[1:], tag, trace_line[:420] + " ...", "\n".join(rows), busy)
open("profiles/%s_final_summary.md" % tag, "w").write(md)
print(md)
