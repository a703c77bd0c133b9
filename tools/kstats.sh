#!/bin/bash
# per-kernel average durations of a short bench run (rocprofv3 --kernel-trace --stats):  tools/kstats.sh <tag> <bench args...>
tag=$1; shift
export TMPDIR=/tmp
out=gpurun_out/kstats_$tag
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python3 bench.py "$@" --no-extras --no-cpu-baseline --no-psnr > $out/bench.json 2> $out/err.txt
f=$(find $out -name "p_kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:12]:
    print("%-70s calls %6s avg %9.1f us  %5.1f%%" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
