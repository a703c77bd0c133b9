#!/bin/bash
# per-kernel averages of a short bench run for several library builds:  tools/kstats_lib.sh "<bench args>" libA.so libB.so ...
args=$1; shift
export TMPDIR=/tmp
for lib in "$@"; do
  out=gpurun_out/kl_${lib%.so}
  rm -rf $out; mkdir -p $out
  export BRIEF_LIB=$PWD/brief_pytorch_amd/$lib
  rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python3 bench.py $args --no-extras --no-cpu-baseline --no-psnr > $out/bench.json 2> $out/err.txt
  f=$(find $out -name "p_kernel_stats.csv" | head -1)
  echo "== $lib"
  python3 - "$f" "$out/bench.json" <<'PY'
import csv, sys, json
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:6]:
    if float(r["Percentage"]) > 1: print("  %-60s calls %6s avg %9.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
o = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); print("  ms_per_step %.4f" % o["ms_per_step"])
PY
  find $out -name "p_kernel_trace.csv" -delete
done
