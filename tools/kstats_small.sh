#!/bin/bash
# per-kernel averages of narrow-net steps for several library builds:  tools/kstats_small.sh "L F n" libA.so libB.so ...   (e.g. "5 22 262144")
cfg=$1; shift
set -- $cfg "$@"
L=$1; F=$2; n=$3; shift 3
export TMPDIR=/tmp
for lib in "$@"; do
  out=gpurun_out/ks_${lib%.so}
  rm -rf $out; mkdir -p $out
  BRIEF_LIB=$PWD/brief_pytorch_amd/$lib rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python3 tools/step_time.py $L $F fp32 $n 300 > $out/run.txt 2> $out/err.txt
  f=$(find $out -name "p_kernel_stats.csv" | head -1)
  echo "== $lib ($L x $F, n = $n)"
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:4]:
    print("  %-60s calls %6s avg %9.2f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  rm -rf $out
done
