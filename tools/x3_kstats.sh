#!/bin/bash
# per-kernel averages of 300 BRIEF_PREC_BF16X3 steps (4x256, 100 000 samples) for library builds / BRIEF_DIAG values:
#   tools/x3_kstats.sh lib.so:diag [lib.so:diag ...]     (rocprofv3 --kernel-trace --stats, one run each)
export TMPDIR=/tmp
for spec in "$@"; do
  lib=${spec%%:*}; diag=${spec##*:}
  out=gpurun_out/x3k_${lib%.so}_$diag
  rm -rf $out; mkdir -p $out
  export BRIEF_LIB=$PWD/brief_pytorch_amd/$lib BRIEF_DIAG=$diag
  rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python3 tools/one_x3.py 300 > $out/out.txt 2> $out/err.txt || { tail -5 $out/err.txt; exit 1; }
  f=$(find $out -name "p_kernel_stats.csv" | head -1)
  echo "== $lib BRIEF_DIAG=$diag"
  python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
for r in rows[:5]:
    print("  %-60s calls %6s avg %9.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
