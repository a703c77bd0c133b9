#!/bin/bash
# what k_reduce's two kinds of blocks cost alone (diagnostics build: BRIEF_REDUCE_PART = 1 hidden-parameter blocks only, 2 skinny blocks only), with and without
# the vectorised hidden path:  tools/reduce_parts.sh "<bench args>"
args=$1
export TMPDIR=/tmp
export BRIEF_LIB=$PWD/brief_pytorch_amd/libbrief_hip_diag.so
for vec in 1 0; do for part in 0 1 2; do
  out=gpurun_out/rp_${vec}_${part}
  rm -rf $out; mkdir -p $out
  BRIEF_REDUCE_VEC=$vec BRIEF_REDUCE_PART=$part rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python3 bench.py $args --no-extras --no-cpu-baseline --no-psnr > $out/bench.json 2> $out/err.txt
  f=$(find $out -name "p_kernel_stats.csv" | head -1)
  echo "== vec $vec part $part: $(python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if r["Name"].startswith("k_reduce"): print("k_reduce avg %.1f us (%s calls)" % (float(r["AverageNs"]) / 1e3, r["Calls"]))
PY
)"
  rm -rf $out
done; done
