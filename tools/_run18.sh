export TMPDIR=/tmp
for d in 0 1 0 1; do
  export BRIEF_DIAG=$d
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sd_$d -o p -- python3 tools/one_net16.py 5 256 fp32 60 > /dev/null 2>&1
  python3 - <<PY
import csv
for r in list(csv.DictReader(open('gpurun_out/sd_$d/p_kernel_stats.csv')))[:2]:
    print('diag=$d   %-36s calls %s avg %.1f us' % (r['Name'][:36], r['Calls'], float(r['AverageNs'])/1e3))
PY
  rm -rf gpurun_out/sd_$d
done
