export TMPDIR=/tmp
for nw in 4 8; do for dg in 0 1; do
  export BRIEF_K16_NW=$nw BRIEF_DIAG=$dg
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/dg_${nw}_${dg} -o p -- python3 tools/one_net16.py 9 512 bf16 20 > /dev/null 2>&1
  python3 - <<PY
import csv
for r in list(csv.DictReader(open('gpurun_out/dg_${nw}_${dg}/p_kernel_stats.csv')))[:4]:
    print('nw=$nw diag=$dg  %-42s calls %s avg %.1f us' % (r['Name'][:42], r['Calls'], float(r['AverageNs'])/1e3))
PY
done; done
