"""scratch: average PMC values per kernel from rocprofv3 counter_collection csv files given on the command line"""
import collections, csv, sys
for f in sys.argv[1:]:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name'][:40]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in acc.items():
        if any(t in k for t in ('k_fused', 'k_wgrad', 'k_small', 'k16', 'k_reduce', 'kb16', 'k_wg16')):
            print(k, {c: '%.4g' % (sum(x) / len(x)) for c, x in v.items()})
