"""Aggregate a rocprofv3 --pmc counter_collection.csv per kernel name."""
import csv, sys, collections, glob
path = sys.argv[1]
files = glob.glob(path + '/**/*counter_collection.csv', recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for f in files:
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].replace('void ','').replace('(anonymous namespace)::','').split('(')[0][:46]
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == list(agg[k].keys())[0]: calls[k] += 1
names = sorted({c for v in agg.values() for c in v})
print('kernel'.ljust(46), 'calls', ' '.join(n[-18:].rjust(18) for n in names))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get('SQ_WAVE_CYCLES', kv[1].get(names[0], 0))):
    print(k.ljust(46), str(calls[k]).rjust(5), ' '.join(('%.4g' % v.get(n, 0)).rjust(18) for n in names))
