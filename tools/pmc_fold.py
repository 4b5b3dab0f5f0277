import csv,glob,sys,collections,re
d=sys.argv[1]; flt=sys.argv[2]
f=glob.glob(d+'/*/*counter_collection.csv')[0]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k=re.sub(r"\(anonymous namespace\)::","",r['Kernel_Name']).replace('void ','').split('(')[0]
    if flt not in k: continue
    agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in agg.items():
    print(k)
    for c,vals in sorted(v.items()):
        print(f"   {c:32s} n={len(vals):3d} avg {sum(vals)/len(vals):16.1f}")
