import csv, sys, collections
for f in sys.argv[1:]:
    rows = list(csv.DictReader(open(f)))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        agg[r['Kernel_Name'][:28]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, d in agg.items():
        if not any(x in k for x in ('sketch', 'scatter', 'sort_rank', 'probe')): continue
        print(k, {c: round(sum(v)/len(v)) for c, v in d.items()})
