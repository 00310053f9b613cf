"""Summarises a rocprofv3 --pmc counter_collection.csv: per kernel name, mean of each counter."""
import csv, sys, collections
path = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(path)):
    name = r['Kernel_Name']
    if 'voc::' not in name:
        continue
    short = name.split('(')[0].replace('void voc::', '')
    acc[short][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    print(k)
    for cn, vals in d.items():
        print(f'   {cn:28s} mean {sum(vals)/len(vals):.4g}  (n={len(vals)})')
