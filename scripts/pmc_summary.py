import csv, glob, collections, sys
d = sys.argv[1]
f = glob.glob(d + '/*/*counter_collection.csv')[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    agg[(r['Kernel_Name'][:60], r['Dispatch_Id'])][r['Counter_Name']] += float(r['Counter_Value'])
for k, v in agg.items():
    if 'path_kernel' in k[0] or 'kernel' in k[0]:
        print(k)
        for n, x in sorted(v.items()): print("   %-28s %.4g" % (n, x))
        if 'SQ_THREAD_CYCLES_VALU' in v and 'SQ_ACTIVE_INST_VALU' in v:
            print("   VALU lane utilisation %.1f%%" % (100 * v['SQ_THREAD_CYCLES_VALU'] / (64 * v['SQ_ACTIVE_INST_VALU'])))
