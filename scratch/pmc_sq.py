import sys, csv, glob, collections
path = glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True)[0]
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
for r in csv.DictReader(open(path)):
    k = r['Kernel_Name'].split('(')[0]
    tot[k][r['Counter_Name']] += float(r['Counter_Value'])
for k, v in tot.items():
    if not k.startswith('rt_'): continue
    w = v.get('SQ_WAVES', 1) or 1
    print(k, ' '.join('%s=%.3g' % (c, x) for c, x in sorted(v.items())))
    print('   per wave:', ' '.join('%s=%.0f' % (c, x / w) for c, x in sorted(v.items()) if c != 'SQ_WAVES'))
