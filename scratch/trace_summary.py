import sys, csv, glob, collections
# per-dispatch durations of the last frame, in launch order
path = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last frame = after the last rt_generate
last = max(i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('rt_generate'))
frame = rows[last:]
t0 = int(frame[0]['Start_Timestamp'])
tot = collections.defaultdict(float)
line = []
for r in frame:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1000.0
    name = r['Kernel_Name'].split('(')[0]
    tot[name] += d
    line.append('%s %.0f' % (name.replace('rt_', ''), d))
    if name == 'rt_shade' or name == 'rt_generate': print(' | '.join(line)); line = []
print(' | '.join(line))
print('span %.1f us' % ((int(frame[-1]['End_Timestamp']) - t0) / 1000.0), {k: round(v, 1) for k, v in tot.items()})
