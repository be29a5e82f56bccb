'''GPU box, after `rocprofv3 --kernel-trace -d <dir> -o r -- python3 bench.py ...`: which kernels run right before / after every
__amd_rocclr_copyBuffer launch (who issues the small copies of a step).'''
import collections
import csv
import glob
import sys

rows = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r['Start_Timestamp']), r['Kernel_Name']))
rows.sort()
pairs = collections.Counter()
for i, (t, name) in enumerate(rows):
    if 'copyBuffer' in name:
        prev = rows[i - 1][1][:60] if i else '-'
        nxt = rows[i + 1][1][:60] if i + 1 < len(rows) else '-'
        pairs[(prev, nxt)] += 1
for k, v in pairs.most_common(15):
    print(v, k)
