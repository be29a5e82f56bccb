'''Print per-kernel averages of every counter found in rocprofv3 --pmc CSV passes under <dir>.
usage: python tools/pmc_kernel.py <dir> [name filter]'''
import csv, glob, re, sys, collections
base = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ''
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob(base + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r'\(anonymous namespace\)::|void ', '', r['Kernel_Name'])
        name = re.sub(r'\(ConvArgs\)|\(.*', '', name).strip()
        if flt and flt not in name:
            continue
        a = agg[name][r['Counter_Name']]
        a[0] += 1
        a[1] += float(r['Counter_Value'])
for name, cs in sorted(agg.items()):
    print(name)
    for cn, (n, s) in sorted(cs.items()):
        print('   %-32s n=%-4d avg=%.4g' % (cn, n, s / n))
