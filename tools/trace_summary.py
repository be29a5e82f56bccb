'''Per-kernel time of a rocprofv3 --kernel-trace run (rocpd .db):  python tools/trace_summary.py <dir or .db> <steps in the trace> [top]
Prints ms/step, calls/step and average microseconds per kernel, sorted by time; template arguments kept, namespaces dropped.'''
import collections, glob, os, re, sqlite3, sys
trace, nstep = sys.argv[1], float(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 45
db = trace if trace.endswith('.db') else sorted(glob.glob(os.path.join(trace, '**', '*.db'), recursive=True))[0]
cur = sqlite3.connect(db).cursor()
tables = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
kt = 'kernels' if 'kernels' in tables else [t for t in tables if 'kernel' in t.lower()][0]
agg = collections.defaultdict(lambda: [0, 0.0])
for n, s, e in cur.execute('select name, start, end from %s' % kt):
    n = re.sub(r'\(anonymous namespace\)::|void ', '', n)
    n = re.sub(r'\(ConvArgs\)$|\((?:[^()]|\([^()]*\))*\)$', '', n).strip()
    a = agg[n]
    a[0] += 1
    a[1] += (e - s) / 1e6
tot = sum(a[1] for a in agg.values())
print('total kernel time %.2f ms/step over %d kernels' % (tot / nstep, len(agg)))
for n, (c, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print('%6.2f%% %8.3f ms/step %7.1f calls/step %9.1f us  %s' % (100 * ms / tot, ms / nstep, c / nstep, 1e3 * ms / c, n[:150]))
