'''Per basic block of one kernel in a hipcc -S listing: counts of the instruction classes that matter (MFMA, LDS, DMA, scratch spills,
VALU ...).  python tools/isa_blocks.py file.s <substring of the mangled kernel name>'''
import re
import sys
from collections import Counter

path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().splitlines()
start = None
for i, l in enumerate(lines):
    if re.match(r'^_Z\S*:', l) and key in l:
        start = i
        break
if start is None:
    sys.exit('kernel not found')
print(lines[start][:140])
blk, cnt, order = 'entry', Counter(), []
def flush():
    if cnt:
        keys = ['v_mfma', 'ds_read', 'ds_write', 'lds_dma', 'global_load', 'global_store', 'scratch_load', 'scratch_store', 'valu', 'salu', 's_waitcnt', 's_barrier']
        print('%-12s %s' % (blk, '  '.join('%s=%d' % (k, cnt[k]) for k in keys if cnt[k])))
for l in lines[start + 1:]:
    t = l.strip()
    if t.startswith('.Lfunc_end'):
        break
    m = re.match(r'^(\.LBB\d+_\d+):', t)
    if m:
        flush()
        blk, cnt = m.group(1), Counter()
        continue
    if not t or t.startswith(('.', ';', '//')):
        continue
    op = t.split()[0]
    if op.startswith('v_mfma'): cnt['v_mfma'] += 1
    elif op.startswith('ds_read') or op.startswith('ds_load'): cnt['ds_read'] += 1
    elif op.startswith('ds_write') or op.startswith('ds_store'): cnt['ds_write'] += 1
    elif op.startswith('global_load_lds') or (op.startswith('buffer_load') and ' lds' in t): cnt['lds_dma'] += 1
    elif op.startswith('global_load') or op.startswith('buffer_load') or op.startswith('flat_load'): cnt['global_load'] += 1
    elif op.startswith('global_store') or op.startswith('buffer_store') or op.startswith('flat_store'): cnt['global_store'] += 1
    elif op.startswith('scratch_load'): cnt['scratch_load'] += 1
    elif op.startswith('scratch_store'): cnt['scratch_store'] += 1
    elif op == 's_waitcnt': cnt['s_waitcnt'] += 1
    elif op == 's_barrier': cnt['s_barrier'] += 1
    elif op.startswith('v_'): cnt['valu'] += 1
    elif op.startswith('s_'): cnt['salu'] += 1
flush()
