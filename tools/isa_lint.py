'''Static checks on the gfx950 code hipcc generates for the kernels (no GPU needed).  The four patterns below were each found in a
hot kernel this round and cost 3-30 %:

  flat     flat_load / flat_store: the address space of a pointer was lost (e.g. a select between a global pointer and a
           `__device__ const` array, which lives in the constant address space).  Flat accesses count in lgkmcnt as well as vmcnt,
           so every wait for an LDS read also waits for them.
  got      @gotpcrel: a device global reached through the GOT -- an s_load + s_waitcnt lgkmcnt(0) at every use.  Pass its address
           as a kernel argument instead.
  scratch  ScratchSize / spilled VGPRs (from -Rpass-analysis=kernel-resource-usage).
  stwait   an s_waitcnt vmcnt(0) within three instructions in front of a global store, many times in one kernel: on gfx9 stores count
           in vmcnt like loads, so a value that MAY come from a load (a branched-around read-modify-write, a bias loaded long
           ago but first used inside the masked store block) makes hipcc wait for the previous store before each next one.

usage: python tools/isa_lint.py [csrc/file.hip ...]      (default: every translation unit of the library)'''
import collections, glob, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'radar-camera-fusion-depth_amd', 'csrc')


def demangle(names):
    out = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True).stdout.split('\n')
    return [re.sub(r'\(anonymous namespace\)::|^void ', '', n)[:140] for n in out]


def lint(src):
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, 'k.s')
        r = subprocess.run(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-I', os.path.join(ROOT, 'include'), '--cuda-device-only',
                            '-Rpass-analysis=kernel-resource-usage', '-S', src, '-o', asm], capture_output=True, text=True)
        if not os.path.exists(asm):
            print(r.stderr[-2000:])
            raise SystemExit('compilation of %s failed' % src)
        lines = open(asm).read().split('\n')
    res = {}
    for blk in re.split(r'remark: Function Name: ', r.stderr)[1:]:
        g = lambda k: int(re.search(k + r': (\d+)', blk).group(1))
        res[blk.split()[0]] = (g(r'ScratchSize \[bytes/lane\]'), g('VGPRs Spill'), g('VGPRs'), g('AGPRs'), g(r'Occupancy \[waves/SIMD\]'))
    stats = collections.OrderedDict()
    kern = None
    for i, l in enumerate(lines):
        m = re.match(r'^(_Z\S+):', l)
        if m:
            kern = m.group(1)
            stats[kern] = collections.Counter()
            continue
        if kern is None:
            continue
        s = stats[kern]
        if re.search(r'\bflat_(load|store)', l):
            s['flat'] += 1
        if 'gotpcrel' in l:
            s['got'] += 1
        if re.search(r'\b(global|buffer)_store', l):
            s['stores'] += 1
            prev = [x for x in lines[max(0, i - 8):i] if x.strip() and not x.strip().startswith((';', '.'))][-3:]
            if any('s_waitcnt vmcnt(0)' in x for x in prev):
                s['stwait'] += 1
    names = [k for k in stats if k in res]
    flagged = 0
    for k, pretty in zip(names, demangle(names)):
        s = stats[k]
        scratch, vspill, vg, ag, occ = res[k]
        why = []
        if s['flat']:
            why.append('%d flat accesses' % s['flat'])
        if s['got']:
            why.append('%d GOT loads' % s['got'])
        if vspill:
            why.append('%d spilled VGPRs (%d B scratch)' % (vspill, scratch))
        if s['stwait'] >= 4:
            why.append('%d of %d stores behind a vmcnt(0)' % (s['stwait'], s['stores']))
        if why:
            flagged += 1
            print('  %-110s vgpr %3d agpr %3d occ %d : %s' % (pretty[:110], vg, ag, occ, '; '.join(why)))
    print('%s: %d kernels, %d flagged' % (os.path.basename(src), len(names), flagged))


if __name__ == '__main__':
    for f in (sys.argv[1:] or sorted(glob.glob(os.path.join(CSRC, '*.hip')))):
        lint(f)
