'''Register / spill / scratch table of every gfx950 kernel in the built library, read from the code objects' metadata (no GPU, no
recompilation): the library's .hip_fatbin section holds one clang offload bundle per translation unit; each is unbundled with
clang-offload-bundler and its AMDGPU metadata note parsed.

    python tools/kernel_meta.py [path/to/librcf_hip.so] [--spills] [--ops]   # --spills: only kernels with spills or private scratch; --ops: count scratch instructions

Used by tests/test_host_logic.py::test_no_hot_kernel_spills_or_uses_scratch (explicit allow-list of cold variants).'''
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = '/opt/rocm/lib/llvm/bin'
MAGIC = b'__CLANG_OFFLOAD_BUNDLE__'


def kernels(lib=None, count_scratch_ops=False):
    '''-> list of dicts: name (demangled), vgpr, agpr, sgpr, vgpr_spill, sgpr_spill, scratch (bytes / lane), lds (static bytes).
    count_scratch_ops: also disassemble the code objects and count the scratch_load / scratch_store instructions of every kernel
    ('scratch_ops'): a kernel whose SGPRs spill into VGPR lanes still reports a private segment size (the frame slots of those
    spills) without ever touching scratch memory -- this tells the two apart.'''
    lib = lib or os.path.join(ROOT, 'radar-camera-fusion-depth_amd', 'librcf_hip.so')
    out = []
    ops = {}
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, 'fat.bin')
        subprocess.run([os.path.join(LLVM, 'llvm-objcopy'), '--dump-section', '.hip_fatbin=' + fat, lib, os.path.join(tmp, 'unused.o')], check=True)
        blob = open(fat, 'rb').read()
        starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
        for i, s in enumerate(starts):
            piece = os.path.join(tmp, 'b%d.bin' % i)
            open(piece, 'wb').write(blob[s:starts[i + 1] if i + 1 < len(starts) else len(blob)])
            co = os.path.join(tmp, 'b%d.co' % i)
            r = subprocess.run([os.path.join(LLVM, 'clang-offload-bundler'), '--unbundle', '--type=o', '--input=' + piece,
                                '--targets=hipv4-amdgcn-amd-amdhsa--gfx950', '--output=' + co], capture_output=True, text=True)
            if r.returncode != 0 or not os.path.exists(co) or os.path.getsize(co) == 0:
                continue
            notes = subprocess.run([os.path.join(LLVM, 'llvm-readelf'), '--notes', co], capture_output=True, text=True, check=True).stdout
            cur = {}
            for line in notes.splitlines():
                m = re.match(r'\s*-?\s*\.(\w+):\s+(.*)$', line)
                if not m:
                    continue
                key, val = m.group(1), m.group(2).strip()
                if key == 'agpr_count' and line.lstrip().startswith('-'):      # first key of a kernel record
                    if 'name' in cur:
                        out.append(cur)
                    cur = {}
                if key in ('agpr_count', 'vgpr_count', 'sgpr_count', 'vgpr_spill_count', 'sgpr_spill_count', 'private_segment_fixed_size',
                           'group_segment_fixed_size'):
                    cur[key] = int(val)
                elif key == 'name':
                    cur['name'] = val.strip("'\"")
            if 'name' in cur:
                out.append(cur)
            if count_scratch_ops:
                dis = subprocess.run([os.path.join(LLVM, 'llvm-objdump'), '-d', '--mcpu=gfx950', co], capture_output=True, text=True).stdout
                name = None
                for line in dis.splitlines():
                    if line.endswith('>:') and '<' in line:
                        name = line[line.index('<') + 1:-2]
                        ops[name] = 0
                    elif name is not None and 'scratch_' in line:
                        ops[name] += 1
    names = [k['name'] for k in out]
    dem = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True, check=True).stdout.splitlines()
    res = []
    for k, d in zip(out, dem):
        d = re.sub(r'\(anonymous namespace\)::|^void ', '', d)
        d = re.sub(r'\((?:[^()]|\([^()]*\))*\)$', '', d).strip()
        res.append({'name': d, 'vgpr': k.get('vgpr_count', 0), 'agpr': k.get('agpr_count', 0), 'sgpr': k.get('sgpr_count', 0),
                    'vgpr_spill': k.get('vgpr_spill_count', 0), 'sgpr_spill': k.get('sgpr_spill_count', 0),
                    'scratch': k.get('private_segment_fixed_size', 0), 'lds': k.get('group_segment_fixed_size', 0),
                    'scratch_ops': ops.get(k['name']) if count_scratch_ops else None})
    return res


if __name__ == '__main__':
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    ks = kernels(args[0] if args else None, count_scratch_ops='--ops' in sys.argv)
    only = '--spills' in sys.argv
    shown = 0
    for k in sorted(ks, key=lambda k: (-k['vgpr_spill'], -k['scratch'], k['name'])):
        if only and not (k['vgpr_spill'] or k['scratch']):   # SGPR spills go to VGPR lanes (v_writelane), not to memory
            continue
        shown += 1
        print('%4d vgpr %4d agpr  spill v%-3d s%-3d scratch %4d B%s  %s' % (k['vgpr'], k['agpr'], k['vgpr_spill'], k['sgpr_spill'], k['scratch'],
                                                                           '' if k['scratch_ops'] is None else ' (%d scratch instructions)' % k['scratch_ops'], k['name'][:170]))
    print('%d kernels in the library, %d listed' % (len(ks), shown))
