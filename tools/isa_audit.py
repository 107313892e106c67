"""Compile every csrc/*.hip to gfx950 assembly (hipcc -S, the product's flags) and report, per kernel, what tends to hide a codegen accident:
FLAT memory instructions (a pointer that lost its address space: FLAT loads count on lgkmcnt as well, so the next LDS wait waits for L2), scratch
(spill) instructions, VGPR / spill counts, MFMAs that sit alone in a basic block (a uniform branch per MFMA: each one waits for its own operands),
the number of s_waitcnt vmcnt(0) and how many of them stand directly in front of a load ("wait, then load": loads that go out one round trip at a
time -- a load under one `if` and its use under another, or two load forms under complementary lane masks into the same registers).
Round 5 finds (profiles/r14k-o): cam_dense_block_kernel (36 lone MFMAs, 54 FLAT loads, two drains per layer), conv2ds_kernel (28-41 wait-then-load),
linear_f32_splitk_kernel (12), fcm_block_kernel<5, 2, true> (15: measured, no effect).
usage: python tools/isa_audit.py [file.hip ...]      (writes /tmp/isa_audit/<file>.s)"""
import glob, os, re, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, 'voiceprintrecognition-pytorch_amd')
sys.path.insert(0, PKG)
import build_native

OUT = '/tmp/isa_audit'


def audit(files=None):
    """-> {source file name: {demangled kernel name: dict(vgpr, vgpr_spill, sgpr_spill, flat, scratch, mfma, lone, vm0, wtl)}}"""
    os.makedirs(OUT, exist_ok=True)
    files = [os.path.join(PKG, 'csrc', f) for f in files] if files else sorted(glob.glob(os.path.join(PKG, 'csrc', '*.hip')))
    procs = []
    for src in files:
        asm = os.path.join(OUT, os.path.basename(src) + '.s')
        cmd = [build_native.HIPCC] + build_native.FLAGS + build_native._file_flags(src) + ['-Wno-inline-asm', '--cuda-device-only', '-S', '-x', 'hip', src, '-o', asm]
        procs.append((src, asm, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    result = {}
    for src, asm, p in procs:
        out, _ = p.communicate()
        if p.returncode:
            raise RuntimeError(f'hipcc failed on {src}:\n' + out.decode()[-2000:])
        kernel, stats = None, {}
        for line in open(asm):
            m = re.match(r'^(_Z\w+):', line)
            if m:
                kernel = m.group(1)
                stats[kernel] = dict(flat=0, scratch=0, mfma=0, lone=0, in_block=0, vm0=0, wtl=0, recent=[])
                continue
            if kernel is None:
                continue
            st = stats[kernel]
            t = line.strip()
            if t.startswith('.LBB') or t.startswith('s_cbranch') or t.startswith('s_branch'):
                if st['in_block'] == 1:
                    st['lone'] += 1
                st['in_block'] = 0
                continue
            if t and not t.startswith((';', '.')):
                if t.startswith(('global_load_dword', 'buffer_load_dword', 'flat_load_dword')) and any(q.startswith('s_waitcnt') and 'vmcnt(0)' in q for q in st['recent'][-4:]):
                    st['wtl'] += 1
                if t.startswith('s_waitcnt') and 'vmcnt(0)' in t:
                    st['vm0'] += 1
                st['recent'] = (st['recent'] + [t])[-4:]
            if t.startswith(('flat_load', 'flat_store', 'flat_atomic')):
                st['flat'] += 1
            elif t.startswith('scratch_'):
                st['scratch'] += 1
            elif t.startswith('v_mfma'):
                st['mfma'] += 1
                st['in_block'] += 1
        meta, cur = {}, None
        for line in open(asm):
            t = line.strip()
            m = re.match(r'\.name:\s+(\S+)', t)
            if m:
                cur = m.group(1)
                meta.setdefault(cur, {})
            m = re.match(r'\.(vgpr_count|vgpr_spill_count|sgpr_spill_count):\s+(\d+)', t)
            if m and cur:
                meta[cur][m.group(1)] = int(m.group(2))
        kernels = {}
        for k, st in stats.items():
            if k not in meta:
                continue
            name = subprocess.run(['c++filt', k], capture_output=True, text=True).stdout.strip().split('(')[0]
            kernels[name] = dict(vgpr=meta[k].get('vgpr_count', 0), vgpr_spill=meta[k].get('vgpr_spill_count', 0), sgpr_spill=meta[k].get('sgpr_spill_count', 0),
                                 flat=st['flat'], scratch=st['scratch'], mfma=st['mfma'], lone=st['lone'], vm0=st['vm0'], wtl=st['wtl'])
        result[os.path.basename(src)] = kernels
    return result


def main():
    for fname, kernels in audit(sys.argv[1:]).items():
        print(fname)
        for name, k in kernels.items():
            flag = ' <--' if k['flat'] or k['scratch'] or k['vgpr_spill'] or k['lone'] > 2 or k['wtl'] > 2 else ''
            print(f"  {name[:90]:90s} vgpr {k['vgpr']:3d} spill v{k['vgpr_spill']} s{k['sgpr_spill']:<3d} flat {k['flat']:3d} scratch {k['scratch']:3d} "
                  f"mfma {k['mfma']:4d} (alone in a block: {k['lone']}) vmcnt(0) {k['vm0']:3d} (wait-then-load: {k['wtl']}){flag}")


if __name__ == '__main__':
    main()
