"""Build mvector/lib/libmvector_hip.so: every csrc/*.hip / *.cpp compiled by hipcc for gfx950 (MI355X).

hipcc cross-compiles without a GPU, so this runs in the build container; the .so is git-ignored but travels
to the GPU box with the tree.  No other architecture, no CPU build of the product library."""
import glob
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OUT_DIR = os.path.join(HERE, 'mvector', 'lib')
OBJ_DIR = os.path.join(HERE, 'build')
LIB = os.path.join(OUT_DIR, 'libmvector_hip.so')
EXPORTS = os.path.join(CSRC, 'exports.map')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-value', '-DNDEBUG', '-fvisibility=hidden', '-I', CSRC]  # <arch/gfx950.h> = csrc/arch


def _file_flags(path):
    """extra hipcc flags a source asks for in a line `// hipcc-flags: ...` (e.g. fbank.hip keeps the SLP vectoriser from
    re-packing scalar fp32 code into v_pk_* ops that need register shuffles)"""
    with open(path) as f:
        for line in f:
            if line.startswith('// hipcc-flags:'):
                return line.split(':', 1)[1].split()
    return []


def _digest(paths):
    h = hashlib.sha1(' '.join(FLAGS).encode())
    for p in paths:
        with open(p, 'rb') as f:
            h.update(f.read())
    return h.hexdigest()


def build(force=False, verbose=False):
    os.makedirs(OUT_DIR, exist_ok=True)
    os.makedirs(OBJ_DIR, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, '*.hip')) + glob.glob(os.path.join(CSRC, '*.cpp')))
    headers = sorted(glob.glob(os.path.join(CSRC, '*.h'))) + sorted(glob.glob(os.path.join(CSRC, 'arch', '*.h'))) + [os.path.join(os.path.dirname(HERE), 'include', 'mvector_hip.h')]
    hdr_digest = _digest(headers)
    procs, objs = [], []
    for s in srcs:
        o = os.path.join(OBJ_DIR, os.path.basename(s) + '.o')
        stamp = o + '.stamp'
        want = _digest([s]) + hdr_digest
        objs.append(o)
        if not force and os.path.exists(o) and os.path.exists(stamp) and open(stamp).read() == want:
            continue
        cmd = [HIPCC] + FLAGS + _file_flags(s) + ['-x', 'hip', '-c', s, '-o', o]
        procs.append((s, cmd, stamp, want, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    rebuilt = False
    for s, cmd, stamp, want, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f'hipcc failed on {s}:\n' + out.decode())
        if verbose and out:
            print(out.decode())
        with open(stamp, 'w') as f:
            f.write(want)
        rebuilt = True
    if rebuilt or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(EXPORTS):
        # exports.map: the dynamic symbol table holds the mv_* entry points of include/mvector_hip.h and nothing else -- hipcc gives every __global__
        # function's host-side handle default visibility whatever -fvisibility says, the version script makes them local too (the HIP runtime
        # registers kernels by address from the library's constructor, not by name)
        subprocess.check_call([HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', f'-Wl,--version-script={EXPORTS}', '-o', LIB] + objs)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv, verbose=True))
