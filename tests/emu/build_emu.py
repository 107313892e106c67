"""Build tests/emu/build/libmvector_emu.so: the csrc/*.hip kernels compiled for the HOST against
hip_emu.h (SIMT emulator).  Test infrastructure only -- see hip_emu.h."""
import fcntl
import glob
import hashlib
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, 'voiceprintrecognition-pytorch_amd', 'csrc')
OUT = os.path.join(HERE, 'build')
CLANG = '/opt/rocm/lib/llvm/bin/clang++'


def build(verbose=False):
    # MV_EMU_SANITIZE=address: the same sources with AddressSanitizer (own build directory; the process that loads the library must have
    # the sanitizer runtime preloaded -- tools/emu_check.py does that)
    sanitize = os.environ.get('MV_EMU_SANITIZE', '')
    assert sanitize in ('', 'address', 'undefined'), sanitize
    OUT = os.path.join(HERE, {'': 'build', 'address': 'build_asan', 'undefined': 'build_ubsan'}[sanitize])
    # undefined: the integer checks that matter for index arithmetic (alignment is left out: the device's vector loads need none)
    checks = 'address' if sanitize == 'address' else 'signed-integer-overflow,shift,integer-divide-by-zero,bounds,null,float-cast-overflow'
    extra = [f'-fsanitize={checks}', '-shared-libsan', '-fno-omit-frame-pointer', '-g1'] if sanitize else ['-g0']
    if sanitize == 'undefined':
        extra.append(f'-fno-sanitize-recover={checks}')
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, 'lock'), 'w') as lock:   # pytest-xdist workers arrive here together: one builds, the others find the stamp
        fcntl.flock(lock, fcntl.LOCK_EX)
        return _build_locked(OUT, extra, sanitize, verbose)


def _build_locked(OUT, extra, sanitize, verbose):
    srcs = sorted(glob.glob(os.path.join(CSRC, '*.hip')) + glob.glob(os.path.join(CSRC, '*.cpp')))
    deps = srcs + sorted(glob.glob(os.path.join(CSRC, '*.h'))) + [os.path.join(HERE, 'hip_emu.h'), os.path.join(HERE, 'arch', 'gfx950.h'),
                                                                   os.path.join(ROOT, 'include', 'mvector_hip.h')]
    h = hashlib.sha1(b'flags: -mfma -ffp-contract=fast')
    for d in deps:
        h.update(open(d, 'rb').read())
    lib = os.path.join(OUT, 'libmvector_emu.so')
    stamp = os.path.join(OUT, 'stamp')
    if os.path.exists(lib) and os.path.exists(stamp) and open(stamp).read() == h.hexdigest():
        return lib
    objs = []
    procs = []
    for s in srcs:
        o = os.path.join(OUT, os.path.basename(s) + '.o')
        # tests/emu in front of csrc on the include path: <arch/gfx950.h> resolves to the emulator's host spellings
        # -mfma -ffp-contract=fast: `a * b + c` rounds once, as in the device build (hipcc contracts by default) -- without them the host build agreed with a
        # torch fp32 reference where the DEVICE does not (r14x: two conv1d fuzz cases that only the GPU failed)
        cmd = [CLANG, '-x', 'c++', '-std=c++17', '-O1', '-fPIC', '-mfma', '-ffp-contract=fast'] + extra + [ '-I', HERE, '-I', CSRC, '-include',
               os.path.join(HERE, 'hip_emu.h'), '-Wno-unused-value', '-c', s, '-o', o]
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(o)
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError('emu build failed:\n' + ' '.join(cmd) + '\n' + out.decode())
        if verbose and out:
            print(out.decode())
    subprocess.check_call([CLANG, '-shared', '-o', lib] + ([e for e in extra if e.startswith('-fsanitize') or e == '-shared-libsan'] if sanitize else []) + objs)
    open(stamp, 'w').write(h.hexdigest())
    return lib


if __name__ == '__main__':
    print(build(verbose=True))
