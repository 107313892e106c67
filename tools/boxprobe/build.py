"""hipcc --offload-arch=gfx950 tools/boxprobe/boxprobe.hip -> tools/probe/libmvector_boxprobe.so (git-ignored, travels with gpurun).
Measurement infrastructure of bench.py's `box` block; `__graft_entry__.build()` calls build()."""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, 'boxprobe.hip')
OUT_DIR = os.path.join(os.path.dirname(HERE), 'probe')
LIB = os.path.join(OUT_DIR, 'libmvector_boxprobe.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
CMD = [HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-fvisibility=hidden', '-x', 'hip', SRC, '-o', LIB]


def build(force=False):
    os.makedirs(OUT_DIR, exist_ok=True)
    with open(SRC, 'rb') as f:
        want = hashlib.sha1(f.read() + ' '.join(CMD).encode()).hexdigest()
    stamp = LIB + '.stamp'
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == want:
        return LIB
    subprocess.check_call(CMD)
    with open(stamp, 'w') as f:
        f.write(want)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
