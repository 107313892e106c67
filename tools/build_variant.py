"""Build an A/B arm of the native library: ONE source recompiled (optionally from another git revision and / or with extra -D flags) and
linked with the product's other objects into tools/probe/<name>.so; tools/bench_*.py pick it up through MV_PROBE_LIB.
usage: python tools/build_variant.py <name> <csrc file>[@<git rev>] [-DFLAG=1 ...] [--replace OLD NEW ...]
       --replace: literal text substitution in the (copied) source before it is compiled -- how a constant or a line is changed for an A/B arm without
       an #if in the product source (every OLD must occur)
e.g.   python tools/build_variant.py libres2_base res2.hip@HEAD~1        (the previous kernel beside the working tree's in one gpurun call)"""
import glob, os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, 'voiceprintrecognition-pytorch_amd')
sys.path.insert(0, PKG)
import build_native

name, spec, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
replaces = []
while '--replace' in flags:
    i = flags.index('--replace')
    replaces.append((flags[i + 1], flags[i + 2]))
    del flags[i:i + 3]
build_native.build()   # the other objects must be current
fname, _, rev = spec.partition('@')
src = os.path.join(PKG, 'csrc', fname)
os.makedirs(os.path.join(REPO, 'tools', 'probe'), exist_ok=True)
if rev:
    text = subprocess.check_output(['git', '-C', REPO, 'show', f'{rev}:voiceprintrecognition-pytorch_amd/csrc/{fname}'])
    src = os.path.join('/tmp', f'variant_{name}_{fname}')
    open(src, 'wb').write(text)
if replaces:
    text = open(src).read()
    for old, new in replaces:
        assert old in text, f'--replace: {old!r} not in {src}'
        text = text.replace(old, new)
    src = os.path.join('/tmp', f'variant_{name}_{fname}')
    open(src, 'w').write(text)
obj = os.path.join('/tmp', f'variant_{name}.o')
cmd = [build_native.HIPCC] + build_native.FLAGS + build_native._file_flags(src) + flags + ['-Wno-inline-asm', '-x', 'hip', '-c', src, '-o', obj]
subprocess.check_call(cmd)
objs = [o for o in glob.glob(os.path.join(PKG, 'build', '*.o')) if os.path.basename(o) != fname + '.o']
out = os.path.join(REPO, 'tools', 'probe', name + '.so')
subprocess.check_call([build_native.HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', out] + objs + [obj])
print('built', out)
