"""ctypes binding + build of the fp32 conv2d yard-stick (tools only; the product library does not export these entry points since ABI 4).

    from yardstick import binding; cdll = binding.load()      # builds tools/probe/libmvector_yardstick.so on first use (hipcc, gfx950)
The library is the product's objects + tools/yardstick/conv2d_f32.hip, so every product entry point is there as well (one process, one library)."""
import ctypes, glob, os, subprocess, sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
PKG = os.path.join(ROOT, 'voiceprintrecognition-pytorch_amd')
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)
from mvector import _hip  # noqa: E402
from mvector._hip import c_f32, c_i32, c_i64, c_vp  # noqa: E402

LIB = os.path.join(ROOT, 'tools', 'probe', 'libmvector_yardstick.so')


class MvConv2dDesc(ctypes.Structure):
    _fields_ = [('x', c_vp), ('x2', c_vp), ('x2_mode', c_i32), ('cin1', c_i32), ('ldx', c_i64), ('ldx2', c_i64),
                ('w', c_vp), ('bias', c_vp), ('res', c_vp), ('res2', c_vp), ('ldres', c_i64), ('ldres2', c_i64),
                ('y', c_vp), ('ldy', c_i64), ('B', c_i32), ('H', c_i32), ('W', c_i32), ('cin16', c_i32),
                ('cout16', c_i32), ('ks', c_i32), ('stride', c_i32), ('epi', c_i32), ('lo', c_f32), ('hi', c_f32),
                ('cin_alg', c_i32), ('cout_alg', c_i32), ('stride_w', c_i32)]


SIGNATURES = {
    'mv_conv2d_packed_elems': (c_i64, [c_i32, c_i32, c_i32]),
    'mv_conv2d_pack_weight': (c_i32, [c_vp, c_vp, c_i32, c_i32, c_i32, c_vp, c_vp]),
    'mv_conv2d_forward': (c_i32, [ctypes.POINTER(MvConv2dDesc), c_vp]),
}


def build():
    import build_native
    build_native.build()
    src = os.path.join(HERE, 'conv2d_f32.hip')
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    obj = os.path.join('/tmp', 'yardstick_conv2d_f32.o')
    subprocess.check_call([build_native.HIPCC] + build_native.FLAGS + ['-I', HERE, '-Wno-inline-asm', '-x', 'hip', '-c', src, '-o', obj])
    objs = glob.glob(os.path.join(PKG, 'build', '*.o'))
    subprocess.check_call([build_native.HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs + [obj])
    return LIB


def load(rebuild=False):
    if rebuild or not os.path.exists(LIB):
        build()
    cdll = _hip.bind(ctypes.CDLL(LIB))
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(cdll, name)
        fn.restype, fn.argtypes = restype, argtypes
    return cdll


if __name__ == '__main__':
    print(build())
