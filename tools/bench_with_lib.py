"""bench.py on a variant build of the library (tools/build_variant.py) -- A/B arms inside one gpurun call:
    python tools/bench_with_lib.py tools/probe/<name>.so --model eres2netv2_w96s4 --batch 64 --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs
The product library stays what `import mvector` loads everywhere else; only this process binds the variant."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'voiceprintrecognition-pytorch_amd')]
lib = os.path.abspath(sys.argv[1])
sys.argv = [os.path.join(ROOT, 'bench.py')] + sys.argv[2:]
from mvector import _hip
_hip._lib = _hip.bind(ctypes.CDLL(lib))
assert _hip._lib.mv_abi_version() == _hip.lib().mv_abi_version()
print(f'# library: {lib}', file=sys.stderr)
import bench
bench.main()
