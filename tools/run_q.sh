cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r1q
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "asp_pool" 2>&1 | grep -E "Error|assert|passed|failed" | head -20
python - <<'PY'
import sys, os
sys.path[:0] = ['.', 'tests', 'voiceprintrecognition-pytorch_amd']
import torch, layer_checks as lc
from mvector import _hip
for cfg in (dict(B=5, T=9, C=72, A=64, ldx=80, centred=False), dict(B=5, T=9, C=72, A=64, ldx=80, centred=True), dict(B=5, T=9, C=64, A=64, ldx=80, centred=False), dict(B=5, T=9, C=128, A=64, ldx=128, centred=False), dict(B=1, T=16, C=64, A=64, centred=False)):
    try:
        print(cfg, lc.asp_pool_case(_hip.lib(), 'cuda', **cfg))
    except AssertionError as e:
        print(cfg, 'FAIL', e)
PY
